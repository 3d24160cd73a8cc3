#!/bin/bash
# fused sweep: the nap between polls of a hand-off wait (32 / 8 / 2), config 5 (one pair, 256 bands per plane) and the batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ac; rm -rf $O; mkdir -p $O
for v in 32 8 2 32; do
  if [ $v = 32 ]; then unset STITCH_LIB; else export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_sleep$v.so; fi
  timeout -k 10 300 python bench.py --frame 16384 --pairs-per-step 1 --batch 1 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-single > $O/c5_$v.json 2> $O/c5_$v.err
  python - $O/c5_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; k=d['kernels']
print('config5 sleep', sys.argv[2], 'ms/pair', c['ms_per_pair_per_gpu'], 'verified', d['outputs_verified'], ' '.join(f"{n} {k[n]['ms_per_pair']:.3f}" for n in k if k[n]['ms_per_pair']>0.3))
PY
  timeout -k 10 240 python bench.py --steps 10 --no-cpu-baseline --no-single > $O/b_$v.json 2> $O/b_$v.err
  python - $O/b_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; k=d['kernels']
print('  batch sleep', sys.argv[2], d['value'], 'verified', d['outputs_verified'], 'one-seq', c['one_sequence_in_flight_ms_per_pair'], 'xbyf', k['vv_xbyf']['ms_per_pair'])
PY
done
