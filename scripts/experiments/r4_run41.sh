#!/bin/bash
# k_vv_xby_m: tile buffers per band and workgroups per CU (4 buffers / 512 workgroups = two per CU; 2 buffers / 768 = three per CU), config 5 and one 6144 x 4096 pair
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default xy2 xy3 default xy2; do
  if [ $v = default ]; then unset STITCH_LIB; else export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_$v.so; fi
  timeout -k 10 300 python bench.py --frame 16384 --pairs-per-step 1 --batch 1 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-single 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; k=d['kernels']
print('$v config5 ms/pair', c['ms_per_pair_per_gpu'], 'verified', d['outputs_verified'], 'xby', round(k['vv_xbyf']['ms_per_pair'],3))"
  echo "$v: $(timeout -k 10 120 python scripts/experiments/exp_single.py 6144 4096 4096 4096 40 pair f32 2>&1 | tail -1 | cut -c1-60)"
done
