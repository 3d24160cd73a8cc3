#!/bin/bash
# rows in flight in the y sweeps' registers (YST stages of 8 rows): 4 (default, 127 VGPRs in the decimating sweep, 4 wavefronts per SIMD), 3, 2 (88 VGPRs, 5 per SIMD)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default yst3 yst2 default yst2 yst3; do
  if [ $v = default ]; then unset STITCH_LIB; else export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_$v.so; fi
  timeout -k 10 240 python bench.py --steps 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; k=d['kernels']
print('$v', d['value'], d['outputs_verified'], 'one-seq', c['one_sequence_in_flight_ms_per_pair'], 'single', c['single_pair_in_flight_ms'], ' '.join(n+' %.4f' % k[n]['ms_per_pair'] for n in ('vv_y_bwd','vv_y_fwd','vv_xbyf')))"
done
