#!/bin/bash
# config 5: persistent workgroups of the fused sweep (fewer than the 1792 bands: shallower pipeline per plane), a third fused level
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "X=0" "STITCH_XBYF_WGS=1024" "STITCH_XBYF_WGS=1280" "STITCH_XBYF_WGS=1536" "STITCH_WAVEFRONT=3" "STITCH_XBYF_WGS=768" "X=0"; do
  ( export $v; timeout -k 10 300 python bench.py --frame 16384 --pairs-per-step 1 --batch 1 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-single 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; k=d['kernels']
print('$v ms/pair', c['ms_per_pair_per_gpu'], 'verified', d['outputs_verified'], ' '.join(f\"{n} {k[n]['ms_per_pair']:.3f}\" for n in ('vv_xbyf','vv_y_bwd','vv_x_fwd_src','vv_x_bwd','vv_y_fwd')))
" )
done
