#!/bin/bash
# round 4, GPU call 6: which side bounds k_vv_x_m -- chain only (STITCH_Y1N=11), mover only (12), both (10)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
for v in "STITCH_Y1N=10" "STITCH_Y1N=11" "STITCH_Y1N=12"; do
for c in "4421 2315 1536 2048"; do
  set -- $c
  n=$(echo "$v" | tr -c 'A-Za-z0-9=' '_')
  ( export $v; rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1_$n -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1_$n.log 2>&1 )
  python scripts/experiments/timeline.py $O/tl_$1_$n > $O/tl_$1_$n.txt; echo "[$v]"; grep "x_m" $O/tl_$1_$n.txt | head -6; rm -rf $O/tl_$1_$n
done
done
