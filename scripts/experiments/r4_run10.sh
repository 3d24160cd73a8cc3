#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
for v in "STITCH_Y1N=20" "STITCH_Y1N=21" "STITCH_Y1N=22" "STITCH_Y1N=23"; do
for c in "4421 2315 1536 2048"; do
  set -- $c
  n=$(echo "$v" | tr -c 'A-Za-z0-9=' '_')
  ( export $v; rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1_$n -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1_$n.log 2>&1 )
  python scripts/experiments/timeline.py $O/tl_$1_$n > $O/tl_$1_$n.txt; echo "[$v]"; grep "dec7" $O/tl_$1_$n.txt | head -4; rm -rf $O/tl_$1_$n
done
done
