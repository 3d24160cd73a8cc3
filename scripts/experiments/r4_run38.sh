#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4am; rm -rf $O; mkdir -p $O
for c in "6144 4096 4096 4096" "4421 2315 2048 2048" "1081 527 384 512"; do
  set -- $c
  rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1.log 2>&1
  python scripts/experiments/timeline.py $O/tl_$1 > $O/tl_$1.txt; tail -1 $O/tl_$1.log; tail -1 $O/tl_$1.txt; rm -rf $O/tl_$1
done
