#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; rm -rf $O; mkdir -p $O
for c in "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  timeout -k 10 120 python scripts/experiments/exp_xbym_ab.py $c 10 2>&1 | grep -v amdgpu.ids; rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && { echo "A/B failed rc=$rc"; exit $rc; }
done
for c in "3072 2048 2048 2048" "6144 4096 4096 4096"; do
  set -- $c
  STITCH_XBYM_STAMP=1 STITCH_XBYM=1 timeout -k 10 120 python scripts/experiments/exp_single.py $c 5 pair f32 2>&1 | grep -v amdgpu.ids
  STITCH_XBYM=1 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1.log 2>&1
  python scripts/experiments/timeline.py $O/tl_$1 > $O/tl_$1.txt; tail -1 $O/tl_$1.log; grep -E "xby_m|dispatches" $O/tl_$1.txt | head -8; rm -rf $O/tl_$1
done
