#!/bin/bash
# round 4, GPU call 8: lone pair with a source-fused level 0 (gathers on the loader wavefront)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_benchpath.py -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed rc=$rc"; tail -60 $O/pytest.log; exit $rc; }
echo "== single pair, ms per call"
for m in pair blend; do
for px in f32 u8; do
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  for v in "STITCH_MOVER=0" "STITCH_SRC_LONE_MPIX=0" "STITCH_SRC_LONE_MPIX=8"; do
    echo -n "[$v] "; env $v timeout -k 10 120 python scripts/experiments/exp_single.py $c 20 $m $px 2>&1 | tail -1 | sed 's/; paths.*//'
  done
done
done
done | tee $O/single.txt
echo "== timelines"
for c in "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  set -- $c
  rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1.log 2>&1
  python scripts/experiments/timeline.py $O/tl_$1 > $O/tl_$1.txt; head -9 $O/tl_$1.txt; tail -3 $O/tl_$1.txt; rm -rf $O/tl_$1
done
