#!/bin/bash
# collapse after the change (per-lane taps on every block, strips padded to 8 blocks): parity, FETCH_SIZE per level under block orders 0/1/2, bench A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ab; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchpath.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for v in 0 1 2; do
  ( export STITCH_C4_SWIZZLE=$v; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$v -- python3 scripts/experiments/exp_collapse_fetch.py default > $O/f$v.log 2>&1 )
  echo "== STITCH_C4_SWIZZLE=$v"; python scripts/experiments/fetch_report.py $O/f$v 8 "k_collapse4"; rm -rf $O/f$v
done
AB_ARGS="--no-single" bash scripts/experiments/ab_env.sh $O "swz1:STITCH_C4_SWIZZLE=1" "swz2:STITCH_C4_SWIZZLE=2" "swz0:STITCH_C4_SWIZZLE=0" "swz1b:STITCH_C4_SWIZZLE=1" "swz2b:STITCH_C4_SWIZZLE=2"
