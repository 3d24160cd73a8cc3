#!/bin/bash
# collapse: per-lane taps (GEN) wherever they bring a block of columns off the one-column path (new default) against the old rule
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4z; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchpath.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for c in "6144 4096 4096 4096" "4421 2315 2048 2048" "1081 527 384 512"; do
  timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_C4_GEN 1 - $c 30 2>&1 | grep -v amdgpu.ids | grep -v uint8
done
AB_ARGS="--no-single" bash scripts/experiments/ab_env.sh $O "old:STITCH_C4_GEN=1" "new:STITCH_X=0" "old2:STITCH_C4_GEN=1" "new2:STITCH_X=0"
for v in 1 2; do
  ( export STITCH_C4_GEN=$v; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$v -- python3 scripts/experiments/exp_collapse_fetch.py default > $O/f$v.log 2>&1 )
  echo "== STITCH_C4_GEN=$v"; python scripts/experiments/fetch_report.py $O/f$v 8 "k_collapse4"; rm -rf $O/f$v
done
