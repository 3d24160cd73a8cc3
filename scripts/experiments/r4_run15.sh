#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; rm -rf $O; mkdir -p $O
for c in "1081 527 384 512" "333 211 150 200" "2000 1100 900 1000" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  timeout -k 10 120 python scripts/experiments/exp_env_ab.py STITCH_COARSE_LDS 0 - $c 20 2>&1 | grep -v amdgpu.ids; rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && { echo "A/B failed rc=$rc"; exit $rc; }
done
timeout -k 10 120 python scripts/experiments/exp_env_ab.py STITCH_XBYM 0 - 6144 4096 4096 4096 10 2>&1 | grep -v amdgpu.ids
for seed in 601 602; do
FUZZ_GENERAL=1 timeout -k 10 400 python scripts/fuzz_pairs.py $seed 150 > $O/fuzz_$seed.log 2>&1; rc=$?; tail -1 $O/fuzz_$seed.log; [ $rc -ne 0 ] && { grep -m5 MISMATCH $O/fuzz_$seed.log; exit $rc; }
done
for c in "1081 527 384 512" "6144 4096 4096 4096"; do
  set -- $c
  rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1.log 2>&1
  python scripts/experiments/timeline.py $O/tl_$1 > $O/tl_$1.txt; tail -1 $O/tl_$1.log; grep -E "xby_m|k_coarse|dispatches" $O/tl_$1.txt | head -8; rm -rf $O/tl_$1
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && tail -60 $O/pytest.log
exit $rc
