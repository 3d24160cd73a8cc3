#!/bin/bash
# collapse: fixed-pattern code for the interior blocks of a per-lane-tap launch (hybrid, default) against per-lane taps everywhere (STITCH_C4_HYBRID=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ap; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchpath.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
AB_ARGS="--no-single" bash scripts/experiments/ab_env.sh $O "gen:STITCH_C4_HYBRID=0" "hyb:STITCH_X=0" "gen2:STITCH_C4_HYBRID=0" "hyb2:STITCH_X=0" "gen3:STITCH_C4_HYBRID=0" "hyb3:STITCH_X=0"
for c in "6144 4096 4096 4096" "4421 2315 2048 2048"; do timeout -k 10 200 python scripts/experiments/exp_env_ab.py STITCH_C4_HYBRID 0 - $c 40 2>&1 | grep "float32"; done
