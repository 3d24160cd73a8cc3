#!/bin/bash
# kernel trace of the default four-lane bench (no stats, no per-launch events): input of scripts/experiments/lane_overlap.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/lanes && mkdir -p gpurun_out/lanes
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lanes -- python3 bench.py --no-cpu-baseline --no-single --no-kernel-events --steps 10 --warmup 3 > gpurun_out/lanes/bench.json 2> gpurun_out/lanes/bench.err
ls -la gpurun_out/lanes/*/ | head
