#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command (same command as the BENCH line, minus the CPU leg)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --no-cpu-baseline --no-single --streams 1 > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/bench_under_rocprof.err
python bench.py --verbose > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
tail -2 gpurun_out/prof/bench.err
