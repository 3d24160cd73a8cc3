#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command with one launch sequence in flight (--streams 1: a launch's duration is the
# kernel's own), minus the CPU leg, the single-pair tail and the output check (--no-verify: its reference runs go through a
# single-pair plan and would add launches of other shapes to the per-symbol averages); then the plain bench line on the same box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --no-cpu-baseline --no-single --no-verify --streams 1 > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/bench_under_rocprof.err
python bench.py --verbose > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err
tail -2 gpurun_out/prof/bench.err
