#!/bin/bash
# is the index plane fetched more than once?  builds where only channel 0 reads it (6) / nobody does (7), same gathers (aligned map)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x; rm -rf $O; mkdir -p $O
run() {  # name, counter, env...
  local name=$1 ctr=$2; shift; shift
  ( export "$@"; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -- python3 scripts/experiments/exp_collapse_fetch.py aligned > $O/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $O/$name.log; }
  echo "== $name ($ctr $*)" | tee -a $O/report.txt
  python scripts/experiments/fetch_report.py $O/$name 8 "k_collapse4<float, true" | tee -a $O/report.txt
  rm -rf $O/$name
}
run product FETCH_SIZE X=0
for a in 6 7; do run ablate$a FETCH_SIZE STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_c4abl$a.so; done
run lockstep FETCH_SIZE STITCH_C4_LOCKSTEP=1
