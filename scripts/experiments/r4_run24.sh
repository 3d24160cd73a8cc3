#!/bin/bash
# the collapse with every read of the four-column path removed (8) and with the one-column part switched off as well (9)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4y; rm -rf $O; mkdir -p $O
run() {  # name, counter, env...
  local name=$1 ctr=$2; shift; shift
  ( export "$@"; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -- python3 scripts/experiments/exp_collapse_fetch.py aligned > $O/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $O/$name.log; }
  echo "== $name ($ctr $*)" | tee -a $O/report.txt
  python scripts/experiments/fetch_report.py $O/$name 8 "k_collapse4<float, true" "k_collapse4<float, false, false, false>" | grep -v "grid    1[23]" | tee -a $O/report.txt
  rm -rf $O/$name
}
for a in 8 9; do run ablate$a FETCH_SIZE STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_c4abl$a.so; done
