#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ag; rm -rf $O; mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  ( export "$@"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$name -- python3 scripts/experiments/exp_collapse_fetch.py default > $O/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $O/$name.log; }
  echo "== $name ($*)" | tee -a $O/report.txt
  python scripts/experiments/fetch_report.py $O/$name 8 "k_collapse4<float, false" | tee -a $O/report.txt
  rm -rf $O/$name
}
run lockstep STITCH_C4_LOCKSTEP=1
run lockstep_gen0 STITCH_C4_LOCKSTEP=1 STITCH_C4_GEN=0
run crows16 STITCH_CROWS_LN=16
run crows8 STITCH_CROWS_LN=8
