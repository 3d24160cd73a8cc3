#!/bin/bash
# levels >= 1: is the mask plane fetched once per channel wavefront?  only channel 0 reads it (11) / nobody (12)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ah; rm -rf $O; mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  ( export "$@"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$name -- python3 scripts/experiments/exp_collapse_fetch.py default > $O/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $O/$name.log; }
  echo "== $name ($*)" | tee -a $O/report.txt
  python scripts/experiments/fetch_report.py $O/$name 8 "k_collapse4<float, false" | tee -a $O/report.txt
  rm -rf $O/$name
}
run product X=0
for a in 11 12; do run ablate$a STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_c4abl$a.so; done
