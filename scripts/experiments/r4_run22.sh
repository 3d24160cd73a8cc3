#!/bin/bash
# the collapse's stores and its FETCH_SIZE: no stores at all / plain instead of non-temporal stores (csrc/ab/libstitch_c4abl{4,5}.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4w; rm -rf $O; mkdir -p $O
run() {  # name, counter, env...
  local name=$1 ctr=$2; shift; shift
  ( export "$@"; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -- python3 scripts/experiments/exp_collapse_fetch.py aligned > $O/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $O/$name.log; }
  echo "== $name ($ctr $*)" | tee -a $O/report.txt
  python scripts/experiments/fetch_report.py $O/$name 8 "k_collapse4<float, true" "k_collapse4<float, false, false, false>" | grep -v "grid    1[23]" | tee -a $O/report.txt
  rm -rf $O/$name
}
run product FETCH_SIZE X=0
for a in 4 5; do run ablate$a FETCH_SIZE STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_c4abl$a.so; done
run product_w WRITE_SIZE X=0
run ablate5_w WRITE_SIZE STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_c4abl5.so
run product_req "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" X=0
