#!/bin/bash
# what the collapse fetches, stream by stream: ablated builds (csrc/ab/libstitch_c4abl{1,2,3}.so: no coarser level / no index plane /
# no mosaic) under the aligned map, then block orders and strip heights of the product build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4u; rm -rf $O; mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  ( export "$@"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$name -- python3 scripts/experiments/exp_collapse_fetch.py aligned > $O/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $O/$name.log; }
  echo "== $name ($*)" | tee -a $O/report.txt
  python scripts/experiments/fetch_report.py $O/$name 8 "k_collapse4<float, true" "k_collapse4<float, false" | tee -a $O/report.txt
  rm -rf $O/$name
}
run product X=0
for a in 1 2 3; do run ablate$a STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_c4abl$a.so; done
run swizzle0 STITCH_C4_SWIZZLE=0
run swizzle1 STITCH_C4_SWIZZLE=1
run crows16 STITCH_CROWS_L0=16 STITCH_CROWS_LN=16
run crows64 STITCH_CROWS_L0=64 STITCH_CROWS_LN=64
run crows128 STITCH_CROWS_L0=128 STITCH_CROWS_LN=128
