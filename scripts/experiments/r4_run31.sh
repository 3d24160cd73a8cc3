#!/bin/bash
# levels >= 1 of the collapse after the change: which reads grew?  product / no coarser level (ablate1) / fixed pattern + one-column edges (STITCH_C4_GEN=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4af; rm -rf $O; mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  ( export "$@"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$name -- python3 scripts/experiments/exp_collapse_fetch.py default > $O/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $O/$name.log; }
  echo "== $name ($*)" | tee -a $O/report.txt
  python scripts/experiments/fetch_report.py $O/$name 8 "k_collapse4" | tee -a $O/report.txt
  rm -rf $O/$name
}
run product X=0
run ablate1 STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_c4abl1.so
run gen0 STITCH_C4_GEN=0
run gen0_ablate1 STITCH_C4_GEN=0 STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_c4abl1.so
run crows64 STITCH_CROWS_LN=64
