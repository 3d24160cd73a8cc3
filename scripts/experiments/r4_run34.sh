#!/bin/bash
# planes 24 MiB apart (level 1 of a 6144 x 4096 canvas): do the streams of a workgroup evict each other?  the same launch on canvases of
# 4096 / 4112 / 4160 / 4224 rows (plane strides of 24 MiB / 24.09 / 24.4 / 24.75)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ai; rm -rf $O; mkdir -p $O
for h in 4096 4112 4160 4224; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/h$h -- python3 scripts/experiments/exp_collapse_fetch.py default $h > $O/h$h.log 2>&1 || { echo "$h failed"; tail -5 $O/h$h.log; }
  echo "== canvas 6144 x $h" | tee -a $O/report.txt
  python scripts/experiments/fetch_report.py $O/h$h 8 "k_collapse4" "k_vv_x_fwd<float, true" "k_vv_xbyf" "k_vv_y_bwd_dec<" | tee -a $O/report.txt
  rm -rf $O/h$h
done
