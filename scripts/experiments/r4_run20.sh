#!/bin/bash
# the level-0 kernels' FETCH_SIZE under four backward maps (aligned / three samples off / slanted / the bench's)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4t; rm -rf $O; mkdir -p $O
for m in aligned shift3 slant default; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$m -- python3 scripts/experiments/exp_collapse_fetch.py $m > $O/$m.log 2>&1 || { echo "$m failed"; tail -5 $O/$m.log; }
  echo "== $m: $(tail -1 $O/$m.log)"
  python scripts/experiments/fetch_report.py $O/$m 8 "k_collapse4<float, true" "k_vv_x_fwd<float, true" "k_collapse4<float, false" | tee -a $O/report.txt
  rm -rf $O/$m
done
