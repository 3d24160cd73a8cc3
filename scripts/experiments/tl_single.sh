#!/bin/bash
# kernel timelines of a single pair in flight at several canvases, with and without STITCH_GATE64 (scripts/experiments/timeline.py)
# usage: scripts/experiments/tl_single.sh <outdir> [extra env assignments...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; shift
mkdir -p $out
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  set -- $c
  for g in 0 1; do
    export STITCH_GATE64=$g
    rocprofv3 --kernel-trace --output-format csv -d $out/$1_g$g -- python3 scripts/experiments/exp_single.py $c 5 ${TL_MODE:-pair} ${TL_PIXEL:-f32} > $out/$1_g$g.log 2>&1
    python scripts/experiments/timeline.py $out/$1_g$g > $out/$1_g$g.timeline.txt
    tail -1 $out/$1_g$g.log; tail -1 $out/$1_g$g.timeline.txt
    rm -rf $out/$1_g$g
  done
done
