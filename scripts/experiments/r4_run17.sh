#!/bin/bash
# dec7: does the loader's depth (chunks in flight) set the rows per second?  A/B builds under csrc/ab/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4q; rm -rf $O; mkdir -p $O
for v in default d7_8_3 d7_12_8 d7_16_10; do
  if [ "$v" = default ]; then unset STITCH_LIB; else export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_$v.so; fi
  for c in "6144 4096 4096 4096" "1081 527 384 512"; do
    set -- $c
    rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1_$v -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1_$v.log 2>&1
    python scripts/experiments/timeline.py $O/tl_$1_$v > $O/tl_$1_$v.txt; echo "$v $1: $(tail -1 $O/tl_$1_$v.log)"; grep -E "dec7" $O/tl_$1_$v.txt | awk '{printf "%s ", $5}'; echo; tail -1 $O/tl_$1_$v.txt; rm -rf $O/tl_$1_$v
  done
done
