#!/bin/bash
# does the bench's output check fail now and then?  N short runs per library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4aa; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  for v in new prev; do
    if [ $v = new ]; then unset STITCH_LIB; else export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_prev.so; fi
    timeout -k 10 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-single --no-kernel-events > $O/b_${v}_$i.json 2> $O/b_${v}_$i.err; rc=$?
    echo "$v run $i rc=$rc $(grep -c 'CHECK FAILED' $O/b_${v}_$i.err) failures; $(grep -m1 'timed out' $O/b_${v}_$i.err | cut -c1-160)"
  done
done
