#!/bin/bash
# per-dispatch durations of one launch sequence for several builds, same box: scripts/experiments/ab_trace.sh <outdir> <filter> <variant>...
# variant = default | old (STITCH_COLLAPSE4=0) | <name> (csrc/ab/libstitch_<name>.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=$1; flt=$2; shift; shift
mkdir -p $out
for v in "$@"; do
  unset STITCH_LIB STITCH_COLLAPSE4
  if [ "$v" = old ]; then export STITCH_COLLAPSE4=0; elif [ "$v" != default ]; then export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_$v.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $out/$v -- python3 scripts/experiments/exp_seq.py ${AB_PAIRS:-8} 4 ${AB_FRAME:-4096} ${AB_PIXEL:-f32} > $out/$v.log 2>&1 || echo "variant $v failed"
  echo "== $v"; python scripts/trace_levels.py $out/$v "$flt" | tail -${AB_TAIL:-6}
done
