#!/bin/bash
# Same-box A/B of builds of the same ABI: the stage benchmark of the projection and the headline bench, default build first.
# usage (on the GPU box): scripts/experiments/ab_lib.sh <name under csrc/ab/ without lib prefix> ...
cd "$(dirname "$0")/../.."
AB=$PWD/computervisionimagestich2_amd/csrc/ab
for lib in default "$@"; do
    if [ "$lib" = default ]; then unset STITCH_LIB; else export STITCH_LIB=$AB/libstitch_$lib.so; fi
    python scripts/experiments/exp_project.py
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   bench', d['value'], d['unit'], 'frac', d['roofline']['frac'])"
done
