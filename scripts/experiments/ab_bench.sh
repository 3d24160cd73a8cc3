#!/bin/bash
# A/B of builds of the same ABI inside ONE gpurun call (boxes differ by up to 8 %): scripts/experiments/ab_bench.sh <outdir> <lib|default>...
# each variant: bench.py --steps 10 --no-cpu-baseline with STITCH_LIB pointing at the build; one summary line per variant.
out=$1; shift
mkdir -p $out
for v in "$@"; do
  if [ "$v" = default ]; then unset STITCH_LIB; else export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_$v.so; fi
  timeout -k 10 240 python bench.py --steps 10 --no-cpu-baseline $AB_ARGS > $out/bench_$v.json 2> $out/bench_$v.err
  echo "variant $v rc=$?"
  python - $out/bench_$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d["config"]; k = d["kernels"]
    print(f"  {sys.argv[2]:10s} {d['value']:9.1f} MPix/s  ms/pair {c['ms_per_pair_per_gpu']:.4f}  one-seq {c['one_sequence_in_flight_ms_per_pair']:.4f}  verified {d['outputs_verified']}  "
          + "  ".join(f"{n} {k[n]['ms_per_pair']:.4f}" for n in ("collapse_l0", "collapse", "vv_xbyf", "vv_x_fwd_src", "vv_x_fwd", "vv_y_bwd")))
except Exception as e:
    print("  no result:", e)
PY
done
