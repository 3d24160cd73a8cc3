#!/bin/bash
# the batch's anticausal y sweep + decimation with the short divide (fastdiv) in its consumer: parity, bench A/B against the previous build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4ao; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchpath.py tests/test_gpu_band.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for v in prev new prev new; do
  if [ $v = new ]; then unset STITCH_LIB; else export STITCH_LIB=$PWD/computervisionimagestich2_amd/csrc/ab/libstitch_prev.so; fi
  timeout -k 10 240 python bench.py --steps 10 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]; k = d["kernels"]
print(f"  {sys.argv[2]:6s} {d['value']:9.1f} MPix/s  ms/pair {c['ms_per_pair_per_gpu']:.4f}  one-seq {c['one_sequence_in_flight_ms_per_pair']:.4f} single {c['single_pair_in_flight_ms']}  verified {d['outputs_verified']}  " + "  ".join(f"{n} {k[n]['ms_per_pair']:.4f}" for n in ("vv_y_bwd", "collapse_l0", "vv_xbyf", "vv_x_fwd_src")))
PY
done
