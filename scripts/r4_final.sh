#!/bin/bash
# round-4 evidence batch: everything profiles/r04_* is made from, in one gpurun call (boxes differ by up to 8 %)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r4final; rm -rf $o; mkdir -p $o
# 1. kernel trace + stats of the bench with one sequence in flight, then the plain default line (same box)
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --no-cpu-baseline --no-single --no-verify --streams 1 > $o/bench_under_rocprof.json 2> $o/bench_under_rocprof.err
cp $(find $o/prof -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv; rm -rf $o/prof
python bench.py --steps 20 --warmup 5 --verbose > $o/bench.json 2> $o/bench.err; tail -3 $o/bench.err
# 2. one rank's share of the 8-GPU run (4 pairs per step), with the RCCL path forced on the one rank; unsigned char frames; A/B of the collapse's block order
python bench.py --steps 20 --warmup 5 --pairs-per-step 4 --no-cpu-baseline > $o/bench_p4.json 2> $o/bench_p4.err
STITCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --pairs-per-step 4 --no-cpu-baseline > $o/bench_p4_forcedist.json 2> $o/bench_p4_forcedist.err
python bench.py --steps 20 --warmup 5 --pixel u8 --no-cpu-baseline > $o/bench_u8.json 2> $o/bench_u8.err
STITCH_C4_SWIZZLE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single > $o/bench_noswizzle.json 2> $o/bench_noswizzle.err
STITCH_BENCH_BACKEND=gloo python bench.py --gpus 2 --pairs-per-step 8 --steps 10 --warmup 2 --no-cpu-baseline > $o/bench_2rank_rehearsal.json 2> $o/bench_2rank_rehearsal.err
for f in bench bench_p4 bench_p4_forcedist bench_u8 bench_noswizzle bench_2rank_rehearsal; do python - $o/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]
    print(sys.argv[1].split("/")[-1], "value", d["value"], "verified", d["outputs_verified"], "ranks", d["n_ranks_seen"], "single_ms", c.get("single_pair_in_flight_ms"), "single_batch_ms", c.get("single_batch_ms"), "one_seq", c.get("one_sequence_in_flight_ms_per_pair"), "roofline", d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"), "pipe", d["pipeline"]["frac_of_hbm_peak"], {k: v["ms_per_pair"] for k, v in d.get("kernels", {}).items() if k.startswith("collapse")})
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
# 3. real canvases, other rows, drop-in config 3
python scripts/bench_realcanvas.py > $o/real_canvases.json 2> $o/real_canvases.err; tail -2 $o/real_canvases.err
python scripts/bench_stages.py > $o/other_rows.json 2> $o/other_rows.err; tail -2 $o/other_rows.err
python scripts/bench_dropin.py > $o/dropin_config3.json 2> $o/dropin_config3.err; tail -2 $o/dropin_config3.err
# 4. single-pair timelines (defaults, and the round-3 kernels: STITCH_MOVER=0 STITCH_DEC7=0 STITCH_Y1S=0)
for v in "" "STITCH_MOVER=0 STITCH_DEC7=0 STITCH_Y1S=0"; do
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  set -- $c
  n=$([ -z "$v" ] && echo r4 || echo r3kernels)
  ( [ -n "$v" ] && export $v; rocprofv3 --kernel-trace --output-format csv -d $o/tl_$1_$n -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $o/tl_$1_$n.log 2>&1 )
  python scripts/experiments/timeline.py $o/tl_$1_$n > $o/tl_$1_$n.txt; tail -1 $o/tl_$1_$n.log; tail -1 $o/tl_$1_$n.txt; rm -rf $o/tl_$1_$n
done
done
ls $o
