#!/bin/bash
# round-3 evidence batch: everything profiles/r03_* is made from, in one gpurun call (boxes differ by up to 8 %)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r3final; rm -rf $o; mkdir -p $o
# 1. kernel trace + stats of the bench with one sequence in flight, then the plain default line (same box)
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --no-cpu-baseline --no-single --no-verify --streams 1 > $o/bench_under_rocprof.json 2> $o/bench_under_rocprof.err
python bench.py --steps 20 --warmup 5 --verbose > $o/bench.json 2> $o/bench.err; tail -3 $o/bench.err
# 2. one rank's share of the 8-GPU run (4 pairs per step): coalesced, not coalesced, with the RCCL path forced on the one rank
python bench.py --steps 20 --warmup 5 --pairs-per-step 4 --no-cpu-baseline > $o/bench_p4.json 2> $o/bench_p4.err
python bench.py --steps 20 --warmup 5 --pairs-per-step 4 --no-cpu-baseline --no-coalesce > $o/bench_p4_nocoalesce.json 2> $o/bench_p4_nocoalesce.err
STITCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --pairs-per-step 4 --no-cpu-baseline > $o/bench_p4_forcedist.json 2> $o/bench_p4_forcedist.err
python bench.py --steps 20 --warmup 5 --pixel u8 --no-cpu-baseline > $o/bench_u8.json 2> $o/bench_u8.err
for f in bench bench_p4 bench_p4_nocoalesce bench_p4_forcedist bench_u8; do python - $o/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]
    print(sys.argv[1].split("/")[-1], "value", d["value"], "verified", d["outputs_verified"], "single_ms", c.get("single_pair_in_flight_ms"), "one_seq", c.get("one_sequence_in_flight_ms_per_pair"), "pairs/seq", c.get("pairs_per_sequence"), "roofline", d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"), "pipe", d["pipeline"]["frac_of_hbm_peak"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
# 3. real canvases, other rows, config 5 bands, drop-in config 3
python scripts/bench_realcanvas.py > $o/real_canvases.json 2> $o/real_canvases.err; tail -2 $o/real_canvases.err
python scripts/bench_stages.py > $o/other_rows.json 2> $o/other_rows.err; tail -2 $o/other_rows.err
python scripts/bench_band.py > $o/config5_band.json 2> $o/config5_band.err; tail -2 $o/config5_band.err
python scripts/bench_dropin.py > $o/dropin_config3.json 2> $o/dropin_config3.err; tail -2 $o/dropin_config3.err
# 4. single-pair timelines
scripts/experiments/tl_single.sh $o/tl 2>&1 | grep -v "tool finalization"
ls $o
