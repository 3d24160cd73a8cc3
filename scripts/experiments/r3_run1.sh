#!/bin/bash
# round-3 measurement batch 1: tests, bench lines (default, one rank's share of 8 GPUs with and without coalescing), reference loop, timelines
cd $GRAFT_REPO_ROOT
o=gpurun_out/r3run1; mkdir -p $o
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; tail -3 $o/tests.log; fi
python bench.py --steps 20 --warmup 3 > $o/bench.json 2> $o/bench.err; tail -2 $o/bench.err
python bench.py --steps 20 --warmup 3 --pairs-per-step 4 --no-cpu-baseline > $o/bench_p4.json 2> $o/bench_p4.err; tail -2 $o/bench_p4.err
python bench.py --steps 20 --warmup 3 --pairs-per-step 4 --no-cpu-baseline --no-coalesce > $o/bench_p4_nocoalesce.json 2> $o/bench_p4_nocoalesce.err
STITCH_FORCE_DIST=1 python bench.py --steps 12 --warmup 3 --pairs-per-step 4 --no-cpu-baseline > $o/bench_p4_forcedist.json 2> $o/bench_p4_forcedist.err; tail -2 $o/bench_p4_forcedist.err
for f in bench bench_p4 bench_p4_nocoalesce bench_p4_forcedist; do python - $o/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]
    print(sys.argv[1].split("/")[-1], "value", d["value"], "verified", d["outputs_verified"], "single_ms", c.get("single_pair_in_flight_ms"), "one_seq", c.get("one_sequence_in_flight_ms_per_pair"), "pairs/seq", c.get("pairs_per_sequence"), "roofline", d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"), "pipe", d["pipeline"]["frac_of_hbm_peak"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
python scripts/bench_reference_config2.py > $o/reference_config2.json 2> $o/reference_config2.err; tail -3 $o/reference_config2.err; grep -A4 mi355x $o/reference_config2.json
scripts/experiments/tl_single.sh $o/tl
