#!/bin/bash
# round-3 measurement batch 2: tests, bench lines, timelines (after the LDS-tiled y sweeps)
cd $GRAFT_REPO_ROOT
o=gpurun_out/r3run2; mkdir -p $o
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; tail -3 $o/tests.log; fi
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $o/bench.json 2> $o/bench.err; tail -2 $o/bench.err
python bench.py --steps 20 --warmup 3 --pairs-per-step 4 --no-cpu-baseline > $o/bench_p4.json 2> $o/bench_p4.err; tail -2 $o/bench_p4.err
for f in bench bench_p4; do python - $o/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]
    print(sys.argv[1].split("/")[-1], "value", d["value"], "verified", d["outputs_verified"], "single_ms", c.get("single_pair_in_flight_ms"), "one_seq", c.get("one_sequence_in_flight_ms_per_pair"), "pairs/seq", c.get("pairs_per_sequence"), "roofline", d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"), "pipe", d["pipeline"]["frac_of_hbm_peak"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
scripts/experiments/tl_single.sh $o/tl
STITCH_YTILE=0 scripts/experiments/tl_single.sh $o/tl_noytile
