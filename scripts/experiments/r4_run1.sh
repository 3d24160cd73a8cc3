#!/bin/bash
# round 4, GPU call 1: the one-column y sweeps (k_sweeps1.inc) and the collapse's lockstep / XCD swizzle, A/B inside one call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { echo "pytest failed rc=$rc"; exit $rc; }
echo "== single pair, ms per call (base = round-3 kernels)"
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  for v in "STITCH_Y1S=0 STITCH_DEC5=0" "STITCH_DEC5=0" "STITCH_Y1S=0" "" "STITCH_Y1S=2"; do
    echo -n "[$v] "; env $v timeout -k 10 120 python scripts/experiments/exp_single.py $c 20 pair f32 2>&1 | tail -1
  done
done | tee $O/single.txt
echo "== timelines (defaults)"
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  set -- $c
  rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1.log 2>&1
  python scripts/experiments/timeline.py $O/tl_$1 > $O/tl_$1.txt; tail -1 $O/tl_$1.txt; rm -rf $O/tl_$1
done
echo "== bench A/B: collapse lockstep / swizzle"
for v in "" "STITCH_C4_LOCKSTEP=0" "STITCH_C4_SWIZZLE=0" "STITCH_C4_LOCKSTEP=0 STITCH_C4_SWIZZLE=0"; do
  n=$(echo "x$v" | tr -c 'A-Za-z0-9=' '_')
  env $v timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline --no-single > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]; k = d["kernels"]
    print(f"  [{sys.argv[2]:45s}] {d['value']:9.1f} MPix/s one-seq {c['one_sequence_in_flight_ms_per_pair']:.4f} verified {d['outputs_verified']} "
          + " ".join(f"{n} {k[n]['ms_per_pair']:.4f}" for n in ("collapse_l0", "collapse", "vv_xbyf", "vv_x_fwd_src", "vv_y_bwd")))
except Exception as e:
    print("  no result:", e)
PY
done | tee $O/ab.txt
echo "== FETCH_SIZE / WRITE_SIZE of the collapse kernels"
for v in "" "STITCH_C4_LOCKSTEP=0 STITCH_C4_SWIZZLE=0" "STITCH_C4_SWIZZLE=0"; do
  n=$(echo "x$v" | tr -c 'A-Za-z0-9=' '_')
  for C in FETCH_SIZE WRITE_SIZE; do
    ( export $v _x=1; rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${n}_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-single --no-verify --streams 1 --batch 16 > $O/pmc_${n}_$C.json 2> $O/pmc_${n}_$C.err )
    python - $O/pmc_${n}_$C $C "$v" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"]
    if "collapse" not in k: continue
    k = k[k.index("k_collapse"):k.index("(")] if "(" in k else k
    tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(tot): print(f"  [{sys.argv[3]:45s}] {sys.argv[2]} {k:45s} {cnt[k]:4d} dispatches, KiB per dispatch {tot[k]/cnt[k]:12.1f}")
PY
    rm -rf $O/pmc_${n}_$C
  done
done | tee $O/pmc.txt
