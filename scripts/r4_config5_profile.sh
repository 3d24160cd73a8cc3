#!/bin/bash
# Config 5 on ONE GPU (one 16384x16384x3 f32 pair -> 24576x16384 canvas): per-kernel times (rocprofv3 --kernel-trace --stats) and HBM
# traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes) of `bench.py --frame 16384 --pairs-per-step 1 --batch 1 --streams 1`
# -> gpurun_out/c5/: kernel_stats.csv, pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv, bench.json.  scripts/r4_config5_report.py turns them
# into profiles/r04_config5_*.  (The program comes directly after `--`.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c5; rm -rf $O; mkdir -p $O
ARGS="--frame 16384 --pairs-per-step 1 --batch 1 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-single --no-verify"
timeout -k 10 500 python3 bench.py $ARGS > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py $ARGS --no-kernel-events > $O/bench_under_rocprof.json 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python3 bench.py $ARGS --no-kernel-events > $O/pmc_$C.json 2> $O/pmc_$C.err
  python - $O/pmc_$C $C > $O/pmc_$C.csv <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"]; k = k[: k.index("(")] if "(" in k else k
    k = k.replace("void ", "").replace("sk::", "")
    tot[k] += float(r["Counter_Value"]); cnt[k] += 1
print("kernel,counter,dispatches,sum_KiB,mean_KiB_per_dispatch")
for k in sorted(tot): print(f'"{k}",{sys.argv[2]},{cnt[k]},{tot[k]:.1f},{tot[k]/cnt[k]:.1f}')
PY
  rm -rf $O/pmc_$C
done
rm -rf $O/trace
head -40 $O/kernel_stats.csv
