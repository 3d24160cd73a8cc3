#!/bin/bash
# round 4, GPU call 5: movers with the load wait ahead of the stores; request-size counters of the collapse
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
echo "== single pair, ms per call"
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  for v in "STITCH_MOVER=0" "STITCH_MOVER=1" "STITCH_MOVER=2"; do
    echo -n "[$v] "; env $v timeout -k 10 120 python scripts/experiments/exp_single.py $c 20 pair f32 2>&1 | tail -1 | sed 's/; paths.*//'
  done
done | tee $O/single.txt
echo "== timelines"
for v in "STITCH_MOVER=1"; do
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  set -- $c
  ( export $v; rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1 -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1.log 2>&1 )
  python scripts/experiments/timeline.py $O/tl_$1 > $O/tl_$1.txt; tail -1 $O/tl_$1.txt; rm -rf $O/tl_$1
done
done
echo "== request sizes of the collapse kernels (TCC_EA0_RDREQ by size)"
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  n=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-single --no-verify --streams 1 --batch 16 > $O/pmc_$n.json 2> $O/pmc_$n.err
  python - $O/pmc_$n <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "collapse4" not in k and "k_vv_x_bwd" not in k: continue
    k = k[k.index("k_"):k.index("(")] if "(" in k else k
    tot[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(tot): print(f"  {k[0]:45s} {k[1]:28s} {cnt[k]:4d} dispatches, per dispatch {tot[k]/cnt[k]:14.1f}")
PY
  rm -rf $O/pmc_$n
done | tee $O/rdreq.txt
