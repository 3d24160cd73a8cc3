#!/bin/bash
# round 4, GPU call 4: mover variants (tiles in flight, one workgroup per CU) + FETCH_SIZE calibration for the repo's access patterns
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
echo "== single pair, ms per call"
for c in "1081 527 384 512" "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  for v in "STITCH_MOVER=0" "STITCH_MOVER=1" "STITCH_MOVER=2" "STITCH_MOVER=1 STITCH_Y1N=1" "STITCH_MOVER=2 STITCH_Y1N=1"; do
    echo -n "[$v] "; env $v timeout -k 10 120 python scripts/experiments/exp_single.py $c 20 pair f32 2>&1 | tail -1
  done
done | tee $O/single.txt
echo "== timelines"
for v in "STITCH_MOVER=2" "STITCH_MOVER=2 STITCH_Y1N=1"; do
for c in "4421 2315 1536 2048" "6144 4096 4096 4096"; do
  set -- $c
  n=$(echo "$v" | tr -c 'A-Za-z0-9=' '_')
  ( export $v; rocprofv3 --kernel-trace --output-format csv -d $O/tl_$1_$n -- python3 scripts/experiments/exp_single.py $c 5 pair f32 > $O/tl_$1_$n.log 2>&1 )
  python scripts/experiments/timeline.py $O/tl_$1_$n > $O/tl_$1_$n.txt; tail -1 $O/tl_$1_$n.txt; rm -rf $O/tl_$1_$n
done
done
echo "== FETCH_SIZE calibration"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib -- scripts/experiments/fetch_calib.bin > $O/calib.log 2>&1
python - $O/calib <<'PY' | tee $O/calib.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].split("(")[0]
    tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(tot): print(f"  {k:10s} {cnt[k]} launches, FETCH_SIZE per launch {tot[k]/cnt[k]:12.1f} KiB = {tot[k]/cnt[k]/1048576:.4f} of the 1048576 KiB read")
PY
rm -rf $O/calib
rocprofv3 --list-avail > $O/avail.txt 2>&1; grep -c "" $O/avail.txt; grep -i -o "TCC_EA0_RD[A-Z_0-9]*" $O/avail.txt | sort -u | head -30
