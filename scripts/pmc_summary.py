#!/usr/bin/env python3
"""Per-kernel sums of every counter in the rocprofv3 --pmc passes of scripts/pmc.sh -> profiles/r04_pmc_summary.csv (or the path given as second argument)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_pmc_summary.csv")
rows = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(os.path.join(src, "*", "runc", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[: k.index("(")] if "(" in k else k
        e = rows[(k.replace("void ", "").replace("sk::", ""), r["Counter_Name"])]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
    for (k, c), (n, s) in sorted(rows.items()):
        w.writerow([k, c, n, s, s / n])
print("wrote", out, len(rows), "rows")
