"""FETCH_SIZE / WRITE_SIZE per dispatch of the kernels of one rocprofv3 --pmc output directory, MB per pair.
usage: python scripts/experiments/fetch_report.py <dir> <pairs per dispatch> [name filter ...]"""
import collections, csv, glob, os, sys
d, n = sys.argv[1], float(sys.argv[2])
flt = sys.argv[3:]
rows = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = (k[: k.index("(")] if "(" in k else k).replace("void ", "").replace("sk::", "")
        if flt and not any(s in k for s in flt):
            continue
        e = rows[(k, r["Counter_Name"], r.get("Grid_Size", ""))]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
for (k, c, g), (cnt, s) in sorted(rows.items()):
    mb = s / cnt * 1024 * (2 if c == "FETCH_SIZE" else 1) / n / 1e6
    print(f"  {k:50s} grid {g:>10s} {c:10s} x{cnt:<3d} {mb:9.1f} MB per pair")
