"""Summarise a rocprofv3 --kernel-trace CSV: average duration per (kernel, grid) over the last N-1 of N repetitions.
usage: python scripts/trace_levels.py <dir-or-csv> [name-filter]"""
import csv, glob, os, sys, collections
p = sys.argv[1]
files = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.OrderedDict()
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        short = name.split("(")[0].replace("void sk::", "")
        key = (short, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")))
        rows.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in rows.items():
    use = v[1:] if len(v) > 1 else v
    avg = sum(use) / len(use)
    tot += avg
    print(f"{k[0][:48]:48s} grid {k[1]:>8s}x{k[2]:>5s}x{k[3]:>3s} vgpr {k[4]:>4s}  n={len(v):3d}  avg {avg:9.1f} us")
print("sum of averages", round(tot, 1), "us")
