"""Timeline of the LAST repetition of a launch sequence in a rocprofv3 --kernel-trace CSV: per dispatch its start (us from the
sequence's first dispatch), duration and the idle gap since the previous dispatch ended.  The sequence starts at the last
dispatch of <first-kernel-substring> (default k_seam ... the first kernel after S1).
usage: python scripts/experiments/timeline.py <dir-or-csv> [first-kernel-substring]"""
import csv, glob, os, sys
p = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_seam"
files = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sk::", ""),
                     r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
i0 = starts[-1]
while i0 > 0 and ("k_src_index" in rows[i0 - 1][2] or "k_compose" in rows[i0 - 1][2] or "k_fill_bytes" in rows[i0 - 1][2] or "k_load_canvases" in rows[i0 - 1][2]):
    i0 -= 1
seq = rows[i0:]
t0, prev_end, busy = seq[0][0], seq[0][0], 0
for s, e, name, gx, gy, gz in seq:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name[:44]:44s} grid {gx}x{gy}x{gz}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"dispatches {len(seq)}, span {(prev_end - t0) / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us")
