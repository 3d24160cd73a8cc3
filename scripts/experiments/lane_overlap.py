#!/usr/bin/env python3
"""How the batches in flight overlap: from a rocprofv3 kernel trace of the bench (scripts/experiments/prof_lanes.sh), the time the
GPU spends with 0, 1, 2, ... kernels running, and for each kernel family the share of the wall time during which one of
its launches is running and how much company it has on average."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "lanes")
f = sorted(glob.glob(os.path.join(src, "*", "*kernel_trace.csv")))[-1]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "sk::" not in n:
        continue
    n = n[n.index("sk::") + 4:]
    n = n[: n.index("(")] if "(" in n else n
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
# the window: from the first to the last moment at which at least three kernels run at once (the bench's region with all
# batches in flight; its one-batch-in-flight regions and the set-up lie outside)
ev0 = sorted([(s_, 1) for s_, e_, n_ in rows] + [(e_, -1) for s_, e_, n_ in rows])
k, t_lo, t_hi = 0, None, None
for t, d in ev0:
    k += d
    if k >= 3:
        t_lo = t if t_lo is None else t_lo
        t_hi = t
ev = []
for s_, e_, n in rows:
    if e_ <= t_lo or s_ >= t_hi:
        continue
    ev.append((max(s_, t_lo), 1, n))
    ev.append((min(e_, t_hi), -1, n))
ev.sort()
active = collections.Counter()
conc_time = collections.Counter()
fam_time = collections.Counter()
fam_company = collections.Counter()
last = ev[0][0]
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        k = sum(active.values())
        conc_time[k] += dt
        for fam, c in active.items():
            if c > 0:
                fam_time[fam] += dt
                fam_company[fam] += dt * (k - 1)
    active[n] += d
    last = t
tot = sum(conc_time.values())
print(f"window {tot / 1e6:.2f} ms of {f}")
print("kernels running at once: " + "  ".join(f"{k}: {100 * v / tot:.1f}%" for k, v in sorted(conc_time.items())))
print(f"{'kernel':40s} {'running %':>10s} {'avg others':>11s}")
for fam, v in sorted(fam_time.items(), key=lambda kv: -kv[1]):
    print(f"{fam[:40]:40s} {100 * v / tot:10.1f} {fam_company[fam] / v:11.2f}")

# ---- where a batch's own time goes: duration of its launches by number of workgroups (coarse pyramid levels = small grids)
rows2 = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "sk::" not in n:
        continue
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e_ <= t_lo or s_ >= t_hi:
        continue
    wgs = 1
    for ax in "XYZ":
        wgs *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
    rows2.append((wgs, e_ - s_))
tot2 = sum(d for _, d in rows2)
print("\nlaunch time by grid size (share of the summed launch durations = of the batches' own time):")
for lo, hi in ((0, 64), (64, 512), (512, 4096), (4096, 1 << 40)):
    sel = [d for w, d in rows2 if lo <= w < hi]
    print(f"  {lo:>5d} <= workgroups < {hi if hi < 1 << 40 else 'inf':>5}: {100 * sum(sel) / tot2:5.1f} %  ({len(sel)} launches, mean {sum(sel) / max(1, len(sel)) / 1e3:.1f} us)")
