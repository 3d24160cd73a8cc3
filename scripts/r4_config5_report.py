#!/usr/bin/env python3
"""gpurun_out/c5/ (scripts/r4_config5_profile.sh) -> profiles/r04_config5_kernel_stats.csv (rocprofv3 --kernel-trace --stats, copied),
profiles/r04_config5_bench.json (the bench line of the same call) and profiles/traffic_config5.json: per kernel symbol the
dispatches per pair, the average duration, the HBM bytes per dispatch from the PMC passes ((2 * FETCH_SIZE + WRITE_SIZE) * 1024: the
gfx950 correction, calibrated for this repo's access widths by scripts/experiments/fetch_calib.hip) and the rate they give."""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "c5")
P = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(P, "r04_config5_kernel_stats.csv"))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(P, "r04_config5_bench.json"), "w"), indent=1)


def short(k):
    k = k.replace("void ", "").replace("sk::", "")
    return k[: k.index("(")] if "(" in k else k


stats = {short(r["Name"]): r for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))}
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(os.path.join(src, f"pmc_{c}.csv"))):
        pmc.setdefault(r["kernel"], {})[c] = (int(r["dispatches"]), float(r["mean_KiB_per_dispatch"]))
seq = stats["k_seam<float>"]["Calls"]  # one per launch sequence = per pair
out = {"workload": bench["config"]["workload"], "ms_per_pair": bench["ms_per_step"], "frac_of_hbm_peak_on_8d_bytes": bench["pipeline"]["frac_of_hbm_peak"],
       "device_copy_GBps": bench["roofline"]["device_copy_GBps"], "launch_sequences_traced": int(seq), "kernels": {}}
for k, s in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
    if not k.startswith("k_"):
        continue
    e = {"dispatches_per_pair": int(s["Calls"]) / int(seq), "avg_ms": float(s["AverageNs"]) / 1e6, "max_ms": float(s["MaxNs"]) / 1e6,
         "ms_per_pair": float(s["TotalDurationNs"]) / 1e6 / int(seq), "share": float(s["Percentage"]) / 100}
    if k in pmc and "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]:
        b = (2 * pmc[k]["FETCH_SIZE"][1] + pmc[k]["WRITE_SIZE"][1]) * 1024
        e["hbm_bytes_per_dispatch"] = int(b)
        e["hbm_TBps_at_avg_duration"] = round(b / (float(s["AverageNs"]) / 1e9) / 1e12, 3)
    out["kernels"][k] = e
json.dump(out, open(os.path.join(P, "traffic_config5.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(f"{k:48s} {e['dispatches_per_pair']:5.1f}/pair  avg {e['avg_ms']:7.3f} ms  {e['ms_per_pair']:7.3f} ms/pair  {e.get('hbm_TBps_at_avg_duration', '')} TB/s of counter traffic")
