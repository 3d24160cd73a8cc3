#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of scripts/pmc.sh (gpurun_out/pmc/) into profiles/traffic.json.

HBM bytes per kernel = 2*FETCH_SIZE + WRITE_SIZE, both counters in KiB.  The factor 2 on FETCH_SIZE is the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md ("FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read"); it is checked here against a kernel whose bytes are known exactly: the level-2 launch of
k_vv_x_bwd sweeps n_pairs*7 planes of 1536 x 1024 floats in place, so 2*FETCH_SIZE and WRITE_SIZE (which needs no
correction) must both come out at 7*n_pairs*1536*1024*4 bytes -- the calibration entry.  (Levels 0 and 1 no longer
qualify: their all-zero tiles are neither stored nor read.)  Counters come from separate passes (FETCH_SIZE and
WRITE_SIZE do not fit one pass)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps_total = None  # derived below: k_seam runs once per step

GROUP = {"k_compose": "compose", "k_src_index": "compose", "k_seam": "seam", "k_mask": "mask", "k_vv_x_fwd": "vv_x_fwd",
         "k_vv_x_fwd<float, false>": "vv_x_fwd", "k_vv_x_fwd<float, true>": "vv_x_fwd_src", "k_vv_x_fwd<unsigned char, false>": "vv_x_fwd",
         "k_vv_x_fwd<unsigned char, true>": "vv_x_fwd_src",
         "k_vv_x_bwd": "vv_x_bwd", "k_vv_y_fwd": "vv_y_fwd", "k_vv_y_fwd1": "vv_y_fwd", "k_vv_y_bwd_dec": "vv_y_bwd", "k_vv_y_bwd": "vv_y_bwd",
         "k_decimate": "decimate", "k_collapse<float, false>": "collapse", "k_collapse<float, true>": "collapse_l0",
         "k_collapse<unsigned char, true>": "collapse_l0", "k_collapse4<float, false>": "collapse", "k_collapse4<float, true>": "collapse_l0",
         "k_collapse4<unsigned char, true>": "collapse_l0", "k_blend_top": "collapse_top", "k_vv_xbyf<false>": "vv_xbyf",
         "k_vv_xbyf<true>": "vv_xbyf", "k_coarse": "coarse"}
# template arguments added later (CKPT of the causal sweep, pixel type and MODE of the fused sweep) do not change the group
GROUP_PREFIX = {"k_vv_x_fwd<float, true": "vv_x_fwd_src", "k_vv_x_fwd<unsigned char, true": "vv_x_fwd_src", "k_vv_x_fwd<": "vv_x_fwd", "k_vv_xbyf<": "vv_xbyf",
                "k_collapse<float, false": "collapse", "k_collapse4<float, false": "collapse", "k_collapse<": "collapse_l0", "k_collapse4<": "collapse_l0"}
# launches per launch sequence: config 2, two fused-sweep levels, levels 8..11 in k_coarse (round 3)
LAUNCH_GROUPS = {"compose": 1, "seam": 1, "mask": 1, "vv_x_fwd": 7, "vv_x_fwd_src": 1, "vv_x_bwd": 6, "vv_y_fwd": 6, "vv_y_bwd": 8, "decimate": 0,
                 "collapse_top": 0, "collapse": 7, "collapse_l0": 1, "vv_xbyf": 2, "coarse": 1}


def kname(full):
    s = full.replace("void ", "")
    s = s[s.index("sk::") + 4:] if "sk::" in s else s
    if "(" in s:
        s = s[: s.index("(")]
    if s.startswith("k_collapse<") or s.startswith("k_collapse4<") or s.startswith("k_vv_xbyf<") or s.startswith("k_vv_x_fwd<"):
        return s
    return s[: s.index("<")] if "<" in s else s


def counter(name):
    f = glob.glob(os.path.join(src, name, "runc", "*counter_collection.csv"))[0]
    tot = collections.defaultdict(float)
    mx = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = kname(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        mx[k] = max(mx[k], float(r["Counter_Value"]))
        cnt[k] += 1
    global steps_total
    steps_total = cnt.get("k_seam", 0) or steps_total
    return tot, mx


fetch, fmax = counter("FETCH_SIZE")
write, wmax = counter("WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    g = GROUP.get(k) or next((v for pre, v in GROUP_PREFIX.items() if k.startswith(pre)), None)
    if g is None:
        continue
    e = out.setdefault(g, {"fetch_kib_per_step": 0.0, "write_kib_per_step": 0.0, "kernels": []})
    e["fetch_kib_per_step"] += fetch.get(k, 0) / steps_total
    e["write_kib_per_step"] += write.get(k, 0) / steps_total
    e["kernels"].append(k)
for g, e in out.items():
    hbm = (2 * e["fetch_kib_per_step"] + e["write_kib_per_step"]) * 1024
    e["hbm_bytes_per_step"] = int(hbm)
    e["hbm_bytes_per_pair"] = int(hbm / batch)
    n = LAUNCH_GROUPS.get(g, 0)
    e["launch_groups_per_step"] = n
    e["hbm_bytes_per_launch"] = int(hbm / n) if n else None
    e["batch"] = batch
out["_meta"] = {"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, averaged over the run's steps",
                "calibration_k_vv_x_bwd_level2": {"FETCH_SIZE_KiB_max": fmax.get("k_vv_x_bwd"), "WRITE_SIZE_KiB_max": wmax.get("k_vv_x_bwd"),
                                                  "expected_KiB_each_way": 7 * batch * 1536 * 1024 * 4 / 1024},
                "source": os.path.relpath(src, ROOT), "batch": batch, "steps": steps_total}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
tot = sum(e["hbm_bytes_per_pair"] for g, e in out.items() if not g.startswith("_"))
for g, e in out.items():
    if not g.startswith("_"):
        print(f"{g:10s} {e['hbm_bytes_per_pair'] / 1e9:7.3f} GB/pair")
print(f"total      {tot / 1e9:7.3f} GB/pair")
