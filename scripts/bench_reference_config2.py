"""The reference's own CImg loop on BASELINE config 2's pair, timed on the GPU box's host cores (north_star: "next to the
reference CImg CPU loop timed on the same box's host cores"): warpingImageByHomography + movingImageByOffset +
blendTwoImages (ImageProcess.cpp:596-620,648-773) from oracle/_ref/libref_hotpath.so -- the reference's sources compiled in
place by oracle/Makefile -- on two 4096x4096x3 unsigned char frames (the reference's pixel type) -> 6144x4096 canvas,
ONE thread (the reference is single-threaded on this path).  The MI355X result of the same pair is compared with it byte
for byte.  ~2-3 minutes of CPU.  Usage: python scripts/bench_reference_config2.py > profiles/r04_reference_config2.json"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from computervisionimagestich2_amd import pipeline

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cw, ch = pipeline.config_canvas(F)
O = oracle_lib.Oracle()
R = oracle_lib.Reference()
A, B = O.synth(F, F, 0, np.uint8), O.synth(F, F, 1, np.uint8)  # pair 0 of bench.py: frame 1 warped, frame 0 the mosaic
p = pipeline.config_map(0, F)
t = {}
t0 = time.perf_counter()
a = R.warp(B, p, 0.0, 0.0, cw, ch)
t["warp_s"] = time.perf_counter() - t0
print(f"[reference] warp {t['warp_s']:.2f} s", file=sys.stderr, flush=True)
t0 = time.perf_counter()
b = R.move(A, 0, 0, cw, ch)
t["move_s"] = time.perf_counter() - t0
print(f"[reference] move {t['move_s']:.2f} s", file=sys.stderr, flush=True)
t0 = time.perf_counter()
out = R.blend(a, b)
t["blend_s"] = time.perf_counter() - t0
print(f"[reference] blend {t['blend_s']:.2f} s", file=sys.stderr, flush=True)
total = sum(t.values())
res = {"what": "reference (chensh236/ComputerVisionImageStich2, compiled in place: oracle/_ref/libref_hotpath.so) on config 2's pair, "
               "unsigned char frames, one thread, on the GPU box's host",
       "frame": [F, F, 3], "canvas": [cw, ch, 3], "seconds": {k: round(v, 3) for k, v in t.items()}, "total_s": round(total, 3),
       "value": round(cw * ch / total / 1e6, 4), "unit": "MPix/s", "cores": 1, "kind": "reference",
       "reference_lines": "ImageProcess.cpp:596-606 (warp), :608-620 (move), :648-773 (blendTwoImages)",
       "sha256": hashlib.sha256(out.tobytes()).hexdigest()}
try:
    import torch
    from computervisionimagestich2_amd import capi
    if torch.cuda.is_available():
        dev = torch.device("cuda:0")
        plan = capi.Plan(cw, ch)
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = plan.pair(torch.from_numpy(B).to(dev) if rep == 0 else dB, p, 0.0, 0.0, torch.from_numpy(A).to(dev) if rep == 0 else dA, 0, 0)
            if rep == 0:
                dB, dA = torch.from_numpy(B).to(dev), torch.from_numpy(A).to(dev)
            plan.status()
            torch.cuda.synchronize()
            gpu_ms = (time.perf_counter() - t0) * 1e3
        res["mi355x_same_pair"] = {"bit_identical_to_the_reference": bool(np.array_equal(got.cpu().numpy(), out)), "ms_single_pair_in_flight": round(gpu_ms, 3),
                                   "speedup_vs_reference_one_thread": round(total * 1e3 / gpu_ms, 1)}
        plan.close()
except Exception as e:  # the reference timing stands on its own
    res["mi355x_same_pair"] = {"error": repr(e)}
print(json.dumps(res, indent=1))
