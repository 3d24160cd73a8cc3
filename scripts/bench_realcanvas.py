"""The canvases the reference really builds (ImageProcess.cpp:206-216) are not multiples of 64: time one stitch step (pair)
and one blendTwoImages (dense canvases) at 1081x527 and 4421x2315, device resident, single-pair plans, with the fast forms
(implicit level-0 mask, source-fused level 0 at any height) and -- STITCH_GATE64=1 -- the round-2 materialised sequence, in one
process on one box (the tuning switches are part of the plan, so both plans coexist).  Also config 3's whole chain.
Usage: python scripts/bench_realcanvas.py > profiles/r03_real_canvases.json"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from computervisionimagestich2_amd import bmp, capi, pipeline  # noqa: E402

dev = torch.device("cuda:0")
G = os.path.join(ROOT, "tests", "golden")


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


res = {"what": "ms per call, device resident, one pair in flight; gate64 = the round-2 sequence (materialised level 0 and mask "
               "because the canvas height is not a multiple of 64)", "cases": []}
for (cw, ch, fw, fh) in [(1081, 527, 384, 512), (4421, 2315, 1536, 2048)]:
    for tdt, name in ((torch.uint8, "u8"), (torch.float32, "f32")):
        F, M = capi.dev_synth(fw, fh, 1, tdt, dev), capi.dev_synth(cw - fw // 2, ch - 7, 2, tdt, dev)
        P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
        A = torch.zeros((3, ch, cw), dtype=tdt, device=dev)
        B = torch.zeros((3, ch, cw), dtype=tdt, device=dev)
        capi.dev_warp(F, P, -0.25, -1.5, A)
        capi.dev_move(M, 0, -2, B)
        row = {"canvas": [cw, ch], "frame": [fw, fh], "pixel": name}
        outs = {}
        for label, gate in (("gate64", "1"), ("fast", None)):
            if gate:
                os.environ["STITCH_GATE64"] = gate
            else:
                os.environ.pop("STITCH_GATE64", None)
            plan = capi.Plan(cw, ch)
            out = torch.empty((3, ch, cw), dtype=tdt, device=dev)
            row[label + "_paths"] = sorted(plan.fast_paths)
            row[label + "_pair_ms"] = round(timed(lambda: plan.pair(F, P, -0.25, -1.5, M, 0, -2, out=out)), 4)
            plan.status()
            o1 = out.clone()
            row[label + "_blend_ms"] = round(timed(lambda: plan.blend(A, B, out=out)), 4)
            plan.status()
            outs[label] = (o1, out.clone())
            plan.close()
        assert torch.equal(outs["gate64"][0], outs["fast"][0]) and torch.equal(outs["gate64"][1], outs["fast"][1])
        assert torch.equal(outs["fast"][0], outs["fast"][1])  # the blend of the warped / moved canvases IS the pair
        row["pair_speedup"] = round(row["gate64_pair_ms"] / row["fast_pair_ms"], 3)
        row["blend_speedup"] = round(row["gate64_blend_ms"] / row["fast_blend_ms"], 3)
        res["cases"].append(row)

# config 3: the reference's own four frames, recorded stitch order, canvases 607x517 -> 838x522 -> 1081x527
J = json.load(open(os.path.join(G, "golden.json")))
frames = [torch.from_numpy(bmp.load_bmp(os.path.join(G, e["file"]))).to(dev) for e in J["input"]]
steps = J["runs"]["4"]["steps"]
chain = {"canvases": [[s["cw"], s["ch"]] for s in steps]}
for label, gate in (("gate64", "1"), ("fast", None)):
    if gate:
        os.environ["STITCH_GATE64"] = gate
    else:
        os.environ.pop("STITCH_GATE64", None)
    plans = {}
    chain[label + "_ms_per_panorama"] = round(timed(lambda: pipeline.stitch_chain(frames, steps, plans=plans), n=40), 4)
    out = pipeline.stitch_chain(frames, steps, plans=plans)
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == J["runs"]["4"]["final_sha256"]
    pipeline.close_plans(plans)
res["config3_chain"] = chain
print(json.dumps(res, indent=1))
