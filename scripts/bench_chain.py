"""Config 3 (the reference's own Input/ frames, recorded stitch order and transforms): time of the whole device-resident
chain -- 4 projections, 3 stitch steps on growing canvases, equalise + luminance mix -- per panorama."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from computervisionimagestich2_amd import bmp, pipeline
G = os.path.join(ROOT, "tests", "golden")
J = json.load(open(os.path.join(G, "golden.json")))
dev = torch.device("cuda:0")
frames = [torch.from_numpy(bmp.load_bmp(os.path.join(G, e["file"]))).to(dev) for e in J["input"]]
steps = J["runs"]["4"]["steps"]
import hashlib
res = {"config": "3 (Input/1..4.bmp, recorded transforms)", "canvases": [[s["cw"], s["ch"]] for s in steps]}
for label, plans in (("workspaces created per panorama", None), ("workspaces reused", {})):
    for _ in range(3):
        out = pipeline.stitch_chain(frames, steps, plans=plans)
    torch.cuda.synchronize()
    t = time.perf_counter()
    N = 20
    for _ in range(N):
        out = pipeline.stitch_chain(frames, steps, plans=plans)
    torch.cuda.synchronize()
    res["ms_per_panorama, " + label] = round((time.perf_counter() - t) / N * 1e3, 3)
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == J["runs"]["4"]["final_sha256"]
    if plans is not None:
        pipeline.close_plans(plans)
res["panorama"] = list(out.shape)
print(json.dumps(res))
