"""Host-pointer entry points (what the C++ adaptor calls), PCIe inclusive: one config-2 pair (f32 and u8) and a
config-3 sized pair, first call and steady state; bytes compared with the device-resident path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from computervisionimagestich2_amd import capi, pipeline
dev = torch.device("cuda:0")
res = {}
for name, F, td in (("f32 4096", 4096, torch.float32), ("u8 4096", 4096, torch.uint8), ("u8 512", 512, torch.uint8)):
    cw, ch = pipeline.config_canvas(F)
    dA, dB = capi.dev_synth(F, F, 0, td, dev), capi.dev_synth(F, F, 1, td, dev)
    A, B = dA.cpu().numpy(), dB.cpu().numpy()
    p = pipeline.config_map(0, F)
    if F != 4096:
        p[3] = -F / 2.0
    ts = []
    for rep in range(4):
        t = time.perf_counter()
        out, seam = capi.pair(B, p, 0.0, 0.0, A, 0, 0, cw, ch)
        ts.append((time.perf_counter() - t) * 1e3)
    plan = capi.Plan(cw, ch)
    ref = plan.pair(dB, p, 0.0, 0.0, dA, 0, 0).cpu().numpy()
    plan.close()
    assert np.array_equal(out.view(np.uint8), ref.view(np.uint8))
    nbytes = A.nbytes + B.nbytes + out.nbytes
    res[f"pair {name} host pointers"] = {"first_ms": round(ts[0], 2), "steady_ms": round(min(ts[1:]), 2), "MB moved": round(nbytes / 1e6, 1),
                                         "GB/s incl. compute": round(nbytes / min(ts[1:]) / 1e6, 1), "MPix/s (canvas)": round(cw * ch / min(ts[1:]) / 1e3, 1)}
capi.lib().stitch_trim()
print(json.dumps(res, indent=1))
