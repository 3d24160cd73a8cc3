"""One pair in flight: the collapse's strip heights (STITCH_CROWS_L0 / STITCH_CROWS_LN).  usage: exp_crows.py cw ch fw fh"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from computervisionimagestich2_amd import capi
cw, ch, fw, fh = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0")
F, M = capi.dev_synth(fw, fh, 1, torch.float32, dev), capi.dev_synth(cw - fw // 2, ch - 7, 2, torch.float32, dev)
P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
out = torch.empty((3, ch, cw), dtype=torch.float32, device=dev)
ref = None
for l0, ln in [(None, None), (None, 4), (None, 6), (None, 8), (None, 12), (16, 8), (8, 8), (16, None), (None, None), (None, 8)]:
    for k, v in (("STITCH_CROWS_L0", l0), ("STITCH_CROWS_LN", ln)):
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = str(v)
    plan = capi.Plan(cw, ch)
    fn = lambda: plan.pair(F, P, -0.25, -1.5, M, 0, -2, out=out)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 40 * 1e3
    plan.status()
    if ref is None:
        ref = out.clone()
    print(f"{cw}x{ch} L0={l0} LN={ln}: {ms:.4f} ms  equal {torch.equal(out, ref)}", flush=True)
    plan.close()
