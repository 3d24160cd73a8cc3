"""One pair in flight with and without the fused lone sweep (k_vv_xby_m): outputs bit for bit, time per call.
usage: python scripts/experiments/exp_xbym_ab.py cw ch fw fh [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from computervisionimagestich2_amd import capi
cw, ch, fw, fh = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
res = {}
for tdt in (torch.float32, torch.uint8):
    F, M = capi.dev_synth(fw, fh, 1, tdt, dev), capi.dev_synth(cw - fw // 2, ch - 7, 2, tdt, dev)
    P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
    outs = {}
    for mode in ("0", "1"):
        os.environ["STITCH_XBYM"] = mode
        plan = capi.Plan(cw, ch)
        out = torch.empty((3, ch, cw), dtype=tdt, device=dev)
        fn = lambda: plan.pair(F, P, -0.25, -1.5, M, 0, -2, out=out)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        plan.status()
        outs[mode] = out.clone()
        print(f"{cw}x{ch} {tdt} STITCH_XBYM={mode}: {ms:.4f} ms per call; forms {sorted(plan.call_forms(1))}", flush=True)
        plan.close()
    same = torch.equal(outs["0"], outs["1"])
    print(f"{cw}x{ch} {tdt}: outputs identical = {same}", flush=True)
    if not same:
        d = (outs["0"].float() - outs["1"].float()).abs()
        nz = torch.nonzero(d)
        print("  differing samples", int((d > 0).sum()), "first", nz[:5].tolist(), "max", float(d.max()))
        sys.exit(1)
