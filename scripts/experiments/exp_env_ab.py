"""One pair in flight under two settings of one tuning switch: outputs bit for bit, time per call.
usage: python scripts/experiments/exp_env_ab.py SWITCH valueA valueB cw ch fw fh [reps]   (value '-' = unset)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from computervisionimagestich2_amd import capi
sw, va, vb = sys.argv[1:4]
cw, ch, fw, fh = (int(v) for v in sys.argv[4:8])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dev = torch.device("cuda:0")
for tdt in (torch.float32, torch.uint8):
    F, M = capi.dev_synth(fw, fh, 1, tdt, dev), capi.dev_synth(cw - fw // 2, ch - 7, 2, tdt, dev)
    P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
    outs = {}
    for v in (va, vb):
        if v == "-":
            os.environ.pop(sw, None)
        else:
            os.environ[sw] = v
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
        outs[v] = out.clone()
        print(f"{cw}x{ch} {tdt} {sw}={v}: {ms:.4f} ms per call; forms {sorted(plan.call_forms(1))}", flush=True)
        plan.close()
    same = torch.equal(outs[va], outs[vb])
    print(f"{cw}x{ch} {tdt}: outputs identical = {same}", flush=True)
    if not same:
        d = (outs[va].float() - outs[vb].float()).abs()
        print("  differing samples", int((d > 0).sum()), "first", torch.nonzero(d)[:5].tolist(), "max", float(d.max()))
        sys.exit(1)
