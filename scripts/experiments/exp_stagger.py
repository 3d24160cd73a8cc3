"""4 lanes x 16 config-2 pairs: does the phase between the lanes matter?  The first launch of every lane is delayed by k * stagger ms
(host sleep); afterwards the lanes run back to back, so the offset persists."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from computervisionimagestich2_amd import capi
cw, ch, fw, fh, n = 6144, 4096, 4096, 4096, 16
dev = torch.device("cuda:0")
tdt = torch.float32
F, M = capi.dev_synth(fw, fh, 1, tdt, dev), capi.dev_synth(cw - fw // 2, ch - 7, 2, tdt, dev)
P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
lanes = [(capi.Plan(cw, ch, max_pairs=n), torch.cuda.Stream(device=dev), [torch.empty((3, ch, cw), dtype=tdt, device=dev) for _ in range(n)]) for _ in range(4)]
def go(stagger_ms=0.0):
    for k, (plan, st, outs_) in enumerate(lanes):
        if stagger_ms and k:
            time.sleep(stagger_ms / 1e3)
        with torch.cuda.stream(st):
            plan.pairs([(F, P, -0.25, -1.5, M, 0, -2, o) for o in outs_])
for trial, stagger in enumerate([0, 16, 0, 8, 0, 16, 4, 0, 16, 0, 32, 0]):
    torch.cuda.synchronize()
    go(stagger)  # sets the phase (not timed)
    t = time.perf_counter()
    reps = 8
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print(f"trial {trial} stagger {stagger:3d} ms: {ms / (4 * n):.4f} ms per pair  {cw * ch / 1e6 / (ms / (4 * n)) * 1e3:9.1f} MPix/s", flush=True)
for plan, st, outs_ in lanes:
    plan.status(n - 1)
