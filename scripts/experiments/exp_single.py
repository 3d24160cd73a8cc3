"""One single-pair plan at a given canvas, a few back-to-back calls (for rocprofv3 --kernel-trace timelines and quick timings).
usage: python scripts/experiments/exp_single.py cw ch fw fh [reps] [pair|blend] [u8|f32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from computervisionimagestich2_amd import capi
cw, ch, fw, fh = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
mode = sys.argv[6] if len(sys.argv) > 6 else "pair"
tdt = torch.uint8 if (len(sys.argv) > 7 and sys.argv[7] == "u8") else torch.float32
dev = torch.device("cuda:0")
F, M = capi.dev_synth(fw, fh, 1, tdt, dev), capi.dev_synth(cw - fw // 2, ch - 7, 2, tdt, dev)
P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
A = torch.zeros((3, ch, cw), dtype=tdt, device=dev)
B = torch.zeros((3, ch, cw), dtype=tdt, device=dev)
capi.dev_warp(F, P, -0.25, -1.5, A)
capi.dev_move(M, 0, -2, B)
plan = capi.Plan(cw, ch)
out = torch.empty((3, ch, cw), dtype=tdt, device=dev)
fn = (lambda: plan.pair(F, P, -0.25, -1.5, M, 0, -2, out=out)) if mode == "pair" else (lambda: plan.blend(A, B, out=out))
for _ in range(3):
    fn()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print(f"{mode} {cw}x{ch} {tdt}: {(time.perf_counter() - t) / reps * 1e3:.4f} ms per call; paths {sorted(plan.fast_paths)}")
plan.status()
plan.close()
