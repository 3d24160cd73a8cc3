"""Where the level-0 kernels' over-fetch comes from: one 8-pair launch sequence at 6144x4096 under a given backward map, to be run
under `rocprofv3 --pmc FETCH_SIZE`.  usage: python3 scripts/experiments/exp_collapse_fetch.py <default|aligned|shift3|slant> [canvas height]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from computervisionimagestich2_amd import capi
MAPS = {
    "default": ((1.0, 0.002, 1e-6, -2048.0, -0.001, 1.0, 5e-7, 1.5), -0.25, -1.5),  # bench.py's map (pipeline.SEED_MAP)
    "aligned": ((1.0, 0.0, 0.0, -2048.0, 0.0, 1.0, 0.0, 0.0), 0.0, 0.0),            # frame rows = canvas rows, columns 128-byte aligned
    "shift3": ((1.0, 0.0, 0.0, -2045.0, 0.0, 1.0, 0.0, 0.0), 0.0, 0.0),             # rows aligned, columns three samples off the lines
    "slant": ((1.0, 0.0, 0.0, -2048.0, -0.001, 1.0, 0.0, 0.0), 0.0, 0.0),           # frame rows slanted by one row per 1000 columns
}
P, offx, offy = MAPS[sys.argv[1]]
cw, ch, fw, fh, n = 6144, (int(sys.argv[2]) if len(sys.argv) > 2 else 4096), 4096, 4096, 8
dev = torch.device("cuda:0")
F = [capi.dev_synth(fw, fh, 1 + i, torch.float32, dev) for i in range(n)]
M = [capi.dev_synth(cw - fw // 2, ch - 7, 100 + i, torch.float32, dev) for i in range(n)]
outs = [torch.empty((3, ch, cw), dtype=torch.float32, device=dev) for _ in range(n)]
plan = capi.Plan(cw, ch, max_pairs=n)
for _ in range(2):
    plan.pairs([(F[i], P, offx, offy, M[i], 0, -2, outs[i]) for i in range(n)])
torch.cuda.synchronize()
plan.status()
print(sys.argv[1], sorted(plan.call_forms(n)))
plan.close()
