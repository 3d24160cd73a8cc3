"""One launch sequence of a batched plan, repeated (for rocprofv3 --kernel-trace: per-dispatch durations by kernel and grid).
usage: python3 scripts/experiments/exp_seq.py [pairs=8] [reps=5] [frame=4096] [pixel=f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from computervisionimagestich2_amd import capi, pipeline
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
F = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
tdt = torch.uint8 if len(sys.argv) > 4 and sys.argv[4] == "u8" else torch.float32
dev = torch.device("cuda:0")
cw, ch = pipeline.config_canvas(F)
plan = capi.Plan(cw, ch, max_pairs=B)
items = [(capi.dev_synth(F, F, 2 * i + 1, tdt, dev), pipeline.config_map(i, F), 0.0, 0.0, capi.dev_synth(F, F, 2 * i, tdt, dev), 0, 0,
          torch.empty((3, ch, cw), dtype=tdt, device=dev)) for i in range(B)]
for _ in range(R):
    plan.pairs(items)
    torch.cuda.synchronize()
for q in range(B):
    plan.status(q)
print("ok")
