"""The band code path as ONE band at config 5's size, a few repetitions (for rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from computervisionimagestich2_amd import capi, pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Ls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
cw, ch = pipeline.config_canvas(F)
A, B = capi.dev_synth(F, F, 0, torch.float32, dev), capi.dev_synth(F, F, 1, torch.float32, dev)
bs = pipeline.BandStitcher(cw, ch, Ls, pipeline.LocalTransport(0, 1, pipeline.LocalTransport.make_queues(1)), dev)
o = None
for _ in range(3):
    o = bs.run(B, pipeline.config_map(0, F), 0.0, 0.0, A, 0, 0, o)
torch.cuda.synchronize()
if len(sys.argv) > 3:  # the single-GPU plan at the same size, for the same trace
    plan = capi.Plan(cw, ch)
    for _ in range(3):
        plan.pair(B, pipeline.config_map(0, F), 0.0, 0.0, A, 0, 0, o)
    torch.cuda.synchronize()
print("ok")
