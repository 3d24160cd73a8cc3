"""Experiment: is the fused sweep bound by the band-to-band hand-off chain?  Same number of bands and tiles in one
launch, but 64, 32 or 16 bands per plane (canvas heights 4096, 2048, 1024 with 4, 8, 16... planes' worth of pairs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from computervisionimagestich2_amd import capi
dev = torch.device("cuda:0")
os.environ["STITCH_WAVEFRONT"] = "1"
for (ch, B) in ((4096, 2), (2048, 4), (4096, 4), (2048, 8)):  # a 6144x1024 canvas has a degenerate pyramid
    cw, fw, fh = 6144, 4096, ch
    plan = capi.Plan(cw, ch, max_pairs=B)
    items = [(capi.dev_synth(fw, fh, 2 * i + 1, torch.float32, dev), [1.0, 0.002, 1e-6, -2048.0 - 8 * i, -0.001, 1.0, 5e-7, 1.5], 0.0, 0.0,
              capi.dev_synth(fw, fh, 2 * i, torch.float32, dev), 0, 0, torch.empty((3, ch, cw), dtype=torch.float32, device=dev)) for i in range(B)]
    for _ in range(3):
        plan.pairs(items)
    torch.cuda.synchronize()
    plan.set_profiling_kernel("vv_xbyf"); plan.set_profiling(True)
    for _ in range(5):
        plan.pairs(items)
    torch.cuda.synchronize()
    ms, n, l0 = plan.read_profile()["vv_xbyf"]
    print(f"canvas 6144x{ch}, {B} pairs: bands/plane {ch // 64}, bands {7 * B * ch // 64}: fused sweep level 0 = {l0 / 5:.3f} ms per launch", flush=True)
    plan.close()
    del items; torch.cuda.empty_cache()
