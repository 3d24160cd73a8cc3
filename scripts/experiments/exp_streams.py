"""Experiment: S batched plans (B pairs each) in flight on S HIP streams."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from computervisionimagestich2_amd import capi, pipeline
F = 4096
cw, ch = pipeline.config_canvas(F)
dev = torch.device("cuda:0")
for S, B in ((1, 4), (2, 2), (2, 4), (3, 4), (4, 2)):
    plans = [capi.Plan(cw, ch, max_pairs=B) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    items = [[(capi.dev_synth(F, F, 2 * (i * B + q) + 1, torch.float32, dev), pipeline.config_map(i * B + q, F), 0.0, 0.0,
               capi.dev_synth(F, F, 2 * (i * B + q), torch.float32, dev), 0, 0,
               torch.empty((3, ch, cw), dtype=torch.float32, device=dev)) for q in range(B)] for i in range(S)]
    def run(K):
        for k in range(K):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    plans[i].pairs(items[i])
        torch.cuda.synchronize()
    run(2)
    t = time.perf_counter(); K = 8; run(K); dt = time.perf_counter() - t
    print(f"S={S} B={B}: {dt/K/S/B*1e3:.3f} ms/pair, {cw*ch*K*S*B/dt/1e6:.0f} MPix/s")
    for p in plans: p.close()
    del plans, items
    torch.cuda.empty_cache()
