"""Experiment: throughput of S independent pairs in flight on S HIP streams (one plan each)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervisionimagestich2_amd import capi, pipeline
F = 4096
cw, ch = pipeline.config_canvas(F)
dev = torch.device("cuda:0")
for S in (1, 2, 3, 4):
    plans = [capi.Plan(cw, ch) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    ins = [(capi.dev_synth(F, F, 2 * i, torch.float32, dev), capi.dev_synth(F, F, 2 * i + 1, torch.float32, dev), pipeline.config_map(i, F)) for i in range(S)]
    outs = [torch.empty((3, ch, cw), dtype=torch.float32, device=dev) for _ in range(S)]
    def run(K):
        for k in range(K):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    plans[i].pair(ins[i][1], ins[i][2], 0.0, 0.0, ins[i][0], 0, 0, outs[i])
        torch.cuda.synchronize()
    run(2)
    t = time.perf_counter(); K = 10; run(K); dt = time.perf_counter() - t
    print(f"S={S}: {dt/K/S*1e3:.3f} ms/pair, {cw*ch*K*S/dt/1e6:.0f} MPix/s")
    for p in plans: p.close()
    del plans, ins, outs
    torch.cuda.empty_cache()
