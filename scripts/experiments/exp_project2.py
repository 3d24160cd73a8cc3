"""Two projections of different frames at once on two streams against the same two back to back: how much of a launch's time
is phases that another launch's workgroups can fill (staging latency vs arithmetic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from computervisionimagestich2_amd import capi
dev = torch.device("cuda:0")
w = h = 4096
for td in (torch.uint8, torch.float32):
    srcs = [capi.dev_synth(w, h, i, td, dev) for i in range(2)]
    dsts = [torch.empty_like(s) for s in srcs]
    sts = [torch.cuda.Stream() for _ in range(2)]
    def both(par):
        for i in range(2):
            if par:
                with torch.cuda.stream(sts[i]):
                    capi.dev_project(srcs[i], 15.0, dsts[i])
            else:
                capi.dev_project(srcs[i], 15.0, dsts[i])
    for par in (False, True):
        for _ in range(3): both(par)
        torch.cuda.synchronize()
        import time
        t = time.perf_counter()
        for _ in range(50): both(par)
        torch.cuda.synchronize()
        print(td, "parallel" if par else "serial", f"{(time.perf_counter() - t) / 50 * 1e3:.4f} ms per two launches")
