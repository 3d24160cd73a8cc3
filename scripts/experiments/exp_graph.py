"""Experiment: time, status and output of every single replay / eager call in the order of exp_graph2."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from computervisionimagestich2_amd import capi, pipeline
F = 4096; B = 3
cw, ch = pipeline.config_canvas(F)
dev = torch.device("cuda:0")
plan = capi.Plan(cw, ch, max_pairs=B)
items = [(capi.dev_synth(F, F, 2 * i + 1, torch.float32, dev), pipeline.config_map(i, F), 0.0, 0.0,
          capi.dev_synth(F, F, 2 * i, torch.float32, dev), 0, 0, torch.empty((3, ch, cw), dtype=torch.float32, device=dev)) for i in range(B)]
plan.pairs(items); torch.cuda.synchronize()
ref = [it[7].clone() for it in items]
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        plan.pairs(items)
def one(name, f):
    for it in items: it[7].zero_()
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 1e3
    try:
        st = [plan.status(i)[0] for i in range(B)]
    except Exception as e:
        st = str(e)[:60]
    eq = all(torch.equal(a, it[7]) for a, it in zip(ref, items))
    print(f"{name}: {ms:.3f} ms status={st} equal={eq}", flush=True)
for k in range(4): one("replay", g.replay)
def burst(f, n):
    def h():
        for _ in range(n): f()
    return h
one("6 replays", burst(g.replay, 6))
one("6 eager", burst(lambda: plan.pairs(items), 6))
for k in range(4): one("replay", g.replay)
for k in range(2): one("eager", lambda: plan.pairs(items))
for k in range(2): one("replay", g.replay)
