"""Config 5 (one 16384 x 16384 x 3 f32 pair -> 24576 x 16384 mosaic) on ONE MI355X: the ordinary single-GPU plan beside the
band-split code path run as 1 band and as N bands on N streams of the same device (ranks as threads; hand-offs on the device,
pipeline.LocalTransport).  N bands on one GPU cannot be faster than one -- the point is what the split
costs: the unfused sweeps, the hand-offs, the gather and the halos; LocalBandGroup = the same bands from one host thread.  usage: bench_band.py [frame=16384] [split=4]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from computervisionimagestich2_amd import capi, pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
Ls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
cw, ch = pipeline.config_canvas(F)
A, B = capi.dev_synth(F, F, 0, torch.float32, dev), capi.dev_synth(F, F, 1, torch.float32, dev)
p = pipeline.config_map(0, F)
res = {"canvas": [cw, ch], "split_levels": Ls}
plan = capi.Plan(cw, ch)
out = torch.empty((3, ch, cw), dtype=torch.float32, device=dev)
for _ in range(2):
    plan.pair(B, p, 0.0, 0.0, A, 0, 0, out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    plan.pair(B, p, 0.0, 0.0, A, 0, 0, out)
torch.cuda.synchronize()
res["single_gpu_plan_ms"] = round((time.perf_counter() - t) / 3 * 1e3, 2)
res["plan_workspace_GB"] = round(plan.workspace_bytes / 1e9, 2)
plan.close()
ref = out
FUSE = {"1": True, "0": False}.get(os.environ.get("BAND_FUSE", ""), None)  # None: BandStitcher's default (fused only without a split)
res["fused_sweeps"] = FUSE
PMIN = int(os.environ["BAND_PLANE_MIN"]) if "BAND_PLANE_MIN" in os.environ else None  # samples per plane from which hand-offs go plane by plane
res["plane_pipeline_min"] = PMIN  # None: hand-offs per level (the default)
for N in (1, 2, 8):
    qs = pipeline.LocalTransport.make_queues(N)
    outs, times = [None] * N, [0.0] * N
    bar = threading.Barrier(N)

    def work(r):
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            bs = pipeline.BandStitcher(cw, ch, Ls, pipeline.LocalTransport(r, N, qs), dev, fuse_sweeps=FUSE, plane_pipeline_min=PMIN)
            o = None
            for rep in range(3):
                if rep == 1:
                    torch.cuda.current_stream().synchronize()
                    bar.wait()
                    t0 = time.perf_counter()
                o = bs.run(B, p, 0.0, 0.0, A, 0, 0, o)
            torch.cuda.current_stream().synchronize()
            times[r] = (time.perf_counter() - t0) / 2
            outs[r] = o
            bs.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(N)]
    [t_.start() for t_ in th]
    [t_.join() for t_ in th]
    got = torch.cat(outs, dim=1)
    res[f"band_path_{N}_band(s)_on_one_gpu_ms"] = round(max(times) * 1e3, 2)
    res[f"band_path_{N}_equals_plan"] = bool(torch.equal(got, ref))
    del outs, got
    torch.cuda.empty_cache()
# the same bands driven by ONE host thread (pipeline.LocalBandGroup): the device sees the pure dependency graph
for N in (2, 8):
    grp = pipeline.LocalBandGroup(cw, ch, Ls, N, dev, fuse_sweeps=FUSE, plane_pipeline_min=PMIN)
    outs = None
    for rep in range(3):
        if rep == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        outs = grp.run(B, p, 0.0, 0.0, A, 0, 0, outs)
    torch.cuda.synchronize()
    res[f"band_group_{N}_bands_one_host_thread_ms"] = round((time.perf_counter() - t0) / 2 * 1e3, 2)
    res[f"band_group_{N}_equals_plan"] = bool(torch.equal(torch.cat(outs, dim=1), ref))
    grp.close()
    del outs, grp
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
