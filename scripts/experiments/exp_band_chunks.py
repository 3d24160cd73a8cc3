"""Config 5's size on ONE MI355X: (1) how a band's y-sweep launches scale with the number of columns they cover (one band of an
8-way split, level 0: stitch_band_reduce_y_fwd_cols / _bwd_cols over the whole pitch, a half, a quarter, ... ), (2) the 8-band
split from one host thread (LocalBandGroup) with the state handed over in 1, 2, 4 and 8 column chunks, each compared with the plan.
usage: exp_band_chunks.py [frame=16384] [split=4]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
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
ref = plan.pair(B, p, 0.0, 0.0, A, 0, 0)
torch.cuda.synchronize()
plan.close()

# (1) one band of 8 (rank 0: no resume), level 0: the two y sweeps over column ranges of shrinking width
band = capi.Band(cw, ch, 0, 8, Ls)
band.compose(B, p, 0.0, 0.0, A, 0, 0)
band.reduce_x(0)
g = band.geom[0]
pitch = g["pitch"]
f64 = dict(dtype=torch.float64, device=dev)
res["band_rows_level0"] = g["rows"]
res["y_sweep_ms_by_columns"] = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for cols in (pitch, pitch // 2, pitch // 4, pitch // 8, pitch // 16, pitch // 32):
    cols = cols // 128 * 128
    st = torch.zeros(4 * 7 * cols, **f64)
    sb = torch.zeros(3 * 7 * cols, **f64)
    rs = torch.zeros(3 * 7 * cols, **f64)  # the last rank's role needs no resume; rank 0 of 8 resumes from below: zeros will do for timing
    for rep in range(3):
        if rep == 1:
            e0.record()
        band.reduce_y_fwd_cols(0, 0, cols, None, st)
    e1.record()
    torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / 2
    for rep in range(3):
        if rep == 1:
            e0.record()
        band.reduce_y_bwd_cols(0, 0, cols, st, rs, sb)
    e1.record()
    torch.cuda.synchronize()
    res["y_sweep_ms_by_columns"][cols] = {"causal_ms": round(tf, 4), "anticausal_dec_ms": round(e0.elapsed_time(e1) / 2, 4)}
band.close()
del band
torch.cuda.empty_cache()

# (2) the 8-band split, one host thread, C column chunks
for C in (1, 2, 4, 8):
    grp = pipeline.LocalBandGroup(cw, ch, Ls, 8, dev, fuse_sweeps=False, col_chunks=C)
    outs = None
    for rep in range(3):
        if rep == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        outs = grp.run(B, p, 0.0, 0.0, A, 0, 0, outs)
    torch.cuda.synchronize()
    res[f"band_group_8_bands_{C}_chunk(s)_ms"] = round((time.perf_counter() - t0) / 2 * 1e3, 2)
    res[f"band_group_8_bands_{C}_chunk(s)_equals_plan"] = bool(torch.equal(torch.cat(outs, dim=1), ref))
    grp.close()
    del outs, grp
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
