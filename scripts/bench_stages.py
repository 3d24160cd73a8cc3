"""Stand-alone timings of the other rows of the path on one MI355X (projection, warp, move, equalise, mix, finish)
at 4096x4096 / 6144x4096, device-resident, torch.cuda events on the current stream."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from computervisionimagestich2_amd import capi, pipeline

dev = torch.device("cuda:0")
F = 4096
cw, ch = pipeline.config_canvas(F)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


res = {}
for td, s in ((torch.uint8, 1), (torch.float32, 4)):
    name = "u8" if s == 1 else "f32"
    src = capi.dev_synth(F, F, 0, td, dev)
    dst = torch.empty_like(src)
    ms = timeit(lambda: capi.dev_project(src, 15.0, dst))
    res[f"project_{name}"] = {"ms": ms, "MPix/s": F * F / ms / 1e3, "GB/s (2*N*3*s)": 2 * F * F * 3 * s / ms / 1e6}
    canvas = torch.zeros((3, ch, cw), dtype=td, device=dev)
    p = pipeline.config_map(0, F)
    ms = timeit(lambda: capi.dev_warp(src, p, 0.0, 0.0, canvas))
    res[f"warp_{name}"] = {"ms": ms, "MPix/s (canvas)": cw * ch / ms / 1e3}
    ms = timeit(lambda: capi.dev_move(src, 0, 0, canvas))
    res[f"move_{name}"] = {"ms": ms, "MPix/s (canvas)": cw * ch / ms / 1e3}
img = capi.dev_synth(cw, ch, 3, torch.uint8, dev)
hist = torch.zeros(256, dtype=torch.int32, device=dev)
work = img.clone()
P = cw * ch
ms = timeit(lambda: capi.dev_equalize(work, hist))
res["equalize_u8"] = {"ms": ms, "MPix/s": P / ms / 1e3, "GB/s (9*P)": 9 * P / ms / 1e6}
eq = work.clone()
ms = timeit(lambda: capi.dev_lummix(work, eq))
res["lummix_u8"] = {"ms": ms, "MPix/s": P / ms / 1e3, "GB/s (9*P)": 9 * P / ms / 1e6}
ms = timeit(lambda: capi.dev_finish(work, 19.0, 20.0, hist))
res["finish_u8 (equalise+mix fused)"] = {"ms": ms, "MPix/s": P / ms / 1e3, "GB/s (9*P)": 9 * P / ms / 1e6}
# the on-disk format either side of the path: 24-bit BMP <-> planar RGB, device-resident
for (w, h, tag) in ((F, F, "frame 4096x4096"), (cw, ch, "mosaic 6144x4096")):
    img = capi.dev_synth(w, h, 1, torch.uint8, dev)
    fil = capi.dev_bmp_encode(img)
    bi = capi.bmp_parse(fil[:54].cpu().numpy().tobytes(), fil.numel())
    back = torch.empty_like(img)
    ms = timeit(lambda: capi.dev_bmp_encode(img, fil))
    res[f"bmp_encode_u8 {tag}"] = {"ms": ms, "MPix/s": w * h / ms / 1e3, "GB/s (file + planar bytes)": (fil.numel() + img.numel()) / ms / 1e6}
    ms = timeit(lambda: capi.dev_bmp_decode(fil, bi, back))
    res[f"bmp_decode_u8 {tag}"] = {"ms": ms, "MPix/s": w * h / ms / 1e3, "GB/s (file + planar bytes)": (fil.numel() + img.numel()) / ms / 1e6}
    assert torch.equal(back, img)
# l-alpha-beta colour transfer: dominated by the serial float running sums the reference prescribes
for (w, h) in ((384, 512), (F, F)):
    a, b = capi.dev_synth(w, h, 1, torch.uint8, dev), capi.dev_synth(w, h, 8, torch.uint8, dev)
    o = torch.empty_like(a)
    ms = timeit(lambda: capi.dev_transfer(a, b, out=o), reps=5)
    res[f"transfer_u8 {w}x{h} (template of the same size)"] = {"ms": ms, "MPix/s": w * h / ms / 1e3}
# host-pointer pair (PCIe inclusive)
import numpy as np, time
A = capi.dev_synth(F, F, 0, torch.float32, dev).cpu().numpy()
B = capi.dev_synth(F, F, 1, torch.float32, dev).cpu().numpy()
capi.pair(B, p, 0.0, 0.0, A, 0, 0, cw, ch)
t = time.perf_counter()
capi.pair(B, p, 0.0, 0.0, A, 0, 0, cw, ch)
dt = time.perf_counter() - t
res["pair_f32 host pointers (PCIe + plan creation inclusive)"] = {"ms": dt * 1e3, "MPix/s (canvas)": cw * ch / dt / 1e6}
print(json.dumps(res, indent=1))
