"""Two bands of one pair driven in ONE process (no torch.distributed): REDUCE of the split levels with the fused and with
the plain anticausal sweep; where do the gathered level-Ls planes differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from computervisionimagestich2_amd import capi
from oracle_lib import Oracle
O = Oracle(); dev = torch.device("cuda:0")
fw, fh, cw, ch, Ls, N = 520, 384, 768, 384, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 2
A, B = O.synth(fw, fh, 4, np.uint8), O.synth(fw, fh, 5, np.uint8)
P = [1.0, 0.002, 1e-6, -(fw // 2) - 40.0, -0.001, 1.0, 5e-7, 1.5]
dA, dB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
f64 = dict(dtype=torch.float64, device=dev)

def reduce_all(plain):
    if plain: os.environ["STITCH_BAND_PLAIN"] = "1"
    else: os.environ.pop("STITCH_BAND_PLAIN", None)
    bands = [capi.Band(cw, ch, r, N, Ls) for r in range(N)]
    for b in bands: b.compose(dB, P, 0.0, 0.0, dA, 0, 0)
    lvl = {}
    for l in range(Ls):
        pitch = bands[0].geom[l]["pitch"]
        for b in bands: b.reduce_x(l)
        for q in range(7):
            stf = [torch.zeros(4 * pitch, **f64) for _ in range(N)]
            for r in range(N):
                bands[r].reduce_y_fwd(l, q, stf[r - 1][:3 * pitch].clone() if r > 0 else None, stf[r])
            stb = [torch.zeros(3 * pitch, **f64) for _ in range(N)]
            for r in range(N - 1, -1, -1):
                bands[r].reduce_y_bwd(l, q, stf[r], stb[r + 1].clone() if r < N - 1 else None, stb[r])
        g = bands[0].geom[l + 1]
        parts = []
        for b in bands:
            buf = torch.empty((7, g["rows"], g["w"]), dtype=torch.float32, device=dev)
            b.rows(l + 1, 2, 0, g["rows"], buf, True)
            parts.append(buf)
        lvl[l + 1] = torch.cat(parts, dim=1).cpu().numpy()
    for b in bands: b.close()
    return lvl

a, b = reduce_all(False), reduce_all(True)
for l in a:
    d = np.argwhere(a[l].view(np.uint32) != b[l].view(np.uint32))
    print("level", l, a[l].shape, "differing samples fused vs plain:", len(d), d[:8].tolist())
    if len(d):
        print("  planes", sorted(set(d[:, 0].tolist())), "rows", sorted(set(d[:, 1].tolist()))[:20], "max abs diff", float(np.abs(a[l] - b[l]).max()))
