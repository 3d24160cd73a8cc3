import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from computervisionimagestich2_amd import capi, pipeline
from oracle_lib import Oracle
o = Oracle(); gpu = torch.device("cuda:0")
def run(world, Ls, cw, ch, fw, fh, dtype, seedA, seedB, stored, **form):
    A, B = o.synth(fw, fh, seedA, dtype), o.synth(fw, fh, seedB, dtype)
    P = [1.0, 0.002, 1e-6, -(cw - fw) + 3.0, -0.001, 1.0, 5e-7, 1.5]
    rc, ref = o.pair(B, P, 0.0, 0.0, A, 0, 0, cw, ch)
    grp = pipeline.LocalBandGroup(cw, ch, Ls, world, gpu, **form)
    if stored:
        for b in grp.bands: b.band.set_level0(False)
    outs = grp.run(torch.from_numpy(B).to(gpu), P, 0.0, 0.0, torch.from_numpy(A).to(gpu), 0, 0)
    got = torch.cat(outs, dim=1).cpu().numpy()
    bad = np.argwhere(got.view(np.uint8) != ref.view(np.uint8))
    d = np.abs(got.astype(np.int64) - ref.astype(np.int64)).max() if len(bad) else 0
    print(f"world {world} Ls {Ls} {cw}x{ch} stored={stored} {form}: rc {rc} mismatches {len(bad)} maxdiff {d}", "rows", (bad[:,1].min(), bad[:,1].max()) if len(bad) else None, "cols", (bad[:,2].min(), bad[:,2].max()) if len(bad) else None)
    grp.close()

for (world, Ls, cw, ch) in [(2, 2, 365, 272), (2, 3, 730, 272)]:
    fw, fh = int(cw * 0.75), ch - 3
    run(world, Ls, cw, ch, fw, fh, np.float32, 47, 87, False, fuse_sweeps=False)
    run(world, Ls, cw, ch, fw, fh, np.float32, 47, 87, True, fuse_sweeps=False)
