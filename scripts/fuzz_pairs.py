"""Randomised parity run: batched pairs with random geometry against the CPU oracle, on canvases where the fused sweep,
the source fusion and the zero-tile flags are all active (forced with STITCH_WAVEFRONT).  Test infrastructure (uses oracle/)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def run(seed, N, verbose=True):
    """-> (pairs compared, mismatches).  STITCH_WAVEFRONT etc. are read when a plan is created: set them before calling."""
    import torch
    import oracle_lib
    from computervisionimagestich2_amd import capi
    O = oracle_lib.Oracle()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    bad = done = 0
    for case in range(N):
        bad, done = _case(case, rng, O, capi, torch, dev, bad, done, verbose)
    return done, bad


SWITCHES = {"STITCH_WAVEFRONT": [None, "0", "1", "2"], "STITCH_RECOMPUTE": [None, "1", "2"], "STITCH_Y2": [None, "1"],
            "STITCH_COLLAPSE4": [None, "0"], "STITCH_NO_ZERO_TILES": [None, "1"], "STITCH_NO_SRC_FUSE": [None, "1"],
            "STITCH_GATE64": [None, None, "1"], "STITCH_COARSE": [None, None, "0", "40", "300"],
            "STITCH_SINGLE_FAST": [None, "1"], "STITCH_ODD_DEC": [None, None, "0"], "STITCH_C4_GEN": [None, None, "0"], "STITCH_COLLAPSE_PX": [None, None, "0"],
            "STITCH_Y1S": [None, None, "0", "2"], "STITCH_DEC7": [None, None, "0"], "STITCH_MOVER": [None, None, "0"], "STITCH_SRC_LONE_MPIX": [None, None, "0", "1"], "STITCH_C4_LOCKSTEP": [None, None, "0"], "STITCH_C4_SWIZZLE": [None, None, "0", "2"], "STITCH_XBYM": [None, "0", "1", "1"], "STITCH_COARSE_LDS": [None, None, "0"]}


def _case(case, rng, O, capi, torch, dev, bad, done, verbose):
    if True:
        general = os.environ.get("FUZZ_GENERAL") == "1"
        if general:
            # any size and parity, every tuning switch drawn per case (none of them may change a bit)
            big = os.environ.get("FUZZ_BIG") == "1"  # canvases up to 3072 x 2048, batches up to 8: the sizes where the fused sweep is chosen by itself
            ch = int(rng.integers(65, 2048 if big else 640))
            cw = int(rng.integers(max(66, ch // 2 + 1), min(2 * ch, 3072) + 1))
            for k, vals in SWITCHES.items():
                v = vals[int(rng.integers(0, len(vals)))]
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        else:
            ch = 64 * int(rng.integers(2, 9))
            cw = 2 * int(rng.integers(max(33, ch // 4 + 1), ch))  # even, within a factor two of the height (else the pyramid degenerates)
        B = int(rng.integers(1, 9 if os.environ.get("FUZZ_BIG") == "1" else 4))
        dtype = np.float32 if rng.random() < 0.6 else np.uint8
        opts = dict(sigma=2.0, blur_kind=0, level_rule=0, seam_rule=0)
        if general and rng.random() < 0.4:  # the ex6 variant's rules and other blur widths (0.3: below both filters' cut-off)
            opts = dict(sigma=float(rng.choice([0.3, 0.8, 1.5, 2.0, 3.5])), blur_kind=int(rng.integers(0, 2)), level_rule=int(rng.integers(0, 2)),
                        seam_rule=int(rng.integers(0, 2)))
        plan = capi.Plan(cw, ch, opts=opts, max_pairs=B)
        if general and rng.random() < 0.25:
            # stitch_dev_blend_*: two dense canvases (what blendTwoImages gets), each empty on one side -- the same plan, the
            # dense-pair form of the source-fused level 0 (or k_load_canvases under the switches that materialise it)
            A, Bc = O.synth(cw, ch, 2 * case, dtype), O.synth(cw, ch, 2 * case + 1, dtype)
            ca, cb = sorted(int(v) for v in rng.integers(0, cw + 1, 2))
            if rng.random() < 0.5:
                A[:, :, cb:] = 0
                Bc[:, :, :ca] = 0
            else:
                A[:, :, :ca] = 0
                Bc[:, :, cb:] = 0
            if rng.random() < 0.3:  # ragged holes: rows of zeros across a canvas (the middle row may become empty)
                r0 = int(rng.integers(0, ch))
                A[:, r0:r0 + int(rng.integers(1, 40)), :] = 0
            rc, ref, rs = O.blend(A, Bc, opts)
            out = plan.blend(torch.from_numpy(A).to(dev), torch.from_numpy(Bc).to(dev))
            try:
                seam = plan.status(0)
                grc = 0
            except capi.StitchError as e:
                grc = e.code
            if grc != rc:
                print("BLEND STATUS MISMATCH", case, grc, rc); bad += 1
            elif rc == 0:
                done += 1
                if seam.as_tuple() != rs.as_tuple() or not np.array_equal(out.cpu().numpy().view(np.uint8), ref.view(np.uint8)):
                    print("BLEND MISMATCH", case, (cw, ch, dtype.__name__), {k: os.environ.get(k) for k in SWITCHES}, opts); bad += 1
            plan.close()
            return bad, done
        items, refs = [], []
        for i in range(B):
            fw, fh = int(rng.integers(40, cw)), int(rng.integers(40, ch + 60))
            mw, mh = int(rng.integers(cw // 3, cw + 40)), int(rng.integers(ch // 2, ch + 40))
            ox, oy = int(rng.integers(-30, 30)), int(rng.integers(-30, 30))
            P = [1 + rng.normal() * 0.02, rng.normal() * 0.02, rng.normal() * 2e-5, -rng.uniform(0, cw - 40), rng.normal() * 0.02, 1 + rng.normal() * 0.02,
                 rng.normal() * 1e-5, rng.normal() * 15]
            offx, offy = (0.0, 0.0) if rng.random() < 0.5 else (float(np.float32(rng.normal() * 3)), float(np.float32(rng.normal() * 3)))
            F, M = O.synth(fw, fh, 2 * case + 1, dtype), O.synth(mw, mh, 2 * case, dtype)
            rc, ref = O.pair(F, P, offx, offy, M, ox, oy, cw, ch, opts)
            items.append((torch.from_numpy(F).to(dev), P, offx, offy, torch.from_numpy(M).to(dev), ox, oy,
                          torch.empty((3, ch, cw), dtype=torch.uint8 if dtype == np.uint8 else torch.float32, device=dev)))
            refs.append((rc, ref))
        outs = plan.pairs(items)
        for i in range(B):
            rc, ref = refs[i]
            try:
                plan.status(i)
                grc = 0
            except capi.StitchError as e:
                grc = e.code
            if grc != rc:
                print("STATUS MISMATCH", case, i, grc, rc); bad += 1
            elif rc == 0:
                same = np.array_equal(outs[i].cpu().numpy().view(np.uint8), ref.view(np.uint8))
                done += 1
                if not same:
                    d = outs[i].cpu().numpy().astype(np.float64) - ref.astype(np.float64)
                    print("PIXEL MISMATCH", case, i, (cw, ch, B, dtype.__name__), np.abs(d).max(), (d != 0).sum(),
                          {k: os.environ.get(k) for k in SWITCHES}, opts); bad += 1
        plan.close()
        return bad, done


if __name__ == "__main__":
    if os.environ.get("FUZZ_GENERAL") != "1":
        os.environ.setdefault("STITCH_WAVEFRONT", "2")
        os.environ.setdefault("STITCH_SINGLE_FAST", "1")  # the throughput forms at these small canvases too
    done, bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 7, int(sys.argv[2]) if len(sys.argv) > 2 else 30)
    print(f"fuzz: {done} pairs compared bit for bit, {bad} mismatches, fused levels forced = {os.environ.get('STITCH_WAVEFRONT')}")
    sys.exit(1 if bad else 0)
