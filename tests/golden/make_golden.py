#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Runs only in the container that has /root/reference: it drives oracle/_ref/libref_hotpath.so (the reference's
own functions compiled in place by oracle/Makefile) and oracle/_ref/libref_record.so (argument recorder), and
writes
    golden.json      hashes, sizes, recorded transforms, seam integers, scalar facts
    golden.npz       small arrays: primitive known-answer vectors, small full outputs, crops
    input/1..4.bmp   the reference's Input/ frames (data files; inputs of BASELINE configs 1 and 3)
Nothing here is reference source text.  The oracle is NOT used to produce expected values (except where a
quantity is not observable in the reference -- the 256 histogram bins -- which is then marked "from": "oracle"
and is only trusted because the equalised image it leads to equals the reference's byte for byte).

    python tests/golden/make_golden.py
"""
import ctypes as C
import hashlib
import json
import os
import re
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_DIR = "/root/reference"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    # the recorder must be in the global namespace BEFORE the reference library is loaded (symbol interposition)
    rec = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_record.so"), mode=C.RTLD_GLOBAL)
    from oracle_lib import REF_SO, Oracle, Reference
    R, O = Reference(), Oracle()
    J, Z = {}, {}
    rng = np.random.default_rng(20260101)

    # ---- inputs ---------------------------------------------------------------------------------------------
    os.makedirs(os.path.join(HERE, "input"), exist_ok=True)
    frames = []
    for i in range(1, 5):
        shutil.copyfile(f"{REF_DIR}/Input/{i}.bmp", os.path.join(HERE, "input", f"{i}.bmp"))
        frames.append(R.load_bmp(f"{REF_DIR}/Input/{i}.bmp"))
    J["input"] = [{"file": f"input/{i + 1}.bmp", "shape": list(f.shape), "sha256": sha(f), "sum": int(f.sum())}
                  for i, f in enumerate(frames)]

    # ---- P1/P2 projection -------------------------------------------------------------------------------------
    proj = [R.project(f) for f in frames]
    J["project_input"] = [{"sha256": sha(p), "sum": int(p.sum())} for p in proj]
    for i, p in enumerate(proj):
        Z[f"project_input{i + 1}_crop"] = p[:, 224:288, 160:224].copy()
    land_src = R.load_bmp(f"{REF_DIR}/Input2/1.bmp")[:, 300:500, 400:700].copy()  # W > H branch (Projection.cpp:30-49)
    Z["project_landscape_src"] = land_src
    Z["project_landscape_out"] = R.project(land_src)
    J["project_synth"] = []
    for (w, h, f) in [(257, 129, 3), (128, 300, 4), (64, 64, 5), (5, 3, 6), (2, 2, 7), (1, 7, 8)]:
        src = O.synth(w, h, f)  # the synthetic generator is an INPUT recipe, not an expected value
        out = R.project(src)
        J["project_synth"].append({"w": w, "h": h, "frame_id": f, "src_sha256": sha(src), "out_sha256": sha(out)})
        if w * h <= 40000:
            Z[f"project_synth_{w}x{h}"] = out
    # toGrayScale of the projected frames, the image SIFT consumes
    J["gray_input"] = [{"sha256": sha(R.gray(p)), "sum": int(R.gray(p).sum())} for p in proj]
    # canvas sizing + feature updates of the recorded 4-frame run are pinned through the runs' cw/ch below; explicit
    # vectors for the sizing function:
    J["canvas"] = []
    for (fw, fh, rw, rh, pp) in [(384, 512, 384, 512, [0.9724, -0.0398, 0.000149, 206.67, 0.00141, 1.00076, -1.2e-06, 4.55]),
                                 (384, 512, 607, 517, [1.08, 0.03, -0.0003, -232.1, -0.0045, 0.997, 7.3e-06, -3.9]),
                                 (600, 800, 1500, 820, [1.01, 0.2, 1e-5, 700.5, 0.1, 0.98, 2e-5, -30.25])]:
        J["canvas"].append({"fw": fw, "fh": fh, "rw": rw, "rh": rh, "p": pp, "out": list(R.canvas(fw, fh, pp, rw, rh))})
    # scalar bilinear taps
    src = O.synth(31, 17, 9)
    pts = [(0.0, 0.0), (29.5, 15.25), (30.0, 16.0), (12.75, 3.5), (29.999, 15.999), (0.25, 15.75)]
    J["bilinear"] = {"w": 31, "h": 17, "frame_id": 9,
                     "taps": [[x, y, c, int(R.bilinear(src, x, y, c))] for (x, y) in pts for c in range(3)]}

    # ---- recorded runs (W1-W3, B1-B6 on the real canvases, M1) --------------------------------------------------
    tmp = tempfile.mkdtemp()
    runs = {}
    for n in (4, 2):
        logp = os.path.join(tmp, f"log{n}.txt")
        dump = os.path.join(tmp, f"dump{n}")
        os.makedirs(dump)
        assert rec.rec_init(REF_SO.encode(), logp.encode(), dump.encode()) == 0
        final = R.pipeline(f"{REF_DIR}/Input/", n)
        rec.rec_close()
        steps = []
        for ln in open(logp).read().strip().split("\n"):
            kv = dict(re.findall(r"(\w+)=([^ ]+)", ln))
            if ln.startswith("warp"):
                cur = {"p": [float(v) for v in kv["p"].split(",")], "offx": float(kv["offx"]), "offy": float(kv["offy"]),
                       "cw": int(kv["cw"]), "ch": int(kv["ch"]), "fw": int(kv["sw"]), "fh": int(kv["sh"])}
            elif ln.startswith("move"):
                cur.update({"ox": int(kv["ox"]), "oy": int(kv["oy"]), "mw": int(kv["sw"]), "mh": int(kv["sh"])})
            else:
                k = len(steps)
                cw, ch = cur["cw"], cur["ch"]
                ld = lambda tag, w_, h_: np.fromfile(os.path.join(dump, f"step{k}_{tag}_{w_}x{h_}.raw"), np.uint8).reshape(3, h_, w_)
                fr, mo = ld("frame", cur["fw"], cur["fh"]), ld("mosaic", cur["mw"], cur["mh"])
                a, b, out = ld("a", cw, ch), ld("b", cw, ch), ld("out", cw, ch)
                # which projected input frame was warped / which one started the chain
                cur["src"] = next(i for i, p in enumerate(proj) if p.shape == fr.shape and np.array_equal(p, fr))
                if k == 0:
                    cur["start"] = next(i for i, p in enumerate(proj) if p.shape == mo.shape and np.array_equal(p, mo))
                cur.update({"a_sha256": sha(a), "b_sha256": sha(b), "out_sha256": sha(out), "out_mean": float(out.mean())})
                if n == 4:
                    Z[f"run4_step{k}_out_crop"] = out[:, ch // 2 - 32:ch // 2 + 32, cw // 2 - 48:cw // 2 + 48].copy()
                    if k == 2:
                        pre_final = out.copy()
                steps.append(cur)
        runs[n] = {"steps": steps, "final_shape": list(final.shape), "final_sha256": sha(final), "final_mean": float(final.mean())}
        if n == 4:
            Z["run4_final_crop"] = final[:, 200:328, 400:656].copy()
            final4 = final
    J["runs"] = {str(k): v for k, v in runs.items()}
    shutil.rmtree(tmp)

    # ---- E1-E3 on the real pre-equalisation mosaic and on a saturating synthetic image -----------------------------
    eq = R.equalize(pre_final)
    oeq, ohist, olut = O.equalize(pre_final)
    assert np.array_equal(eq, oeq), "oracle equalisation differs from the reference: bins would be untrustworthy"
    J["equalize_real"] = {"in_sha256": sha(pre_final), "out_sha256": sha(eq), "hist": ohist.tolist(), "lut": olut.tolist(),
                          "hist_lut_from": "oracle (not observable in the reference; pinned by out_sha256, which IS the reference's)"}
    sat = O.synth(300, 200, 11)
    sat[1] = np.maximum(sat[1], 240)
    sat[:, :60, :90] = 0
    eqs = R.equalize(sat)
    oeqs, oh2, ol2 = O.equalize(sat)
    assert np.array_equal(eqs, oeqs)
    Z["equalize_sat_out"] = eqs
    J["equalize_sat"] = {"w": 300, "h": 200, "frame_id": 11, "out_sha256": sha(eqs), "hist": oh2.tolist(), "lut": ol2.tolist(),
                         "hist_lut_from": "oracle, pinned by out_sha256"}
    # M1 is inline in matching(): observable only as final = mix(pre_final, equalize(pre_final))
    J["lummix_real"] = {"result_sha256": sha(pre_final), "final_sha256": sha(final4)}

    # ---- B3/B3' line filters and B4/B5 resizes: CImg primitives -----------------------------------------------------
    for N in (2, 3, 4, 5, 17, 540, 1081):
        x = (rng.random((1, 1, N)) * 255).astype(np.float32)
        Z[f"line_{N}_in"] = x
        Z[f"vanvliet_{N}_out"] = R.cimg_blur(x, 2.0, True)
        Z[f"deriche_{N}_out"] = R.cimg_blur(x, 2.0, False)
    x = (rng.random((2, 9, 17)) * 255).astype(np.float32)
    Z["blur2d_in"] = x
    Z["blur2d_vanvliet_out"] = R.cimg_blur(x, 2.0, True)
    Z["blur2d_deriche_out"] = R.cimg_blur(x, 2.0, False)
    x = np.zeros((1, 1, 700), np.float32)
    x[0, 0, :40] = 255.0  # long zero tail: the recursion decays through the float denormal range
    Z["line_denormal_in"] = x
    Z["vanvliet_denormal_out"] = R.cimg_blur(x, 2.0, True)
    for (w, h, w2, h2) in [(1081, 1, 540, 1), (67, 33, 33, 16), (4, 2, 2, 1), (3, 3, 1, 1), (16, 8, 8, 4), (135, 65, 67, 32)]:
        src = (rng.random((1, h, w)) * 255).astype(np.float32)
        Z[f"decimate_{w}x{h}_in"] = src
        Z[f"decimate_{w}x{h}_out"] = R.cimg_resize(src, w2, h2) if h2 != h else R.cimg_resize(src, w2, h)
    for (w, h, w2, h2) in [(2, 1, 4, 2), (33, 16, 67, 33), (540, 1, 1081, 1), (1, 1, 3, 3), (8, 4, 16, 8), (67, 32, 135, 65)]:
        src = (rng.random((1, h, w)) * 255).astype(np.float32)
        Z[f"expand_{w}x{h}_in"] = src
        Z[f"expand_{w}x{h}_out"] = R.cimg_resize(src, w2, h2)
    J["expand_tables"] = {}
    for (n_src, n_dst) in [(2, 4), (33, 67), (540, 1081), (3, 6), (263, 527)]:
        # the table is observable through the resize of a ramp: out[x] = idx + alpha for src[i] = i (exact in float here)
        ramp = np.arange(n_src, dtype=np.float32).reshape(1, 1, n_src)
        J["expand_tables"][f"{n_src}->{n_dst}"] = R.cimg_resize(ramp, n_dst, 1)[0, 0].astype(np.float64).tolist()

    # ---- B1-B6 whole blend on synthetic canvases (inputs are recipes; outputs are the reference's) ---------------------
    J["blend_synth"] = []
    for (w, h, fa, fb) in [(67, 33, 5, 6), (270, 131, 1, 2), (100, 64, 7, 8), (33, 67, 9, 10), (512, 512, 3, 4), (4, 2, 11, 12)]:
        for a_left in (True, False):
            A, B = O.synth(w, h, fa), O.synth(w, h, fb)
            if a_left:
                A[:, :, (2 * w) // 3:] = 0
                B[:, :, : w // 3] = 0
            else:
                A[:, :, : w // 3] = 0
                B[:, :, (2 * w) // 3:] = 0
            out = R.blend(A, B)
            rcs, s = O.seam(A, B)  # integers are plain counts of the inputs; the branch is pinned by out_sha256
            J["blend_synth"].append({"w": w, "h": h, "fa": fa, "fb": fb, "a_left": a_left, "out_sha256": sha(out),
                                     "seam": list(s.as_tuple())})
            if w * h <= 9000:
                Z[f"blend_{w}x{h}_{int(a_left)}_out"] = out

    # ---- the on-disk format: CImg::save_bmp bytes and CImg::load_bmp of layout variants --------
    # inputs are recipes (synthetic frame ids, oracle_lib.make_bmp knobs); hashes are of the reference's outputs
    from oracle_lib import make_bmp
    tmp = tempfile.mkdtemp()
    J["bmp_save"] = [{"w": w, "h": h, "frame": f, "file_sha256": hashlib.sha256(R.save_bmp_bytes(O.synth(w, h, f), tmp)).hexdigest()}
                     for (w, h, f) in [(257, 129, 5), (5, 3, 5), (1, 7, 5), (64, 64, 5), (1025, 4, 5), (1368, 17, 2)]]
    J["bmp_save"].append({"input": 1, "file_sha256": hashlib.sha256(R.save_bmp_bytes(frames[0], tmp)).hexdigest(),
                          "same_as_input_file": R.save_bmp_bytes(frames[0], tmp) == open(f"{REF_DIR}/Input/1.bmp", "rb").read()})
    J["bmp_load"] = []
    for (w, h) in [(257, 129), (5, 3), (1027, 3)]:
        for kw in [dict(), dict(bpp=32), dict(top_down=True), dict(header_size=108), dict(extra_gap=10), dict(size_field=0),
                   dict(truncate=7), dict(truncate=3 * w + 5), dict(bpp=32, top_down=True, header_size=124, extra_gap=3)]:
            img = R.load_bmp_bytes(make_bmp(O.synth(w, h, 3), **kw), tmp)
            J["bmp_load"].append({"w": w, "h": h, "frame": 3, "knobs": kw, "shape": list(img.shape), "sha256": sha(img)})
    shutil.rmtree(tmp)

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **Z)
    json.dump(J, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "golden.json"), os.path.getsize(os.path.join(HERE, "golden.json")), "bytes;",
          "golden.npz", os.path.getsize(os.path.join(HERE, "golden.npz")), "bytes")


if __name__ == "__main__":
    main()
