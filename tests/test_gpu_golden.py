"""GPU: the HIP path against the golden vectors the reference produced (tests/golden/), device-resident entry
points against the oracle, and BASELINE.json's full-size configuration against the oracle (the oracle finishes a
6144x4096 pair in seconds on the GPU box's host cores, so the full-size check is a direct comparison)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
TOL_F32 = 1e-4  # north_star tolerance for float pixels


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def J():
    return json.load(open(os.path.join(G, "golden.json")))


@pytest.fixture(scope="module")
def Z():
    return np.load(os.path.join(G, "golden.npz"))


@pytest.fixture(scope="module")
def frames(J):
    from computervisionimagestich2_amd import bmp
    return [bmp.load_bmp(os.path.join(G, e["file"])) for e in J["input"]]


def test_projection_of_input_frames(st, gpu, J, Z, frames):
    for i, (f, e) in enumerate(zip(frames, J["project_input"])):
        p = st.project(f)
        assert sha(p) == e["sha256"] and int(p.sum()) == e["sum"]
    assert np.array_equal(st.project(Z["project_landscape_src"]), Z["project_landscape_out"])


@pytest.mark.parametrize("n", ["2", "4"])
def test_recorded_panorama_chain(st, gpu, J, frames, n):
    """BASELINE configs 1 and 3: Input/ frames, the reference's recorded stitch order and transforms; every
    intermediate mosaic and the final equalised panorama must carry the reference's hashes."""
    import torch
    from computervisionimagestich2_amd import capi
    run = J["runs"][n]
    proj = [capi.dev_project(torch.from_numpy(f).to(gpu)) for f in frames]
    result = proj[run["steps"][0]["start"]]
    for st_ in run["steps"]:
        plan = capi.Plan(st_["cw"], st_["ch"])
        result = plan.pair(proj[st_["src"]], st_["p"], st_["offx"], st_["offy"], result, st_["ox"], st_["oy"])
        plan.status()
        plan.close()
        assert sha(result.cpu().numpy()) == st_["out_sha256"]
    hist = torch.zeros(256, dtype=torch.int32, device=gpu)
    capi.dev_finish(result, 19.0, 20.0, hist)
    final = result.cpu().numpy()
    assert list(final.shape) == run["final_shape"] and sha(final) == run["final_sha256"]
    if n == "4":
        assert hist.cpu().numpy().tolist() == J["equalize_real"]["hist"]  # bit-exact histogram bins
    # the same chain through the host-side driver
    from computervisionimagestich2_amd import pipeline
    again = pipeline.stitch_chain([torch.from_numpy(f).to(gpu) for f in frames], run["steps"])
    assert sha(again.cpu().numpy()) == run["final_sha256"]



def test_whole_stitch_step_random_maps(st, gpu, oracle):
    """stitch_dev_step_* on seeded random forward / backward maps and frame / mosaic sizes (FUZZ_STEPS=n FUZZ_STEPS_SEED=s: a
    campaign): canvas, warp offsets and move offsets against the oracle's canvas_bbox (pinned to the reference), the mosaic
    against the oracle's pair on that geometry, both pixel types; point updates of the same step included."""
    import torch
    from computervisionimagestich2_amd import capi
    rng = np.random.default_rng(int(os.environ.get("FUZZ_STEPS_SEED", "20261009")))
    ran = 0
    n_cases = int(os.environ.get("FUZZ_STEPS", "16"))
    for case in range(n_cases):
        fw, fh = int(rng.integers(120, 420)), int(rng.integers(150, 520))
        mw, mh = int(rng.integers(fw, 2 * fw)), fh + int(rng.integers(-4, 12))
        # the new frame lands to the right of (or to the left of) the running mosaic, slightly rotated and sheared
        tx = float(rng.uniform(0.3, 0.8) * mw) if case % 3 else float(-rng.uniform(0.3, 0.7) * fw)
        ty = float(rng.uniform(-6, 6))
        a, b_, c, d = 1.0 + rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03), rng.uniform(-0.02, 0.02), 1.0 + rng.uniform(-0.02, 0.02)
        p_fwd = [a, b_, float(rng.uniform(-5e-5, 5e-5)), tx, c, d, float(rng.uniform(-5e-5, 5e-5)), ty]
        det = a * d - b_ * c
        ia, ib, ic, id_ = d / det, -b_ / det, -c / det, a / det  # inverse of the affine part: what RANSAC's second fit approximates
        p_bwd = [ia, ib, float(rng.uniform(-2e-6, 2e-6)), -(ia * tx + ib * ty), ic, id_, float(rng.uniform(-2e-6, 2e-6)), -(ic * tx + id_ * ty)]
        dtype = np.uint8 if case % 2 == 0 else np.float32
        F, M = oracle.synth(fw, fh, 50 + case, dtype), oracle.synth(mw, mh, 90 + case, dtype)
        min_x, min_y, cw, ch = oracle.canvas_bbox(fw, fh, p_fwd, mw, mh)
        g = capi.step_geometry(fw, fh, p_fwd, mw, mh)
        assert (g.min_x, g.min_y, g.cw, g.ch, g.ox, g.oy) == (min_x, min_y, cw, ch, int(min_x), int(min_y)), (case, p_fwd)
        if cw * ch > 6_000_000 or cw < 2 or ch < 2:
            continue
        rc, ref = oracle.pair(F, p_bwd, min_x, min_y, M, int(min_x), int(min_y), cw, ch)
        try:
            out, g2, seam = capi.dev_step(torch.from_numpy(F).to(gpu), p_fwd, p_bwd, torch.from_numpy(M).to(gpu))
            assert rc == 0, (case, rc)
            assert (g2.cw, g2.ch, g2.ox, g2.oy) == (cw, ch, int(min_x), int(min_y))
            assert np.array_equal(out.cpu().numpy().view(np.uint8), ref.view(np.uint8)), (case, cw, ch, str(dtype), p_fwd, p_bwd)
            ran += 1
        except capi.StitchError as e:
            assert rc != 0 and e.code == rc, (case, e.code, rc)
        x, y = rng.uniform(0, fw, 20).astype(np.float32), rng.uniform(0, fh, 20).astype(np.float32)
        for u, v in zip(capi.map_points(x, y, p_fwd, min_x, min_y), oracle.map_points(x, y, p_fwd, min_x, min_y)):
            assert np.array_equal(u, v)
    assert ran >= n_cases // 2, ran


@pytest.mark.parametrize("n", ["2", "4"])
def test_whole_stitch_step_from_the_forward_map(st, gpu, oracle, J, frames, n):
    """one device-resident call per stitch step, starting from the FORWARD map the reference's RANSAC
    produced (recorded, golden.json "p_fwd"): canvas sizing (ImageProcess.cpp:206-216), warp, move, blend, then the feature
    updates (:226-227).  Canvas, offsets and every intermediate mosaic must be the recorded run's."""
    import torch
    from computervisionimagestich2_amd import capi
    run = J["runs"][n]
    proj = [capi.dev_project(torch.from_numpy(f).to(gpu)) for f in frames]
    result = proj[run["steps"][0]["start"]]
    rng = np.random.default_rng(7)
    for st_ in run["steps"]:
        fr = proj[st_["src"]]
        g = capi.step_geometry(fr.shape[2], fr.shape[1], st_["p_fwd"], result.shape[2], result.shape[1])
        assert (g.cw, g.ch, g.min_x, g.min_y, g.ox, g.oy) == (st_["cw"], st_["ch"], np.float32(st_["offx"]), np.float32(st_["offy"]), st_["ox"], st_["oy"])
        result, g2, seam = capi.dev_step(fr, st_["p_fwd"], st_["p"], result)
        assert (g2.cw, g2.ch, g2.ox, g2.oy) == (g.cw, g.ch, g.ox, g.oy) and list(result.shape) == [3, st_["ch"], st_["cw"]]
        assert sha(result.cpu().numpy()) == st_["out_sha256"]
        # the feature updates of the same step, against the oracle (pinned to the reference in test_oracle_vs_reference.py)
        x, y = rng.uniform(0, fr.shape[2], 50).astype(np.float32), rng.uniform(0, fr.shape[1], 50).astype(np.float32)
        for a, b in zip(capi.map_points(x, y, st_["p_fwd"], g.min_x, g.min_y), oracle.map_points(x, y, st_["p_fwd"], g.min_x, g.min_y)):
            assert np.array_equal(a, b)
        for a, b in zip(capi.shift_points(x, y, g.ox, g.oy), oracle.shift_points(x, y, g.ox, g.oy)):
            assert np.array_equal(a, b)
    capi.dev_finish(result, 19.0, 20.0)
    assert sha(result.cpu().numpy()) == run["final_sha256"]
    # a buffer that is too small is refused before anything runs
    small = torch.empty(10, dtype=torch.uint8, device=gpu)
    rc = capi.lib().stitch_dev_step_u8(capi._dp(proj[0]), 384, 512, capi._map8(run["steps"][0]["p_fwd"]), capi._map8(run["steps"][0]["p"]),
                                       capi._dp(proj[1]), 384, 512, None, capi._dp(small), C.c_size_t(10), None, None, None)
    assert rc == capi.ERR_ARG


def test_blend_ex6_goldens_on_device(st, gpu, oracle, J):
    """The src/ex6 variant's whole blend on the HIP path (blur_kind = level_rule = seam_rule = 1) against the bytes the
    variant's own compiled function produced (golden.json "blend_ex6")."""
    from oracle_lib import EX6_OPTS
    for e in J["blend_ex6"]:
        w, h = e["w"], e["h"]
        A, B = oracle.synth(w, h, e["fa"]), oracle.synth(w, h, e["fb"])
        if e["a_left"]:
            A[:, :, (2 * w) // 3:] = 0
            B[:, :, : w // 3] = 0
        else:
            A[:, :, : w // 3] = 0
            B[:, :, (2 * w) // 3:] = 0
        got, _ = st.blend(A, B, EX6_OPTS)
        assert sha(got) == e["out_sha256"], e


def test_blend_synthetic_goldens(st, gpu, oracle, J):
    for e in J["blend_synth"]:
        w, h = e["w"], e["h"]
        A, B = oracle.synth(w, h, e["fa"]), oracle.synth(w, h, e["fb"])
        if e["a_left"]:
            A[:, :, (2 * w) // 3:] = 0
            B[:, :, : w // 3] = 0
        else:
            A[:, :, : w // 3] = 0
            B[:, :, (2 * w) // 3:] = 0
        out, seam = st.blend(A, B)
        assert list(seam.as_tuple()) == e["seam"] and sha(out) == e["out_sha256"], (w, h)


def test_equalize_golden(st, gpu, oracle, J, Z):
    e = J["equalize_sat"]
    sat = oracle.synth(e["w"], e["h"], e["frame_id"])
    sat[1] = np.maximum(sat[1], 240)
    sat[:, :60, :90] = 0
    out, hist = st.equalize(sat)
    assert hist.tolist() == e["hist"] and np.array_equal(out, Z["equalize_sat_out"])


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_device_synth_matches_oracle(st, gpu, oracle, dtype):
    import torch
    from computervisionimagestich2_amd import capi
    td = torch.uint8 if dtype == np.uint8 else torch.float32
    for (w, h, f) in [(300, 200, 0), (257, 129, 63), (4096, 16, 7)]:
        got = capi.dev_synth(w, h, f, td, gpu).cpu().numpy()
        ref = oracle.synth(w, h, f, dtype)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))


def test_device_resident_entry_points(st, gpu, oracle):
    import torch
    from computervisionimagestich2_amd import capi
    src = oracle.synth(320, 240, 2)
    t = torch.from_numpy(src).to(gpu)
    assert np.array_equal(capi.dev_project(t).cpu().numpy(), oracle.project(src))
    P = [1.01, 0.01, -1e-5, -60.5, 0.002, 0.99, 1e-6, 3.25]
    canvas = torch.zeros((3, 260, 400), dtype=torch.uint8, device=gpu)
    capi.dev_warp(t, P, -3.5, 2.25, canvas)
    assert np.array_equal(canvas.cpu().numpy(), oracle.warp(src, P, -3.5, 2.25, 400, 260))
    canvas.zero_()
    capi.dev_move(t, -17, 9, canvas)
    assert np.array_equal(canvas.cpu().numpy(), oracle.move(src, -17, 9, 400, 260))
    eq_ref, hist_ref, _ = oracle.equalize(src)
    hist = torch.zeros(256, dtype=torch.int32, device=gpu)
    eq = capi.dev_equalize(t.clone(), hist)
    assert np.array_equal(eq.cpu().numpy(), eq_ref) and np.array_equal(hist.cpu().numpy(), hist_ref)
    mixed = capi.dev_lummix(t.clone(), eq)
    assert np.array_equal(mixed.cpu().numpy(), oracle.lummix(src, eq_ref))
    f = torch.from_numpy(oracle.synth(100, 50, 3, np.float32) * 1.01).to(gpu)
    q = capi.dev_quantize(f)
    assert np.array_equal(q.cpu().numpy(), f.cpu().numpy().astype(np.uint8))


def test_two_plans_on_two_streams(st, gpu, oracle):
    """Independent pairs in flight on separate HIP streams (the batch configs) give the serial results."""
    import torch
    from computervisionimagestich2_amd import capi, pipeline
    fw, fh = 512, 384
    cw, ch = pipeline.config_canvas(fw)[0], fh
    plans = [capi.Plan(cw, ch) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    ins, outs = [], []
    for i in range(2):
        ins.append((oracle.synth(fw, fh, 2 * i, np.float32), oracle.synth(fw, fh, 2 * i + 1, np.float32), pipeline.config_map(i, fw)))
    dev_in = [(torch.from_numpy(a).to(gpu), torch.from_numpy(b).to(gpu)) for a, b, _ in ins]
    torch.cuda.synchronize()
    for rep in range(3):
        outs = []
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs.append(plans[i].pair(dev_in[i][1], ins[i][2], 0.0, 0.0, dev_in[i][0], 0, 0))
        torch.cuda.synchronize()
    for i in range(2):
        plans[i].status()
        rc, ref = oracle.pair(ins[i][1], ins[i][2], 0.0, 0.0, ins[i][0], 0, 0, cw, ch)
        assert rc == 0 and np.array_equal(outs[i].cpu().numpy().view(np.uint32), ref.view(np.uint32))
        plans[i].close()


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
def test_config2_full_size_against_oracle(st, gpu, oracle, dtype):
    """BASELINE.json configs[1]: one 4096x4096x3 pair -> 6144x4096 canvas, GPU vs CPU oracle, 1e-4 tolerance
    (float frames) / bit-exact (unsigned char frames)."""
    import torch
    from computervisionimagestich2_amd import capi, pipeline
    F = 4096
    cw, ch = pipeline.config_canvas(F)
    td = torch.float32 if dtype == np.float32 else torch.uint8
    A, B = capi.dev_synth(F, F, 0, td, gpu), capi.dev_synth(F, F, 1, td, gpu)
    p = pipeline.config_map(0, F)
    plan = capi.Plan(cw, ch)
    out = plan.pair(B, p, 0.0, 0.0, A, 0, 0)
    seam = plan.status()
    got = out.cpu().numpy()
    plan.close()
    rc, ref = oracle.pair(B.cpu().numpy(), p, 0.0, 0.0, A.cpu().numpy(), 0, 0, cw, ch)
    assert rc == 0
    assert seam.branch == 1 and 3000 < seam.start < 3200  # else-branch, seam near x ~ 3069
    if dtype == np.uint8:
        assert np.array_equal(got, ref)
    else:
        err = float(np.abs(got - ref).max())
        assert err <= TOL_F32, err
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"within tolerance ({err}) but not bit-equal"
        # size-independent properties of the float mosaic: clamped range, pure-a / pure-b regions far from the seam
        assert got.min() >= 0.0 and got.max() <= 255.0


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("n", [1, 3, 4])
def test_batched_plan_matches_single_pairs(st, gpu, oracle, dtype, n):
    """stitch_dev_pairs_*: n independent pairs through ONE launch sequence = n single-pair results (config 4's
    per-GPU shard).  Pairs differ in frames, maps, offsets; one of them has an odd frame size."""
    import torch
    from computervisionimagestich2_amd import capi
    cw, ch = 768, 384
    plan = capi.Plan(cw, ch, max_pairs=4)
    items, refs = [], []
    for i in range(n):
        fw, fh = (512, 384) if i != 1 else (501, 377)
        A, B = oracle.synth(fw, fh, 2 * i, dtype), oracle.synth(fw, fh, 2 * i + 1, dtype)
        P = [1.0, 0.002, 1e-6, -256.0 - 8.0 * i, -0.001, 1.0, 5e-7, 1.5]
        offx, offy, ox, oy = (0.0, 0.0, 0, 0) if i != 2 else (-3.25, 1.5, -3, 1)
        rc, ref = oracle.pair(B, P, offx, offy, A, ox, oy, cw, ch)
        assert rc == 0
        refs.append(ref)
        out = torch.empty((3, ch, cw), dtype=torch.uint8 if dtype == np.uint8 else torch.float32, device=gpu)
        items.append((torch.from_numpy(B).to(gpu), P, offx, offy, torch.from_numpy(A).to(gpu), ox, oy, out))
    outs = plan.pairs(items)
    for i in range(n):
        plan.status(i)
        got = outs[i].cpu().numpy()
        assert np.array_equal(got.view(np.uint8), refs[i].view(np.uint8)), i
    # a failing pair is reported for its own index only
    if n >= 2 and dtype == np.uint8:
        bad = list(items[1])
        bad[4] = torch.zeros_like(items[1][4])  # empty mosaic -> zero overlap for pair 1
        plan.pairs([items[0], tuple(bad)])
        plan.status(0)
        with pytest.raises(st.StitchError) as e:
            plan.status(1)
        assert e.value.code == st.capi.ERR_ZERO_OVERLAP
    plan.close()


def test_uint8_copy_of_float_mosaics(st, gpu, oracle):
    """stitch_pair_desc.out_u8: the level-0 collapse writes the float mosaic and, in the same pass, its unsigned char cast
    (CImg<unsigned char>(CImg<float>), truncation) -- what a sharded batch gathers; both collapse kernels (a canvas width that
    is a multiple of 4 and one that is not)."""
    import torch
    from computervisionimagestich2_amd import capi
    for (fw, fh, cw, ch) in [(704, 512, 1024, 512), (520, 384, 770, 384)]:
        plan = capi.Plan(cw, ch, max_pairs=2)
        items = []
        for i in range(2):
            A, B = oracle.synth(fw, fh, 2 * i, np.float32), oracle.synth(fw, fh, 2 * i + 1, np.float32)
            P = [1.0, 0.002, 1e-6, -(fw // 2) - 20.0 - 8.0 * i, -0.001, 1.0, 5e-7, 1.5]
            items.append((torch.from_numpy(B).to(gpu), P, 0.0, 0.0, torch.from_numpy(A).to(gpu), 0, 0,
                          torch.empty((3, ch, cw), dtype=torch.float32, device=gpu), torch.full((3, ch, cw), 7, dtype=torch.uint8, device=gpu)))
        plan.pairs(items)
        for q, it in enumerate(items):
            plan.status(q)
            rc, ref = oracle.pair(it[0].cpu().numpy(), it[1], 0.0, 0.0, it[4].cpu().numpy(), 0, 0, cw, ch)
            assert rc == 0 and np.array_equal(it[7].cpu().numpy().view(np.uint32), ref.view(np.uint32))
            assert np.array_equal(it[8].cpu().numpy(), ref.astype(np.uint8)) and torch.equal(it[8], capi.dev_quantize(it[7]))
        plan.close()
    # a single-level pyramid (3 x 2 canvas): the result is the top-level blend itself, out_u8 is written there too; and the
    # status of a pair beyond the LAST call's n is refused instead of answering with an earlier call's record
    plan = capi.Plan(3, 2, max_pairs=2)
    assert plan.levels == 1
    A, B = oracle.synth(3, 2, 0, np.float32), oracle.synth(3, 2, 1, np.float32)
    P = [1.0, 0.0, 0.0, -1.0, 0.0, 1.0, 0.0, 0.0]
    mk = lambda: (torch.from_numpy(B).to(gpu), P, 0.0, 0.0, torch.from_numpy(A).to(gpu), 0, 0,
                  torch.empty((3, 2, 3), dtype=torch.float32, device=gpu), torch.full((3, 2, 3), 7, dtype=torch.uint8, device=gpu))
    items = [mk(), mk()]
    plan.pairs(items)
    plan.status(1)
    rc, ref = oracle.pair(B, P, 0.0, 0.0, A, 0, 0, 3, 2)
    assert rc == 0
    for it in items:
        assert np.array_equal(it[7].cpu().numpy().view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(it[8].cpu().numpy(), ref.astype(np.uint8))
    plan.pairs(items[:1])
    plan.status(0)
    with pytest.raises(st.StitchError) as e:
        plan.status(1)  # pair 1 belongs to the call before the last one
    assert e.value.code == st.capi.ERR_ARG
    plan.close()


def test_config5_size_single_gpu_against_oracle(st, gpu, oracle):
    """BASELINE.json configs[4]'s problem size (one 16384x16384x3 f32 pair -> 24576x16384 canvas, 14 levels) on ONE
    GPU ("replicas only" for the multi-GPU form): the 30 GB plan fits HBM; compared in full with the
    oracle (about half a minute on the box's cores)."""
    import torch
    from computervisionimagestich2_amd import capi, pipeline
    F = 16384
    cw, ch = pipeline.config_canvas(F)
    A, B = capi.dev_synth(F, F, 0, torch.float32, gpu), capi.dev_synth(F, F, 1, torch.float32, gpu)
    p = pipeline.config_map(0, F)  # p[3] = -8192
    plan = capi.Plan(cw, ch)
    assert plan.levels == 14 and plan.level_w[-1] == 3 and plan.level_h[-1] == 2
    out = plan.pair(B, p, 0.0, 0.0, A, 0, 0)
    seam = plan.status()
    got = out.cpu().numpy()
    plan.close()
    Ah, Bh = A.cpu().numpy(), B.cpu().numpy()
    del A, B, out
    torch.cuda.empty_cache()
    rc, ref = oracle.pair(Bh, p, 0.0, 0.0, Ah, 0, 0, cw, ch)
    assert rc == 0 and seam.branch == 1
    err = float(np.abs(got - ref).max())
    assert err <= TOL_F32, err
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"within tolerance ({err}) but not bit-equal"


@pytest.mark.parametrize("fw,fh,cw,ch", [(1500, 1100, 2200, 1100), (1500, 1088, 2176, 1088)])
def test_fused_sweep_under_concurrent_load(st, gpu, oracle, fw, fh, cw, ch):
    """The band-pipeline sweep (inter-workgroup granule hand-offs) while the chip is busy: three batched plans of
    three pairs each in flight on three HIP streams, repeated, at sizes where the fused sweep is chosen automatically
    (levels >= 1024 x 1024): 2200 x 1100 has a partial last band and column block and a materialised level 0;
    2176 x 1088 (a multiple of 64 rows) runs source-fused with the implicit mask and the zero-tile flags.  Every output
    of every repetition must equal the oracle's."""
    import torch
    from computervisionimagestich2_amd import capi
    S, B, REPS = 3, 3, 4
    plans = [capi.Plan(cw, ch, max_pairs=B) for _ in range(S)]
    assert plans[0].fused_sweep_levels >= 1
    streams = [torch.cuda.Stream() for _ in range(S)]
    items, refs = [], []
    for s_ in range(S):
        lane = []
        for q in range(B):
            i = s_ * B + q
            A, Bf = oracle.synth(fw, fh, 2 * i, np.float32), oracle.synth(fw, fh, 2 * i + 1, np.float32)
            P = [1.0, 0.002, 1e-6, -700.0 - 8.0 * i, -0.001, 1.0, 5e-7, 1.5]
            rc, ref = oracle.pair(Bf, P, 0.0, 0.0, A, 0, 0, cw, ch)
            assert rc == 0
            refs.append(ref)
            lane.append((torch.from_numpy(Bf).to(gpu), P, 0.0, 0.0, torch.from_numpy(A).to(gpu), 0, 0,
                         torch.empty((3, ch, cw), dtype=torch.float32, device=gpu)))
        items.append(lane)
    torch.cuda.synchronize()
    for rep in range(REPS):
        for lane in items:
            for it in lane:
                it[7].fill_(-1.0)
        for s_ in range(S):
            with torch.cuda.stream(streams[s_]):
                plans[s_].pairs(items[s_])
        torch.cuda.synchronize()
        for s_ in range(S):
            for q in range(B):
                plans[s_].status(q)
                got = items[s_][q][7].cpu().numpy()
                assert np.array_equal(got.view(np.uint32), refs[s_ * B + q].view(np.uint32)), (rep, s_, q)
    for p in plans:
        p.close()


def test_captured_graph_replays_follow_their_inputs(st, gpu, oracle):
    """stitch_dev_pairs_* is capture-safe: its launch sequence (fused sweep included) recorded once into a HIP graph and
    replayed on NEW frame contents in the same buffers gives the oracle's result for the new contents, every time.  Nothing
    may survive from the captured or the previous run: hand-off tags, band-queue heads and the abort flag are cleared by
    kernels of the sequence itself, not told apart by per-launch arguments (which a replay freezes)."""
    import torch
    from computervisionimagestich2_amd import capi
    fw, fh, cw, ch = 1500, 1100, 2200, 1100
    B = 2
    plan = capi.Plan(cw, ch, max_pairs=B)
    assert plan.fused_sweep_levels >= 1
    P = [[1.0, 0.002, 1e-6, -700.0 - 8.0 * i, -0.001, 1.0, 5e-7, 1.5] for i in range(B)]
    items = [(torch.zeros((3, fh, fw), dtype=torch.float32, device=gpu), P[i], 0.0, 0.0,
              torch.zeros((3, fh, fw), dtype=torch.float32, device=gpu), 0, 0,
              torch.empty((3, ch, cw), dtype=torch.float32, device=gpu)) for i in range(B)]

    def load(seed):
        refs = []
        for i, it in enumerate(items):
            A, Bf = oracle.synth(fw, fh, seed + 2 * i, np.float32), oracle.synth(fw, fh, seed + 2 * i + 1, np.float32)
            rc, ref = oracle.pair(Bf, P[i], 0.0, 0.0, A, 0, 0, cw, ch)
            assert rc == 0
            refs.append(ref)
            it[0].copy_(torch.from_numpy(Bf))
            it[4].copy_(torch.from_numpy(A))
            it[7].fill_(-1.0)
        torch.cuda.synchronize()
        return refs

    load(0)
    s, g = torch.cuda.Stream(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        plan.pairs(items)  # warm: nothing is allocated or compiled inside the capture
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            plan.pairs(items)
    for rep, seed in enumerate((0, 10, 20, 10)):
        refs = load(seed)
        with torch.cuda.stream(s):
            g.replay()
        torch.cuda.synchronize()
        for i in range(B):
            plan.status(i)
            got = items[i][7].cpu().numpy()
            assert np.array_equal(got.view(np.uint32), refs[i].view(np.uint32)), (rep, i)
    plan.close()


def test_gray_and_fused_projection(st, gpu, oracle, J, frames):
    """toGrayScale + SIFT float staging, stand-alone and fused into the projection kernel."""
    for f, e, eg in zip(frames, J["project_input"], J["gray_input"]):
        proj, g, gf = st.capi.project_gray(f)
        assert sha(proj) == e["sha256"] and sha(g) == eg["sha256"]
        assert np.array_equal(gf, g.astype(np.float32))
        g2, gf2 = st.capi.gray(proj)
        assert np.array_equal(g2, g) and np.array_equal(gf2, gf)
    img = oracle.synth(301, 97, 12)
    og, ogf = oracle.gray(img)
    g, gf = st.capi.gray(img)
    assert np.array_equal(g, og) and np.array_equal(gf, ogf)


def test_bmp_golden_on_device(st, gpu, J, frames):
    """SURVEY.md 8(f) row 3 against the reference's own bytes (tests/golden: CImg::load_bmp / save_bmp outputs)."""
    import hashlib
    from computervisionimagestich2_amd import capi
    from oracle_lib import Oracle, make_bmp
    O = Oracle()
    for e, f in zip(J["input"], frames):
        got = capi.bmp_decode(open(os.path.join(G, e["file"]), "rb").read())
        assert sha(got) == e["sha256"]
    for e in J["bmp_load"]:
        got = capi.bmp_decode(make_bmp(O.synth(e["w"], e["h"], e["frame"]), **e["knobs"]))
        assert list(got.shape) == e["shape"] and sha(got) == e["sha256"], e
    for e in J["bmp_save"]:
        img = frames[e["input"] - 1] if "input" in e else O.synth(e["w"], e["h"], e["frame"])
        assert hashlib.sha256(capi.bmp_encode(img)).hexdigest() == e["file_sha256"], e
