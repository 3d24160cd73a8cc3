"""GPU parity: the HIP path, called through the C ABI (include/stitch.h), against the oracle on the same seeded
inputs.  Bar: bit-exact for unsigned-char images, histogram bins and seam integers; for float frames the
north-star tolerance is 1e-4 per channel -- the tests assert bit-equality first and report the max error."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

TOL_F32 = 1e-4  # BASELINE.json north_star: "within 1e-4 per channel for the warped/blended float pixels"

MAP2 = [1.0, 0.002, 1e-6, -2048.0, -0.001, 1.0, 5e-7, 1.5]  # config-2 backward map


def small_map(shift):
    return [1.0, 0.002, 1e-6, -float(shift), -0.001, 1.0, 5e-7, 1.5]


def two_canvases(oracle, w, h, fa, fb, dtype, a_left=True):
    A, B = oracle.synth(w, h, fa, dtype), oracle.synth(w, h, fb, dtype)
    if a_left:
        A[:, :, (2 * w) // 3:] = 0
        B[:, :, : w // 3] = 0
    else:
        A[:, :, : w // 3] = 0
        B[:, :, (2 * w) // 3:] = 0
    return A, B


@pytest.mark.parametrize("w,h", [(384, 512), (1210, 907), (257, 129), (128, 300), (64, 64), (5, 3), (2, 2), (1, 7), (1000, 1000), (1212, 907),
                                 (2048, 1024), (36, 20)])
def test_project_u8(st, gpu, oracle, w, h):
    src = oracle.synth(w, h, 3, np.uint8)
    assert np.array_equal(st.project(src), oracle.project(src))


@pytest.mark.parametrize("w,h", [(384, 512), (640, 360), (33, 67), (1000, 1000), (128, 300), (4, 4), (1028, 2050), (2052, 1026)])
def test_project_f32(st, gpu, oracle, w, h):
    src = oracle.synth(w, h, 5, np.float32)
    got, ref = st.project(src), oracle.project(src)
    assert np.abs(got - ref).max() <= TOL_F32
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_warp_and_move(st, gpu, oracle, dtype):
    src = oracle.synth(384, 512, 1, dtype)
    P = [1.0283414336297387, 0.040912708233406321, -0.00015778685586447313, -212.52122664118863,
         -0.0014527911613998329, 0.99923600064491935, 1.2704617705933406e-06, -4.5458940849449379]
    for offx, offy, cw, ch in [(0.0, 0.0, 607, 517), (-13.7, -3.2, 700, 530), (-230.579239, -4.68064785, 838, 522)]:
        got = st.warp(src, P, offx, offy, np.zeros((3, ch, cw), dtype))
        assert np.array_equal(got, oracle.warp(src, P, offx, offy, cw, ch))
        # read-modify-write: untouched pixels keep the caller's value
        pre = np.full((3, ch, cw), 7, dtype)
        got = st.warp(src, P, offx, offy, pre.copy())
        assert np.array_equal(got, oracle.warp(src, P, offx, offy, cw, ch, canvas=pre.copy()))
    for ox, oy in [(0, 0), (-13, -3), (40, 25), (-500, 0)]:
        got = st.move(src, ox, oy, np.zeros((3, 530, 700), dtype))
        assert np.array_equal(got, oracle.move(src, ox, oy, 700, 530))


def test_warp_nonfinite_map(st, gpu, oracle):
    src = oracle.synth(64, 48, 2, np.uint8)
    for P in ([float("nan")] + [0.0] * 7, [1e30, 0, 0, 0, 0, 1, 0, 0], [1, 0, 0, -0.5, 0, 1, 0, -0.5]):
        got = st.warp(src, P, 0.0, 0.0, np.zeros((3, 50, 70), np.uint8))
        assert np.array_equal(got, oracle.warp(src, P, 0.0, 0.0, 70, 50))


# (959: a width 4k+3 whose last workgroup of the seven-wavefront decimating sweep starts on an even column and straddles the row pitch --
# the round-4 fuzz campaign found its loader clamping the four-column group instead of leaving it in place)
@pytest.mark.parametrize("w,h", [(67, 33), (270, 131), (512, 512), (100, 64), (33, 67), (607, 517), (1081, 527), (4, 2), (3, 3), (959, 549), (1211, 515)])
@pytest.mark.parametrize("a_left", [True, False])
def test_blend_u8(st, gpu, oracle, w, h, a_left):
    A, B = two_canvases(oracle, w, h, 5, 6, np.uint8, a_left)
    rc, ref, rs = oracle.blend(A, B)
    assert rc == 0
    got, s = st.blend(A, B)
    assert s.as_tuple() == rs.as_tuple()
    assert np.array_equal(got, ref), f"{(got != ref).sum()} bytes differ"


@pytest.mark.parametrize("w,h", [(67, 33), (270, 131), (512, 512), (1081, 527), (959, 549)])
def test_blend_f32(st, gpu, oracle, w, h):
    A, B = two_canvases(oracle, w, h, 7, 8, np.float32)
    rc, ref, rs = oracle.blend(A, B)
    assert rc == 0
    got, s = st.blend(A, B)
    assert s.as_tuple() == rs.as_tuple()
    err = np.abs(got - ref).max()
    assert err <= TOL_F32, err
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"f32 not bit-equal, max err {err}"


@pytest.mark.parametrize("w", [256, 512, 768, 1024, 1280, 1536, 2048, 3072])
def test_collapse_row_end_tap_even_widths(st, gpu, oracle, w):
    """k_collapse4 covers whole rows of every even-width level with per-lane tap offsets (round 4): the first group of a row has the
    taps 0, 0, 0, 1, the last one ends on the source row's last sample, whose second tap is that sample again (CImg.h:29648) although the
    16-byte window behind it reaches into the next row.  Widths whose levels are multiples of 256, float canvases bit for bit (a wrong
    second tap under alpha = 0 shows only as the sign of a zero), with a black band so that exact zeros of both signs occur."""
    h = w // 2 + 2  # (the shorter side must reach the pyramid's top: ImageProcess.cpp:675-684)
    A, B = two_canvases(oracle, w, h, 11, 12, np.float32)
    A[:, 8:20, :] = 0.0  # (away from the middle row, which the seam scan needs)
    B[:, h - 26:h - 12, w // 2:] = 0.0
    rc, ref, rs = oracle.blend(A, B)
    assert rc == 0
    got, s = st.blend(A, B)
    assert s.as_tuple() == rs.as_tuple()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"{(got.view(np.uint32) != ref.view(np.uint32)).sum()} samples differ"


@pytest.mark.parametrize("w", [256, 512, 1024, 1280, 1536, 2048, 3072, 6144, 1030, 4422])
@pytest.mark.parametrize("probe", [0, 1])
def test_collapse_per_lane_taps_equal_the_reference_taps(st, gpu, w, probe):
    """One collapse level on synthetic planes by k_collapse4 (per-lane tap offsets over whole rows, the form every plan takes) and by
    k_collapse (every tap pair formed as CImg.h:29648 forms it), bit for bit.  probe 1 is built so that a column whose first tap is the
    source row's last sample comes out -0.0f only if that sample is taken twice: a build that takes the sample behind it in the window
    instead (-DSTITCH_C4_NO_ENDTAP) fails it at every width whose last tap is clamped (256, 1024, 1280, 1536, 2048, 3072: 3 x 72 samples each, verified)."""
    compared, bad = st.capi.dev_check_collapse_taps(w, 72, probe)
    assert compared == 3 * w * 72
    assert bad == 0


def test_collapse_covers_whole_rows(st, gpu):
    """The four-column path of the collapse covers every 256-column block of every level wide enough for it -- the one-column path it used
    to fall back to at the first and the last block of every even-width level is what the collapse over-fetched through in round 3 (101 MB
    per pair at level 0).  Even widths: [0, w rounded down to 256) with per-lane taps; an odd width (4421): all but the ragged end; with
    STITCH_C4_GEN=0 the fixed pattern leaves the first and the last block out (the A/B form)."""
    from computervisionimagestich2_amd import capi
    for (cw, ch) in [(6144, 4096), (2048, 1024), (4421, 2315), (1024, 512)]:
        plan = capi.Plan(cw, ch)
        for l in range(plan.levels - 1):
            w = plan.level_w[l]
            if w < 512:
                break
            xa, xb, gen = plan.collapse_range(l)
            assert (xa, xb, gen) == (0, (w - (w % 4 != 0) * 4) // 256 * 256 if w % 2 else w // 256 * 256, 1), (cw, l, w, xa, xb, gen)
        plan.close()


def test_blend_black_regions_denormals(st, gpu, oracle):
    """Large empty areas make the recursive filter decay through the float denormal range."""
    w, h = 1500, 600
    A = np.zeros((3, h, w), np.uint8)
    B = np.zeros((3, h, w), np.uint8)
    A[:, 250:350, 20:120] = oracle.synth(100, 100, 1)
    B[:, 250:350, 60:160] = oracle.synth(100, 100, 2)
    rc, ref, rs = oracle.blend(A, B)
    assert rc == 0
    got, s = st.blend(A, B)
    assert np.array_equal(got, ref)
    Af, Bf = A.astype(np.float32), B.astype(np.float32)
    rc, reff, _ = oracle.blend(Af, Bf)
    gotf, _ = st.blend(Af, Bf)
    assert np.array_equal(gotf.view(np.uint32), reff.view(np.uint32))


@pytest.mark.parametrize("opts", ["ex6"])
@pytest.mark.parametrize("w,h", [(270, 131), (600, 800)])
def test_blend_ex6_variant(st, gpu, oracle, opts, w, h):
    from oracle_lib import EX6_OPTS
    A, B = two_canvases(oracle, w, h, 9, 10, np.uint8)
    rc, ref, rs = oracle.blend(A, B, EX6_OPTS)
    assert rc == 0
    got, s = st.blend(A, B, EX6_OPTS)
    assert s.as_tuple() == rs.as_tuple()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_pair(st, gpu, oracle, dtype):
    fw, fh, cw, ch = 512, 384, 768, 384
    A, B = oracle.synth(fw, fh, 0, dtype), oracle.synth(fw, fh, 1, dtype)
    P = small_map(256)
    rc, ref = oracle.pair(B, P, 0.0, 0.0, A, 0, 0, cw, ch)
    assert rc == 0
    got, s = st.pair(B, P, 0.0, 0.0, A, 0, 0, cw, ch)
    if dtype == np.uint8:
        assert np.array_equal(got, ref)
    else:
        assert np.abs(got - ref).max() <= TOL_F32
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("gate64,single_fast", [("0", "0"), ("0", "1"), ("1", "0")])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("cw,ch,fw,fh", [(1081, 527, 384, 512), (607, 517, 384, 512), (838, 522, 384, 512), (1106, 579, 600, 500)])
def test_reference_canvas_sizes_take_the_fast_path(st, gpu, oracle, cw, ch, fw, fh, dtype, gate64, single_fast, monkeypatch):
    """The canvases the reference builds (ImageProcess.cpp:206-216: ceil of warped corners -- 607x517, 838x522, 1081x527
    for its own Input/ frames; 4421x2315 is covered at a quarter of its size) are never multiples of 64: the implicit level-0
    mask and the source-fused level 0 run per-plane bands with a masked partial last band and hold for ANY canvas size,
    for pairs and for stitch_blend_* (dense canvases read in place).  A lone pair of this size takes the implicit mask but
    keeps the materialised level 0 (shorter chains); STITCH_SINGLE_FAST=1 runs the source-fused forms (what a batch of such
    canvases runs).  STITCH_GATE64=1 is the old materialised sequence: same bits, and the oracle's."""
    import torch
    from computervisionimagestich2_amd import capi
    monkeypatch.setenv("STITCH_GATE64", gate64)
    monkeypatch.setenv("STITCH_SINGLE_FAST", single_fast)
    plan = capi.Plan(cw, ch)
    want = set() if gate64 == "1" else {"implicit_mask", "source_fused"}
    assert plan.fast_paths & {"implicit_mask", "source_fused"} == want, plan.fast_paths
    # a stitch step: frame warped in at the right, running mosaic on the left
    F, M = oracle.synth(fw, fh, 1, dtype), oracle.synth(cw - fw // 2, ch - 7, 2, dtype)
    P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
    rc, ref = oracle.pair(F, P, -0.25, -1.5, M, 0, -2, cw, ch)
    assert rc == 0
    out = plan.pair(torch.from_numpy(F).to(gpu), P, -0.25, -1.5, torch.from_numpy(M).to(gpu), 0, -2)
    plan.status()
    assert np.array_equal(out.cpu().numpy().view(np.uint8), ref.view(np.uint8))
    # blendTwoImages on the canvases themselves (what the C++ adaptor's blendTwoImages binds), device and host entry points
    A = oracle.warp(F, P, -0.25, -1.5, cw, ch)
    B = oracle.move(M, 0, -2, cw, ch)
    rc, refb, rs = oracle.blend(A, B)
    assert rc == 0 and np.array_equal(refb.view(np.uint8), ref.view(np.uint8))
    outb = plan.blend(torch.from_numpy(A).to(gpu), torch.from_numpy(B).to(gpu))
    assert plan.status().as_tuple() == rs.as_tuple()
    assert np.array_equal(outb.cpu().numpy().view(np.uint8), refb.view(np.uint8))
    plan.close()
    got, s = st.blend(A, B)
    assert s.as_tuple() == rs.as_tuple() and np.array_equal(got.view(np.uint8), refb.view(np.uint8))


@pytest.mark.parametrize("coarse", ["0", "24", "40", "146", "400"])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_coarse_levels_in_one_launch(st, gpu, oracle, dtype, coarse, monkeypatch):
    """k_coarse: every pyramid level from the first one with both sides <= STITCH_COARSE on (default 40) runs in ONE launch,
    one workgroup per pair -- REDUCE to the top, the top blend, the collapse back up -- instead of about six launches per
    level.  Same bits as the per-level launches (STITCH_COARSE=0) and as the oracle: landscape and portrait canvases, odd
    sizes (three-tap decimation rows / columns), a level of width 1 at the top, thresholds that give work-items one line
    (146), several lines (400: the coarse launch starts at 292 x 131 / 200 x 350 / 350 x 175) or a few samples (24), batches."""
    import torch
    from computervisionimagestich2_amd import capi
    monkeypatch.setenv("STITCH_COARSE", coarse)
    for (cw, ch, n) in [(1081, 527, 1), (400, 700, 2), (585, 263, 3), (200, 350, 1), (1170, 600, 1)]:
        plan = capi.Plan(cw, ch, max_pairs=n)
        L = plan.levels
        lw, lh = plan.level_w, plan.level_h
        want = next((l for l in range(1, L) if max(lw[l], lh[l]) <= int(coarse)), L)
        assert plan.coarse_from == (want if int(coarse) > 0 and want <= L - 2 else 0), (cw, ch, coarse, plan.coarse_from)
        assert ("coarse_levels" in plan.fast_paths) == (plan.coarse_from > 0)
        items, refs = [], []
        for i in range(n):
            fw, fh = max(40, cw * 2 // 3 - 7 * i), max(40, ch - 11 * i)
            F, M = oracle.synth(fw, fh, 2 * i + 1, dtype), oracle.synth(cw - fw // 2, ch, 2 * i, dtype)
            P = [1.0, 0.002, 1e-6, -(cw - fw - 2.0), -0.001, 1.0, 5e-7, 1.5 * i]
            rc, ref = oracle.pair(F, P, 0.0, 0.0, M, 0, 0, cw, ch)
            assert rc == 0, (cw, ch, i, rc)
            refs.append(ref)
            items.append((torch.from_numpy(F).to(gpu), P, 0.0, 0.0, torch.from_numpy(M).to(gpu), 0, 0,
                          torch.empty((3, ch, cw), dtype=torch.uint8 if dtype == np.uint8 else torch.float32, device=gpu)))
        outs = plan.pairs(items)
        for i in range(n):
            plan.status(i)
            assert np.array_equal(outs[i].cpu().numpy().view(np.uint8), refs[i].view(np.uint8)), (cw, ch, coarse, i)
        plan.close()


def test_blend_errors(st, gpu, oracle):
    A, B = two_canvases(oracle, 128, 64, 1, 2, np.uint8)
    A0 = A.copy()
    A0[0, 32, :] = 0
    with pytest.raises(st.StitchError) as e:
        st.blend(A0, B)
    assert e.value.code == st.capi.ERR_EMPTY_MIDROW
    B0 = B.copy()
    B0[0, 32, :] = 0
    with pytest.raises(st.StitchError) as e:
        st.blend(A, B0)
    assert e.value.code == st.capi.ERR_ZERO_OVERLAP
    with pytest.raises(st.StitchError) as e:
        st.blend(np.ones((3, 4, 256), np.uint8), np.ones((3, 4, 256), np.uint8))
    assert e.value.code == st.capi.ERR_PYRAMID
    assert oracle.blend(A0, B)[0] == -2 and oracle.blend(A, B0)[0] == -3


def test_project_column_strip_kernel_equals_pixel_kernel(st, gpu, oracle, monkeypatch):
    """k_project4 (four columns per work-item, per-column constants hoisted) against k_project (STITCH_PROJECT1=1) and the
    oracle, fused gray outputs included, on a frame whose rows are not a multiple of the strip height."""
    for (w, h) in [(1236, 1019), (1236, 1243)]:  # landscape (the axes swap roles) and portrait
        src = oracle.synth(w, h, 9, np.uint8)
        monkeypatch.delenv("STITCH_PROJECT1", raising=False)
        a = st.capi.project_gray(src)
        monkeypatch.setenv("STITCH_PROJECT1", "1")
        b = st.capi.project_gray(src)
        ref = oracle.project(src)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert np.array_equal(a[0], ref)


def test_project_degenerate_angles(st, gpu, oracle):
    """Cylinder angles at which r is infinite, (almost) zero, negative or not a number: k leaves the range the tiled kernels'
    hoisted-reciprocal quotient is exact for -- such frames never take those kernels (their source boxes exceed LDS, or the
    box test fails) -- and the result must still be the reference's arithmetic, NaN comparisons included."""
    for (w, h) in [(384, 512), (512, 384), (130, 77)]:
        for fov in [0.0, 1e-3, 89.99, 90.0, 120.0, 180.0, -15.0, float("nan"), float("inf")]:
            for dt in (np.uint8, np.float32):
                src = oracle.synth(w, h, 3, dt)
                assert np.array_equal(st.capi.project(src, fov).view(np.uint8), oracle.project(src, fov).view(np.uint8)), (w, h, fov, dt)


def test_project_random_sizes_sweep(st, gpu, oracle):
    """Seeded random frame sizes through every form of the projection: widths that are multiples of 4 (source box staged in
    LDS: pixel-interleaved for unsigned char, planar for float; portrait and landscape kernels) and widths that are not (the
    untiled kernel), tiles cut by the frame edge, frames smaller than one tile; fused gray planes included."""
    rng = np.random.default_rng(20261005)
    sizes = [(4 * int(rng.integers(1, 300)), int(rng.integers(1, 1300))) for _ in range(14)]
    sizes += [(int(rng.integers(1, 1300)), 4 * int(rng.integers(1, 300))) for _ in range(6)]
    sizes += [(int(rng.integers(1, 700)), int(rng.integers(1, 700))) for _ in range(6)]
    sizes += [(128, 16), (128, 17), (132, 15), (256, 33), (64, 32), (68, 31), (2044, 50), (48, 2100)]
    for i, (w, h) in enumerate(sizes):
        fov = (15.0, 15.0, 30.0, 7.5)[i % 4]  # the reference's angle, a wider and a narrower cylinder
        src = oracle.synth(w, h, 100 + i, np.uint8)
        ref = oracle.project(src, fov)
        dst, gray, gray_f32 = st.capi.project_gray(src, fov)
        assert np.array_equal(dst, ref), (w, h, fov)
        g, gf = oracle.gray(ref)
        assert np.array_equal(gray, g) and np.array_equal(gray_f32, gf), (w, h, fov)
        if i % 2 == 0:
            srcf = oracle.synth(w, h, 200 + i, np.float32)
            assert np.array_equal(st.capi.project(srcf, fov).view(np.uint32), oracle.project(srcf, fov).view(np.uint32)), (w, h, fov)
        if i % 3 == 0:  # device buffers at odd byte offsets (the tiled kernels need 4-byte aligned planes: the untiled one takes over)
            import torch
            n, k = 3 * h * w, 1 + i % 3
            buf_in, buf_out = torch.zeros(n + 8, dtype=torch.uint8, device=gpu), torch.zeros(n + 8, dtype=torch.uint8, device=gpu)
            buf_in[k:k + n] = torch.from_numpy(src).to(gpu).flatten()
            o = st.capi.dev_project(buf_in[k:k + n].view(3, h, w), fov, out=buf_out[3 - k:3 - k + n].view(3, h, w))
            assert np.array_equal(o.cpu().numpy(), ref), (w, h, fov, "offset", k)
            assert int(buf_out[:3 - k].sum()) == 0 and int(buf_out[3 - k + n:].sum()) == 0  # nothing written outside the image


@pytest.mark.parametrize("w,h", [(1081, 527), (300, 200), (64, 64), (7, 5), (2048, 1024)])
def test_equalize_lummix_finish(st, gpu, oracle, w, h):
    img = oracle.synth(w, h, 11, np.uint8)
    img[1] = np.maximum(img[1], 200)  # the 0.857 luma typo saturates bright pixels
    img[:, : h // 3, : w // 3] = 0
    ref, rhist, _ = oracle.equalize(img)
    got, hist = st.equalize(img)
    assert np.array_equal(hist, rhist)  # "bit-exact for the histogram bins"
    assert np.array_equal(got, ref)
    mixed = oracle.lummix(img, ref)
    assert np.array_equal(st.lummix(img, got), mixed)
    fin, hist2 = st.finish(img)
    assert np.array_equal(hist2, rhist)
    assert np.array_equal(fin, mixed)
    assert np.array_equal(st.lummix(img, got, 5.0, 6.0), oracle.lummix(img, ref, 5.0, 6.0))


def test_host_entry_points_random_sequence(st, gpu, oracle, monkeypatch):
    """The host-pointer entry points (what the C++ adaptor binds) take their workspaces from the plan cache: a seeded random
    SEQUENCE of blends and stitch steps over a dozen canvas sizes (more than the cache holds: evictions and re-creations), both
    pixel types, the ex6 options now and then, calls that must fail (an empty middle row, no overlap) in between, tuning switches
    flipped between calls (part of the cache key) -- every output, seam and error code against the oracle."""
    from oracle_lib import EX6_OPTS
    rng = np.random.default_rng(int(os.environ.get("FUZZ_HOST_SEED", "20261008")))
    sizes = [(int(rng.integers(96, 700)), int(rng.integers(96, 520))) for _ in range(12)]
    for call in range(int(os.environ.get("FUZZ_HOST", "40"))):
        cw, ch = sizes[int(rng.integers(0, len(sizes)))]
        dtype = np.uint8 if rng.random() < 0.6 else np.float32
        opts = EX6_OPTS if rng.random() < 0.15 else None
        for k in ("STITCH_NO_SRC_FUSE", "STITCH_COARSE", "STITCH_NO_ZERO_TILES"):
            monkeypatch.delenv(k, raising=False)
        flip = rng.random()
        if flip < 0.15:
            monkeypatch.setenv("STITCH_NO_SRC_FUSE", "1")
        elif flip < 0.3:
            monkeypatch.setenv("STITCH_COARSE", "0")
        elif flip < 0.4:
            monkeypatch.setenv("STITCH_NO_ZERO_TILES", "1")
        if rng.random() < 0.5:  # blendTwoImages on two dense canvases
            A, B = two_canvases(oracle, cw, ch, 3000 + call, 4000 + call, dtype, a_left=bool(rng.integers(0, 2)))
            kind = rng.random()
            if kind < 0.1:
                A[0, ch // 2, :] = 0  # empty middle row of a
            elif kind < 0.2:
                B[:, :, :] = 0        # nothing of b on the middle row
            rc, ref, seam = oracle.blend(A, B, opts) if opts is not None else oracle.blend(A, B)
            try:
                got, s_ = st.blend(A, B, opts) if opts is not None else st.blend(A, B)
                assert rc == 0, (call, cw, ch, "oracle refused", rc)
                assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)) and s_.as_tuple() == seam.as_tuple(), (call, "blend", cw, ch, str(dtype), flip)
            except st.capi.StitchError as e:
                assert rc != 0 and e.code == rc, (call, cw, ch, e.code, rc)
        else:  # one stitch step: warp + move + blend
            fw, fh = int(cw * rng.uniform(0.5, 0.8)), ch - int(rng.integers(0, 5))
            F, M = oracle.synth(fw, fh, 1000 + call, dtype), oracle.synth(fw, fh, 2000 + call, dtype)
            P = [1.0, float(rng.uniform(-0.004, 0.004)), float(rng.uniform(-2e-6, 2e-6)), -(cw - fw) + float(rng.uniform(0, 6)), float(rng.uniform(-0.002, 0.002)), 1.0,
                 float(rng.uniform(-1e-6, 1e-6)), float(rng.uniform(-2, 2))]
            offx, offy = float(np.float32(rng.uniform(-1, 1))), float(np.float32(rng.uniform(-1, 1)))
            ox, oy = int(rng.integers(-3, 1)), int(rng.integers(-2, 3))
            rc, ref = oracle.pair(F, P, offx, offy, M, ox, oy, cw, ch)
            try:
                got, s_ = st.pair(F, P, offx, offy, M, ox, oy, cw, ch)
                assert rc == 0, (call, cw, ch, "oracle refused", rc)
                assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (call, "pair", cw, ch, str(dtype), flip)
            except st.capi.StitchError as e:
                assert rc != 0 and e.code == rc, (call, cw, ch, e.code, rc)


def test_device_entry_points_at_odd_byte_offsets(st, gpu, oracle):
    """The device-resident entry points on unsigned char tensors that start 1-3 bytes into an allocation (the four-pixels-per-word
    forms of equalise / mix / finish / quantise, the word stores of the level-0 collapse, the tiled projection and the BMP kernels
    all have an alignment condition and a byte form behind it): results equal the oracle's and the guard bytes either side of
    every output stay untouched.  Plane sizes that are and are not multiples of 4."""
    import torch
    from computervisionimagestich2_amd import capi

    def view_at(arr, k, fill=0):
        """arr's bytes on the device starting k bytes into a buffer that is guarded by 8 bytes of `fill` either side"""
        n = arr.size
        buf = torch.full((n + 16,), fill, dtype=torch.uint8, device=gpu)
        buf[8 + k:8 + k + n] = torch.from_numpy(arr).to(gpu).flatten()
        return buf, buf[8 + k:8 + k + n].view(*arr.shape)

    def guards_intact(buf, k, n, fill):
        return bool((buf[:8 + k] == fill).all()) and bool((buf[8 + k + n:] == fill).all())

    for (w, h) in [(128, 96), (127, 93), (260, 64), (66, 33)]:
        img = oracle.synth(w, h, 31, np.uint8)
        img[1] = np.maximum(img[1], 180)
        ref, rhist, _ = oracle.equalize(img)
        mixed = oracle.lummix(img, ref)
        for k in (1, 2, 3):
            buf, v = view_at(img, k, 0xA5)
            hist = torch.zeros(256, dtype=torch.int32, device=gpu)
            capi.dev_equalize(v, hist)
            assert np.array_equal(v.cpu().numpy(), ref) and np.array_equal(hist.cpu().numpy(), rhist) and guards_intact(buf, k, img.size, 0xA5), ("equalize", w, h, k)
            buf2, v2 = view_at(img, (k + 1) % 4, 0x5A)
            capi.dev_lummix(v2, v)
            assert np.array_equal(v2.cpu().numpy(), mixed) and guards_intact(buf2, (k + 1) % 4, img.size, 0x5A), ("lummix", w, h, k)
            buf3, v3 = view_at(img, k, 0x3C)
            capi.dev_finish(v3)
            assert np.array_equal(v3.cpu().numpy(), mixed) and guards_intact(buf3, k, img.size, 0x3C), ("finish", w, h, k)
            f32 = oracle.synth(w, h, 5, np.float32)
            bufq, vq = view_at(np.zeros((3, h, w), np.uint8), k, 0x77)
            capi.dev_quantize(torch.from_numpy(f32).to(gpu), out=vq)
            assert np.array_equal(vq.cpu().numpy(), f32.astype(np.uint8)) and guards_intact(bufq, k, img.size, 0x77), ("quantize", w, h, k)
    # one stitch step whose output (and inputs) sit at odd offsets: the level-0 collapse stores words only into aligned rows
    for (cw, ch, k) in [(256, 192, 1), (260, 160, 3), (255, 131, 2)]:
        fw, fh = int(cw * 0.7), ch - 2
        F, M = oracle.synth(fw, fh, 61, np.uint8), oracle.synth(fw, fh, 62, np.uint8)
        P = small_map(cw - fw - 4)
        rc, ref = oracle.pair(F, P, 0.25, -0.5, M, 0, 1, cw, ch)
        assert rc == 0
        _, dF = view_at(F, k)
        _, dM = view_at(M, (k + 2) % 4)
        bufo, out = view_at(np.zeros((3, ch, cw), np.uint8), k, 0xC3)
        plan = capi.Plan(cw, ch)
        plan.pair(dF, P, 0.25, -0.5, dM, 0, 1, out=out)
        plan.status()
        assert np.array_equal(out.cpu().numpy(), ref) and guards_intact(bufo, k, 3 * ch * cw, 0xC3), ("pair", cw, ch, k)
        plan.close()


def test_other_rows_random_sizes(st, gpu, oracle):
    """Seeded random sizes through the rows around the blend (FUZZ_ROWS=n FUZZ_ROWS_SEED=s: a campaign): equalise + histogram, mix
    with random weights, fused finish, gray + SIFT staging, colour transfer with its twelve statistics, BMP encode / decode, warp and
    move with random offsets -- sizes of either parity and below one word (the byte forms of the four-pixels-per-word kernels),
    images with saturated and empty regions; every result against the oracle, bit for bit."""
    from computervisionimagestich2_amd import capi
    rng = np.random.default_rng(int(os.environ.get("FUZZ_ROWS_SEED", "20261007")))
    for case in range(int(os.environ.get("FUZZ_ROWS", "24"))):
        w, h = int(rng.integers(1, 420)), int(rng.integers(1, 330))
        if case % 5 == 0:
            w, h = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        img = oracle.synth(w, h, 300 + case, np.uint8)
        if case % 3 == 0:
            img[1] = np.maximum(img[1], 190)
        if case % 4 == 1:
            img[:, : h // 2, : w // 2] = 0
        ref, rhist, _ = oracle.equalize(img)
        got, hist = st.equalize(img)
        assert np.array_equal(hist, rhist) and np.array_equal(got, ref), ("equalize", w, h)
        num, den = float(rng.integers(1, 40)), float(rng.integers(1, 40))
        for (a, b) in ((19.0, 20.0), (num, den)):
            assert np.array_equal(st.lummix(img, got, a, b), oracle.lummix(img, ref, a, b)), ("lummix", w, h, a, b)
        fin, hist2 = st.finish(img)
        assert np.array_equal(hist2, rhist) and np.array_equal(fin, oracle.lummix(img, ref)), ("finish", w, h)
        g, gf = capi.gray(img)
        og, ogf = oracle.gray(img)
        assert np.array_equal(g, og) and np.array_equal(gf, ogf), ("gray", w, h)
        if w * h >= 4:
            tem = oracle.synth(int(rng.integers(2, 200)), int(rng.integers(2, 200)), 700 + case, np.uint8)
            rt, rstat = oracle.transfer(img, tem)
            gt, gstat = capi.transfer(img, tem)
            assert np.array_equal(gt, rt) and np.array_equal(gstat.view(np.uint32), rstat.view(np.uint32)), ("transfer", w, h, tem.shape)
        f = oracle.bmp_encode(img)
        assert capi.bmp_encode(img) == f, ("bmp_encode", w, h)
        rc, back = oracle.bmp_decode(f)
        assert rc == 0 and np.array_equal(capi.bmp_decode(f), back) and np.array_equal(back, img), ("bmp_decode", w, h)
        for dt in (np.uint8, np.float32):
            src = oracle.synth(w, h, 500 + case, dt)
            cw, ch = int(rng.integers(1, 500)), int(rng.integers(1, 400))
            P = [1.0 + rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(-2e-4, 2e-4), rng.uniform(-cw, w),
                 rng.uniform(-0.05, 0.05), 1.0 + rng.uniform(-0.05, 0.05), rng.uniform(-2e-4, 2e-4), rng.uniform(-ch, h)]
            offx, offy = float(np.float32(rng.uniform(-50, 50))), float(np.float32(rng.uniform(-50, 50)))
            pre = oracle.synth(cw, ch, 900 + case, dt)
            assert np.array_equal(st.warp(src, P, offx, offy, pre.copy()), oracle.warp(src, P, offx, offy, cw, ch, canvas=pre.copy())), ("warp", w, h, cw, ch)
            ox, oy = int(rng.integers(-w - 5, cw + 5)), int(rng.integers(-h - 5, ch + 5))
            assert np.array_equal(st.move(src, ox, oy, np.zeros((3, ch, cw), dt)), oracle.move(src, ox, oy, cw, ch)), ("move", w, h, cw, ch, ox, oy)


def test_equalize_every_colour(st, gpu, oracle):
    """The equalisation and mix kernels evaluate the colour transforms of byte pixels in integers (csrc/k_equalize.inc,
    ycc_terms): an image that holds each of the 2^24 colours once (plus a copy permuted so that the equalised partner of a
    pixel varies) must come out of equalise, mix and finish exactly as the oracle's double / float evaluation gives it; 4095
    columns wide as well, so that the byte kernels (plane size not a multiple of 4) see every colour too."""
    v = np.arange(256, dtype=np.uint8)
    R, G, B = np.meshgrid(v, v, v, indexing="ij")
    img = np.stack([R, G, B]).reshape(3, 4096, 4096).copy()
    ref, rhist, _ = oracle.equalize(img)
    got, hist = st.equalize(img)
    assert np.array_equal(hist, rhist) and np.array_equal(got, ref)
    other = np.ascontiguousarray(ref[:, ::-1, ::-1])  # every colour mixed with some other equalised colour
    assert np.array_equal(st.lummix(img, other), oracle.lummix(img, other))
    # the mix divides by `den` through a reciprocal where the host has proven that equal for every possible operand, and by an
    # IEEE divide otherwise (MixK): ordinary and odd parameter pairs alike must give the oracle's bytes
    for num, den in [(7.0, 9.0), (3.0, 4.0), (3.7, 0.9), (1.0e-3, 7.0e5), (5.0, 3.0), (1.0, 1.0), (2.5, -3.0), (0.1, 0.3)]:
        assert np.array_equal(st.lummix(img, other, num, den), oracle.lummix(img, other, num, den)), (num, den)
    fin, _ = st.finish(img)
    assert np.array_equal(fin, oracle.lummix(img, ref))
    odd = np.ascontiguousarray(img[:, :, :4095])
    ref2, rh2, _ = oracle.equalize(odd)
    got2, h2 = st.equalize(odd)
    assert np.array_equal(h2, rh2) and np.array_equal(got2, ref2)
    fin2, _ = st.finish(odd)
    assert np.array_equal(fin2, oracle.lummix(odd, ref2))


def test_blend_random_sizes_sweep(st, gpu, oracle):
    """Random canvas sizes (odd/even at every level -> fused and stand-alone decimation, implicit and materialised
    level-0 mask, both seam branches) for both pixel types, against the oracle."""
    rng = np.random.default_rng(4321)
    done = 0
    while done < 30:
        w, h = int(rng.integers(2, 700)), int(rng.integers(2, 500))
        if rng.random() < 0.3:
            h = max(64, (h // 64) * 64)  # implicit-mask path needs a multiple of 64
        dtype = np.uint8 if done % 2 == 0 else np.float32
        A, B = oracle.synth(w, h, int(rng.integers(0, 50)), dtype), oracle.synth(w, h, int(rng.integers(50, 100)), dtype)
        ca, cb = sorted(int(v) for v in rng.integers(0, w + 1, 2))
        if rng.random() < 0.5:
            A[:, :, cb:] = 0
            B[:, :, :ca] = 0
        else:
            A[:, :, :ca] = 0
            B[:, :, cb:] = 0
        rc, ref, rs = oracle.blend(A, B)
        if rc != 0:
            with pytest.raises(st.StitchError) as e:
                st.blend(A, B)
            assert e.value.code == rc  # same status codes as the oracle (-2 empty mid row, -3 no overlap, -4 pyramid)
            continue
        got, s = st.blend(A, B)
        assert s.as_tuple() == rs.as_tuple(), (w, h)
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (w, h, dtype)
        done += 1


@pytest.mark.parametrize("mode", ["0", "1", "4"])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_fused_sweep_modes_agree(st, gpu, oracle, mode, dtype, monkeypatch):
    """STITCH_WAVEFRONT=<n>: anticausal-x + causal-y sweeps fused (row-band pipeline with inter-workgroup hand-offs)
    for the n finest levels, or kept as separate kernels (0).  Same bits either way; sizes exercise partial bands
    and column blocks, the implicit level-0 mask and the materialised one.
    The host entry point takes its workspace from a cache: the tuning switches are part of the cache key, so the plan
    that ran MUST be one built under this mode -- asserted through stitch_plan_cache_query (which looks the workspace up
    under the current environment) right after the call, with other modes' plans for the same sizes already cached."""
    from computervisionimagestich2_amd import capi
    sizes = [(700, 448), (333, 250), (1200, 128), (130, 2050 // 2)]
    other = "1" if mode != "1" else "0"
    monkeypatch.setenv("STITCH_WAVEFRONT", other)  # plans of ANOTHER mode for the same sizes sit in the cache first
    A, B = two_canvases(oracle, 700, 448, 3, 4, dtype)
    st.blend(A, B)
    n_other, fused_other = capi.plan_cache_query(700, 448)
    assert n_other >= 1 and fused_other == min(int(other), 9 - 1)
    monkeypatch.setenv("STITCH_WAVEFRONT", mode)
    for (w, h) in sizes:
        A, B = two_canvases(oracle, w, h, 3, 4, dtype)
        rc, ref, rs = oracle.blend(A, B)
        if rc != 0:
            continue
        got, s = st.blend(A, B)
        assert s.as_tuple() == rs.as_tuple()
        assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (mode, w, h)
        n, fused = capi.plan_cache_query(w, h)
        levels = capi.pyramid_levels(w, h)[0]
        assert n >= 1, "the plan that just ran is not cached under the current switches: a stale plan was used"
        assert fused == min(int(mode), levels - 1), (mode, w, h, fused)
    plan = capi.Plan(700, 448)
    assert plan.fused_sweep_levels == min(int(mode), plan.levels - 1)
    plan.close()
    monkeypatch.delenv("STITCH_WAVEFRONT")
    plan = capi.Plan(700, 448)
    assert plan.fused_sweep_levels == 0  # auto: only levels of at least 1024 x 1024
    plan.close()
    plan = capi.Plan(6144, 4096, max_pairs=2)
    assert plan.fused_sweep_levels == 2
    plan.close()
    plan = capi.Plan(2048, 1024)  # a lone pair is faster with separate sweeps
    assert plan.fused_sweep_levels == 0
    plan.close()


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_lone_fused_sweep_equals_the_oracle(st, gpu, oracle, dtype, monkeypatch):
    """STITCH_XBYM=1: ONE pair in flight runs the anticausal-x + causal-y sweep of its first four levels as one launch of
    five-wavefront bands (k_vv_xby_m: x chain, y chain, loader, storer, courier; y state band -> band through the granules of
    k_vv_xbyf) -- by default only from 20 MPix per plane.  Sizes: partial last band and tile, one band only, odd sides, a plane
    narrower than a tile row of bands is wide; pairs (source-fused or not) and dense blends; against the oracle bit for bit,
    and the plan reports the form."""
    import torch
    from computervisionimagestich2_amd import capi
    monkeypatch.setenv("STITCH_XBYM", "1")
    for (w, h) in [(700, 448), (333, 250), (1200, 130), (130, 1025), (959, 549), (64, 64), (257, 63)]:
        A, B = two_canvases(oracle, w, h, 3, 4, dtype)
        rc, ref, rs = oracle.blend(A, B)
        if rc != 0:
            continue
        plan = capi.Plan(w, h)
        assert "fused_sweep" in plan.call_forms(1), (w, h)
        got = plan.blend(torch.from_numpy(A).to(gpu), torch.from_numpy(B).to(gpu))
        assert plan.status().as_tuple() == rs.as_tuple()
        assert np.array_equal(got.cpu().numpy().view(np.uint8), ref.view(np.uint8)), (w, h, dtype)
        plan.close()
    monkeypatch.setenv("STITCH_XBYM", "0")
    plan = capi.Plan(6144, 4096)
    assert "fused_sweep" not in plan.call_forms(1)
    plan.close()
    monkeypatch.delenv("STITCH_XBYM")
    plan = capi.Plan(6144, 4096)  # auto: level 0 of 25 MPix
    assert "fused_sweep" in plan.call_forms(1)
    plan.close()
    plan = capi.Plan(4421, 2315)  # 10 MPix: separate sweeps
    assert "fused_sweep" not in plan.call_forms(1)
    plan.close()


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_dev_blend_into_one_of_its_inputs(st, gpu, oracle, dtype, monkeypatch):
    """stitch_dev_blend_* with the output buffer on top of an input (ImageProcess.cpp:230 assigns the blend's result to the
    panorama it was computed from).  A source-fused level 0 reads both canvases again while the collapse writes the output, so a
    call whose output overlaps an input takes the materialised level 0 (k_load_canvases copies first): same bits as the oracle,
    in place over a, in place over b, and into a buffer that overlaps a by one row."""
    import torch
    from computervisionimagestich2_amd import capi
    monkeypatch.setenv("STITCH_SRC_LONE_MPIX", "1")  # one pair in flight is source-fused from 1 MPix of canvas
    w, h = 1400, 800
    A, B = two_canvases(oracle, w, h, 7, 8, dtype)
    rc, ref, rs = oracle.blend(A, B)
    assert rc == 0
    plan = capi.Plan(w, h)
    assert "source_fused" in plan.call_forms(1)
    for where in ("separate", "over_a", "over_b", "one_row_into_a"):
        a, b = torch.from_numpy(A).to(gpu), torch.from_numpy(B).to(gpu)
        if where == "one_row_into_a":
            big = torch.empty(2 * 3 * h * w, dtype=a.dtype, device=gpu)
            a2 = big[3 * h * w - w:2 * 3 * h * w - w].view(3, h, w)
            a2.copy_(a)
            out = big[:3 * h * w].view(3, h, w)  # its last row is a2's first
            got = plan.blend(a2, b, out=out)
        else:
            out = {"separate": None, "over_a": a, "over_b": b}[where]
            got = plan.blend(a, b, out=out)
        assert plan.status().as_tuple() == rs.as_tuple()
        assert np.array_equal(got.cpu().numpy().view(np.uint8), ref.view(np.uint8)), where
    plan.close()


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_lone_fused_sweep_with_zero_tile_flags(st, gpu, oracle, dtype, monkeypatch):
    """k_vv_xby_m with zero-tile flags (a plan that owns flags -- here a batched one -- asked for ONE pair under STITCH_XBYM=1, at a
    size where a level has more bands than the chain + loader + storer sweeps take: > 1024 at the source-fused level 0, > 340 above):
    the loader reads flagged tiles from the plan's zero page, the storer records full all-zero tiles instead of storing them, the
    anticausal y sweep + decimation reads the flags.  A tall narrow canvas (148 / 74 bands per plane at levels 0 / 1; partial last
    band and tile) whose frame covers a third of it, so that most tiles of three planes are zero; against the oracle bit for bit."""
    import torch
    from computervisionimagestich2_amd import capi
    monkeypatch.setenv("STITCH_XBYM", "1")
    cw, ch, fw, fh = 1100, 9450, 420, 9000
    F, M = oracle.synth(fw, fh, 21, dtype), oracle.synth(cw - 300, ch - 5, 22, dtype)
    P = [1.0, 0.002, 1e-6, -(cw - fw - 3.0), -0.001, 1.0, 5e-7, -3.5]
    opts = dict(sigma=2.0, blur_kind=0, level_rule=1, seam_rule=0)  # levels from the SHORTER side (the canvas is nine times as tall as wide)
    rc, ref = oracle.pair(F, P, -0.25, -1.5, M, 0, -2, cw, ch, opts=opts)
    assert rc == 0, rc
    plan = capi.Plan(cw, ch, opts=opts, max_pairs=2)
    assert plan.fused_sweep_levels >= 1 and "fused_sweep" in plan.call_forms(1)
    out = torch.empty((3, ch, cw), dtype=torch.from_numpy(F).dtype, device=gpu)
    plan.pairs([(torch.from_numpy(F).to(gpu), P, -0.25, -1.5, torch.from_numpy(M).to(gpu), 0, -2, out)])
    plan.status(0)
    assert np.array_equal(out.cpu().numpy().view(np.uint8), ref.view(np.uint8))
    plan.close()


@pytest.mark.parametrize("lds", ["0", None])
def test_coarse_levels_in_lds_and_in_global_memory_agree(st, gpu, oracle, lds, monkeypatch):
    """The coarse levels of a pyramid run in one launch per pair: k_coarse_lds (every level in LDS, the default where they fit) or
    k_coarse (STITCH_COARSE_LDS=0, planes in global memory; also the form for thresholds whose levels do not fit: STITCH_COARSE=300).
    Same bits, against the oracle."""
    if lds is None:
        monkeypatch.delenv("STITCH_COARSE_LDS", raising=False)
    else:
        monkeypatch.setenv("STITCH_COARSE_LDS", lds)
    for coarse in (None, "64", "300"):
        if coarse is None:
            monkeypatch.delenv("STITCH_COARSE", raising=False)
        else:
            monkeypatch.setenv("STITCH_COARSE", coarse)
        for (w, h) in [(1081, 527), (333, 250), (130, 1025), (67, 33), (40, 40), (41, 23)]:
            for dtype in (np.uint8, np.float32):
                A, B = two_canvases(oracle, w, h, 5, 6, dtype)
                rc, ref, rs = oracle.blend(A, B)
                if rc != 0:
                    continue
                got, s = st.blend(A, B)
                assert s.as_tuple() == rs.as_tuple()
                assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (lds, coarse, w, h, dtype)


@pytest.mark.parametrize("no_src", ["0", "1"])
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("seam_rule", [0, 1])
def test_source_fused_level0_edge_cases(st, gpu, oracle, dtype, seam_rule, no_src, monkeypatch):
    """Pairs on a canvas whose height is a multiple of 64 run source-fused (level 0 gathered from the frames through
    k_src_index's offsets, k_compose never runs); STITCH_NO_SRC_FUSE=1 keeps the materialised level 0.  Both must give
    the oracle's bits for: a frame that maps partly outside the canvas and the canvas partly outside the frame, odd
    frame sizes, negative and positive integer shifts of the mosaic, fractional warp offsets, a strongly non-affine
    map (xy term), a mosaic smaller than the canvas, and the three-channel seam rule."""
    import torch
    from computervisionimagestich2_amd import capi
    monkeypatch.setenv("STITCH_NO_SRC_FUSE", no_src)
    monkeypatch.setenv("STITCH_SINGLE_FAST", "1")  # the source-fused form at this small size too (by default chosen from 8 MPix per launch on)
    cw, ch = 832, 448
    opts = dict(sigma=2.0, blur_kind=0, level_rule=0, seam_rule=seam_rule)
    cases = [
        # fw, fh, map, offx, offy, mw, mh, ox, oy
        (512, 448, [1.0, 0.002, 1e-6, -300.0, -0.001, 1.0, 5e-7, 1.5], 0.0, 0.0, 512, 448, 0, 0),
        (501, 377, [0.97, 0.01, 2e-5, -250.5, 0.02, 1.03, -1e-5, -20.25], -3.25, 1.5, 600, 400, -7, 5),
        (640, 300, [1.1, -0.05, 0.0, -330.0, 0.0, 0.8, 0.0, 40.0], 2.75, -0.5, 450, 470, 11, -9),
    ]
    plan = capi.Plan(cw, ch, opts=opts, max_pairs=len(cases))
    items, refs = [], []
    for i, (fw, fh, P, offx, offy, mw, mh, ox, oy) in enumerate(cases):
        F, M = oracle.synth(fw, fh, 2 * i + 1, dtype), oracle.synth(mw, mh, 2 * i, dtype)
        rc, ref = oracle.pair(F, P, offx, offy, M, ox, oy, cw, ch, opts=opts)
        assert rc == 0, (i, rc)
        refs.append(ref)
        out = torch.empty((3, ch, cw), dtype=torch.uint8 if dtype == np.uint8 else torch.float32, device=gpu)
        items.append((torch.from_numpy(F).to(gpu), P, offx, offy, torch.from_numpy(M).to(gpu), ox, oy, out))
    outs = plan.pairs(items)
    for i in range(len(cases)):
        plan.status(i)
        assert np.array_equal(outs[i].cpu().numpy().view(np.uint8), refs[i].view(np.uint8)), i
    # a single pair through the same plan (batch of one)
    out1 = plan.pairs(items[1:2])
    plan.status(0)
    assert np.array_equal(out1[0].cpu().numpy().view(np.uint8), refs[1].view(np.uint8))
    plan.close()


BMP_VARIANTS = [dict(), dict(bpp=32), dict(top_down=True), dict(header_size=108), dict(extra_gap=10), dict(size_field=0),
                dict(truncate=7), dict(truncate="row"), dict(bpp=32, top_down=True, header_size=124, extra_gap=3), dict(extra_gap=1)]


@pytest.mark.parametrize("w,h", [(257, 129), (5, 3), (1, 7), (64, 64), (2, 2), (1, 1), (1024, 5), (1025, 4), (2051, 3), (1368, 17)])
def test_bmp_decode_encode(st, gpu, oracle, w, h):
    """the on-disk format either side of the path.  Decode = CImg::load_bmp for every header
    layout its 24/32-bit branch distinguishes (and files that end early), encode = the bytes CImg::save_bmp writes;
    widths around the 1024-pixel workgroup segment and with every row padding; odd pixel-data alignment."""
    import torch
    from computervisionimagestich2_amd import capi
    from oracle_lib import make_bmp
    img = oracle.synth(w, h, 5)
    for kw in BMP_VARIANTS:
        kw = dict(kw)
        if kw.get("truncate") == "row":
            kw["truncate"] = 3 * w + 5
        data = make_bmp(img, **kw)
        if len(data) < 54:
            continue
        rc, ref = oracle.bmp_decode(data)
        assert rc == 0
        got = capi.bmp_decode(data)
        assert got.shape == ref.shape and np.array_equal(got, ref), (w, h, kw)
        if not kw.get("truncate"):
            assert np.array_equal(ref, img)
    ref_file = oracle.bmp_encode(img)
    assert capi.bmp_encode(img) == ref_file
    # device-resident twins, and the round trip
    d_file = capi.dev_bmp_encode(torch.from_numpy(img).to(gpu))
    assert d_file.cpu().numpy().tobytes() == ref_file
    back = capi.dev_bmp_decode(d_file, capi.bmp_parse(ref_file, len(ref_file)))
    assert np.array_equal(back.cpu().numpy(), img)
    # a file image that does not start on a 4-byte boundary of device memory
    shifted = torch.empty(d_file.numel() + 1, dtype=torch.uint8, device=gpu)
    shifted[1:].copy_(d_file)
    back = capi.dev_bmp_decode(shifted[1:], capi.bmp_parse(ref_file, len(ref_file)))
    assert np.array_equal(back.cpu().numpy(), img)


def test_bmp_refusals(st, gpu):
    from computervisionimagestich2_amd import capi
    from oracle_lib import make_bmp
    img = np.zeros((3, 4, 4), np.uint8)
    good = bytearray(make_bmp(img))
    for patch in [(0, b"XM"), (0x1C, bytes([8, 0])), (0x1E, bytes([1, 0, 0, 0])), (0x12, bytes([0, 0, 0, 0]))]:
        bad = bytearray(good)
        bad[patch[0]:patch[0] + len(patch[1])] = patch[1]
        with pytest.raises(st.StitchError) as e:
            capi.bmp_decode(bytes(bad))
        assert e.value.code == st.capi.ERR_ARG
    with pytest.raises(st.StitchError):
        capi.bmp_decode(bytes(good[:40]))


@pytest.mark.parametrize("sw,sh,tw,th", [(384, 512, 384, 512), (300, 200, 97, 61), (257, 129, 640, 360), (16, 16, 5, 3), (1, 1, 2, 2), (1000, 1000, 333, 77)])
def test_colour_transfer(st, gpu, oracle, sw, sh, tw, th):
    """SURVEY.md 8(f) row 4 (transfer.cpp; parity unpinned, see include/stitch.h): the HIP path equals the CPU
    restatement bit for bit -- the uchar result AND the twelve float statistics, whose serial float running sums are the
    order-sensitive part (sizes that are not multiples of the 256-sample staging block included)."""
    import torch
    from computervisionimagestich2_amd import capi
    src, tem = oracle.synth(sw, sh, 1), oracle.synth(tw, th, 8)
    tem[0] //= 2  # a template with a different colour balance
    ref, rst = oracle.transfer(src, tem)
    got, gst = capi.transfer(src, tem)
    assert np.array_equal(gst.view(np.uint32), rst.view(np.uint32)), (gst, rst)
    assert np.array_equal(got, ref)
    # device-resident, in place
    d = torch.from_numpy(src).to(gpu)
    capi.dev_transfer(d, torch.from_numpy(tem).to(gpu), out=d)
    assert np.array_equal(d.cpu().numpy(), ref)


def test_colour_transfer_degenerate(st, gpu, oracle):
    """A constant source channel has standard deviation 0: the reference divides by it (transfer.cpp:168-170); the
    clamps of LabToRGB map the NaN/inf that follows to 0/255.  Same bytes on both sides."""
    from computervisionimagestich2_amd import capi
    src = np.full((3, 40, 30), 77, np.uint8)
    tem = oracle.synth(64, 48, 3)
    ref, _ = oracle.transfer(src, tem)
    got, _ = capi.transfer(src, tem)
    assert np.array_equal(got, ref)
    src = oracle.synth(30, 40, 2)
    src[:, :3, :] = 0  # black pixels: l = m = s = 0 is replaced by 1 (transfer.cpp:184-186)
    ref, rst = oracle.transfer(src, tem)
    got, gst = capi.transfer(src, tem)
    assert np.array_equal(got, ref) and np.array_equal(gst.view(np.uint32), rst.view(np.uint32))


@pytest.mark.parametrize("cw,ch", [(1024, 512), (1080, 527), (836, 300), (1081, 527), (1335, 640)])
@pytest.mark.parametrize("no_zero_tiles", ["0", "1"])
@pytest.mark.parametrize("negative", [False, True])
def test_zero_tile_flags(st, gpu, oracle, no_zero_tiles, negative, cw, ch, monkeypatch):
    """Sparse canvases: where the fused sweep runs, all-(+0) tiles of the blur scratch are recorded in a flag instead of
    being stored and re-read (ZeroTiles in csrc/k_compose.inc).  Same bits as the oracle with the flags on and off
    (STITCH_NO_ZERO_TILES=1), on canvases that are mostly empty for one image, with a frame that lies entirely inside
    a few tiles, and -- float frames -- with negative samples, whose decaying tails end in -0.0f (such tiles must not
    be taken for zero tiles).  Canvas heights: 512 (levels 0 and 1, 512 and 256 rows, whole bands), 527 (a partial last band
    of 15 rows at level 0, 263 rows at level 1: the fused anticausal-y + decimation looks its flags up per row, and level 1 has
    an odd height), 300 (44-row last band; 150 rows at level 1).  Widths 1081 and 1335 are odd at level 0 (1335 x 640: 667 at level 1 too):
    the decimating sweep advances by 126 columns there and a wavefront's columns lie in up to three flag tiles."""
    import torch
    from computervisionimagestich2_amd import capi
    monkeypatch.setenv("STITCH_WAVEFRONT", "2")
    monkeypatch.setenv("STITCH_SINGLE_FAST", "1")  # source-fused (index-tile flags) at this small size too
    monkeypatch.setenv("STITCH_NO_ZERO_TILES", no_zero_tiles)
    sy = ch / 512.0
    cases = [
        # fw, fh, map, mosaic w, h, ox, oy
        (300, 200, [1.0, 0.0, 0.0, -350.0, 0.0, 1.0, 0.0, -100.0 * sy], 400, ch, 0, 0),       # small frame in the middle, mosaic at the left, right half of the canvas empty
        (512, 512, [1.0, 0.002, 1e-6, -500.0, -0.001, 1.0, 5e-7, 1.5], 600, 300, 0, int(-100 * sy)),   # half overlap, mosaic short
        (128, 64, [1.0, 0.0, 0.0, -480.0, 0.0, 1.0, 0.0, -230.0 * sy], cw, ch, 0, 0),          # frame inside one or two tiles
    ]
    plan = capi.Plan(cw, ch, max_pairs=len(cases))
    assert plan.fused_sweep_levels == 2
    items, refs = [], []
    for i, (fw, fh, P, mw, mh, ox, oy) in enumerate(cases):
        F, M = oracle.synth(fw, fh, 2 * i + 1, np.float32), oracle.synth(mw, mh, 2 * i, np.float32)
        if negative:
            F, M = F - 300.0, M - 300.0
        rc, ref = oracle.pair(F, P, 0.0, 0.0, M, ox, oy, cw, ch)
        assert rc == 0, (i, rc)
        refs.append(ref)
        items.append((torch.from_numpy(F).to(gpu), P, 0.0, 0.0, torch.from_numpy(M).to(gpu), ox, oy,
                      torch.empty((3, ch, cw), dtype=torch.float32, device=gpu)))
    for rep in range(2):  # the flag buffer is reused: a second run must not see the first one's flags
        outs = plan.pairs(items if rep == 0 else items[::-1])
        order = list(range(len(cases))) if rep == 0 else list(range(len(cases)))[::-1]
        for slot, i in enumerate(order):
            plan.status(slot)
            assert np.array_equal(outs[slot].cpu().numpy().view(np.uint32), refs[i].view(np.uint32)), (rep, i)
    plan.close()


@pytest.mark.parametrize("wavefront,seed", [("2", 3), ("0", 4)])
def test_random_pairs_against_oracle(st, gpu, oracle, wavefront, seed, monkeypatch):
    """Randomised geometry (scripts/fuzz_pairs.py): batches of 1-3 pairs, random canvas sizes (heights multiples of 64),
    frame / mosaic sizes, bilinear maps, fractional and integer offsets, both pixel types; outputs AND error codes
    (empty middle row, zero overlap) must equal the oracle's.  With the fused sweep forced on two levels every fast path is
    active at these sizes (source fusion, implicit mask, zero-tile flags, fused sweep); with 0 the separate sweeps run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_pairs", os.path.join(os.path.dirname(HERE), "scripts", "fuzz_pairs.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    monkeypatch.setenv("STITCH_WAVEFRONT", wavefront)
    monkeypatch.setenv("STITCH_SINGLE_FAST", "1")  # every fast path at these sizes (the default picks the forms by canvas area per launch)
    done, bad = fz.run(seed, 16)
    assert bad == 0 and done >= 4, (done, bad)  # the other cases ended in the same error code on both sides


def test_random_pairs_any_size_every_switch(st, gpu, oracle, monkeypatch):
    """The general mode of the same generator: canvases of any size and parity, every tuning switch of include/stitch.h drawn
    per case (fused levels, re-run form of the causal sweep, column width of the y sweep, collapse form, zero-tile flags,
    source fusion) and, for four cases in ten, the ex6 variant's rules, both blurs and other sigmas.  A switch may never
    change a bit.  (3 000 cases of this run were compared once per round with scripts/fuzz_pairs.py directly.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_pairs", os.path.join(os.path.dirname(HERE), "scripts", "fuzz_pairs.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    monkeypatch.setenv("FUZZ_GENERAL", "1")
    for k in fz.SWITCHES:  # the generator sets and clears them itself: restore whatever was there afterwards
        monkeypatch.setenv(k, "0")
    done, bad = fz.run(5, 80)
    assert bad == 0 and done >= 40, (done, bad)


def test_bad_arguments_are_refused(st, gpu):
    """Null pointers, non-positive sizes, a null map, a too-small output buffer, a bad rank: 58 calls over every family of
    entry points (scripts/argprobe.py holds the table) -- each must return STITCH_ERR_ARG with a message and touch nothing.
    (The reference has no error paths at all here: SURVEY.md 5.)"""
    import ctypes as C
    import importlib.util
    spec = importlib.util.spec_from_file_location("argprobe_table", os.path.join(os.path.dirname(HERE), "scripts", "argprobe.py"))
    src = open(spec.origin).read().split("if len(sys.argv) > 1:")[0]  # the table only, not the driver below it
    ns = {"__file__": spec.origin, "__name__": "argprobe_table"}
    exec(compile(src, spec.origin, "exec"), ns)
    L = st.capi.lib()
    buf = (C.c_uint8 * (1 << 20))()
    table = ns["calls"](L, buf)
    assert len(table) >= 58
    for name, call in table.items():
        rc = call()
        assert rc == st.capi.ERR_ARG, (name, rc)
        assert L.stitch_last_error(), name
    assert not any(buf), "a refused call wrote into a buffer"


def test_host_entry_points_reuse_workspaces_and_trim(st, gpu, oracle, monkeypatch):
    """The host-pointer entry points keep idle workspaces (LRU by canvas and options), device staging and pinned buffers
    between calls; results must not depend on whether a call found a cached plan, a new one, or ran after stitch_trim(), and
    large copies (the chunked, threaded path: >= 4 MB) must arrive intact."""
    rng = np.random.default_rng(3)
    sizes = [(300, 200), (640, 480), (300, 200), (64, 64), (640, 480)]
    want = {}
    for rep in range(2):
        for (w, h) in sizes:
            A, B = two_canvases(oracle, w, h, 3, 4, np.uint8)
            got, _ = st.blend(A, B)
            if (w, h) not in want:
                rc, ref, _ = oracle.blend(A, B)
                assert rc == 0
                want[(w, h)] = ref
            assert np.array_equal(got, want[(w, h)]), (rep, w, h)
        st.capi.lib().stitch_trim()
    # a pair whose frames and mosaic are well above the threaded-copy threshold, both pixel types, twice
    fw, fh, cw, ch = 1408, 1024, 2048, 1024
    for dtype in (np.float32, np.uint8):
        A, B = oracle.synth(fw, fh, 2, dtype), oracle.synth(fw, fh, 3, dtype)
        P = [1.0, 0.002, 1e-6, -660.0, -0.001, 1.0, 5e-7, 1.5]
        rc, ref = oracle.pair(B, P, 0.0, 0.0, A, 0, 0, cw, ch)
        assert rc == 0
        for rep in range(2):
            got, _ = st.pair(B, P, 0.0, 0.0, A, 0, 0, cw, ch)
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (dtype, rep)
    monkeypatch.setenv("STITCH_PLAN_CACHE", "0")  # read once per process: this only documents the switch
    st.capi.lib().stitch_trim()


def test_host_entry_points_from_two_threads(st, gpu, oracle):
    """Two host threads in the blend at once (different canvas sizes, one of them above the threaded-copy threshold): the
    workspace LRU hands every call its own plan, the staging copier serialises, every result equals the oracle's."""
    import threading
    jobs = []
    for (w, h, fa) in [(640, 480, 3), (1600, 1200, 5)]:
        A, B = two_canvases(oracle, w, h, fa, fa + 1, np.uint8)
        rc, ref, _ = oracle.blend(A, B)
        assert rc == 0
        jobs.append((A, B, ref))
    errs = []

    def work(job):
        A, B, ref = job
        try:
            for _ in range(3):
                got, _ = st.blend(A, B)
                if not np.array_equal(got, ref):
                    errs.append("mismatch %s" % (A.shape,))
        except Exception as e:
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(j,)) for j in jobs]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not errs, errs


def test_host_entry_points_from_six_threads_mixed(st, gpu, oracle):
    """Six host threads at once, each running its own seeded mix of host-pointer calls -- blends and stitch steps on a handful
    of canvas sizes shared between the threads (they compete for the same cached workspaces), projections, equalisations,
    warps -- against results the oracle computed beforehand; stitch_trim() is called by one thread in the middle of it."""
    import threading
    from computervisionimagestich2_amd import capi
    sizes = [(320, 240), (500, 300), (333, 257), (640, 384)]
    jobs = {}
    for i, (w, h) in enumerate(sizes):
        for dt in (np.uint8, np.float32):
            A, B = two_canvases(oracle, w, h, 11 + i, 21 + i, dt)
            rc, ref, _ = oracle.blend(A, B)
            assert rc == 0
            jobs[("blend", i, dt)] = (A, B, ref)
        fw, fh = int(w * 0.7), h - 3
        F, M = oracle.synth(fw, fh, 31 + i, np.uint8), oracle.synth(fw, fh, 41 + i, np.uint8)
        P = small_map(w - fw - 4)
        rc, ref = oracle.pair(F, P, 0.5, -0.25, M, 0, 1, w, h)
        assert rc == 0
        jobs[("pair", i)] = (F, P, M, w, h, ref)
        img = oracle.synth(w, h, 51 + i, np.uint8)
        jobs[("project", i)] = (img, oracle.project(img))
        jobs[("equalize", i)] = (img, oracle.equalize(img)[0])
    keys = sorted(jobs.keys(), key=str)
    errs = []

    def work(tid):
        rng = np.random.default_rng(100 + tid)
        try:
            for n in range(30):
                k = keys[int(rng.integers(0, len(keys)))]
                j = jobs[k]
                if k[0] == "blend":
                    got, _ = st.blend(j[0], j[1])
                    ok = np.array_equal(got.view(np.uint8), j[2].view(np.uint8))
                elif k[0] == "pair":
                    got, _ = st.pair(j[0], j[1], 0.5, -0.25, j[2], 0, 1, j[3], j[4])
                    ok = np.array_equal(got, j[5])
                elif k[0] == "project":
                    ok = np.array_equal(st.project(j[0]), j[1])
                else:
                    ok = np.array_equal(st.equalize(j[0])[0], j[1])
                if not ok:
                    errs.append(("mismatch", tid, n, str(k)))
                if tid == 0 and n == 15:
                    capi.trim()
        except Exception as e:  # noqa: BLE001
            errs.append((tid, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not any(t.is_alive() for t in th), "a thread is stuck"
    assert not errs, errs[:5]


@pytest.mark.parametrize("w", [2, 3, 5, 7, 63, 64, 100, 263, 527, 540, 1081, 2210, 2315, 4096, 4421, 6144, 16384, 24576, 16777215])
def test_fastdiv_equals_ieee_divide(st, w):
    """The decimation's divide by a level's width / height in its short form (hoisted reciprocal refinement, five instructions,
    k_sweeps1.inc) against the IEEE divide for EVERY numerator bit pattern inside the short form's range."""
    tested, bad = st.capi.dev_check_fastdiv(float(w))
    assert tested > 2 ** 31 and bad == 0, (w, tested, bad)
