"""CPU, only where oracle/_ref/libref_hotpath.so exists (built from /root/reference by oracle/Makefile):
the oracle against the reference's own functions on fresh seeded inputs.  Skipped on the GPU box when the
prebuilt library did not travel."""
import numpy as np
import pytest

from oracle_lib import Reference, have_reference

pytestmark = pytest.mark.skipif(not have_reference(), reason="oracle/_ref/libref_hotpath.so not built (needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    return Reference()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("w,h,f", [(384, 512, 1), (400, 300, 2), (97, 61, 3), (61, 97, 4), (2, 2, 5), (1, 9, 6), (9, 1, 7)])
def test_project(oracle, ref, w, h, f):
    src = oracle.synth(w, h, f)
    assert np.array_equal(oracle.project(src), ref.project(src))


def test_map_and_bbox(oracle, ref):
    rng = np.random.default_rng(3)
    for _ in range(200):
        p = [1 + rng.normal() * 0.05, rng.normal() * 0.05, rng.normal() * 1e-4, rng.normal() * 300,
             rng.normal() * 0.05, 1 + rng.normal() * 0.05, rng.normal() * 1e-5, rng.normal() * 20]
        x, y = np.float32(rng.uniform(-500, 1500)), np.float32(rng.uniform(-500, 1500))
        assert oracle.map_xy(x, y, p) == ref.map_xy(x, y, p)


@pytest.mark.parametrize("seed", range(4))
def test_warp_move_random_maps(oracle, ref, seed):
    rng = np.random.default_rng(seed)
    src = oracle.synth(200, 150, seed)
    p = [1 + rng.normal() * 0.03, rng.normal() * 0.03, rng.normal() * 1e-4, rng.normal() * 80,
         rng.normal() * 0.03, 1 + rng.normal() * 0.03, rng.normal() * 1e-5, rng.normal() * 10]
    offx, offy = np.float32(rng.uniform(-40, 0.99)), np.float32(rng.uniform(-9, 0.99))
    assert np.array_equal(oracle.warp(src, p, offx, offy, 333, 177), ref.warp(src, p, offx, offy, 333, 177))
    ox, oy = int(rng.integers(-50, 50)), int(rng.integers(-20, 20))
    assert np.array_equal(oracle.move(src, ox, oy, 333, 177), ref.move(src, ox, oy, 333, 177))


def test_warp_truncation_toward_zero(oracle, ref):
    """coordinates in (-1,0) truncate to 0 and are therefore INSIDE the source (ImageProcess.cpp:598-600)"""
    src = oracle.synth(40, 30, 1)
    p = [1, 0, 0, -0.5, 0, 1, 0, -0.75]
    a, b = oracle.warp(src, p, 0.0, 0.0, 50, 40), ref.warp(src, p, 0.0, 0.0, 50, 40)
    assert np.array_equal(a, b) and a[0, 0, 0] == src[0, 0, 0]


@pytest.mark.parametrize("w,h,c", [(540, 3, 1), (17, 9, 3), (2, 2, 1), (3, 5, 2), (64, 1, 1), (1, 64, 1), (300, 200, 3)])
@pytest.mark.parametrize("gauss", [True, False])
def test_blur(oracle, ref, w, h, c, gauss):
    x = (np.random.default_rng(w * h).random((c, h, w)) * 255).astype(np.float32)
    assert np.array_equal(bits(oracle.blur(x, 2.0, 0 if gauss else 1)), bits(ref.cimg_blur(x, 2.0, gauss)))


@pytest.mark.parametrize("w,h", [(1081, 527), (67, 33), (4, 2), (3, 3), (135, 65), (600, 800)])
def test_resize(oracle, ref, w, h):
    rng = np.random.default_rng(w + h)
    w2, h2 = max(1, w // 2), max(1, h // 2)
    x = (rng.random((3, h, w)) * 255).astype(np.float32)
    assert np.array_equal(bits(oracle.decimate(x, w2, h2)), bits(ref.cimg_resize(x, w2, h2)))
    y = (rng.random((3, h2, w2)) * 255).astype(np.float32)
    assert np.array_equal(bits(oracle.expand(y, w, h)), bits(ref.cimg_resize(y, w, h)))


@pytest.mark.parametrize("w,h", [(67, 33), (270, 131), (333, 222), (100, 64), (33, 67), (640, 480)])
def test_blend(oracle, ref, w, h):
    for a_left in (True, False):
        A, B = oracle.synth(w, h, 21), oracle.synth(w, h, 22)
        if a_left:
            A[:, :, (2 * w) // 3:] = 0
            B[:, :, : w // 3] = 0
        else:
            A[:, :, : w // 3] = 0
            B[:, :, (2 * w) // 3:] = 0
        rc, out, _ = oracle.blend(A, B)
        assert rc == 0 and np.array_equal(out, ref.blend(A, B))


def test_equalize(oracle, ref):
    for f in range(3):
        img = oracle.synth(311, 173, 30 + f)
        img[f % 3] = np.maximum(img[f % 3], 150 + 40 * f)
        img[:, :50, :70] = 0
        assert np.array_equal(oracle.equalize(img)[0], ref.equalize(img))


def test_blend_random_sizes_sweep(oracle, ref):
    """40 random canvas sizes (odd/even mixes at every pyramid level, both seam branches, ragged content): the oracle
    equals the reference byte for byte or reports the degenerate-pyramid case the reference cannot express."""
    rng = np.random.default_rng(1234)
    done = 0
    while done < 40:
        w, h = int(rng.integers(2, 260)), int(rng.integers(2, 200))
        n, _, _ = oracle.pyramid_levels(w, h)
        A, B = oracle.synth(w, h, int(rng.integers(0, 50))), oracle.synth(w, h, int(rng.integers(50, 100)))
        ca, cb = sorted(int(v) for v in rng.integers(0, w + 1, 2))
        if rng.random() < 0.5:
            A[:, :, cb:] = 0
            B[:, :, :ca] = 0
        else:
            A[:, :, :ca] = 0
            B[:, :, cb:] = 0
        rc, out, seam = oracle.blend(A, B)
        if n < 0:
            assert rc == -4
            continue
        if rc != 0:
            assert rc in (-2, -3)  # empty mid row / no overlap: the reference hangs or divides 0/0 here
            continue
        assert np.array_equal(out, ref.blend(A, B)), (w, h)
        done += 1


def test_bbox(oracle, ref):
    # the reference's canvas sizing inputs (ImageProcess.cpp:206-216) -- used by stitch_canvas_bbox's test as well
    p = [0.9724, -0.0398, 0.000149, 206.67, 0.00141, 1.00076, -1.2e-06, 4.55]
    mn_x, mn_y, mx_x, mx_y = ref.bbox(384, 512, p)
    corners = [oracle.map_xy(np.float32(x), np.float32(y), p) for x in (0, 383) for y in (0, 511)]
    assert mn_x == min(c[0] for c in corners) and mx_x == max(c[0] for c in corners)
    assert mn_y == min(c[1] for c in corners) and mx_y == max(c[1] for c in corners)


def test_gray_bbox_features(oracle, ref):
    """toGrayScale, canvas sizing, feature updates."""
    rng = np.random.default_rng(7)
    for f in range(3):
        img = oracle.synth(123, 77, 40 + f)
        g, gf = oracle.gray(img)
        assert np.array_equal(g, ref.gray(img)) and np.array_equal(gf, g.astype(np.float32))
    for _ in range(100):
        p = [1 + rng.normal() * 0.05, rng.normal() * 0.05, rng.normal() * 2e-4, rng.normal() * 300,
             rng.normal() * 0.05, 1 + rng.normal() * 0.05, rng.normal() * 2e-5, rng.normal() * 20]
        fw, fh, rw, rh = (int(v) for v in rng.integers(50, 900, 4))
        assert oracle.canvas_bbox(fw, fh, p, rw, rh) == ref.canvas(fw, fh, p, rw, rh)
    p = [0.9724, -0.0398, 0.000149, 206.67, 0.00141, 1.00076, -1.2e-06, 4.55]
    x, y = rng.uniform(0, 384, 200).astype(np.float32), rng.uniform(0, 512, 200).astype(np.float32)
    a, b = oracle.map_points(x, y, p, -230.579239, -4.68064785), ref.update_features(x, y, p, -230.579239, -4.68064785, 0, 0, False)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))
    a, b = oracle.shift_points(x, y, -230, -4), ref.update_features(x, y, p, 0, 0, -230, -4, True)
    assert all(np.array_equal(u, v) for u, v in zip(a, b))


@pytest.mark.parametrize("w,h", [(257, 129), (5, 3), (1, 7), (64, 64), (2, 2), (1027, 3), (1368, 17)])
def test_bmp_load_save(oracle, ref, w, h, tmp_path):
    """oracle_bmp_decode/encode against CImg::load_bmp / save_bmp for every 24/32-bit header
    layout the loader distinguishes, including files that end early."""
    from oracle_lib import make_bmp
    img = oracle.synth(w, h, 3)
    for kw in [dict(), dict(bpp=32), dict(top_down=True), dict(header_size=108), dict(extra_gap=10), dict(extra_gap=1), dict(size_field=0),
               dict(size_field=60), dict(truncate=7), dict(truncate=3 * w + 5), dict(bpp=32, top_down=True, header_size=124, extra_gap=3)]:
        data = make_bmp(img, **kw)
        if len(data) < 54:
            continue
        rc, got = oracle.bmp_decode(data)
        want = ref.load_bmp_bytes(data, tmp_path)
        assert rc == 0 and got.shape == want.shape and np.array_equal(got, want), kw
    assert oracle.bmp_encode(img) == ref.save_bmp_bytes(img, tmp_path)


# ---- the src/ex6 variant of the blend (Deriche blur, depth from min(w,h), three-channel seam scan with double ratios) ----
def _ex6():
    import os
    from oracle_lib import REF6_SO, ReferenceEx6
    if not os.path.exists(REF6_SO):
        pytest.skip("oracle/_ref/libref6_hotpath.so not built (needs /root/reference/src/ex6)")
    return ReferenceEx6()


@pytest.mark.parametrize("w,h", [(67, 33), (270, 131), (333, 222), (100, 64), (33, 67), (640, 480), (600, 800)])
def test_blend_ex6_whole_function(oracle, w, h):
    """oracle.blend with the variant options against the variant's own ImageProcess::blend
    (src/ex6/ImageProcess.cpp:638-742, compiled in place): byte for byte, both seam branches."""
    from oracle_lib import EX6_OPTS
    ref6 = _ex6()
    for a_left in (True, False):
        A, B = oracle.synth(w, h, 21), oracle.synth(w, h, 22)
        if a_left:
            A[:, :, (2 * w) // 3:] = 0
            B[:, :, : w // 3] = 0
        else:
            A[:, :, : w // 3] = 0
            B[:, :, (2 * w) // 3:] = 0
        rc, out, _ = oracle.blend(A, B, EX6_OPTS)
        assert rc == 0 and np.array_equal(out, ref6.blend(A, B)), (w, h, a_left)


def test_blend_ex6_random_sizes_sweep(oracle):
    from oracle_lib import EX6_OPTS
    ref6 = _ex6()
    rng = np.random.default_rng(99)
    done = 0
    while done < 25:
        w, h = int(rng.integers(4, 260)), int(rng.integers(4, 200))
        A, B = oracle.synth(w, h, int(rng.integers(0, 50))), oracle.synth(w, h, int(rng.integers(50, 100)))
        ca, cb = sorted(int(v) for v in rng.integers(0, w + 1, 2))
        if rng.random() < 0.5:
            A[:, :, cb:] = 0
            B[:, :, :ca] = 0
        else:
            A[:, :, :ca] = 0
            B[:, :, cb:] = 0
        if rng.random() < 0.3:  # a channel that is empty where the others are not: the three-channel "non-empty" rule differs from the root's
            A[int(rng.integers(0, 3)), :, ca:ca + 5] = 0
        rc, out, seam = oracle.blend(A, B, EX6_OPTS)
        if rc != 0:
            continue  # empty mid row / no overlap / degenerate pyramid: the variant hangs or divides 0/0 there
        assert np.array_equal(out, ref6.blend(A, B)), (w, h)
        done += 1
