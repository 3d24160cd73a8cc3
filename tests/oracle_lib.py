"""ctypes bindings for the parity oracle (oracle/libstitch_oracle.so) and, where it has been built, for the
reference itself (oracle/_ref/libref_hotpath.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package.

Images are numpy arrays of shape (3, H, W), C-contiguous = CImg's planar layout (CImg.h:11787-11793).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libstitch_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_hotpath.so")
REF6_SO = os.path.join(ROOT, "oracle", "_ref", "libref6_hotpath.so")  # the src/ex6 variant (oracle/ref6_harness.cpp)


class BlendOpts(C.Structure):
    _fields_ = [("sigma", C.c_float), ("blur_kind", C.c_int), ("level_rule", C.c_int), ("seam_rule", C.c_int)]


class Seam(C.Structure):
    _fields_ = [("sum_a_x", C.c_int), ("n_a", C.c_int), ("sum_ov_x", C.c_int), ("n_ov", C.c_int),
                ("ratio", C.c_float), ("ov", C.c_float), ("branch", C.c_int), ("start", C.c_int)]

    def as_tuple(self):
        return (self.sum_a_x, self.n_a, self.sum_ov_x, self.n_ov, self.branch, self.start)


class BmpInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bpp", C.c_int32), ("top_down", C.c_int32),
                ("data_pos", C.c_uint64), ("stride", C.c_uint64), ("data_bytes", C.c_uint64)]


def make_bmp(img, bpp=24, top_down=False, header_size=40, extra_gap=0, size_field=None, truncate=0, alpha=0x5A):
    """A BMP file image (bytes) with the layout knobs CImg's loader distinguishes (CImg.h:48413-48441): bits per
    pixel 24/32, negative height, a larger info header, a gap before the pixel data, the file-size field (None =
    true size, 0 = "unknown"), and missing bytes at the end.  Test-side builder, independent of the encoders."""
    import struct
    img = np.asarray(img, np.uint8)
    _, h, w = img.shape
    bp = bpp // 8
    stride = (w * bp + 3) & ~3
    rows = np.zeros((h, stride), np.uint8)
    px = np.full((h, w, bp), alpha, np.uint8)
    px[:, :, 0], px[:, :, 1], px[:, :, 2] = img[2], img[1], img[0]
    rows[:, : w * bp] = (px if top_down else px[::-1]).reshape(h, w * bp)
    offset = 14 + header_size + extra_gap
    total = offset + stride * h
    hdr = b"BM" + struct.pack("<IHHI", total if size_field is None else size_field, 0, 0, offset)
    info = struct.pack("<IiiHHIIiiII", header_size, w, -h if top_down else h, 1, bpp, 0, stride * h, 2835, 2835, 0, 0)
    info += bytes((i * 7 + 3) & 0xFF for i in range(header_size - 40)) + bytes((i * 5 + 1) & 0xFF for i in range(extra_gap))
    data = hdr + info + rows.tobytes()
    return data[: len(data) - truncate] if truncate else data


ROOT_OPTS = dict(sigma=2.0, blur_kind=0, level_rule=0, seam_rule=0)   # root variant (ImageProcess.cpp)
EX6_OPTS = dict(sigma=2.0, blur_kind=1, level_rule=1, seam_rule=1)    # src/ex6 variant


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _img(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    assert a.ndim == 3 and a.shape[0] == 3, a.shape
    return a


def build_oracle():
    if not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(os.path.join(ROOT, "oracle", f)) > os.path.getmtime(ORACLE_SO)
            for f in ("stitch_oracle.c", "stitch_oracle.h", "stitch_oracle_px.inc")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


class Oracle:
    """The CPU restatement.  Method names mirror include/stitch.h without the prefix."""

    def __init__(self):
        build_oracle()
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        L.oracle_threads.restype = C.c_int
        for n in ("u8", "f32"):
            for f in ("project", "warp", "move", "seam", "blend", "pair"):
                getattr(L, f"oracle_{f}_{n}").restype = C.c_int
        L.oracle_pyramid_levels.restype = C.c_int
        L.oracle_equalize_u8.restype = C.c_int
        L.oracle_lummix_u8.restype = C.c_int

        # OpenMP's thread count is a per-thread setting that defaults to every CPU the machine has (256 on the GPU box, whose
        # share is 16 cores: an 18 s pair instead of 0.3 s) and that other libraries in the process change; the wrapper
        # therefore sets it, in the calling thread, before every parallel entry point
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        self._n = max(1, min(avail, int(os.environ.get("STITCH_CPU_THREADS", "16"))))

    def threads(self):
        return self.lib.oracle_threads()

    def set_threads(self, n):
        self._n = int(n)
        self.lib.oracle_set_threads(C.c_int(n))

    def __getattribute__(self, name):
        if name in ("project", "warp", "move", "blur", "decimate", "expand", "blend", "pair", "equalize", "lummix", "gray", "transfer", "synth"):
            object.__getattribute__(self, "lib").oracle_set_threads(C.c_int(object.__getattribute__(self, "_n")))
        return object.__getattribute__(self, name)

    @staticmethod
    def _sfx(a):
        return "u8" if a.dtype == np.uint8 else "f32"

    def project(self, src, fov_deg=15.0):
        src = _img(src, src.dtype)
        dst = np.empty_like(src)
        _, h, w = src.shape
        rc = getattr(self.lib, "oracle_project_" + self._sfx(src))(_p(src), w, h, C.c_float(fov_deg), _p(dst))
        assert rc == 0, rc
        return dst

    def map_xy(self, x, y, p):
        X, Y = C.c_float(), C.c_float()
        pp = (C.c_double * 8)(*p)
        self.lib.oracle_map_xy(C.c_float(x), C.c_float(y), pp, C.byref(X), C.byref(Y))
        return X.value, Y.value

    def warp(self, src, p, offx, offy, cw, ch, canvas=None):
        src = _img(src, src.dtype)
        if canvas is None:
            canvas = np.zeros((3, ch, cw), src.dtype)
        _, sh, sw = src.shape
        pp = (C.c_double * 8)(*p)
        rc = getattr(self.lib, "oracle_warp_" + self._sfx(src))(_p(src), sw, sh, pp, C.c_float(offx), C.c_float(offy),
                                                               _p(canvas), cw, ch)
        assert rc == 0, rc
        return canvas

    def move(self, src, ox, oy, cw, ch, canvas=None):
        src = _img(src, src.dtype)
        if canvas is None:
            canvas = np.zeros((3, ch, cw), src.dtype)
        _, sh, sw = src.shape
        rc = getattr(self.lib, "oracle_move_" + self._sfx(src))(_p(src), sw, sh, int(ox), int(oy), _p(canvas), cw, ch)
        assert rc == 0, rc
        return canvas

    def seam(self, a, b, seam_rule=0):
        a, b = _img(a, a.dtype), _img(b, a.dtype)
        _, h, w = a.shape
        s = Seam()
        rc = getattr(self.lib, "oracle_seam_" + self._sfx(a))(_p(a), _p(b), w, h, seam_rule, C.byref(s))
        return rc, s

    def pyramid_levels(self, w, h, level_rule=0):
        lw = (C.c_int * 32)()
        lh = (C.c_int * 32)()
        n = self.lib.oracle_pyramid_levels(w, h, level_rule, lw, lh)
        if n < 0:
            return n, [], []
        return n, list(lw[:n]), list(lh[:n])

    def blur(self, img, sigma=2.0, blur_kind=0):
        img = np.array(img, dtype=np.float32, order="C", copy=True)
        c, h, w = img.shape
        self.lib.oracle_blur_f32(_p(img), w, h, c, C.c_float(sigma), blur_kind)
        return img

    def vanvliet_coeffs(self, sigma):
        f = (C.c_double * 4)()
        self.lib.oracle_vanvliet_coeffs(C.c_float(sigma), f)
        return list(f)

    def decimate(self, img, w2, h2):
        img = np.ascontiguousarray(img, np.float32)
        c, h, w = img.shape
        out = np.empty((c, h2, w2), np.float32)
        self.lib.oracle_decimate_f32(_p(img), w, h, c, _p(out), w2, h2)
        return out

    def expand(self, img, w2, h2):
        img = np.ascontiguousarray(img, np.float32)
        c, h, w = img.shape
        out = np.empty((c, h2, w2), np.float32)
        self.lib.oracle_expand_f32(_p(img), w, h, c, _p(out), w2, h2)
        return out

    def expand_table(self, n_src, n_dst):
        idx = np.empty(n_dst, np.int32)
        alpha = np.empty(n_dst, np.float64)
        self.lib.oracle_expand_table(n_src, n_dst, _p(idx), _p(alpha))
        return idx, alpha

    def blend(self, a, b, opts=ROOT_OPTS, want_f32=False):
        """-> (rc, out, seam[, out_f32 for u8 inputs])"""
        a, b = _img(a, a.dtype), _img(b, a.dtype)
        _, h, w = a.shape
        o = BlendOpts(**opts)
        s = Seam()
        out = np.zeros_like(a)
        if a.dtype == np.uint8:
            f = np.zeros(a.shape, np.float32) if want_f32 else None
            rc = self.lib.oracle_blend_u8(_p(a), _p(b), w, h, C.byref(o), _p(out), _p(f) if want_f32 else None, C.byref(s))
            return (rc, out, s, f) if want_f32 else (rc, out, s)
        rc = self.lib.oracle_blend_f32(_p(a), _p(b), w, h, C.byref(o), _p(out), C.byref(s))
        return rc, out, s

    def pair(self, frame, p, offx, offy, mosaic, ox, oy, cw, ch, opts=ROOT_OPTS):
        frame, mosaic = _img(frame, frame.dtype), _img(mosaic, frame.dtype)
        o = BlendOpts(**opts)
        out = np.zeros((3, ch, cw), frame.dtype)
        pp = (C.c_double * 8)(*p)
        rc = getattr(self.lib, "oracle_pair_" + self._sfx(frame))(
            _p(frame), frame.shape[2], frame.shape[1], pp, C.c_float(offx), C.c_float(offy), _p(mosaic), mosaic.shape[2],
            mosaic.shape[1], int(ox), int(oy), cw, ch, C.byref(o), _p(out))
        return rc, out

    def equalize(self, img):
        img = np.array(_img(img, np.uint8), copy=True)
        _, h, w = img.shape
        hist = np.zeros(256, np.int32)
        lut = np.zeros(256, np.int32)
        rc = self.lib.oracle_equalize_u8(_p(img), w, h, _p(hist), _p(lut))
        assert rc == 0, rc
        return img, hist, lut

    def lummix(self, result, equalized, num=19.0, den=20.0):
        result = np.array(_img(result, np.uint8), copy=True)
        equalized = _img(equalized, np.uint8)
        _, h, w = result.shape
        rc = self.lib.oracle_lummix_u8(_p(result), _p(equalized), w, h, C.c_double(num), C.c_double(den))
        assert rc == 0, rc
        return result

    def gray(self, rgb):
        rgb = _img(rgb, np.uint8)
        _, h, w = rgb.shape
        g, f = np.empty((h, w), np.uint8), np.empty((h, w), np.float32)
        assert self.lib.oracle_gray_u8(_p(rgb), w, h, _p(g), _p(f)) == 0
        return g, f

    def canvas_bbox(self, fw, fh, p, rw, rh):
        mx, my, nw, nh = C.c_float(), C.c_float(), C.c_int(), C.c_int()
        assert self.lib.oracle_canvas_bbox(fw, fh, (C.c_double * 8)(*p), rw, rh, C.byref(mx), C.byref(my), C.byref(nw), C.byref(nh)) == 0
        return mx.value, my.value, nw.value, nh.value

    def map_points(self, x, y, p, offx, offy):
        x, y = np.array(x, np.float32), np.array(y, np.float32)
        ix, iy = np.empty(x.size, np.int32), np.empty(x.size, np.int32)
        self.lib.oracle_map_points(_p(x), _p(y), _p(ix), _p(iy), x.size, (C.c_double * 8)(*p), C.c_float(offx), C.c_float(offy))
        return x, y, ix, iy

    def shift_points(self, x, y, ox, oy):
        x, y = np.array(x, np.float32), np.array(y, np.float32)
        ix, iy = np.empty(x.size, np.int32), np.empty(x.size, np.int32)
        self.lib.oracle_shift_points(_p(x), _p(y), _p(ix), _p(iy), x.size, int(ox), int(oy))
        return x, y, ix, iy

    def synth(self, w, h, frame_id, dtype=np.uint8):
        out = np.empty((3, h, w), dtype)
        getattr(self.lib, "oracle_synth_" + ("u8" if dtype == np.uint8 else "f32"))(_p(out), w, h, int(frame_id))
        return out

    def transfer(self, src, tem, use_libm=False):
        """-> (out, stats[12]); use_libm: this platform's logf/pow instead of include/stitch_elem.h"""
        src, tem = _img(src, np.uint8), _img(tem, np.uint8)
        out, st = np.empty_like(src), np.zeros(12, np.float32)
        self.lib.oracle_transfer_u8.restype = C.c_int
        rc = self.lib.oracle_transfer_u8(_p(src), src.shape[2], src.shape[1], _p(tem), tem.shape[2], tem.shape[1], _p(out), _p(st),
                                         int(use_libm))
        assert rc == 0, rc
        return out, st

    def bmp_decode(self, data):
        """bytes of a BMP file -> (rc, planar (3,H,W) uint8 or None)"""
        buf = np.frombuffer(bytes(data), np.uint8)
        bi = BmpInfo()
        self.lib.oracle_bmp_parse.restype = C.c_int
        rc = self.lib.oracle_bmp_parse(_p(buf), C.c_size_t(buf.size), C.byref(bi))
        if rc:
            return rc, None
        out = np.empty((3, bi.height, bi.width), np.uint8)
        self.lib.oracle_bmp_decode_u8.restype = C.c_int
        rc = self.lib.oracle_bmp_decode_u8(_p(buf), C.c_size_t(buf.size), _p(out))
        return rc, out

    def bmp_encode(self, img):
        img = _img(img, np.uint8)
        _, h, w = img.shape
        self.lib.oracle_bmp_file_bytes.restype = C.c_size_t
        n = self.lib.oracle_bmp_file_bytes(w, h)
        out = np.empty(n, np.uint8)
        self.lib.oracle_bmp_encode_u8.restype = C.c_int
        rc = self.lib.oracle_bmp_encode_u8(_p(img), w, h, _p(out), C.c_size_t(n))
        assert rc == 0, rc
        return out.tobytes()


def have_reference():
    return os.path.exists(REF_SO)


class ReferenceEx6:
    """The `src/ex6` variant's own blend (src/ex6/ImageProcess.cpp:638-742, oracle/ref6_harness.cpp)."""

    def __init__(self):
        self.lib = C.CDLL(REF6_SO)

    def blend(self, a, b):
        a, b = _img(a, np.uint8), _img(b, np.uint8)
        out = np.empty_like(a)
        assert self.lib.ref6_blend_u8(_p(a), _p(b), a.shape[2], a.shape[1], _p(out)) == 0
        return out


class Reference:
    """The reference's own functions (oracle/ref_harness.cpp).  Only for pinning the oracle and for
    generating tests/golden; exists only where /root/reference was available to oracle/Makefile."""

    def __init__(self):
        self.lib = C.CDLL(REF_SO)

    def project(self, src):
        src = _img(src, np.uint8)
        dst = np.empty_like(src)
        rc = self.lib.ref_project_u8(_p(src), src.shape[2], src.shape[1], _p(dst))
        assert rc == 0
        return dst

    def bilinear(self, src, x, y, c):
        src = _img(src, np.uint8)
        return self.lib.ref_bilinear_u8(_p(src), src.shape[2], src.shape[1], C.c_float(x), C.c_float(y), c)

    def map_xy(self, x, y, p):
        X, Y = C.c_float(), C.c_float()
        self.lib.ref_map_xy(C.c_float(x), C.c_float(y), (C.c_double * 8)(*p), C.byref(X), C.byref(Y))
        return X.value, Y.value

    def bbox(self, w, h, p):
        out = (C.c_float * 4)()
        self.lib.ref_bbox(w, h, (C.c_double * 8)(*p), out)
        return list(out)

    def warp(self, src, p, offx, offy, cw, ch, canvas=None):
        src = _img(src, np.uint8)
        if canvas is None:
            canvas = np.zeros((3, ch, cw), np.uint8)
        self.lib.ref_warp_u8(_p(src), src.shape[2], src.shape[1], (C.c_double * 8)(*p), C.c_float(offx), C.c_float(offy),
                             _p(canvas), cw, ch)
        return canvas

    def move(self, src, ox, oy, cw, ch, canvas=None):
        src = _img(src, np.uint8)
        if canvas is None:
            canvas = np.zeros((3, ch, cw), np.uint8)
        self.lib.ref_move_u8(_p(src), src.shape[2], src.shape[1], int(ox), int(oy), _p(canvas), cw, ch)
        return canvas

    def blend(self, a, b):
        a, b = _img(a, np.uint8), _img(b, np.uint8)
        out = np.empty_like(a)
        rc = self.lib.ref_blend_u8(_p(a), _p(b), a.shape[2], a.shape[1], _p(out))
        assert rc == 0
        return out

    def equalize(self, img):
        img = np.array(_img(img, np.uint8), copy=True)
        self.lib.ref_equalize_u8(_p(img), img.shape[2], img.shape[1])
        return img

    def gray(self, rgb):
        rgb = _img(rgb, np.uint8)
        _, h, w = rgb.shape
        g = np.empty((h, w), np.uint8)
        assert self.lib.ref_gray_u8(_p(rgb), w, h, _p(g)) == 0
        return g

    def canvas(self, fw, fh, p, rw, rh):
        mx, my, nw, nh = C.c_float(), C.c_float(), C.c_int(), C.c_int()
        self.lib.ref_canvas(fw, fh, (C.c_double * 8)(*p), rw, rh, C.byref(mx), C.byref(my), C.byref(nw), C.byref(nh))
        return mx.value, my.value, nw.value, nh.value

    def update_features(self, x, y, p, offx, offy, ox, oy, by_offset):
        x, y = np.array(x, np.float32), np.array(y, np.float32)
        ix, iy = np.empty(x.size, np.int32), np.empty(x.size, np.int32)
        self.lib.ref_update_features(_p(x), _p(y), _p(ix), _p(iy), x.size, (C.c_double * 8)(*p), C.c_float(offx), C.c_float(offy),
                                     int(ox), int(oy), int(by_offset))
        return x, y, ix, iy

    def cimg_blur(self, img, sigma=2.0, is_gaussian=True):
        img = np.array(img, dtype=np.float32, order="C", copy=True)
        c, h, w = img.shape
        self.lib.ref_cimg_blur_f32(_p(img), w, h, c, C.c_float(sigma), int(is_gaussian))
        return img

    def cimg_resize(self, img, w2, h2, c2=None, interp=3):
        img = np.ascontiguousarray(img, np.float32)
        c, h, w = img.shape
        c2 = c if c2 is None else c2
        out = np.empty((c2, h2, w2), np.float32)
        rc = self.lib.ref_cimg_resize_f32(_p(img), w, h, c, _p(out), w2, h2, c2, interp)
        assert rc == 0
        return out

    def pipeline(self, directory, n):
        w, h = C.c_int(), C.c_int()
        buf = np.zeros(64 << 20, np.uint8)
        d = directory if directory.endswith("/") else directory + "/"
        rc = self.lib.ref_pipeline(d.encode(), n, _p(buf), buf.size, C.byref(w), C.byref(h))
        assert rc == 0
        return buf[:w.value * h.value * 3].reshape(3, h.value, w.value).copy()

    def load_bmp(self, path):
        w, h = C.c_int(), C.c_int()
        buf = np.zeros(64 << 20, np.uint8)
        rc = self.lib.ref_load_bmp(path.encode(), _p(buf), buf.size, C.byref(w), C.byref(h))
        assert rc == 0, rc
        return buf[:w.value * h.value * 3].reshape(3, h.value, w.value).copy()

    def load_bmp_bytes(self, data, tmpdir):
        """CImg's loader on a file image given as bytes (written to tmpdir first)."""
        path = os.path.join(str(tmpdir), "ref_in.bmp")
        with open(path, "wb") as f:
            f.write(data)
        return self.load_bmp(path)

    def save_bmp_bytes(self, img, tmpdir):
        """CImg<unsigned char>::save_bmp on a planar image -> the bytes of the file it wrote."""
        img = _img(img, np.uint8)
        path = os.path.join(str(tmpdir), "ref_out.bmp")
        self.lib.ref_save_bmp(_p(img), img.shape[2], img.shape[1], path.encode())
        with open(path, "rb") as f:
            return f.read()
