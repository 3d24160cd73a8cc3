"""ctypes binding of libstitch_hip.so (include/stitch.h) -- the product path.

Two families, mirroring the header:
  * host-buffer calls on numpy arrays of shape (3, H, W) (CImg's planar layout, CImg.h:11787-11793);
  * device-resident calls on torch CUDA(=HIP) tensors, enqueued on torch's current stream.
PyTorch supplies device memory and streams only; every computation is a hand-written HIP kernel in csrc/.
There is no fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STITCH_LIB", os.path.join(_HERE, "libstitch_hip.so"))  # STITCH_LIB: A/B builds of the same ABI

# kernel ids of stitch_plan_read_profile -> the HIP kernel symbol each one times (rocprofv3 reports the same names)
KERNELS = ("compose", "seam", "mask", "vv_x_fwd", "vv_x_bwd", "vv_y_fwd", "vv_y_bwd", "decimate", "collapse_top", "collapse",
           "collapse_l0", "vv_xbyf", "vv_x_fwd_src", "coarse")
KERNEL_SYMBOLS = {"compose": "k_src_index (source-fused) / k_compose", "seam": "k_seam", "mask": "k_mask", "vv_x_fwd": "k_vv_x_fwd<T, false, false>", "vv_x_bwd": "k_vv_x_bwd",
                  "vv_y_fwd": "k_vv_y_fwd1 / k_vv_y_fwd", "vv_y_bwd": "k_vv_y_bwd_dec", "decimate": "k_decimate", "collapse_top": "k_blend_top",
                  "collapse": "k_collapse<float, false>", "collapse_l0": "k_collapse<T, true>", "vv_xbyf": "k_vv_xbyf<false, float, 0, false>", "vv_x_fwd_src": "k_vv_x_fwd<T, true, false>",
                  "coarse": "k_coarse"}


class StitchError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"stitch error {code}: {text}")
        self.code = code


OK, ERR_ARG, ERR_EMPTY_MIDROW, ERR_ZERO_OVERLAP, ERR_PYRAMID, ERR_HIP, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5, -6


class BlendOpts(C.Structure):
    """stitch_blend_opts.  Defaults = root variant (ImageProcess.cpp:648-773)."""
    _fields_ = [("sigma", C.c_float), ("blur_kind", C.c_int), ("level_rule", C.c_int), ("seam_rule", C.c_int)]

    def __init__(self, sigma=2.0, blur_kind=0, level_rule=0, seam_rule=0):
        super().__init__(sigma, blur_kind, level_rule, seam_rule)


ROOT_OPTS = dict(sigma=2.0, blur_kind=0, level_rule=0, seam_rule=0)
EX6_OPTS = dict(sigma=2.0, blur_kind=1, level_rule=1, seam_rule=1)


class Seam(C.Structure):
    _fields_ = [("sum_a_x", C.c_int32), ("n_a", C.c_int32), ("sum_ov_x", C.c_int32), ("n_ov", C.c_int32),
                ("ratio", C.c_float), ("ov", C.c_float), ("branch", C.c_int32), ("start", C.c_int32)]

    def as_tuple(self):
        return (self.sum_a_x, self.n_a, self.sum_ov_x, self.n_ov, self.branch, self.start)


class PairDesc(C.Structure):
    """stitch_pair_desc: one independent stitch step of a batch (device pointers)."""
    _fields_ = [("frame", C.c_void_p), ("fw", C.c_int), ("fh", C.c_int), ("p", C.c_double * 8), ("offx", C.c_float),
                ("offy", C.c_float), ("mosaic", C.c_void_p), ("mw", C.c_int), ("mh", C.c_int), ("ox", C.c_int), ("oy", C.c_int),
                ("out", C.c_void_p), ("out_u8", C.c_void_p)]


_lib = None


def lib():
    """Load libstitch_hip.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C computervisionimagestich2_amd/csrc). There is no CPU fallback.")
        try:
            # One HIP runtime per process: torch bundles its own libamdhip64.so.7; loading it first makes the
            # loader bind this library's NEEDED libamdhip64.so.7 to the same copy, so tensors, streams and the
            # kernels below share one runtime (loading /opt/rocm's copy first leaves torch without a device).
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.stitch_last_error.restype = C.c_char_p
        L.stitch_plan_workspace_bytes.restype = C.c_size_t
        L.stitch_plan_workspace_bytes.argtypes = [C.c_void_p]
        L.stitch_plan_collapse_range.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.stitch_plan_workspace_base.restype = C.c_void_p
        L.stitch_plan_workspace_base.argtypes = [C.c_void_p]
        L.stitch_plan_fast_paths.argtypes = [C.c_void_p]
        L.stitch_plan_call_forms.argtypes = [C.c_void_p, C.c_int]
        L.stitch_plan_coarse_from.argtypes = [C.c_void_p]
        L.stitch_plan_destroy.restype = None
        L.stitch_plan_destroy.argtypes = [C.c_void_p]
        L.stitch_blend_opts_default.restype = None
        L.stitch_bmp_file_bytes.restype = C.c_size_t
        _lib = L
    return _lib


def _chk(rc):
    if rc < 0:
        raise StitchError(rc, lib().stitch_last_error().decode())
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _sfx(dtype):
    if dtype == np.uint8:
        return "u8"
    if dtype == np.float32:
        return "f32"
    raise TypeError(f"unsupported pixel type {dtype}")


def _img(a):
    a = np.ascontiguousarray(a)
    if a.ndim != 3 or a.shape[0] != 3:
        raise ValueError(f"expected a (3,H,W) planar image, got {a.shape}")
    return a


def _map8(p):
    p = [float(v) for v in p]
    if len(p) != 8:
        raise ValueError("the bilinear map has 8 parameters {H00,H01,H02,H10,H11,H12,H20,H21}")
    return (C.c_double * 8)(*p)


def _opts(opts):
    if opts is None:
        return BlendOpts()
    if isinstance(opts, BlendOpts):
        return opts
    return BlendOpts(**opts)


def device_count():
    return lib().stitch_device_count()


def plan_cache_query(cw, ch, opts=None):
    """(idle cached workspaces a host-buffer call for this canvas would take under the CURRENT environment's tuning
    switches, fused_sweep_levels of the first) -- stitch_plan_cache_query."""
    fused = C.c_int(-1)
    n = _chk(lib().stitch_plan_cache_query(int(cw), int(ch), C.byref(_opts(opts)), C.byref(fused)))
    return n, fused.value


def trim():
    lib().stitch_trim()


def pyramid_levels(w, h, level_rule=0):
    lw, lh = (C.c_int * 32)(), (C.c_int * 32)()
    n = _chk(lib().stitch_pyramid_levels(int(w), int(h), int(level_rule), lw, lh))
    return n, list(lw[:n]), list(lh[:n])


# ---- host-buffer entry points (numpy) ------------------------------------------------------------------------
def project(src, fov_deg=15.0):
    """Projection::imageProjection (Projection.cpp:20-73)."""
    src = _img(src)
    dst = np.empty_like(src)
    _, h, w = src.shape
    _chk(getattr(lib(), "stitch_project_" + _sfx(src.dtype))(_p(src), w, h, C.c_float(fov_deg), _p(dst)))
    return dst


def warp(src, p, offx, offy, canvas):
    """ImageProcess::warpingImageByHomography (ImageProcess.cpp:596-606); writes into `canvas` in place."""
    src = _img(src)
    assert canvas.flags.c_contiguous and canvas.dtype == src.dtype and canvas.shape[0] == 3
    _chk(getattr(lib(), "stitch_warp_" + _sfx(src.dtype))(_p(src), src.shape[2], src.shape[1], _map8(p), C.c_float(offx),
                                                          C.c_float(offy), _p(canvas), canvas.shape[2], canvas.shape[1]))
    return canvas


def move(src, ox, oy, canvas):
    """ImageProcess::movingImageByOffset (ImageProcess.cpp:608-620); writes into `canvas` in place."""
    src = _img(src)
    assert canvas.flags.c_contiguous and canvas.dtype == src.dtype and canvas.shape[0] == 3
    _chk(getattr(lib(), "stitch_move_" + _sfx(src.dtype))(_p(src), src.shape[2], src.shape[1], int(ox), int(oy), _p(canvas),
                                                          canvas.shape[2], canvas.shape[1]))
    return canvas


def blend(a, b, opts=None):
    """ImageProcess::blendTwoImages (ImageProcess.cpp:648-773) -> (out, Seam)."""
    a, b = _img(a), _img(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    out = np.empty_like(a)
    s = Seam()
    o = _opts(opts)
    _chk(getattr(lib(), "stitch_blend_" + _sfx(a.dtype))(_p(a), _p(b), a.shape[2], a.shape[1], C.byref(o), _p(out), C.byref(s)))
    return out, s


def pair(frame, p, offx, offy, mosaic, ox, oy, cw, ch, opts=None):
    """One stitch step (ImageProcess.cpp:218-230): warp `frame`, move `mosaic`, blend -> (out, Seam)."""
    frame, mosaic = _img(frame), _img(mosaic)
    assert frame.dtype == mosaic.dtype
    out = np.empty((3, ch, cw), frame.dtype)
    s = Seam()
    o = _opts(opts)
    _chk(getattr(lib(), "stitch_pair_" + _sfx(frame.dtype))(
        _p(frame), frame.shape[2], frame.shape[1], _map8(p), C.c_float(offx), C.c_float(offy), _p(mosaic), mosaic.shape[2],
        mosaic.shape[1], int(ox), int(oy), int(cw), int(ch), C.byref(o), _p(out), C.byref(s)))
    return out, s


def equalize(img):
    """equalization::equalization(img, 1) (equalization.cpp:4-25,74-131) -> (equalised copy, 256 Y bins)."""
    img = np.array(_img(img), dtype=np.uint8, copy=True)
    hist = np.zeros(256, np.int32)
    _chk(lib().stitch_equalize_u8(_p(img), img.shape[2], img.shape[1], _p(hist)))
    return img, hist


def lummix(result, equalized, num=19.0, den=20.0):
    """Luminance mix of ImageProcess::matching (ImageProcess.cpp:240-268) -> new array."""
    result = np.array(_img(result), dtype=np.uint8, copy=True)
    equalized = np.ascontiguousarray(equalized, np.uint8)
    _chk(lib().stitch_lummix_u8(_p(result), _p(equalized), result.shape[2], result.shape[1], C.c_double(num), C.c_double(den)))
    return result


def finish(result, num=19.0, den=20.0):
    """Tail of matching() in one call (ImageProcess.cpp:237-268): equalise a copy, mix -> (new array, Y bins)."""
    result = np.array(_img(result), dtype=np.uint8, copy=True)
    hist = np.zeros(256, np.int32)
    _chk(lib().stitch_finish_u8(_p(result), result.shape[2], result.shape[1], C.c_double(num), C.c_double(den), _p(hist)))
    return result, hist


def gray(rgb):
    """ImageProcess::toGrayScale + SIFT float staging (ImageProcess.cpp:27-40,47-51) -> (gray uint8 (H,W), float32 (H,W))."""
    rgb = np.ascontiguousarray(_img(rgb), np.uint8)
    _, h, w = rgb.shape
    g, f = np.empty((h, w), np.uint8), np.empty((h, w), np.float32)
    _chk(lib().stitch_gray_u8(_p(rgb), w, h, _p(g), _p(f)))
    return g, f


def project_gray(src, fov_deg=15.0):
    """readFile's per-image chain in one kernel (ImageProcess.cpp:18-20) -> (projected, gray, gray_f32)."""
    src = np.ascontiguousarray(_img(src), np.uint8)
    _, h, w = src.shape
    dst, g, f = np.empty_like(src), np.empty((h, w), np.uint8), np.empty((h, w), np.float32)
    _chk(lib().stitch_project_gray_u8(_p(src), w, h, C.c_float(fov_deg), _p(dst), _p(g), _p(f)))
    return dst, g, f


def transfer(src, tem):
    """transfer::transfer (transfer.cpp:3-13,125-225): l-alpha-beta colour transfer of tem's statistics onto src
    -> (out (3,H,W) uint8, stats float32[12] = mean/sd of src, mean/sd of tem)."""
    src, tem = np.ascontiguousarray(_img(src), np.uint8), np.ascontiguousarray(_img(tem), np.uint8)
    out, st = np.empty_like(src), np.zeros(12, np.float32)
    _chk(lib().stitch_transfer_u8(_p(src), src.shape[2], src.shape[1], _p(tem), tem.shape[2], tem.shape[1], _p(out), _p(st)))
    return out, st


def dev_transfer(d_src, d_tem, out=None, stats=None):
    """Device-resident colour transfer; out may be d_src itself."""
    import torch
    for t in (d_src, d_tem):
        assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.shape[0] == 3
    out = torch.empty_like(d_src) if out is None else out
    _chk(lib().stitch_dev_transfer_u8(_dp(d_src), int(d_src.shape[2]), int(d_src.shape[1]), _dp(d_tem), int(d_tem.shape[2]),
                                      int(d_tem.shape[1]), _dp(out), _dp(stats) if stats is not None else None, _stream()))
    return out


class BmpInfo(C.Structure):
    """stitch_bmp_info: what CImg's loader derives from the 54-byte header (CImg.h:48413-48441)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bpp", C.c_int32), ("top_down", C.c_int32),
                ("data_pos", C.c_uint64), ("stride", C.c_uint64), ("data_bytes", C.c_uint64)]


def bmp_parse(header, file_bytes):
    """Header arithmetic of CImg's load_bmp on the first 54 bytes of a file of `file_bytes` bytes.  Host only."""
    h = np.frombuffer(bytes(header[:54]), np.uint8)
    bi = BmpInfo()
    _chk(lib().stitch_bmp_parse(_p(h), C.c_size_t(int(file_bytes) if h.size >= 54 else h.size), C.byref(bi)))
    return bi


def bmp_decode(data):
    """Bytes of a 24/32-bit BMP file -> planar (3,H,W) uint8 RGB, as CImg<unsigned char>::load_bmp gives it."""
    buf = np.frombuffer(bytes(data), np.uint8)
    bi = bmp_parse(buf[:54].tobytes(), buf.size)
    out = np.empty((3, bi.height, bi.width), np.uint8)
    _chk(lib().stitch_bmp_decode_u8(_p(buf), C.c_size_t(buf.size), _p(out)))
    return out


def bmp_encode(img):
    """Planar (3,H,W) uint8 RGB -> the bytes CImg<unsigned char>::save_bmp writes."""
    img = np.ascontiguousarray(_img(img), np.uint8)
    _, h, w = img.shape
    n = lib().stitch_bmp_file_bytes(w, h)
    out = np.empty(n, np.uint8)
    _chk(lib().stitch_bmp_encode_u8(_p(img), w, h, _p(out), C.c_size_t(n)))
    return out.tobytes()


def dev_bmp_decode(d_file, info, out=None):
    """Device-resident file image (1-D uint8 tensor) -> planar (3,H,W) uint8 tensor; `info` from bmp_parse."""
    import torch
    assert d_file.is_cuda and d_file.dtype == torch.uint8 and d_file.is_contiguous()
    out = torch.empty((3, info.height, info.width), dtype=torch.uint8, device=d_file.device) if out is None else out
    _chk(lib().stitch_dev_bmp_decode_u8(_dp(d_file), C.c_size_t(d_file.numel()), C.byref(info), _dp(out), _stream()))
    return out


def dev_bmp_encode(d_img, out=None):
    """Planar (3,H,W) uint8 tensor -> device-resident file image (1-D uint8 tensor), byte-identical to save_bmp."""
    import torch
    assert d_img.is_cuda and d_img.dtype == torch.uint8 and d_img.is_contiguous() and d_img.shape[0] == 3
    _, h, w = d_img.shape
    n = lib().stitch_bmp_file_bytes(int(w), int(h))
    out = torch.empty(n, dtype=torch.uint8, device=d_img.device) if out is None else out
    _chk(lib().stitch_dev_bmp_encode_u8(_dp(d_img), int(w), int(h), _dp(out), C.c_size_t(out.numel()), _stream()))
    return out


def canvas_bbox(fw, fh, p_fwd, result_w, result_h):
    """Canvas of one stitch step (ImageProcess.cpp:206-216) -> (min_x, min_y, new_w, new_h).  Host arithmetic."""
    mx, my, nw, nh = C.c_float(), C.c_float(), C.c_int(), C.c_int()
    _chk(lib().stitch_canvas_bbox(int(fw), int(fh), _map8(p_fwd), int(result_w), int(result_h), C.byref(mx), C.byref(my),
                                  C.byref(nw), C.byref(nh)))
    return mx.value, my.value, nw.value, nh.value


class StepGeom(C.Structure):
    """stitch_step_geom: canvas and offsets of one stitch step (ImageProcess.cpp:206-216, :224)."""
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("cw", C.c_int), ("ch", C.c_int), ("ox", C.c_int), ("oy", C.c_int)]


def step_geometry(fw, fh, p_fwd, mw, mh):
    g = StepGeom()
    _chk(lib().stitch_step_geometry(int(fw), int(fh), _map8(p_fwd), int(mw), int(mh), C.byref(g)))
    return g


def dev_step(frame, p_fwd, p_bwd, mosaic, opts=None):
    """One stitch step of matching() from the FORWARD map, device resident (ImageProcess.cpp:206-230): canvas sizing,
    warp, move, blend -> (new mosaic tensor, StepGeom, Seam)."""
    import torch
    frame, mosaic = _timg(frame), _timg(mosaic)
    g = step_geometry(frame.shape[2], frame.shape[1], p_fwd, mosaic.shape[2], mosaic.shape[1])
    out = torch.empty((3, g.ch, g.cw), dtype=frame.dtype, device=frame.device)
    g2, s, o = StepGeom(), Seam(), _opts(opts)
    _chk(getattr(lib(), "stitch_dev_step_" + _tsfx(frame))(
        _dp(frame), frame.shape[2], frame.shape[1], _map8(p_fwd), _map8(p_bwd), _dp(mosaic), mosaic.shape[2], mosaic.shape[1], C.byref(o),
        _dp(out), C.c_size_t(out.numel()), C.byref(g2), C.byref(s), _stream()))
    return out, g2, s


def map_points(x, y, p_fwd, offx, offy):
    """updateFeaturesByHomography (ImageProcess.cpp:622-631) -> (x, y, ix, iy)."""
    x, y = np.array(x, np.float32), np.array(y, np.float32)
    ix, iy = np.empty(x.size, np.int32), np.empty(x.size, np.int32)
    _chk(lib().stitch_map_points(_p(x), _p(y), _p(ix), _p(iy), x.size, _map8(p_fwd), C.c_float(offx), C.c_float(offy)))
    return x, y, ix, iy


def shift_points(x, y, ox, oy):
    """updateFeaturesByOffset (ImageProcess.cpp:633-640) -> (x, y, ix, iy)."""
    x, y = np.array(x, np.float32), np.array(y, np.float32)
    ix, iy = np.empty(x.size, np.int32), np.empty(x.size, np.int32)
    _chk(lib().stitch_shift_points(_p(x), _p(y), _p(ix), _p(iy), x.size, int(ox), int(oy)))
    return x, y, ix, iy


# ---- device-resident entry points (torch tensors on the HIP device) --------------------------------------------
def _tsfx(t):
    import torch
    if t.dtype == torch.uint8:
        return "u8"
    if t.dtype == torch.float32:
        return "f32"
    raise TypeError(f"unsupported tensor dtype {t.dtype}")


def _timg(t):
    if not t.is_cuda or not t.is_contiguous() or t.dim() != 3 or t.shape[0] != 3:
        raise ValueError("expected a contiguous (3,H,W) tensor on the HIP device")
    return t


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dp(t):
    return C.c_void_p(t.data_ptr())


def dev_project(src, fov_deg=15.0, out=None):
    import torch
    src = _timg(src)
    out = torch.empty_like(src) if out is None else out
    _, h, w = src.shape
    _chk(getattr(lib(), "stitch_dev_project_" + _tsfx(src))(_dp(src), w, h, C.c_float(fov_deg), _dp(out), _stream()))
    return out


def dev_warp(src, p, offx, offy, canvas):
    src, canvas = _timg(src), _timg(canvas)
    _chk(getattr(lib(), "stitch_dev_warp_" + _tsfx(src))(_dp(src), src.shape[2], src.shape[1], _map8(p), C.c_float(offx),
                                                         C.c_float(offy), _dp(canvas), canvas.shape[2], canvas.shape[1], _stream()))
    return canvas


def dev_move(src, ox, oy, canvas):
    src, canvas = _timg(src), _timg(canvas)
    _chk(getattr(lib(), "stitch_dev_move_" + _tsfx(src))(_dp(src), src.shape[2], src.shape[1], int(ox), int(oy), _dp(canvas),
                                                         canvas.shape[2], canvas.shape[1], _stream()))
    return canvas


def dev_equalize(img, hist=None):
    """In place on a uint8 device tensor; `hist` (int32[256] device tensor, optional) receives the Y bins."""
    img = _timg(img)
    _chk(lib().stitch_dev_equalize_u8(_dp(img), img.shape[2], img.shape[1], _dp(hist) if hist is not None else None, _stream()))
    return img


def dev_lummix(result, equalized, num=19.0, den=20.0):
    result, equalized = _timg(result), _timg(equalized)
    _chk(lib().stitch_dev_lummix_u8(_dp(result), _dp(equalized), result.shape[2], result.shape[1], C.c_double(num),
                                    C.c_double(den), _stream()))
    return result


def dev_finish(result, num=19.0, den=20.0, hist=None):
    result = _timg(result)
    _chk(lib().stitch_dev_finish_u8(_dp(result), result.shape[2], result.shape[1], C.c_double(num), C.c_double(den),
                                    _dp(hist) if hist is not None else None, _stream()))
    return result


def dev_synth(w, h, frame_id, dtype, device=None):
    """Synthetic benchmark frame generated on the device -> (3,h,w) tensor."""
    import torch
    out = torch.empty((3, h, w), dtype=dtype, device=device or torch.device("cuda", torch.cuda.current_device()))
    _chk(getattr(lib(), "stitch_dev_synth_" + _tsfx(out))(_dp(out), int(w), int(h), int(frame_id), _stream()))
    return out


def dev_check_fastdiv(w):
    """stitch_dev_check_fastdiv: (numerators tested, quotients that differ from the IEEE divide) for denominator w."""
    lib().stitch_dev_check_fastdiv.argtypes = [C.c_float, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    t, m = C.c_ulonglong(), C.c_ulonglong()
    _chk(lib().stitch_dev_check_fastdiv(C.c_float(w), C.byref(t), C.byref(m)))
    return t.value, m.value


def dev_check_collapse_taps(w, h, probe):
    """stitch_dev_check_collapse_taps: (samples compared, samples that differ between k_collapse4's per-lane taps and k_collapse)."""
    lib().stitch_dev_check_collapse_taps.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    t, m = C.c_ulonglong(), C.c_ulonglong()
    _chk(lib().stitch_dev_check_collapse_taps(int(w), int(h), int(probe), C.byref(t), C.byref(m)))
    return t.value, m.value


def dev_quantize(src, out=None):
    """float mosaic -> unsigned char by truncation (CImg.h:11167-11182, behind ImageProcess.cpp:772)."""
    import torch
    assert src.is_cuda and src.is_contiguous() and src.dtype == torch.float32
    out = torch.empty(src.shape, dtype=torch.uint8, device=src.device) if out is None else out
    _chk(lib().stitch_dev_quantize_u8(_dp(src), _dp(out), C.c_size_t(src.numel()), _stream()))
    return out


class Plan:
    """stitch_plan: the device workspace of one canvas size (pyramids, scratch, tables, seam record)."""

    def __init__(self, cw, ch, opts=None, max_pairs=1):
        self._h = C.c_void_p()
        self.cw, self.ch, self.max_pairs = int(cw), int(ch), int(max_pairs)
        o = _opts(opts)
        _chk(lib().stitch_plan_create_batched(self.cw, self.ch, C.byref(o), self.max_pairs, C.byref(self._h)))
        lw, lh = (C.c_int * 32)(), (C.c_int * 32)()
        n = _chk(lib().stitch_plan_levels(self._h, lw, lh))
        self.level_w, self.level_h = list(lw[:n]), list(lh[:n])
        self.fused_sweep_levels = lib().stitch_plan_fused_sweep_levels(self._h)

    @property
    def levels(self):
        return len(self.level_w)

    @property
    def coarse_from(self):
        return lib().stitch_plan_coarse_from(self._h)

    @property
    def fast_paths(self):
        """stitch_plan_fast_paths as a set of names."""
        f = lib().stitch_plan_fast_paths(self._h)
        names = ("implicit_mask", "source_fused", "fused_sweep", "zero_tiles", "fused_decimate", "coarse_levels")
        return {n for i, n in enumerate(names) if f & (1 << i)}

    def call_forms(self, n_pairs=1):
        """stitch_plan_call_forms: the per-call forms ("source_fused", "fused_sweep") a call with n_pairs pairs runs."""
        f = lib().stitch_plan_call_forms(self._h, int(n_pairs))
        return {n for b, n in ((2, "source_fused"), (4, "fused_sweep")) if f & b}

    @property
    def workspace_bytes(self):
        return lib().stitch_plan_workspace_bytes(self._h)

    def collapse_range(self, level):
        """stitch_plan_collapse_range: (xa, xb, per_lane_taps) of the collapse of `level`."""
        xa, xb, g = C.c_int(), C.c_int(), C.c_int()
        _chk(lib().stitch_plan_collapse_range(self._h, int(level), C.byref(xa), C.byref(xb), C.byref(g)))
        return xa.value, xb.value, g.value

    @property
    def workspace_base(self):
        return lib().stitch_plan_workspace_base(self._h) or 0

    def close(self):
        if self._h:
            lib().stitch_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def blend(self, a, b, out=None):
        import torch
        a, b = _timg(a), _timg(b)
        assert a.shape == b.shape == (3, self.ch, self.cw)
        out = torch.empty_like(a) if out is None else out
        _chk(getattr(lib(), "stitch_dev_blend_" + _tsfx(a))(self._h, _dp(a), _dp(b), _dp(out), _stream()))
        return out

    def pair(self, frame, p, offx, offy, mosaic, ox, oy, out=None):
        """Enqueue warp+move+blend of one pair on torch's current stream; `out` is (3,ch,cw)."""
        import torch
        frame, mosaic = _timg(frame), _timg(mosaic)
        if out is None:
            out = torch.empty((3, self.ch, self.cw), dtype=frame.dtype, device=frame.device)
        _chk(getattr(lib(), "stitch_dev_pair_" + _tsfx(frame))(
            self._h, _dp(frame), frame.shape[2], frame.shape[1], _map8(p), C.c_float(offx), C.c_float(offy), _dp(mosaic),
            mosaic.shape[2], mosaic.shape[1], int(ox), int(oy), _dp(out), _stream()))
        return out

    def pairs(self, items):
        """Enqueue n <= max_pairs independent pairs as ONE launch sequence.  items: iterable of
        (frame, p, offx, offy, mosaic, ox, oy, out[, out_u8]) with device tensors; out_u8 (float frames only) receives the
        mosaic as unsigned char as well.  Returns the list of `out` tensors."""
        items = list(items)
        arr = (PairDesc * len(items))()
        sfx = None
        for d, it in zip(arr, items):
            frame, p, offx, offy, mosaic, ox, oy, out = it[:8]
            out8 = it[8] if len(it) > 8 else None
            frame, mosaic, out = _timg(frame), _timg(mosaic), _timg(out)
            assert tuple(out.shape) == (3, self.ch, self.cw) and out.dtype == frame.dtype == mosaic.dtype
            sfx = _tsfx(frame) if sfx is None else sfx
            assert sfx == _tsfx(frame), "one pixel type per batch"
            d.frame, d.fw, d.fh = frame.data_ptr(), frame.shape[2], frame.shape[1]
            d.p = _map8(p)
            d.offx, d.offy = float(offx), float(offy)
            d.mosaic, d.mw, d.mh = mosaic.data_ptr(), mosaic.shape[2], mosaic.shape[1]
            d.ox, d.oy, d.out = int(ox), int(oy), out.data_ptr()
            if out8 is not None:
                import torch
                assert out8.is_cuda and out8.is_contiguous() and out8.dtype == torch.uint8 and tuple(out8.shape) == (3, self.ch, self.cw)
                d.out_u8 = out8.data_ptr()
        _chk(getattr(lib(), "stitch_dev_pairs_" + sfx)(self._h, arr, len(items), _stream()))
        return [it[7] for it in items]

    def status(self, index=0):
        """Wait for the last call and return pair `index`'s Seam; raises StitchError for an empty mid row / zero
        overlap of that pair."""
        s = Seam()
        _chk(lib().stitch_plan_status_at(self._h, int(index), C.byref(s)))
        return s

    def clear_fault(self):
        """Acknowledge the fused sweep's sticky time-out report (stitch_plan_clear_fault)."""
        _chk(lib().stitch_plan_clear_fault(self._h))

    def set_handoff_spin_limit(self, polls):
        _chk(lib().stitch_plan_set_handoff_spin_limit(self._h, C.c_uint(int(polls))))

    def set_profiling_kernel(self, name):
        _chk(lib().stitch_plan_set_profiling_kernel(self._h, KERNELS.index(name)))

    def set_profiling(self, on):
        _chk(lib().stitch_plan_set_profiling(self._h, int(bool(on))))

    def read_profile(self):
        """-> {kernel: (total_ms, launches, level0_ms)} since profiling was enabled / last read."""
        ms = (C.c_double * len(KERNELS))()
        n = (C.c_int * len(KERNELS))()
        l0 = (C.c_double * len(KERNELS))()
        _chk(lib().stitch_plan_read_profile(self._h, ms, n, l0))
        return {KERNELS[i]: (ms[i], n[i], l0[i]) for i in range(len(KERNELS))}


class Band:
    """stitch_band: this rank's row band of ONE pair split over several GPUs (include/stitch.h, "one pair split into row
    bands").  Computation only; what crosses ranks is moved by pipeline.BandStitcher."""

    def __init__(self, cw, ch, rank, nranks, split_levels, opts=None):
        self._h = C.c_void_p()
        o = _opts(opts)
        _chk(lib().stitch_band_create(int(cw), int(ch), int(rank), int(nranks), int(split_levels), C.byref(o), C.byref(self._h)))
        self.cw, self.ch, self.rank, self.nranks, self.split_levels = int(cw), int(ch), int(rank), int(nranks), int(split_levels)
        tot = C.c_int()
        lib().stitch_band_levels(self._h, C.byref(tot))
        self.levels = tot.value
        self.geom = []
        for l in range(self.split_levels + 1):
            g = (C.c_int * 6)()
            _chk(lib().stitch_band_geometry(self._h, l, g))
            self.geom.append(dict(w=g[0], rows=g[1], row0=g[2], pitch=g[3], h=g[4], halo=g[5]))

    def close(self):
        if self._h:
            lib().stitch_band_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compose(self, frame, p, offx, offy, mosaic, ox, oy):
        frame, mosaic = _timg(frame), _timg(mosaic)
        _chk(getattr(lib(), "stitch_band_compose_" + _tsfx(frame))(
            self._h, _dp(frame), frame.shape[2], frame.shape[1], _map8(p), C.c_float(offx), C.c_float(offy), _dp(mosaic), mosaic.shape[2],
            mosaic.shape[1], int(ox), int(oy), _stream()))

    def set_level0(self, source_fused):
        """Level 0 source-fused (default) or materialised (what the one-plane-at-a-time sweeps need); before the next compose."""
        _chk(lib().stitch_band_set_level0(self._h, int(bool(source_fused))))

    def reduce_x(self, level):
        _chk(lib().stitch_band_reduce_x(self._h, int(level), _stream()))

    def reduce_xy_fwd(self, level, resume, state_out):
        """reduce_x + reduce_y_fwd(plane -1) with the anticausal x and causal y sweeps fused (one pass over the level)."""
        _chk(lib().stitch_band_reduce_xy_fwd(self._h, int(level), _dp(resume) if resume is not None else None, _dp(state_out), _stream()))

    def reduce_y_fwd(self, level, plane, resume, state_out):
        _chk(lib().stitch_band_reduce_y_fwd(self._h, int(level), int(plane), _dp(resume) if resume is not None else None, _dp(state_out), _stream()))

    def reduce_y_bwd(self, level, plane, fwd_state, resume, state_out):
        _chk(lib().stitch_band_reduce_y_bwd(self._h, int(level), int(plane), _dp(fwd_state), _dp(resume) if resume is not None else None,
                                            _dp(state_out), _stream()))

    def reduce_y_fwd_cols(self, level, x0, x1, resume, state_out):
        _chk(lib().stitch_band_reduce_y_fwd_cols(self._h, int(level), int(x0), int(x1), _dp(resume) if resume is not None else None, _dp(state_out), _stream()))

    def reduce_y_bwd_cols(self, level, x0, x1, fwd_state, resume, state_out):
        _chk(lib().stitch_band_reduce_y_bwd_cols(self._h, int(level), int(x0), int(x1), _dp(fwd_state), _dp(resume) if resume is not None else None,
                                                 _dp(state_out), _stream()))

    def rows(self, level, kind, first_row, nrows, buf, to_buffer):
        assert buf.is_cuda and buf.is_contiguous() and buf.dtype.is_floating_point and buf.element_size() == 4
        _chk(lib().stitch_band_rows(self._h, int(level), int(kind), int(first_row), int(nrows), _dp(buf), int(bool(to_buffer)), _stream()))

    def top(self, g7):
        assert g7.is_cuda and g7.is_contiguous()
        _chk(lib().stitch_band_top(self._h, _dp(g7), _stream()))

    def collapse(self, level, out=None):
        if level == 0:
            _chk(getattr(lib(), "stitch_band_collapse_" + _tsfx(out))(self._h, 0, _dp(out), _stream()))
        else:
            _chk(lib().stitch_band_collapse_f32(self._h, int(level), None, _stream()))

    def status(self):
        s = Seam()
        _chk(lib().stitch_band_status(self._h, C.byref(s)))
        return s
