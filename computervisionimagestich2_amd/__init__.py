"""MI355X-native per-pixel stitching hot path of chensh236/ComputerVisionImageStich2.

Product path: hand-written HIP kernels (csrc/) behind the C ABI of include/stitch.h, bound here with ctypes
(capi.py).  The host-side mirror of the reference's operators lives in capi (project / warp / move / blend /
pair / equalize / lummix / finish) and in pipeline (panorama chains, batches of pairs across GPUs).
"""
import os as _os
import sys as _sys

# Process start-up, the Python form of stitch_init() (include/stitch.h): the HIP runtime reads GPU_MAX_HW_QUEUES when it
# initialises, so the default goes into the environment here, at import -- before this package loads the library and, when
# the package is imported first, before torch brings the runtime up.  A runtime that is already up keeps what it had.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    _t = _sys.modules.get("torch")
    if _t is not None and getattr(_t, "cuda", None) is not None and _t.cuda.is_initialized():
        import warnings as _w
        _w.warn("computervisionimagestich2_amd: the HIP runtime was initialised before this package was imported, so "
                "GPU_MAX_HW_QUEUES=8 comes too late for this process (launch sequences on more than 4 streams will share "
                "hardware queues); import the package, or set the variable, before the first torch.cuda call", RuntimeWarning)

from . import capi  # noqa: F401,E402
from .capi import (BlendOpts, Plan, Seam, StitchError, blend, device_count, equalize, finish, lummix, move, pair, project,  # noqa: F401,E402
                   pyramid_levels, warp)

__all__ = ["capi", "BlendOpts", "Plan", "Seam", "StitchError", "blend", "device_count", "equalize", "finish", "lummix",
           "move", "pair", "project", "pyramid_levels", "warp"]
