"""MI355X-native per-pixel stitching hot path of chensh236/ComputerVisionImageStich2.

Product path: hand-written HIP kernels (csrc/) behind the C ABI of include/stitch.h, bound here with ctypes
(capi.py).  The host-side mirror of the reference's operators lives in capi (project / warp / move / blend /
pair / equalize / lummix / finish) and in pipeline (panorama chains, batches of pairs across GPUs).
"""
from . import capi  # noqa: F401
from .capi import (BlendOpts, Plan, Seam, StitchError, blend, device_count, equalize, finish, lummix, move, pair, project,  # noqa: F401
                   pyramid_levels, warp)

__all__ = ["capi", "BlendOpts", "Plan", "Seam", "StitchError", "blend", "device_count", "equalize", "finish", "lummix",
           "move", "pair", "project", "pyramid_levels", "warp"]
