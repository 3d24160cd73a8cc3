// Hand-written gfx950 (CDNA4) kernels of the stitching hot path.  Included only by stitch_hip.hip.
//
// Numerics contract: the reference semantics are x86-64 SSE2 -- IEEE, strict left to right,
// no FMA -- with the C++ promotions exactly as the reference writes them.  This translation unit is compiled
// with -ffp-contract=off (and the pragma below), float and double divide / sqrt are the correctly rounded
// forms, and every transcendental (tan, exp) is evaluated on the host and passed in.  Operation order inside
// each function follows the cited reference lines; what is free is the mapping of samples to work-items.
//
// Layout: the blend's planes live in a pitched layout -- row pitch = width rounded up to 64 floats (256 B),
// so every row starts on a 256-byte boundary and any 64-column tile can be moved with 16-byte-per-lane
// accesses without bounds tests.  User-facing frames and canvases stay dense CImg planar buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "stitch_elem.h"

#pragma clang fp contract(off)

namespace sk {

constexpr int WAVE = 64;
typedef float f4 __attribute__((ext_vector_type(4)));  // native 16-byte vector (keeps prefetch arrays in VGPRs)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct MapP {  // bilinear-map parameters, ImageProcess.h:58-73 order
    double p[8];
};

struct VVK {  // Van Vliet constants, CImg.h:34889-34903 / :35053-35065 (computed on the host)
    double f1, f2, f3, sumsq, sum, den;
    double M[9];
};

struct DRK {  // Deriche constants, CImg.h:34801-34816,34840-34841 (all float)
    float a0, a1, a2, a3, b1, b2, coefp, coefn;
};

struct SeamDev {  // written by k_seam, read by k_mask / the host
    int32_t sum_a_x, n_a, sum_ov_x, n_ov;
    float ratio, ov;
    int32_t branch, start;
    double thr;      // branch 0: mask = 1 where (double)x < thr
    int32_t status;  // 0 ok, -2 empty mid row, -3 zero overlap
    int32_t pad;
};

// ---- pixel helpers ---------------------------------------------------------------------------------------
template <typename PX>
__device__ __forceinline__ PX px_store(float f);
template <>
__device__ __forceinline__ uint8_t px_store<uint8_t>(float f) {
    return (uint8_t)(int)f;  // C-cast truncation (values are in [0,256))
}
template <>
__device__ __forceinline__ float px_store<float>(float f) {
    return f;
}

// Projection::bilinearInterpolation, Projection.cpp:3-18: ((1-a)(1-b))*ld + (a(1-b))*rd + (ab)*rt + ((1-a)b)*lt,
// summed left to right in float.  All three channels share the weights (the reference recomputes them per channel).
template <typename PX>
__device__ __forceinline__ void bilinear3(const PX* __restrict__ src, int w, int h, float x, float y, PX out[3]) {
    const int xf = (int)floorf(x), yf = (int)floorf(y);
    const float cx = ceilf(x), cy = ceilf(y);
    const int xc = cx >= (float)(w - 1) ? (w - 1) : (int)cx;
    const int yc = cy >= (float)(h - 1) ? (h - 1) : (int)cy;
    const float a = x - (float)xf, b = y - (float)yf;
    const float w_ld = (1 - a) * (1 - b), w_rd = a * (1 - b), w_rt = a * b, w_lt = (1 - a) * b;
    const size_t pl = (size_t)w * h;
    const size_t o_ld = (size_t)yf * w + xf, o_lt = (size_t)yc * w + xf, o_rd = (size_t)yf * w + xc,
                 o_rt = (size_t)yc * w + xc;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const PX* p = src + c * pl;
        const float ld = (float)p[o_ld], lt = (float)p[o_lt], rd = (float)p[o_rd], rt = (float)p[o_rt];
        out[c] = px_store<PX>(w_ld * ld + w_rd * rd + w_rt * rt + w_lt * lt);
    }
}

#include "k_formats.inc"  // gray + SIFT staging, BMP decode/encode, colour transfer
#include "k_geometry.inc"  // cylindrical projection, the bilinear map, stand-alone warp and move (P1, P2, W1-W3)
#include "k_gather.inc"  // pair descriptors of a batched launch, range-checked raw-buffer accesses
#include "k_project_lds.inc"  // the projection with source rows staged in LDS (P1, P2)
#include "k_compose.inc"  // S1: level-0 planes or their source index (source-fused level 0), seam scan and mask (B1)
#include "k_sweeps.inc"  // Van Vliet recursive Gaussian: causal / anticausal x and y sweeps, fused anticausal-y + decimation (B3, B4)
#include "k_sweeps1.inc"  // the y sweeps with one column per lane and scalar row offsets: launches that leave SIMDs idle (one pair in flight)
#include "k_fused_sweep.inc"  // the fused anticausal-x + causal-y sweep: row bands as pipeline stages (k_vv_xbyf)
#include "k_fused_sweep1.inc"  // the same for one pair in flight: five wavefronts per band (k_vv_xby_m)
#include "k_pyramid.inc"  // Deriche blur, stand-alone decimation, expand / Laplacian / blend / collapse (B3', B4-B6)
#include "k_coarse.inc"  // all levels below a size threshold in one launch: REDUCE to the top, top blend, collapse back up
#include "k_equalize.inc"  // equalisation and luminance mix (E1-E3, M1)
#include "k_synth.inc"  // synthetic frames and small utility kernels

}  // namespace sk
