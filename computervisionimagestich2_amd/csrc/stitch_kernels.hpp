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

// ---- gray conversion + SIFT input staging ---------------------------------------------
// ImageProcess::toGrayScale, ImageProcess.cpp:27-40: gray = 0.299*R + 0.587*G + 0.114*B evaluated in double on
// float-cast pixels, stored to unsigned char by truncation; siftAlgorithm stages it as float (:47-51).
__device__ __forceinline__ uint8_t gray_ref(uint8_t r, uint8_t g, uint8_t b) {
    return (uint8_t)(int)(0.299 * (double)(float)r + 0.587 * (double)(float)g + 0.114 * (double)(float)b);
}
__global__ __launch_bounds__(256) void k_gray(const uint8_t* __restrict__ rgb, size_t n, uint8_t* __restrict__ gray,
                                              float* __restrict__ gray_f32) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint8_t v = gray_ref(rgb[i], rgb[i + n], rgb[i + 2 * n]);
        if (gray) gray[i] = v;
        if (gray_f32) gray_f32[i] = (float)v;
    }
}

// ---- BMP <-> planar RGB (CImg.h:48395-48566 _load_bmp, :52614-52700 _save_bmp) ------------------------------
// The on-disk format either side of the path: rows bottom-up (or top-down when the header height is negative), bytes
// B G R [X], rows padded to 4 bytes.  One workgroup moves 1024 pixels of one file row; the file side is accessed in
// A-byte units (A = the alignment the file position allows, usually 2 because pixel data starts at byte 54), staged
// through LDS so that both the interleaved file bytes and the three planar rows are moved by coalesced accesses.
struct BmpGeom {
    int w, h, bp, top_down;  // bp = bytes per pixel in the file (3 or 4)
    unsigned long long data_pos, stride, data_bytes;
    int planar_vec;  // planar rows start 4-byte aligned (w % 4 == 0 and an aligned base): uchar4 accesses
};
constexpr int BMP_SEG = 1024;  // pixels per workgroup
template <int A>
struct BmpUnit;
template <>
struct BmpUnit<1> { typedef uint8_t type; };
template <>
struct BmpUnit<2> { typedef uint16_t type; };
template <>
struct BmpUnit<4> { typedef uint32_t type; };

template <int A>
__global__ __launch_bounds__(256) void k_bmp_decode(const uint8_t* __restrict__ file, BmpGeom g, uint8_t* __restrict__ planar) {
    typedef typename BmpUnit<A>::type U;
    __shared__ __attribute__((aligned(16))) uint8_t seg[BMP_SEG * 4];
    const int x0 = blockIdx.x * BMP_SEG, r = blockIdx.y;  // r = row in file order
    const int npx = min(BMP_SEG, g.w - x0), nbytes = npx * g.bp;
    const unsigned long long row_off = (unsigned long long)r * g.stride + (unsigned long long)x0 * g.bp;  // within the pixel data
    const uint8_t* src = file + g.data_pos + row_off;
    for (int u = threadIdx.x; u * A < nbytes; u += 256) {
        const unsigned long long o = row_off + (unsigned long long)u * A;
        U v = 0;
        if (o + A <= g.data_bytes)
            v = reinterpret_cast<const U*>(src)[u];
        else  // a file that ends early: the reference's zero-filled buffer (CImg.h:48445)
            for (int k = 0; k < A; ++k)
                if (o + k < g.data_bytes) v |= (U)((U)src[(size_t)u * A + k] << (8 * k));
        reinterpret_cast<U*>(seg)[u] = v;
    }
    __syncthreads();
    const int x = 4 * threadIdx.x;
    if (x >= npx) return;
    const int y = g.top_down ? r : g.h - 1 - r;  // CImg.h:48536 (rows arrive last-first), :48563 (mirror when dy < 0)
    const size_t pl = (size_t)g.w * g.h, o = (size_t)y * g.w + x0 + x;
    const uint8_t* p = seg + x * g.bp;
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // channel c = byte 2 - c of the pixel (CImg.h:48540-48542)
        if (g.planar_vec) {
            uchar4 v;
            v.x = p[2 - c];
            v.y = p[g.bp + 2 - c];
            v.z = p[2 * g.bp + 2 - c];
            v.w = p[3 * g.bp + 2 - c];
            *reinterpret_cast<uchar4*>(planar + c * pl + o) = v;
        } else
            for (int j = 0; j < 4 && x + j < npx; ++j) planar[c * pl + o + j] = p[j * g.bp + 2 - c];
    }
}

struct BmpHeader {
    uint8_t b[56];
};
template <int A>
__global__ __launch_bounds__(256) void k_bmp_encode(const uint8_t* __restrict__ planar, BmpGeom g, BmpHeader hdr, uint8_t* __restrict__ file) {
    typedef typename BmpUnit<A>::type U;
    __shared__ __attribute__((aligned(16))) uint8_t seg[BMP_SEG * 3 + 16];
    const int x0 = blockIdx.x * BMP_SEG, r = blockIdx.y;
    if (blockIdx.x == 0 && r == 0 && threadIdx.x < 54) file[threadIdx.x] = hdr.b[threadIdx.x];
    const int y = g.h - 1 - r;  // bottom-up (CImg.h:52672-52674, :52698)
    const size_t pl = (size_t)g.w * g.h;
    {
        const int x = x0 + 4 * threadIdx.x;
        uint8_t px[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};  // pixels beyond the row stay 0 = the row padding
        const size_t o = (size_t)y * g.w + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (g.planar_vec && x + 3 < g.w) {
                const uchar4 v = *reinterpret_cast<const uchar4*>(planar + c * pl + o);
                px[c][0] = v.x;
                px[c][1] = v.y;
                px[c][2] = v.z;
                px[c][3] = v.w;
            } else
                for (int j = 0; j < 4; ++j)
                    if (x + j < g.w) px[c][j] = planar[c * pl + o + j];
        }
        uint8_t* p = seg + 12 * threadIdx.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p[3 * j] = px[2][j];
            p[3 * j + 1] = px[1][j];
            p[3 * j + 2] = px[0][j];
        }
    }
    __syncthreads();
    // bytes of this row segment in the file: up to the end of the row including its padding
    const unsigned long long seg_off = (unsigned long long)x0 * 3;
    const int nbytes = (int)min((unsigned long long)BMP_SEG * 3, g.stride - seg_off);
    uint8_t* dst = file + 54 + (unsigned long long)r * g.stride + seg_off;
    for (int u = threadIdx.x; u * A < nbytes; u += 256) reinterpret_cast<U*>(dst)[u] = reinterpret_cast<const U*>(seg)[u];
}

// ---- l-alpha-beta colour transfer, transfer.cpp:3-13,125-225 (dead code in the reference) --
// The per-pixel arithmetic with every float/double promotion where the C++ puts it; std::log(float) and
// std::pow(10, float) are the specified functions of include/stitch_elem.h (see there).  Constants that the reference
// obtains from sqrt() are evaluated on the host.
struct TrK {
    float a1, b1, c1;  // 1/sqrt(3), 1/sqrt(6), 1/sqrt(2) as float (transfer.cpp:193-195)
    float a2, b2, c2;  // sqrt(3)/3, sqrt(6)/6, sqrt(2)/2 as float (transfer.cpp:204-206)
    double ln10;       // log(10)
};
__device__ __forceinline__ void tr_rgb_to_lab(const TrK& k, float R, float G, float B, float& L, float& a, float& b) {
    float l = (float)(0.3811 * (double)R + 0.5783 * (double)G + 0.0402 * (double)B);
    float m = (float)(0.1967 * (double)R + 0.7244 * (double)G + 0.0782 * (double)B);
    float s = (float)(0.0241 * (double)R + 0.1288 * (double)G + 0.8444 * (double)B);
    if (l == 0) l = 1;
    if (m == 0) m = 1;
    if (s == 0) s = 1;
    l = (float)((double)stitch_elem_logf(l) / k.ln10);
    m = (float)((double)stitch_elem_logf(m) / k.ln10);
    s = (float)((double)stitch_elem_logf(s) / k.ln10);
    L = k.a1 * ((l + m) + s);
    a = (float)((double)(k.b1 * l + k.b1 * m) - (2.0 * (double)k.b1) * (double)s);
    b = k.c1 * l - k.c1 * m;
}
__device__ __forceinline__ void tr_lab_to_rgb(const TrK& k, float L, float a, float b, float& R, float& G, float& B) {
    float l = (k.a2 * L + k.b2 * a) + k.c2 * b;
    float m = (k.a2 * L + k.b2 * a) - k.c2 * b;
    float s = (float)((double)(k.a2 * L) - (2.0 * (double)k.b2) * (double)a);
    l = (float)stitch_elem_pow10((double)l);
    m = (float)stitch_elem_pow10((double)m);
    s = (float)stitch_elem_pow10((double)s);
    const float r = (float)((4.4679 * (double)l - 3.5873 * (double)m) + 0.1193 * (double)s);
    const float g = (float)(((-1.2186) * (double)l + 2.3809 * (double)m) - 0.1624 * (double)s);
    const float bb = (float)((0.0497 * (double)l - 0.2439 * (double)m) + 1.2045 * (double)s);
    R = r > 0.0f ? (r < 255.0f ? r : 255.0f) : 0.0f;
    G = g > 0.0f ? (g < 255.0f ? g : 255.0f) : 0.0f;
    B = bb > 0.0f ? (bb < 255.0f ? bb : 255.0f) : 0.0f;
}
__global__ __launch_bounds__(256) void k_tr_to_lab(const uint8_t* __restrict__ rgb, size_t n, TrK k, float* __restrict__ lab) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float L, a, b;
        tr_rgb_to_lab(k, (float)rgb[i], (float)rgb[i + n], (float)rgb[i + 2 * n], L, a, b);
        lab[i] = L;
        lab[i + n] = a;
        lab[i + 2 * n] = b;
    }
}
// transfer.cpp:128-164: mean and standard deviation with FLOAT accumulators in raster order.  A float running sum is
// not associative, so the order is kept: one wavefront per (image, channel) chain walks its plane serially; the 64
// lanes only fetch (256 samples ahead, double-buffered in LDS) and square, every lane then adds the same samples in
// the same order.  Six chains run side by side.  stats = [mean_src[3], sd_src[3], mean_tem[3], sd_tem[3]].
__global__ __launch_bounds__(64) void k_tr_stats(const float* __restrict__ lab_s, size_t ns, float cnt_s, const float* __restrict__ lab_t,
                                                 size_t nt, float cnt_t, float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float buf[2][256];
    const int chain = blockIdx.x, c = chain % 3, lane = threadIdx.x;
    const bool is_t = chain >= 3;
    const size_t n = is_t ? nt : ns;
    const float* __restrict__ p = (is_t ? lab_t : lab_s) + (size_t)c * n;
    const float cnt = is_t ? cnt_t : cnt_s;
    const size_t nblk = (n + 255) / 256;
    float mean = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        // out-of-range slots add exactly nothing: 0 in the sum pass, (mean - mean)^2 = 0 in the squares pass
        const float pad = pass ? mean : 0.f;
        float r[4];
        auto fetch = [&](size_t b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t i = b * 256 + (size_t)j * 64 + lane;
                r[j] = i < n ? p[i] : pad;
            }
        };
        auto stash = [&](int which) {
#pragma unroll
            for (int j = 0; j < 4; ++j) buf[which][j * 64 + lane] = pass ? (r[j] - mean) * (r[j] - mean) : r[j];
        };
        fetch(0);
        stash(0);
        __syncthreads();
        float acc = 0.f;
        for (size_t b = 0; b < nblk; ++b) {
            if (b + 1 < nblk) fetch(b + 1);
            // the add chain is the critical path (one dependent v_add_f32 per sample): LDS reads run one 16-sample
            // group ahead of it, in registers
            const f4* q = reinterpret_cast<const f4*>(buf[b & 1]);
            f4 cur[4], nxt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = q[j];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (g + 1 < 16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) nxt[j] = q[4 * (g + 1) + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc += cur[j].x;
                    acc += cur[j].y;
                    acc += cur[j].z;
                    acc += cur[j].w;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
            }
            if (b + 1 < nblk) stash((int)((b + 1) & 1));
            __syncthreads();
        }
        if (pass == 0)
            mean = acc / cnt;
        else if (lane == 0) {
            stats[(is_t ? 6 : 0) + c] = mean;
            stats[(is_t ? 9 : 3) + c] = sqrtf(acc / cnt);
        }
    }
}
__global__ __launch_bounds__(256) void k_tr_apply(const float* __restrict__ lab, size_t n, const float* __restrict__ stats, TrK k,
                                                  uint8_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float st[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) st[i] = stats[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v[3], R, G, B;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (lab[i + (size_t)c * n] - st[c]) * st[9 + c] / st[3 + c] + st[6 + c];  // transfer.cpp:168-170
        tr_lab_to_rgb(k, v[0], v[1], v[2], R, G, B);
        out[i] = px_store<uint8_t>(R);  // the CImg<float> -> CImg<unsigned char> assignment of transfer.cpp:12
        out[i + n] = px_store<uint8_t>(G);
        out[i + 2 * n] = px_store<uint8_t>(B);
    }
}

// ---- P1: cylindrical projection, Projection.cpp:20-73 ------------------------------------------------------
// One output pixel (three channels) per work-item; r is computed on the host (tan).  Writes 0 where the
// source coordinate falls outside, so no memset pass is needed.
template <typename PX>
__global__ __launch_bounds__(256) void k_project(const PX* __restrict__ src, PX* __restrict__ dst, int w, int h,
                                                 int flag, int width, int height, float r, uint8_t* __restrict__ gray,
                                                 float* __restrict__ gray_f32) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const float dst_x = (float)((flag ? y : x) - width / 2);
    const float dst_y = (float)((flag ? x : y) - height / 2);
    const double rd = (double)r, dx = (double)dst_x;
    const float k = (float)(rd / sqrt(rd * rd + dx * dx));
    const float src_x = dst_x / k, src_y = dst_y / k;
    const float u = src_x + (float)(width / 2);
    const float v = src_y + (float)(height / 2);
    PX o[3] = {PX(0), PX(0), PX(0)};
    if (u >= 0 && u < (float)width && v >= 0 && v < (float)height) {
        if (flag)
            bilinear3<PX>(src, w, h, v, u, o);
        else
            bilinear3<PX>(src, w, h, u, v, o);
    }
    const size_t pl = (size_t)w * h, off = (size_t)y * w + x;
    dst[off] = o[0];
    dst[off + pl] = o[1];
    dst[off + 2 * pl] = o[2];
    if (sizeof(PX) == 1 && (gray || gray_f32)) {  // readFile's next step on the same pixel (ImageProcess.cpp:20)
        const uint8_t v = gray_ref((uint8_t)o[0], (uint8_t)o[1], (uint8_t)o[2]);
        if (gray) gray[off] = v;
        if (gray_f32) gray_f32[off] = (float)v;
    }
}

// ---- W1: the bilinear map, ImageProcess.cpp:465-471 --------------------------------------------------------
// double: ((p0*x + p1*y) + (p2*x)*y) + p3, rounded to float; then `int newX = float` truncation.  Values that
// do not fit an int (x86 gives INT_MIN, which fails the range test) and NaN are reported as "outside".
__device__ __forceinline__ bool map_to_src(const MapP& m, float fx, float fy, int sw, int sh, int& nx, int& ny) {
    const double dx = (double)fx, dy = (double)fy;
    const float X = (float)(m.p[0] * dx + m.p[1] * dy + m.p[2] * dx * dy + m.p[3]);
    const float Y = (float)(m.p[4] * dx + m.p[5] * dy + m.p[6] * dx * dy + m.p[7]);
    if (!(X > -2147483648.0f && X < 2147483648.0f) || !(Y > -2147483648.0f && Y < 2147483648.0f)) return false;
    nx = (int)X;
    ny = (int)Y;
    return nx >= 0 && nx < sw && ny >= 0 && ny < sh;
}

// The degenerate bilinear call of the warp (ImageProcess.cpp:602): a = b = 0, so the value is
// ((1*1)*ld + (0*1)*rd + (0*0)*rt + (1*0)*lt) = ld*1 + rd*0 + rt*0 + lt*0 with all four taps = ld.
// For every float that expression equals 0*ld + ld rounded once: a finite ld gives 1*ld = ld and 0*ld = a zero of ld's sign,
// and ld plus zeros of its own sign is ld (-0 included); an infinite or NaN ld gives NaN either way.  One v_fma_f32 instead
// of seven operations -- the causal x sweep and the collapse of a source-fused level 0 evaluate it for every sample, and the
// sweep is bound by instruction issue.
__device__ __forceinline__ float warp_tap(float ld) { return __builtin_fmaf(0.f, ld, ld); }
// px_store<PX>(warp_tap(t)) as a float, t a sample of a PX frame: a byte value comes through both steps unchanged
template <typename PX>
__device__ __forceinline__ float warped_px(float t);
template <>
__device__ __forceinline__ float warped_px<float>(float t) {
    return warp_tap(t);
}
template <>
__device__ __forceinline__ float warped_px<uint8_t>(float t) {
    return t;
}

// ---- W2 / W3 stand-alone (read-modify-write canvases of the C++ seam) ---------------------------------------
template <typename PX>
__global__ __launch_bounds__(256) void k_warp(const PX* __restrict__ src, int sw, int sh, MapP m, float offx,
                                              float offy, PX* __restrict__ canvas, int cw, int ch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= cw) return;
    int nx, ny;
    if (!map_to_src(m, (float)x + offx, (float)y + offy, sw, sh, nx, ny)) return;
    const size_t spl = (size_t)sw * sh, so = (size_t)ny * sw + nx, cpl = (size_t)cw * ch, co = (size_t)y * cw + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) canvas[co + c * cpl] = px_store<PX>(warp_tap((float)src[so + c * spl]));
}

template <typename PX>
__global__ __launch_bounds__(256) void k_move(const PX* __restrict__ src, int sw, int sh, int ox, int oy,
                                              PX* __restrict__ canvas, int cw, int ch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= cw) return;
    const long long nx = (long long)x + ox, ny = (long long)y + oy;
    if (nx < 0 || nx >= sw || ny < 0 || ny >= sh) return;
    const size_t spl = (size_t)sw * sh, so = (size_t)ny * sw + nx, cpl = (size_t)cw * ch, co = (size_t)y * cw + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) canvas[co + c * cpl] = src[so + c * spl];
}

// ---- S1: compose = warp + move + value cast, straight into the level-0 planes -------------------------------
// ImageProcess.cpp:218-224 + :680-681.  Level-0 planes of pair b: [a0 a1 a2 b0 b1 b2 mask], pitched, plane
// stride ps, pairs stacked (7*ps apart).  The zero canvases of the reference are implicit: an out-of-range pixel
// is written as 0.  A launch covers every pair of the batch (blockIdx.z = pair).
constexpr int MAXB = 16;  // pairs per plan / launch
template <typename PX>
struct PairArgs {
    const PX* frame[MAXB];
    const PX* mosaic[MAXB];
    PX* out[MAXB];
    MapP map[MAXB];
    int fw[MAXB], fh[MAXB], mw[MAXB], mh[MAXB], ox[MAXB], oy[MAXB];
    float offx[MAXB], offy[MAXB];
};

// Range-checked gathers of input samples: a raw buffer descriptor over one channel plane; a byte offset beyond the
// plane (the index plane's "outside") reads as 0, which is the reference's untouched zero canvas.
template <typename PX>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const PX* plane, size_t elems) {
    // the inputs ARE wave-uniform (kernel arguments indexed by block-derived scalars); readfirstlane makes that provable,
    // otherwise every buffer load is wrapped in a waterfall loop
    const unsigned long long a = reinterpret_cast<unsigned long long>(plane);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned bytes = __builtin_amdgcn_readfirstlane((unsigned)(elems * sizeof(PX)));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<PX*>(((unsigned long long)hi << 32) | lo), (short)0, (int)bytes, 0x00020000);
}
template <typename PX>
__device__ __forceinline__ float buf_px(__amdgpu_buffer_rsrc_t r, unsigned byte_off);
template <>
__device__ __forceinline__ float buf_px<float>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
template <>
__device__ __forceinline__ float buf_px<uint8_t>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return (float)__builtin_amdgcn_raw_buffer_load_b8(r, byte_off, 0, 0);
}
// the same loads, value left as raw bits (converted where it is consumed, so that nothing waits on the load early)
template <typename PX>
__device__ __forceinline__ float buf_raw(__amdgpu_buffer_rsrc_t r, unsigned byte_off);
template <>
__device__ __forceinline__ float buf_raw<float>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
template <>
__device__ __forceinline__ float buf_raw<uint8_t>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, byte_off, 0, 0));
}
template <typename PX>
__device__ __forceinline__ float raw_to_px(float raw);
template <>
__device__ __forceinline__ float raw_to_px<float>(float raw) {
    return raw;
}
template <>
__device__ __forceinline__ float raw_to_px<uint8_t>(float raw) {
    return (float)__float_as_uint(raw);
}
// "outside": a byte offset no plane reaches (the host keeps planes below 0xfffffff0 bytes), aligned for the access
template <typename PX>
__device__ __forceinline__ constexpr unsigned off_outside() {
    return 0u - (unsigned)sizeof(PX);
}

// The same projection for portrait / square frames (the reference's `flag == 0` branch), source rows tiled into LDS.
// k_project is bound by the vector L1: twelve tap loads per pixel, each touching its own cache line(s).  Here a workgroup
// owns an output tile of TW columns x TH rows.  The cylinder only stretches (1/k = sqrt(r^2 + dx^2)/r >= 1 and grows with
// |dx|), so the source samples of the tile lie in a box that its corner columns and rows bound: columns floor(u(x_first))
// .. ceil(u(x_last)), rows between the extremes of v over the corners.  The workgroup fetches that box once with coalesced
// 16-byte loads into LDS (three channels), then every work-item -- one output column, TH / (256 / TW) rows, so that the
// double-precision k of its column is evaluated once -- takes its four taps per channel from LDS.  Arithmetic per pixel and
// its order are k_project's / bilinear3's.  The host checks that the largest box of the frame fits the LDS budget.
// (Measured at 4096 x 4096: 0.069 ms u8 / 0.113 ms f32 against k_project's 0.107 / 0.146.  Tried and slower: four columns per
// work-item without LDS, 0.134 / 0.183 -- more cache lines per load instruction; the column terms from a table written by a
// kernel of its own, 0.073 / 0.134 -- the extra launch costs more than the redundant double-precision work it removes.)
template <typename PX>
struct Px4;
template <>
struct Px4<uint8_t> {
    typedef unsigned type;
    static __device__ __forceinline__ type pack(const uint8_t v[4]) { return (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24); }
};
template <>
struct Px4<float> {
    typedef f4 type;
    static __device__ __forceinline__ type pack(const float v[4]) { return f4{v[0], v[1], v[2], v[3]}; }
};
struct ProjCol {
    float k, u;
};
__device__ __forceinline__ ProjCol proj_col(int x, int w, float r) {  // Projection.cpp:33-37 for one column
    const float dst_x = (float)(x - w / 2);
    const double rd = (double)r, dx = (double)dst_x;
    ProjCol c;
    c.k = (float)(rd / sqrt(rd * rd + dx * dx));
    c.u = dst_x / c.k + (float)(w / 2);
    return c;
}
__device__ __forceinline__ float proj_v(int y, int h, float k) { return (float)(y - h / 2) / k + (float)(h / 2); }
constexpr int PJ_CHUNK = 16;  // bytes per staging access
// output tile per workgroup: columns x rows, per pixel type (256 work-items: TH / (256 / TW) rows per work-item)
#ifndef STITCH_PJ_TW_U8
#define STITCH_PJ_TW_U8 128
#endif
#ifndef STITCH_PJ_TH_U8
#define STITCH_PJ_TH_U8 16
#endif
#ifndef STITCH_PJ_TW_F32
#define STITCH_PJ_TW_F32 64
#endif
#ifndef STITCH_PJ_TH_F32
#define STITCH_PJ_TH_F32 32
#endif
constexpr int PJ_TW_U8 = STITCH_PJ_TW_U8, PJ_TH_U8 = STITCH_PJ_TH_U8, PJ_TW_F32 = STITCH_PJ_TW_F32, PJ_TH_F32 = STITCH_PJ_TH_F32;
// Staging of a tile's source box (three channels, rows r0 .. r0+nrow-1, columns c0a .. c0a+ncol-1) into LDS with 16-byte
// loads: 32 lanes across a row's chunks, 8 rows per pass (no division by the run-time chunk count).  All loads of a work-item
// are issued before the first LDS store -- a loop of load -> store iterations put up to nine HBM round trips of a workgroup
// one behind the other, which was half of the float kernel's time (0.134 -> 0.067 ms at 4096 x 4096 with the staging ablated).
// A chunk that is not this lane's reads at an offset beyond the buffer: no traffic, the value is dropped.
// PJ_MAXP row passes are held in registers (boxes of up to 8 PJ_MAXP rows and 32 chunks per row; larger ones loop): the box of a
// TH-row tile has about 1.1 TH + 3 rows.
template <typename PX, int PJ_MAXP>
__device__ __forceinline__ void pj_stage(const PX* __restrict__ src, size_t pl, int w, int r0, int nrow, int c0a, int ncol, uint8_t* smem) {
    constexpr int CPX = PJ_CHUNK / (int)sizeof(PX);
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc(src, 3 * pl);  // a chunk that runs past the last row reads 0
    const int cpr = ncol / CPX, lc = threadIdx.x & 31, lr = threadIdx.x >> 5;
    if (cpr <= 32 && nrow <= 8 * PJ_MAXP) {
        u4 v[3][PJ_MAXP];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < PJ_MAXP; ++p) {
                const int rr = lr + 8 * p;
                const bool mine = rr < nrow && lc < cpr;
                const unsigned off = mine ? (unsigned)((c * pl + (size_t)(r0 + rr) * w + c0a + lc * CPX) * sizeof(PX)) : 0xfffffff0u;
                v[c][p] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < PJ_MAXP; ++p) {
                const int rr = lr + 8 * p;
                if (rr < nrow && lc < cpr) *reinterpret_cast<u4*>(smem + ((size_t)(c * nrow + rr) * ncol + lc * CPX) * sizeof(PX)) = v[c][p];
            }
        return;
    }
    for (int c = 0; c < 3; ++c)
        for (int rr = lr; rr < nrow; rr += 8)
            for (int cc = lc; cc < cpr; cc += 32) {
                const unsigned off = (unsigned)((c * pl + (size_t)(r0 + rr) * w + c0a + cc * CPX) * sizeof(PX));
                *reinterpret_cast<u4*>(smem + ((size_t)(c * nrow + rr) * ncol + cc * CPX) * sizeof(PX)) =
                    __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
}

template <typename PX, int TW, int TH>
__global__ __launch_bounds__(256) void k_project_lds(const PX* __restrict__ src, PX* __restrict__ dst, int w, int h, float r,
                                                     uint8_t* __restrict__ gray, float* __restrict__ gray_f32, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pj_smem[];
    __shared__ ProjCol corner[4];
    constexpr int CPX = PJ_CHUNK / (int)sizeof(PX);  // pixels per staging chunk
    constexpr int RPT = TH / (256 / TW);             // rows per work-item
    const int xa = blockIdx.x * TW, ya = blockIdx.y * TH;
    const int xb = min(xa + TW, w) - 1, yb = min(ya + TH, h) - 1;
    // ---- the tile's source box: four columns decide it (first, last, nearest to and farthest from the axis), evaluated by
    // four work-items and shared ----
    if (threadIdx.x < 4) {
        const int xmid = w / 2, xnear = xa <= xmid && xmid <= xb ? xmid : (abs(xa - xmid) < abs(xb - xmid) ? xa : xb),
                  xfar = abs(xa - xmid) > abs(xb - xmid) ? xa : xb;
        const int xs = threadIdx.x == 0 ? xa : threadIdx.x == 1 ? xb : threadIdx.x == 2 ? xnear : xfar;
        corner[threadIdx.x] = proj_col(xs, w, r);
    }
    // meanwhile every work-item evaluates its own column (one output column, RPT rows per work-item)
    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int x = min(xa + tx, xb);
    const ProjCol col = proj_col(x, w, r);
    __syncthreads();
    int c0 = (int)floorf(corner[0].u), c1 = (int)ceilf(corner[1].u);  // u grows with x
    c0 = max(c0, 0);
    c1 = min(c1, w - 1);
    // for a row, |v - h/2| grows with |dx|: the columns nearest to / farthest from the axis bound v over the tile
    const float kn = corner[2].k, kf = corner[3].k;
    const float v00 = proj_v(ya, h, kn), v01 = proj_v(ya, h, kf), v10 = proj_v(yb, h, kn), v11 = proj_v(yb, h, kf);
    int r0 = (int)floorf(fminf(fminf(v00, v01), fminf(v10, v11))), r1 = (int)ceilf(fmaxf(fmaxf(v00, v01), fmaxf(v10, v11)));
    r0 = max(r0, 0);
    r1 = min(r1, h - 1);
    const int c0a = c0 / CPX * CPX;                          // chunk-aligned first column
    const int ncol = ((c1 - c0a + 1) + CPX - 1) / CPX * CPX;  // staged columns per row
    const int nrow = max(r1 - r0 + 1, 0);
    const size_t pl = (size_t)w * h;
    const bool fits = c1 >= c0 && (size_t)3 * nrow * ncol * sizeof(PX) <= (size_t)lds_bytes;  // the host sized lds_bytes for every tile
    PX* tile = reinterpret_cast<PX*>(pj_smem);
    if (fits) {
        pj_stage<PX, ((TH + TH / 8 + 11) / 8 > 4 ? (TH + TH / 8 + 11) / 8 : 4)>(src, pl, w, r0, nrow, c0a, ncol, pj_smem);
    }
    __syncthreads();
    if (xa + tx > xb) return;
    const float u = col.u;
    const bool xin = u >= 0 && u < (float)w;
    const int xf = (int)floorf(u);
    const float cx = ceilf(u);
    const int xc = cx >= (float)(w - 1) ? (w - 1) : (int)cx;
    const float a = u - (float)xf;
    const int lx = xin ? xf - c0a : 0, sx = xc != xf ? 1 : 0;
    for (int q = 0; q < RPT; ++q) {
        const int y = ya + ty * RPT + q;
        if (y > yb) break;
        const float v = proj_v(y, h, col.k);
        const bool in = xin && v >= 0 && v < (float)h;
        PX o[3] = {PX(0), PX(0), PX(0)};
        if (in) {
            const int yf = (int)floorf(v);
            const float cy = ceilf(v);
            const int yc = cy >= (float)(h - 1) ? (h - 1) : (int)cy;
            const float b = v - (float)yf;
            const float w_ld = (1 - a) * (1 - b), w_rd = a * (1 - b), w_rt = a * b, w_lt = (1 - a) * b;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float ld, rd_, lt, rt;
                if (fits) {
                    const PX* t = tile + (size_t)(c * nrow + (yf - r0)) * ncol + lx;
                    const PX* t2 = t + (size_t)(yc - yf) * ncol;
                    ld = (float)t[0], rd_ = (float)t[sx], lt = (float)t2[0], rt = (float)t2[sx];
                } else {  // never taken when the host sized the LDS from the frame's largest box; kept for safety
                    const PX* pc = src + c * pl;
                    ld = (float)pc[(size_t)yf * w + xf], rd_ = (float)pc[(size_t)yf * w + xc];
                    lt = (float)pc[(size_t)yc * w + xf], rt = (float)pc[(size_t)yc * w + xc];
                }
                o[c] = px_store<PX>(w_ld * ld + w_rd * rd_ + w_rt * rt + w_lt * lt);
            }
        }
        const size_t off = (size_t)y * w + x;
        dst[off] = o[0];
        dst[off + pl] = o[1];
        dst[off + 2 * pl] = o[2];
        if constexpr (sizeof(PX) == 1) {
            if (gray || gray_f32) {  // readFile's next step on the same pixel (ImageProcess.cpp:20)
                const uint8_t gv = gray_ref((uint8_t)o[0], (uint8_t)o[1], (uint8_t)o[2]);
                if (gray) gray[off] = gv;
                if (gray_f32) gray_f32[off] = (float)gv;
            }
        }
    }
}

// Landscape frames (the reference's `flag == 1` branch, Projection.cpp:24-26,30-49: the roles of the axes swap): k and the
// source ROW u depend on the output row alone, the source column v = (x - w/2)/k(y) + w/2 on both.  Same tiling; the per-row
// terms of a tile's TH rows are evaluated by TH work-items and shared through LDS, the tile's source box is bounded by its
// first and last row (u grows with y) and, for the columns, by its rows nearest to / farthest from the axis.
template <typename PX, int TW, int TH>
__global__ __launch_bounds__(256) void k_project_lds_t(const PX* __restrict__ src, PX* __restrict__ dst, int w, int h, float r,
                                                       uint8_t* __restrict__ gray, float* __restrict__ gray_f32, int lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pj_smem[];
    __shared__ ProjCol corner[4], rowc[TH];
    constexpr int CPX = PJ_CHUNK / (int)sizeof(PX);
    constexpr int RPT = TH / (256 / TW);
    const int xa = blockIdx.x * TW, ya = blockIdx.y * TH;
    const int xb = min(xa + TW, w) - 1, yb = min(ya + TH, h) - 1;
    if (threadIdx.x < 4) {
        const int ymid = h / 2, ynear = ya <= ymid && ymid <= yb ? ymid : (abs(ya - ymid) < abs(yb - ymid) ? ya : yb),
                  yfar = abs(ya - ymid) > abs(yb - ymid) ? ya : yb;
        const int ys = threadIdx.x == 0 ? ya : threadIdx.x == 1 ? yb : threadIdx.x == 2 ? ynear : yfar;
        corner[threadIdx.x] = proj_col(ys, h, r);  // the "width" of this branch is the image height
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + TH) rowc[threadIdx.x - 64] = proj_col(min(ya + (int)threadIdx.x - 64, yb), h, r);
    __syncthreads();
    int r0 = (int)floorf(corner[0].u), r1 = (int)ceilf(corner[1].u);  // u grows with y
    r0 = max(r0, 0);
    r1 = min(r1, h - 1);
    const float kn = corner[2].k, kf = corner[3].k;
    const float v00 = proj_v(xa, w, kn), v01 = proj_v(xa, w, kf), v10 = proj_v(xb, w, kn), v11 = proj_v(xb, w, kf);
    int c0 = (int)floorf(fminf(fminf(v00, v01), fminf(v10, v11))), c1 = (int)ceilf(fmaxf(fmaxf(v00, v01), fmaxf(v10, v11)));
    c0 = max(c0, 0);
    c1 = min(c1, w - 1);
    const int c0a = c0 / CPX * CPX;
    const int ncol = ((c1 - c0a + 1) + CPX - 1) / CPX * CPX;
    const int nrow = max(r1 - r0 + 1, 0);
    const size_t pl = (size_t)w * h;
    const bool fits = c1 >= c0 && (size_t)3 * nrow * ncol * sizeof(PX) <= (size_t)lds_bytes;
    PX* tile = reinterpret_cast<PX*>(pj_smem);
    if (fits) {
        pj_stage<PX, ((TH + TH / 8 + 11) / 8 > 4 ? (TH + TH / 8 + 11) / 8 : 4)>(src, pl, w, r0, nrow, c0a, ncol, pj_smem);
    }
    __syncthreads();
    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int x = xa + tx;
    if (x > xb) return;
    for (int q = 0; q < RPT; ++q) {
        const int y = ya + ty * RPT + q;
        if (y > yb) break;
        const ProjCol rc = rowc[y - ya];
        const float u = rc.u, v = proj_v(x, w, rc.k);
        const bool in = u >= 0 && u < (float)h && v >= 0 && v < (float)w;  // width = h, height = w in this branch
        PX o[3] = {PX(0), PX(0), PX(0)};
        if (in) {  // bilinear3(src, w, h, x = v, y = u)
            const int xf = (int)floorf(v), yf = (int)floorf(u);
            const float cx = ceilf(v), cy = ceilf(u);
            const int xc = cx >= (float)(w - 1) ? (w - 1) : (int)cx;
            const int yc = cy >= (float)(h - 1) ? (h - 1) : (int)cy;
            const float a = v - (float)xf, b = u - (float)yf;
            const float w_ld = (1 - a) * (1 - b), w_rd = a * (1 - b), w_rt = a * b, w_lt = (1 - a) * b;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float ld, rd_, lt, rt;
                if (fits) {
                    const PX* t = tile + (size_t)(c * nrow + (yf - r0)) * ncol + (xf - c0a);
                    const PX* t2 = t + (size_t)(yc - yf) * ncol;
                    ld = (float)t[0], rd_ = (float)t[xc - xf], lt = (float)t2[0], rt = (float)t2[xc - xf];
                } else {
                    const PX* pc = src + c * pl;
                    ld = (float)pc[(size_t)yf * w + xf], rd_ = (float)pc[(size_t)yf * w + xc];
                    lt = (float)pc[(size_t)yc * w + xf], rt = (float)pc[(size_t)yc * w + xc];
                }
                o[c] = px_store<PX>(w_ld * ld + w_rd * rd_ + w_rt * rt + w_lt * lt);
            }
        }
        const size_t off = (size_t)y * w + x;
        dst[off] = o[0];
        dst[off + pl] = o[1];
        dst[off + 2 * pl] = o[2];
        if constexpr (sizeof(PX) == 1) {
            if (gray || gray_f32) {
                const uint8_t gv = gray_ref((uint8_t)o[0], (uint8_t)o[1], (uint8_t)o[2]);
                if (gray) gray[off] = gv;
                if (gray_f32) gray_f32[off] = (float)gv;
            }
        }
    }
}

// The level-0 planes of one pair as a FUNCTION of the inputs -- exactly the values k_compose stores (planes 0..2 the
// warped frame, 3..5 the moved mosaic, 0 where the reference leaves its zeroed canvas untouched).  The consumers of
// level 0 (seam scan, causal x sweep, level-0 collapse) evaluate it in place when the plan runs "source-fused", so the
// six level-0 planes are never written to or read from HBM.  The double-precision map is evaluated once per canvas
// pixel by k_src_index, which leaves the byte offset of frame sample (nx, ny) within a channel plane (or "outside") in
// the slot of level-0 plane 0; the three channel sweeps and the collapse gather through that index.  use_src = 0: the planes are in memory (k_compose ran).
template <typename PX>
struct PairSrc {
    const PX* __restrict__ frame;
    const PX* __restrict__ mosaic;
    MapP map;
    int fw, fh, mw, mh, ox, oy;
    float offx, offy;
    size_t fpl, mpl;
    __device__ __forceinline__ PairSrc(const PairArgs<PX>& pa, int pr)
        : frame(pa.frame[pr]), mosaic(pa.mosaic[pr]), map(pa.map[pr]), fw(pa.fw[pr]), fh(pa.fh[pr]), mw(pa.mw[pr]), mh(pa.mh[pr]),
          ox(pa.ox[pr]), oy(pa.oy[pr]), offx(pa.offx[pr]), offy(pa.offy[pr]), fpl((size_t)pa.fw[pr] * pa.fh[pr]),
          mpl((size_t)pa.mw[pr] * pa.mh[pr]) {}
    __device__ __forceinline__ bool frame_at(int x, int y, size_t& so) const {
        int nx, ny;
        if (!map_to_src(map, (float)x + offx, (float)y + offy, fw, fh, nx, ny)) return false;
        so = (size_t)ny * fw + nx;
        return true;
    }
    __device__ __forceinline__ bool mosaic_at(int x, int y, size_t& so) const {
        const long long mx = (long long)x + ox, my = (long long)y + oy;
        if (mx < 0 || mx >= mw || my < 0 || my >= mh) return false;
        so = (size_t)my * mw + mx;
        return true;
    }
    __device__ __forceinline__ float frame_val(size_t so, int c) const { return warped_px<PX>((float)frame[so + c * fpl]); }
    __device__ __forceinline__ float mosaic_val(size_t so, int c) const { return (float)mosaic[so + c * mpl]; }
    // plane q (0..5) at canvas pixel (x, y), 0 <= x < cw
    __device__ __forceinline__ float plane(int q, int x, int y) const {
        size_t so;
        if (q < 3) return frame_at(x, y, so) ? frame_val(so, q) : 0.f;
        return mosaic_at(x, y, so) ? mosaic_val(so, q - 3) : 0.f;
    }
};

struct NoPairArgs {};  // levels >= 1 carry no pair description
template <typename OUT, bool DENSE>
struct CollapseSrc {
    typedef NoPairArgs type;
};
template <typename OUT>
struct CollapseSrc<OUT, true> {
    typedef PairArgs<OUT> type;
};
// What one block of the causal x sweep needs of the pair when level 0 is source-fused: a range-checked descriptor of
// its channel plane of the frame (q < 3, gathered through the index plane) or of the mosaic (a pure shift).
struct XShift {
    int mw, mh, ox, oy;
};
template <typename PX>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x_rsrc(const PairArgs<PX>& pa, int pr, int q) {
    const bool fr = q < 3;
    const PX* base = fr ? pa.frame[pr] : pa.mosaic[pr];
    const size_t elems = fr ? (size_t)pa.fw[pr] * pa.fh[pr] : (size_t)pa.mw[pr] * pa.mh[pr];
    return plane_rsrc(base + (size_t)(fr ? q : (q < 6 ? q - 3 : 0)) * elems, elems);
}
template <typename PX>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x_rsrc(const NoPairArgs&, int, int) {
    return plane_rsrc<PX>(nullptr, 0);
}
template <typename PX>
__device__ __forceinline__ XShift x_shift(const PairArgs<PX>& pa, int pr) {
    return XShift{pa.mw[pr], pa.mh[pr], pa.ox[pr], pa.oy[pr]};
}
template <typename PX>
__device__ __forceinline__ XShift x_shift(const NoPairArgs&, int) {
    return XShift{0, 0, 0, 0};
}
// mosaic byte offset of canvas (x, y) = row part + column part; either part 0xffffffff = outside (a valid part is
// below 0xfffffff0).  Branch-free.
template <typename PX>
__device__ __forceinline__ unsigned mosaic_col(const XShift& m, int x, int w) {
    const long long mx = (long long)x + m.ox;
    const bool ok = x < w && mx >= 0 && mx < m.mw;
    return ok ? (unsigned)mx * (unsigned)sizeof(PX) : 0xffffffffu;
}
template <typename PX>
__device__ __forceinline__ unsigned mosaic_row(const XShift& m, int y) {
    const long long my = (long long)y + m.oy;
    const bool ok = my >= 0 && my < m.mh;
    const unsigned r = (unsigned)my * (unsigned)m.mw * (unsigned)sizeof(PX);
    return ok ? r : 0xffffffffu;
}
template <typename PX>
__device__ __forceinline__ unsigned mosaic_offset(unsigned row, unsigned col) {
    // saturating add: an "outside" part (all ones) drags the sum to all ones; two valid parts never reach it (planes stay below
    // 0xfffffff0 bytes).  Masked down to the access size, all ones IS off_outside<PX>().  Two operations instead of five per
    // sample of the sweep's mosaic planes.
    return __builtin_elementwise_add_sat(row, col) & (0u - (unsigned)sizeof(PX));
}

// Sparse canvases.  A stitched canvas is mostly empty for either image (the reference blurs and decimates the zeros
// like everything else), and the recursive filters leave exact +0.0f wherever the input was zero and the state has
// died out (about 150 samples past the data; the tails of these images are non-negative).  For the levels the fused
// sweep covers (heights that are multiples of 64), a kernel that produces a 64x64 tile of the blur scratch T made of
// +0.0f only records one byte instead of storing the tile, and the next kernel takes the zeros from the flag instead
// of from HBM.  The arithmetic is unchanged (zeros are swept like any other sample); only stores and loads of
// zeros are skipped.  The test is on the bit pattern, so a -0.0f keeps its tile "non-zero".
struct ZeroTiles {
    uint8_t* flags;  // [planes][NC][NR] (bands of one tile column are contiguous), nullptr = feature off for this level
    int h, NR, NC;   // rows per plane, 64-row bands per plane, 64-column tiles per row
    __device__ __forceinline__ size_t index(long plane, int band, int tile) const { return ((size_t)plane * NC + tile) * NR + band; }
};

constexpr int SI_ROWS = 8;  // rows per workgroup of k_src_index (a one-row workgroup is launch-bound: 393k workgroups per batch)
template <typename PX>
__global__ __launch_bounds__(256) void k_src_index(PairArgs<PX> pa, float* __restrict__ g0_all, int cw, int ch, int pitch, size_t ps,
                                                   ZeroTiles zi) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, pr = blockIdx.z;
    if (x >= pitch) return;
    const MapP m = pa.map[pr];
    const float offx = pa.offx[pr], offy = pa.offy[pr];
    const int fw = pa.fw[pr], fh = pa.fh[pr];
    unsigned* __restrict__ idx = reinterpret_cast<unsigned*>(g0_all + (size_t)pr * 7 * ps);
    const int y0 = blockIdx.y * SI_ROWS, y1 = min(y0 + SI_ROWS, ch);
    for (int y = y0; y < y1; ++y) {
        unsigned v = off_outside<PX>();
        int nx, ny;
        if (x < cw && map_to_src(m, (float)x + offx, (float)y + offy, fw, fh, nx, ny))
            v = ((unsigned)ny * (unsigned)fw + (unsigned)nx) * (unsigned)sizeof(PX);  // the host checked that a plane fits 32 bits
        // a wavefront is 64 consecutive pixels of one row = one row of one 64x64 tile (the pitch is a multiple of 64): the
        // tile's "every pixel outside the frame" flag, preset to 1, is cleared by any row that holds a sample (same byte,
        // same value)
        if (zi.flags && __ballot(v != off_outside<PX>()) != 0 && (threadIdx.x & 63) == 0) zi.flags[zi.index(pr, y >> 6, x >> 6)] = 0;
        idx[(size_t)y * pitch + x] = v;
    }
}

// row0: canvas row of the planes' row 0 (0 for a whole canvas; the first row of a rank's band when one pair is split over GPUs)
template <typename PX>
__global__ __launch_bounds__(256) void k_compose(PairArgs<PX> pa, float* __restrict__ g0_all, int cw, int ch, int pitch,
                                                 size_t ps, int row0) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, yl = blockIdx.y, y = yl + row0, pr = blockIdx.z;
    if (x >= pitch) return;
    float* g0 = g0_all + (size_t)pr * 7 * ps;
    float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f};
    if (x < cw) {
        const PX* __restrict__ frame = pa.frame[pr];
        const PX* __restrict__ mosaic = pa.mosaic[pr];
        const int fw = pa.fw[pr], fh = pa.fh[pr], mw = pa.mw[pr], mh = pa.mh[pr];
        int nx, ny;
        if (map_to_src(pa.map[pr], (float)x + pa.offx[pr], (float)y + pa.offy[pr], fw, fh, nx, ny)) {
            const size_t spl = (size_t)fw * fh, so = (size_t)ny * fw + nx;
#pragma unroll
            for (int c = 0; c < 3; ++c) a[c] = warped_px<PX>((float)frame[so + c * spl]);
        }
        const long long mx = (long long)x + pa.ox[pr], my = (long long)y + pa.oy[pr];
        if (mx >= 0 && mx < mw && my >= 0 && my < mh) {
            const size_t spl = (size_t)mw * mh, so = (size_t)my * mw + mx;
#pragma unroll
            for (int c = 0; c < 3; ++c) b[c] = (float)mosaic[so + c * spl];
        }
    }
    const size_t o = (size_t)yl * pitch + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        g0[o + c * ps] = a[c];
        g0[o + (3 + c) * ps] = b[c];
    }
}

// dense canvases a, b (already warped / moved by the caller) -> level-0 planes (stitch_blend_*)
template <typename PX>
__global__ __launch_bounds__(256) void k_load_canvases(const PX* __restrict__ a, const PX* __restrict__ b,
                                                       float* __restrict__ g0, int cw, int ch, int pitch, size_t ps) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= pitch) return;
    const size_t cpl = (size_t)cw * ch, co = (size_t)y * cw + x, o = (size_t)y * pitch + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        g0[o + c * ps] = x < cw ? (float)a[co + c * cpl] : 0.f;
        g0[o + (3 + c) * ps] = x < cw ? (float)b[co + c * cpl] : 0.f;
    }
}

// ---- B1: seam scan, ImageProcess.cpp:659-671,686-698 --------------------------------------------------------
// One workgroup walks the middle row of the level-0 planes; integer sums are reduced with wavefront shuffles
// and one LDS exchange.  Thread 0 derives ratio / ov / branch / start exactly as the reference does.
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

template <typename PX>
__global__ __launch_bounds__(1024) void k_seam(const float* __restrict__ g0_all, int cw, int ch, int pitch, size_t ps,
                                               int seam_rule, SeamDev* __restrict__ out_all, PairArgs<PX> pa, int use_src) {
    __shared__ int red[4][16];
    const int mid = ch / 2;
    const float* g0 = g0_all + (size_t)blockIdx.x * 7 * ps;  // one workgroup per pair
    SeamDev* out = out_all + blockIdx.x;
    const float* a0 = g0 + (size_t)mid * pitch;
    const float* b0 = a0 + 3 * ps;
    const PairSrc<PX> src(pa, blockIdx.x);
    int s_a = 0, n_a = 0, s_o = 0, n_o = 0;
    for (int x = threadIdx.x; x < cw; x += blockDim.x) {
        bool a_on, b_on;
        if (use_src) {  // source-fused plan: the middle row straight from the inputs
            size_t fo = 0, mo = 0;
            const bool fin = src.frame_at(x, mid, fo), min_ = src.mosaic_at(x, mid, mo);
            a_on = fin && src.frame_val(fo, 0) != 0.f;
            b_on = min_ && src.mosaic_val(mo, 0) != 0.f;
            if (seam_rule) {
                a_on = a_on && src.frame_val(fo, 1) != 0.f && src.frame_val(fo, 2) != 0.f;
                b_on = b_on && src.mosaic_val(mo, 1) != 0.f && src.mosaic_val(mo, 2) != 0.f;
            }
        } else {
            a_on = a0[x] != 0.f;
            b_on = b0[x] != 0.f;
            if (seam_rule) {
                a_on = a_on && a0[x + ps] != 0.f && a0[x + 2 * ps] != 0.f;
                b_on = b_on && b0[x + ps] != 0.f && b0[x + 2 * ps] != 0.f;
            }
        }
        if (a_on) {
            s_a += x;
            ++n_a;
            if (b_on) {
                s_o += x;
                ++n_o;
            }
        }
    }
    s_a = wave_sum(s_a);
    n_a = wave_sum(n_a);
    s_o = wave_sum(s_o);
    n_o = wave_sum(n_o);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[0][wid] = s_a;
        red[1][wid] = n_a;
        red[2][wid] = s_o;
        red[3][wid] = n_o;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        s_a = n_a = s_o = n_o = 0;
        for (int i = 0; i < nw; ++i) {
            s_a += red[0][i];
            n_a += red[1][i];
            s_o += red[2][i];
            n_o += red[3][i];
        }
        SeamDev s;
        s.sum_a_x = s_a;
        s.n_a = n_a;
        s.sum_ov_x = s_o;
        s.n_ov = n_o;
        s.ratio = s.ov = 0.f;
        s.branch = 1;
        s.start = cw;  // neutral mask (all zero) when the scan fails
        s.thr = 0.0;
        s.status = 0;
        s.pad = 0;
        if (n_a == 0)
            s.status = -2;
        else if (n_o == 0)
            s.status = -3;
        else if (seam_rule == 0) {
            const float ratio = (float)(1.0 * (double)s_a / (double)n_a);
            const float ov = (float)(1.0 * (double)s_o / (double)n_o);
            s.ratio = ratio;
            s.ov = ov;
            s.branch = (ratio < ov) ? 0 : 1;
            s.start = (int)(ov + 1.f);
            s.thr = (double)ov;
        } else {
            const double ratio = (double)s_a / (double)n_a, ov = (double)s_o / (double)n_o;
            s.ratio = (float)ratio;
            s.ov = (float)ov;
            s.branch = (ratio < ov) ? 0 : 1;
            s.start = (int)(ov + 1.0);
            s.thr = ov;
        }
        *out = s;
    }
}

// mask level 0: a vertical step (ImageProcess.cpp:682,690-698), plane 6 of level 0
__global__ __launch_bounds__(256) void k_mask(float* __restrict__ g0_all, int cw, int pitch, size_t ps,
                                              const SeamDev* __restrict__ seam_all) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= pitch) return;
    float* m0 = g0_all + ((size_t)blockIdx.z * 7 + 6) * ps;
    const SeamDev* seam = seam_all + blockIdx.z;
    float v = 0.f;
    if (x < cw) v = seam->branch == 0 ? ((double)x < seam->thr ? 1.f : 0.f) : (x >= seam->start ? 1.f : 0.f);
    m0[(size_t)y * pitch + x] = v;
}

// ---- B3: Van Vliet recursive Gaussian, CImg.h:34887-34932 ---------------------------------------------------
// The recurrence is strictly sequential along a line (double accumulators that keep their unrounded value,
// every output stored to float), so the parallel axis is "lines": one line per work-item.
//
// x pass: a wavefront owns 64 consecutive rows.  Rows are moved between HBM and LDS as 64x64 tiles with
// 16-byte-per-lane accesses (a lane loads 4 consecutive samples of one row, 16 lanes cover a 256-byte row
// segment); in LDS a row is padded to 68 floats so that both the row-wise tile traffic and the per-lane
// ds_read_b128 of "my row" are bank-conflict free.  The next tile is prefetched into registers while the
// current one runs its 64 recurrence steps.  The forward kernel leaves (v1,v2,v3,iplus) of every line in
// `state`; the backward kernel starts from the Triggs boundary values computed from them.
constexpr int TS = 64;  // tile edge
constexpr int TP = 68;  // padded LDS row (floats)

__device__ __forceinline__ void tile_load(const float* __restrict__ base, int pitch, int c0, int lane, f4 pre[16]) {
    // lane -> (row group, 4-column group): rows lane/16 + 4*i, columns 4*(lane%16)..+3
    const float* p = base + (size_t)(lane >> 4) * pitch + c0 + ((lane & 15) << 2);
#pragma unroll
    for (int i = 0; i < 16; ++i) pre[i] = *reinterpret_cast<const f4*>(p + (size_t)(4 * i) * pitch);
}
__device__ __forceinline__ void tile_to_lds(float* tile, int lane, const f4 pre[16]) {
    float* t = tile + (lane >> 4) * TP + ((lane & 15) << 2);
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<f4*>(t + (4 * i) * TP) = pre[i];
}
__device__ __forceinline__ void tile_store(float* __restrict__ base, int pitch, int c0, int lane, const float* tile) {
    float* p = base + (size_t)(lane >> 4) * pitch + c0 + ((lane & 15) << 2);
    const float* t = tile + (lane >> 4) * TP + ((lane & 15) << 2);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        __builtin_nontemporal_store(*reinterpret_cast<const f4*>(t + (4 * i) * TP), reinterpret_cast<f4*>(p + (size_t)(4 * i) * pitch));
}

// Level-0 mask without a level-0 mask plane.  The reference's mask[0] is a vertical step (ImageProcess.cpp:690-698):
// every row is the same function of x, so (when the level height is a multiple of 64, i.e. a 64-row block never
// straddles planes) the x sweeps generate the step on the fly, compute only the first 64 of its identical rows,
// leave the x-blurred row in a small side buffer, and the causal y sweep reads that one row for every y.  The
// collapse kernel evaluates the step directly.  Saves one plane write and six plane reads/writes of level 0.
struct MaskL0 {
    const SeamDev* seam;  // per pair
    float* side;          // [pairs][pitch]: the x-blurred mask row
    int h;                // level height = lines per plane
    int enabled;
};
__device__ __forceinline__ float mask_step(const SeamDev& sd, int x) {
    return sd.branch == 0 ? ((double)x < sd.thr ? 1.f : 0.f) : (x >= sd.start ? 1.f : 0.f);
}

__device__ __forceinline__ void tile_store_rows(float* __restrict__ base, int pitch, int c0, int lane, const float* tile, int nrows) {
    float* p = base + (size_t)(lane >> 4) * pitch + c0 + ((lane & 15) << 2);
    const float* t = tile + (lane >> 4) * TP + ((lane & 15) << 2);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((lane >> 4) + 4 * i < nrows)
            __builtin_nontemporal_store(*reinterpret_cast<const f4*>(t + (4 * i) * TP), reinterpret_cast<f4*>(p + (size_t)(4 * i) * pitch));
}

// Source-fused level 0: plane byte offsets of one 64x64 tile in tile_load's element-to-lane layout (or "outside"),
// and the gather through them.  k_compose's two stores: the warped sample goes through the degenerate bilinear call,
// the moved one is a copy.
template <typename PX>
__device__ __forceinline__ void mosaic_indices(const XShift& ms, int c0, int lane, int y0, int w, f4 nidx[16]) {
    const int c = c0 + ((lane & 15) << 2);
    const unsigned c0i = mosaic_col<PX>(ms, c, w), c1i = mosaic_col<PX>(ms, c + 1, w), c2i = mosaic_col<PX>(ms, c + 2, w),
                   c3i = mosaic_col<PX>(ms, c + 3, w);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned row = mosaic_row<PX>(ms, y0 + (lane >> 4) + 4 * i);
        f4 v;
        v.x = __uint_as_float(mosaic_offset<PX>(row, c0i));
        v.y = __uint_as_float(mosaic_offset<PX>(row, c1i));
        v.z = __uint_as_float(mosaic_offset<PX>(row, c2i));
        v.w = __uint_as_float(mosaic_offset<PX>(row, c3i));
        nidx[i] = v;
    }
}
template <typename PX>
__device__ __forceinline__ float src_px(__amdgpu_buffer_rsrc_t rs, bool warped, unsigned byte_off) {
    const float v = buf_px<PX>(rs, byte_off);
    return warped ? warped_px<PX>(v) : v;
}
template <typename PX>
__device__ __forceinline__ void src_gather(__amdgpu_buffer_rsrc_t rs, const f4 nidx[16], f4 pre[16]) {  // raw bits
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f4 id = nidx[i];
        f4 v;
        v.x = buf_raw<PX>(rs, __float_as_uint(id.x));
        v.y = buf_raw<PX>(rs, __float_as_uint(id.y));
        v.z = buf_raw<PX>(rs, __float_as_uint(id.z));
        v.w = buf_raw<PX>(rs, __float_as_uint(id.w));
        pre[i] = v;
    }
}
template <typename PX>
__device__ __forceinline__ void src_finish(bool warped, f4 pre[16]) {  // raw bits -> the value k_compose would have stored
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        f4 v = pre[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = raw_to_px<PX>(v[j]);
            v[j] = warped ? warped_px<PX>(t) : t;
        }
        pre[i] = v;
    }
}

// `lines` = rows of all planes stacked (plane stride = pitch*h, so line L starts at L*pitch); the buffers are
// allocated with 64 spare rows so that a partial last block may touch rows >= lines without leaving them.
// Level 0 of a source-fused plan (use_src; needs mk.enabled, i.e. a level height that is a multiple of 64): the input
// tile is not read from `in` but evaluated from the pair's frames (PairSrc), so S1 never materialises.
// CKPT (a level whose anticausal sweep re-runs the causal one tile by tile, k_vv_xbyf MODE 1/2): the swept samples are not
// written at all; what is kept is the recurrence state in front of every tile (ckpt[3][tiles][lines], doubles), from which
// the consumer reproduces the tile's samples exactly -- 24 bytes per 64 samples instead of 256 written and read back.
template <typename PX, bool SRC, bool CKPT = false>
__global__ __launch_bounds__(64) void k_vv_x_fwd(const float* __restrict__ in, float* __restrict__ out, int w, int pitch,
                                                  long lines, VVK k, double* __restrict__ state, MaskL0 mk,
                                                  typename CollapseSrc<PX, SRC>::type pa, ZeroTiles zt, ZeroTiles zi,
                                                  double* __restrict__ ckpt) {
    __shared__ __attribute__((aligned(16))) float tile[TS * TP];
    const int lane = threadIdx.x;
    long blk = blockIdx.x;
    if (SRC && mk.enabled && (mk.h / TS) % 8 == 0) {
        // Source-fused level 0: the three channel sweeps of a band read the same index tiles.  Workgroups go round-robin
        // over the 8 XCDs (each with its own L2), so the planes of a pair are interleaved in groups of 8 bands: the seven
        // blocks of one band are then 8 apart -- same XCD, dispatched together -- so that a re-read of an index tile can
        // be served on the chip.  (Plain order: plane-major, the channels of a band 64 blocks apart in time.)  Measured:
        // the kernel is 1.5 % faster, but FETCH_SIZE (traffic leaving the L2) is unchanged -- at 8 workgroups per CU a
        // 4 MB L2 does not keep a tile from one channel sweep to the next; what helps is the memory-side cache.
        const long nr = mk.h / TS, per_pair = 7 * nr, pr_ = blk / per_pair, r = blk % per_pair;
        const long band = (r / 56) * 8 + r % 8, q = (r / 8) % 7;
        blk = pr_ * per_pair + q * nr + band;
    }
    const long line0 = blk * TS, line = line0 + lane;
    const float* ib = in + (size_t)line0 * pitch;
    float* ob = out + (size_t)line0 * pitch;
    const int ntiles = (w + TS - 1) / TS;
    const bool live = line < lines;
    bool gen_mask = false;  // wave-uniform: this block's 64 lines are rows of a level-0 mask plane
    SeamDev sd;
    int src_q = 0, src_y0 = 0, src_pr = 0;  // wave-uniform: plane, first row and pair of this block (use_src)
    if (mk.enabled) {
        const long plane = line0 / mk.h;
        if (plane % 7 == 6) {
            if (line0 % mk.h >= TS) return;  // rows 64.. of the step are copies of rows 0..63 and are never read
            gen_mask = true;
            sd = mk.seam[plane / 7];
        }
        src_q = (int)(plane % 7);
        src_pr = (int)(plane / 7);
        src_y0 = (int)(line0 % mk.h);
    }
    const bool gen_frame = SRC && !gen_mask && src_q < 3, gen_mosaic = SRC && !gen_mask && src_q >= 3;
    const __amdgpu_buffer_rsrc_t rs = x_rsrc<PX>(pa, src_pr, src_q);
    const XShift ms = x_shift<PX>(pa, src_pr);
    // rows src_y0.. of the pair's index plane (the slot of level-0 plane 0)
    const float* idx_rows = in + ((size_t)src_pr * 7 * mk.h + src_y0) * pitch;
    // Element indices of the next tile, produced one tile ahead of the gather that consumes them: loaded from the index
    // plane (frame channels) or computed (the mosaic is a pure shift).
    f4 nidx[SRC ? 16 : 1];
    // index tile t of this block's band: from the index plane, or all "outside" when k_src_index left the tile's flag set
    // frame channels: tile t of this band lies outside the frame altogether (k_src_index's flag): its samples are +0
    auto tile_outside = [&](int t) -> bool {
        if constexpr (SRC) return gen_frame && zi.flags && zi.flags[zi.index(src_pr, src_y0 >> 6, t)];
        return false;
    };
    auto index_tile = [&](int t) {
        if constexpr (SRC)
            if (!tile_outside(t)) tile_load(idx_rows, pitch, t * TS, lane, nidx);  // an outside tile is not gathered at all
    };
    auto gen_tile = [&](int c0, f4 pre[16]) {
        f4 v;
        const int c = c0 + ((lane & 15) << 2);
        v.x = mask_step(sd, c);
        v.y = mask_step(sd, c + 1);
        v.z = mask_step(sd, c + 2);
        v.w = mask_step(sd, c + 3);
#pragma unroll
        for (int i = 0; i < 16; ++i) pre[i] = v;
    };
    double iplus = 0.0;  // CImg.h:34906
    if (live)
        iplus = gen_mask     ? (double)mask_step(sd, w - 1)
                : gen_frame  ? (double)src_px<PX>(rs, true, reinterpret_cast<const unsigned*>(idx_rows)[(size_t)lane * pitch + (w - 1)])
                : gen_mosaic ? (double)src_px<PX>(rs, false, mosaic_offset<PX>(mosaic_row<PX>(ms, src_y0 + lane), mosaic_col<PX>(ms, w - 1, w)))
                             : (double)in[(size_t)line * pitch + (w - 1)];
    double v1 = 0, v2 = 0, v3 = 0;
    f4 pre[16];
// tile T into `pre`; a source-fused block gathers through the indices produced during the previous fetch, then produces the next
#define STITCH_X_FETCH(T)                                                      \
    do {                                                                       \
        if (gen_mask)                                                          \
            gen_tile((T) * TS, pre);                                           \
        else if constexpr (SRC) {                                              \
            if (tile_outside(T)) {                                             \
                _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) pre[i_] = f4{0.f, 0.f, 0.f, 0.f}; \
            } else                                                             \
                src_gather<PX>(rs, nidx, pre);                                 \
            if ((T) + 1 < ntiles) {                                            \
                if (gen_frame)                                                 \
                    index_tile((T) + 1);                                       \
                else                                                           \
                    mosaic_indices<PX>(ms, ((T) + 1) * TS, lane, src_y0, w, nidx); \
            }                                                                  \
        } else                                                                 \
            tile_load(ib, pitch, (T) * TS, lane, pre);                         \
    } while (0)
    if constexpr (SRC) {
        if (gen_frame)
            index_tile(0);
        else if (gen_mosaic)
            mosaic_indices<PX>(ms, 0, lane, src_y0, w, nidx);
    }
    STITCH_X_FETCH(0);
    for (int t = 0; t < ntiles; ++t) {
        // Zeros in under a zero state: every product and sum of the recurrence is +0 again, the state stays as it is and the
        // tile would be recorded as all +0 -- recorded at once.  (In a stitch step the frame's canvas is empty left of the
        // frame: a third of the frame channels' tiles in config 2.)
        if (!CKPT && zt.flags && tile_outside(t) &&
            (t == 0 || __ballot((__double_as_longlong(v1) | __double_as_longlong(v2) | __double_as_longlong(v3)) != 0) == 0)) {
            if (t + 1 < ntiles) STITCH_X_FETCH(t + 1);
            if (lane == 0) zt.flags[zt.index(line0 / zt.h, (int)((line0 % zt.h) / TS), t)] = 1;
            continue;
        }
        if constexpr (SRC)
            if (!gen_mask) src_finish<PX>(gen_frame, pre);
        tile_to_lds(tile, lane, pre);
        if (t + 1 < ntiles) STITCH_X_FETCH(t + 1);
        __syncthreads();  // one wave per workgroup: orders the tile writes before the per-lane row reads
        float* row = tile + lane * TP;
        const int jmax = min(TS, w - t * TS);
        if (t == 0) v1 = v2 = v3 = (double)row[0] / k.sumsq;  // CImg.h:34909
        if (CKPT && t > 0 && live) {
            double* c = ckpt + (size_t)(3 * t) * lines + line;
            c[0] = v1;
            c[lines] = v2;
            c[2 * lines] = v3;
        }
        const int jfull = jmax & ~15;
        unsigned nz = 0;  // OR of the bit patterns this lane stores: 0 <=> every sample is +0.0f
        for (int jb = 0; jb < jfull; jb += 16) {
            float xs[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(xs + 4 * q) = *reinterpret_cast<const f4*>(row + jb + 4 * q);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                double v0 = (double)xs[u];
                v0 += v1 * k.f1;
                v0 += v2 * k.f2;
                v0 += v3 * k.f3;
                xs[u] = (float)v0;
                nz |= __float_as_uint(xs[u]);
                v3 = v2;
                v2 = v1;
                v1 = v0;
            }
            if (!CKPT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(row + jb + 4 * q) = *reinterpret_cast<const f4*>(xs + 4 * q);
            }
        }
        for (int j = jfull; j < jmax; ++j) {  // ragged tail of the last tile, straight from LDS
            double v0 = (double)row[j];
            v0 += v1 * k.f1;
            v0 += v2 * k.f2;
            v0 += v3 * k.f3;
            const float f = (float)v0;
            if (!CKPT) row[j] = f;
            nz |= __float_as_uint(f);
            v3 = v2;
            v2 = v1;
            v1 = v0;
        }
        __syncthreads();
        if (zt.flags) {  // an all-(+0) tile is recorded instead of written: its consumers never read it (ZeroTiles)
            const bool zero = __ballot(nz != 0) == 0;
            if (lane == 0) zt.flags[zt.index(line0 / zt.h, (int)((line0 % zt.h) / TS), t)] = zero ? 1 : 0;
            if (zero) continue;
        }
        if (!CKPT) tile_store(ob, pitch, t * TS, lane, tile);
    }
    if (live) {
        state[line] = v1;
        state[lines + line] = v2;
        state[2 * lines + line] = v3;
        state[3 * lines + line] = iplus;
    }
}

#undef STITCH_X_FETCH

__device__ __forceinline__ void triggs(const VVK& k, double iplus, double& v1, double& v2, double& v3, float& first) {
    // CImg.h:34911-34922
    const double uplus = iplus / k.den, vplus = uplus / k.den, unp = v1 - uplus, unp1 = v2 - uplus, unp2 = v3 - uplus;
    const double n0 = (k.M[0] * unp + k.M[1] * unp1 + k.M[2] * unp2 + vplus) * k.sum;
    const double n1 = (k.M[3] * unp + k.M[4] * unp1 + k.M[5] * unp2 + vplus) * k.sum;
    const double n2 = (k.M[6] * unp + k.M[7] * unp1 + k.M[8] * unp2 + vplus) * k.sum;
    first = (float)n0;
    v3 = n2;
    v2 = n1;
    v1 = n0;
}

__global__ __launch_bounds__(64) void k_vv_x_bwd(float* __restrict__ data, int w, int pitch, long lines, VVK k,
                                                  const double* __restrict__ state, MaskL0 mk) {
    __shared__ __attribute__((aligned(16))) float tile[TS * TP];
    const int lane = threadIdx.x;
    const long line0 = (long)blockIdx.x * TS, line = line0 + lane;
    float* side_row = nullptr;  // level-0 mask plane: row 0 of this block is also left in the side buffer
    if (mk.enabled) {
        const long plane = line0 / mk.h;
        if (plane % 7 == 6) {
            if (line0 % mk.h >= TS) return;
            side_row = mk.side + (size_t)(plane / 7) * pitch;
        }
    }
    float* base = data + (size_t)line0 * pitch;
    const int ntiles = (w + TS - 1) / TS;
    const bool live = line < lines;
    double v1 = 0, v2 = 0, v3 = 0, iplus = 0;
    if (live) {
        v1 = state[line];
        v2 = state[lines + line];
        v3 = state[2 * lines + line];
        iplus = state[3 * lines + line];
    }
    float first;
    triggs(k, iplus, v1, v2, v3, first);
    f4 pre[16];
    tile_load(base, pitch, (ntiles - 1) * TS, lane, pre);
    for (int t = ntiles - 1; t >= 0; --t) {
        tile_to_lds(tile, lane, pre);
        if (t > 0) tile_load(base, pitch, (t - 1) * TS, lane, pre);
        __syncthreads();
        float* row = tile + lane * TP;
        int jtop = min(TS, w - t * TS);  // samples [0,jtop) of this tile, processed from jtop-1 down to 0
        if (t == ntiles - 1) {
            row[jtop - 1] = first;  // sample N-1 takes the boundary value (CImg.h:34920)
            --jtop;
        }
        // whole 16-sample groups below jtop, then the ragged top group first (descending order overall)
        const int jfull = jtop & ~15;
        if (jtop > jfull) {
            for (int j = jtop - 1; j >= jfull; --j) {
                double v0 = (double)row[j];
                v0 *= k.sum;
                v0 += v1 * k.f1;
                v0 += v2 * k.f2;
                v0 += v3 * k.f3;
                row[j] = (float)v0;
                v3 = v2;
                v2 = v1;
                v1 = v0;
            }
        }
        for (int jb = jfull - 16; jb >= 0; jb -= 16) {
            float xs[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(xs + 4 * q) = *reinterpret_cast<const f4*>(row + jb + 4 * q);
#pragma unroll
            for (int u = 15; u >= 0; --u) {
                double v0 = (double)xs[u];
                v0 *= k.sum;
                v0 += v1 * k.f1;
                v0 += v2 * k.f2;
                v0 += v3 * k.f3;
                xs[u] = (float)v0;
                v3 = v2;
                v2 = v1;
                v1 = v0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f4*>(row + jb + 4 * q) = *reinterpret_cast<const f4*>(xs + 4 * q);
        }
        __syncthreads();
        tile_store(base, pitch, t * TS, lane, tile);
        if (side_row && lane < 16) *reinterpret_cast<f4*>(side_row + t * TS + (lane << 2)) = *reinterpret_cast<const f4*>(tile + (lane << 2));
    }
}

// overlap weights of the moving-average resize (used by the fused anticausal pass and by k_decimate)
struct Taps {  // up to 4 overlaps per output sample (3 when n_src = 2*n_dst+1, 2 when n_src = 2*n_dst)
    int s0, n;
    float d[4];
};
__device__ __forceinline__ Taps make_taps(int t, int n_src, int n_dst) {
    Taps r;
    const long long pos = (long long)t * n_src;
    int s = (int)(pos / n_dst);
    int c_left = (int)((long long)(s + 1) * n_dst - pos);
    int remaining = n_src;
    r.s0 = s;
    r.n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = remaining < c_left ? remaining : c_left;
        r.d[i] = (float)(unsigned)d;
        if (d > 0) r.n = i + 1;
        remaining -= d;
        c_left = n_dst;
    }
    return r;
}

// y pass: two adjacent columns per work-item (8-byte accesses: a wavefront moves 512 contiguous bytes of one row
// per step, and the two independent recurrences interleave in the fp64 pipe).  The recurrence is cheap next to an
// HBM round trip, so rows are fetched far ahead of their use: YST stages of YCH rows rotate through registers
// (YCH*(YST-1) = 24 rows in flight per wavefront while one chunk computes).  In place.
// grid = (ceil(pitch/128), planes); pitch is a multiple of 64, a last half-empty block is masked off.
constexpr int YCH = 8, YST = 4, YCOLS = 2 * WAVE;

// Calls f(y, value) for rows y_begin, y_begin +/- 1, ... (count rows), loading each row's value long before.
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool DOWN, typename V = f2, typename F>
__device__ __forceinline__ void stream_rows(const float* __restrict__ p, int pitch, int y_begin, int count, F&& f) {
    V buf[YST][YCH];
    // row index is clamped instead of predicated: the tail re-reads the last row, the loop body stays branch-free
    auto ld = [&](int i) {
        const int ic = i < count ? i : count - 1;
        return *reinterpret_cast<const V*>(p + (size_t)(DOWN ? y_begin + ic : y_begin - ic) * pitch);
    };
#pragma unroll
    for (int s = 0; s < YST - 1; ++s)
#pragma unroll
        for (int u = 0; u < YCH; ++u) buf[s][u] = ld(s * YCH + u);
    int c0 = 0;
    for (; c0 + YST * YCH <= count; c0 += YST * YCH) {  // whole groups: no predicates at all
#pragma unroll
        for (int s = 0; s < YST; ++s) {
#pragma unroll
            for (int u = 0; u < YCH; ++u) buf[(s + YST - 1) % YST][u] = ld(c0 + (s + YST - 1) * YCH + u);
#pragma unroll
            for (int u = 0; u < YCH; ++u) {
                const int i = c0 + s * YCH + u;
                f(DOWN ? y_begin + i : y_begin - i, buf[s][u]);
            }
        }
    }
    if (c0 < count) {  // ragged last group
#pragma unroll
        for (int s = 0; s < YST; ++s) {
#pragma unroll
            for (int u = 0; u < YCH; ++u) buf[(s + YST - 1) % YST][u] = ld(c0 + (s + YST - 1) * YCH + u);
#pragma unroll
            for (int u = 0; u < YCH; ++u) {
                const int i = c0 + s * YCH + u;
                if (i < count) f(DOWN ? y_begin + i : y_begin - i, buf[s][u]);
            }
        }
    }
}

// recurrence state of the two columns a work-item owns
struct Y2 {
    double a1, a2, a3, b1, b2, b3;
};
__device__ __forceinline__ f2 y2_step_fwd(Y2& s, const VVK& k, f2 x) {
    double a0 = (double)x.x, b0 = (double)x.y;
    a0 += s.a1 * k.f1;
    b0 += s.b1 * k.f1;
    a0 += s.a2 * k.f2;
    b0 += s.b2 * k.f2;
    a0 += s.a3 * k.f3;
    b0 += s.b3 * k.f3;
    s.a3 = s.a2;
    s.a2 = s.a1;
    s.a1 = a0;
    s.b3 = s.b2;
    s.b2 = s.b1;
    s.b1 = b0;
    f2 r;
    r.x = (float)a0;
    r.y = (float)b0;
    return r;
}
__device__ __forceinline__ f2 y2_step_bwd(Y2& s, const VVK& k, f2 x) {
    double a0 = (double)x.x, b0 = (double)x.y;
    a0 *= k.sum;
    b0 *= k.sum;
    a0 += s.a1 * k.f1;
    b0 += s.b1 * k.f1;
    a0 += s.a2 * k.f2;
    b0 += s.b2 * k.f2;
    a0 += s.a3 * k.f3;
    b0 += s.b3 * k.f3;
    s.a3 = s.a2;
    s.a2 = s.a1;
    s.a1 = a0;
    s.b3 = s.b2;
    s.b2 = s.b1;
    s.b1 = b0;
    f2 r;
    r.x = (float)a0;
    r.y = (float)b0;
    return r;
}
// state buffer: [4][planes][pitch] doubles -- v1, v2, v3, iplus of every column
__device__ __forceinline__ size_t ystate_index(int plane, int pitch, int x) { return (size_t)plane * pitch + x; }

// resume (one pair split into row bands over several GPUs): the rows above this band belong to another rank, which left its
// recurrence state (v1, v2, v3 per column, [3][planes][pitch]) behind: the sweep continues from it instead of starting at
// the image's first row.  The state this kernel leaves is exactly what the rank below resumes from.
__global__ __launch_bounds__(64) void k_vv_y_fwd(float* __restrict__ data, int h, int pitch, size_t ps, VVK k,
                                                  double* __restrict__ state, MaskL0 mk, const double* __restrict__ resume) {
    const int x = (blockIdx.x * WAVE + threadIdx.x) * 2;
    if (x >= pitch) return;
    float* p = data + blockIdx.y * ps + x;
    // level-0 mask plane: every input row is the one x-blurred row kept in the side buffer (row stride 0)
    const bool side = mk.enabled && (blockIdx.y % 7 == 6);
    const float* src = side ? mk.side + (size_t)(blockIdx.y / 7) * pitch + x : p;
    const int spitch = side ? 0 : pitch;
    const f2 last = *reinterpret_cast<const f2*>(src + (size_t)(h - 1) * spitch);  // iplus, read before the sweep
    const f2 x0 = *reinterpret_cast<const f2*>(src);
    Y2 s;
    s.a1 = s.a2 = s.a3 = (double)x0.x / k.sumsq;
    s.b1 = s.b2 = s.b3 = (double)x0.y / k.sumsq;
    if (resume) {
        const size_t n = (size_t)gridDim.y * pitch, i = ystate_index(blockIdx.y, pitch, x);
        s.a1 = resume[i], s.b1 = resume[i + 1];
        s.a2 = resume[n + i], s.b2 = resume[n + i + 1];
        s.a3 = resume[2 * n + i], s.b3 = resume[2 * n + i + 1];
    }
    stream_rows<true>(src, spitch, 0, h, [&](int y, f2 xv) {
        *reinterpret_cast<f2*>(p + (size_t)y * pitch) = y2_step_fwd(s, k, xv);
    });
    const size_t n = (size_t)gridDim.y * pitch, i = ystate_index(blockIdx.y, pitch, x);
    state[i] = s.a1;
    state[i + 1] = s.b1;
    state[n + i] = s.a2;
    state[n + i + 1] = s.b2;
    state[2 * n + i] = s.a3;
    state[2 * n + i + 1] = s.b3;
    state[3 * n + i] = (double)last.x;
    state[3 * n + i + 1] = (double)last.y;
}

// The same sweep with ONE column per work-item, for launches that leave SIMDs idle (a single pair: 7 planes x 6144 columns are
// 336 wavefronts of two columns on 1024 SIMDs).  A double-precision operation occupies its SIMD for 8 cycles per wavefront and
// hands its result on after about 24: two interleaved chains are bound by issue (16 operations = 128 cycles per row), one
// chain by latency (4 dependent operations = 96 cycles) -- and there are twice as many wavefronts to spread.
__global__ __launch_bounds__(64) void k_vv_y_fwd1(float* __restrict__ data, int h, int pitch, size_t ps, VVK k,
                                                   double* __restrict__ state, MaskL0 mk, const double* __restrict__ resume) {
    const int x = blockIdx.x * WAVE + threadIdx.x;  // pitch is a multiple of 64
    float* p = data + blockIdx.y * ps + x;
    const bool side = mk.enabled && (blockIdx.y % 7 == 6);
    const float* src = side ? mk.side + (size_t)(blockIdx.y / 7) * pitch + x : p;
    const int spitch = side ? 0 : pitch;
    const float last = src[(size_t)(h - 1) * spitch];
    const float x0 = src[0];
    double v1, v2, v3;
    v1 = v2 = v3 = (double)x0 / k.sumsq;
    const size_t n = (size_t)gridDim.y * pitch, i = ystate_index(blockIdx.y, pitch, x);
    if (resume) {
        v1 = resume[i];
        v2 = resume[n + i];
        v3 = resume[2 * n + i];
    }
    stream_rows<true, float>(src, spitch, 0, h, [&](int y, float xv) {
        double v0 = (double)xv;
        v0 += v1 * k.f1;
        v0 += v2 * k.f2;
        v0 += v3 * k.f3;
        v3 = v2;
        v2 = v1;
        v1 = v0;
        p[(size_t)y * pitch] = (float)v0;
    });
    state[i] = v1;
    state[n + i] = v2;
    state[2 * n + i] = v3;
    state[3 * n + i] = (double)last;
}

__device__ __forceinline__ void y2_triggs(const VVK& k, const double* __restrict__ state, size_t n, size_t i, Y2& s, f2& first) {
    float fa, fb;
    s.a1 = state[i];
    s.a2 = state[n + i];
    s.a3 = state[2 * n + i];
    triggs(k, state[3 * n + i], s.a1, s.a2, s.a3, fa);
    s.b1 = state[i + 1];
    s.b2 = state[n + i + 1];
    s.b3 = state[2 * n + i + 1];
    triggs(k, state[3 * n + i + 1], s.b1, s.b2, s.b3, fb);
    first.x = fa;
    first.y = fb;
}

// Anticausal y pass, stand-alone (odd source widths): stores the blurred rows in place.
// resume / state_out: as in k_vv_y_fwd, for the band BELOW (the anticausal sweep runs bottom-up): with `resume` the last row of
// this band is an ordinary step from the state the rank below left after its first row, not the Triggs boundary value.
__device__ __forceinline__ void y2_load(const double* __restrict__ st, size_t n, size_t i, Y2& s) {
    s.a1 = st[i], s.b1 = st[i + 1];
    s.a2 = st[n + i], s.b2 = st[n + i + 1];
    s.a3 = st[2 * n + i], s.b3 = st[2 * n + i + 1];
}
__device__ __forceinline__ void y2_store(double* __restrict__ st, size_t n, size_t i, const Y2& s) {
    st[i] = s.a1, st[i + 1] = s.b1;
    st[n + i] = s.a2, st[n + i + 1] = s.b2;
    st[2 * n + i] = s.a3, st[2 * n + i + 1] = s.b3;
}
__global__ __launch_bounds__(64) void k_vv_y_bwd(float* __restrict__ data, int h, int pitch, size_t ps, VVK k,
                                                  const double* __restrict__ state, const double* __restrict__ resume,
                                                  double* __restrict__ state_out) {
    const int x = (blockIdx.x * WAVE + threadIdx.x) * 2;
    if (x >= pitch) return;
    float* p = data + blockIdx.y * ps + x;
    const size_t n = (size_t)gridDim.y * pitch, i = ystate_index(blockIdx.y, pitch, x);
    Y2 s;
    if (resume) {
        y2_load(resume, n, i, s);
        stream_rows<false>(p, pitch, h - 1, h, [&](int y, f2 xv) {
            *reinterpret_cast<f2*>(p + (size_t)y * pitch) = y2_step_bwd(s, k, xv);
        });
    } else {
        f2 first;
        y2_triggs(k, state, n, i, s, first);
        *reinterpret_cast<f2*>(p + (size_t)(h - 1) * pitch) = first;
        stream_rows<false>(p, pitch, h - 2, h - 1, [&](int y, f2 xv) {
            *reinterpret_cast<f2*>(p + (size_t)y * pitch) = y2_step_bwd(s, k, xv);
        });
    }
    if (state_out) y2_store(state_out, n, i, s);
}

// Anticausal y pass fused with the decimation (even source width): the blurred level is never written.
// A lone wavefront is bound by its own instruction issue (recurrence + IEEE divides), so the work is split over
// the two wavefronts of a workgroup:
//   wave 0 (producer)  runs the recurrence for 128 columns (two per lane), rows bottom-up, and drops each blurred
//                      row into an LDS ring (two slots of YCH rows);
//   wave 1 (consumer)  owns one column PAIR per lane: x-decimates each row (columns 2t, 2t+1, both overlaps = w2,
//                      CImg.h:29542-29555: acc = 0; acc += s0*d; acc += s1*d; acc /= W), keeps the last two
//                      x-decimated rows, and for every output row y-decimates (CImg.h:29557-29575: accumulation in
//                      INCREASING source row although rows arrive bottom-up; overlaps {h2,h2} for even h,
//                      {h2-t, h2, t+1} for odd h) and stores 256 bytes of the next pyramid level.
// One workgroup barrier per YCH rows hands a slot over.  rc = h-1-y counts rows in processing order.
__global__ __launch_bounds__(128) void k_vv_y_bwd_dec(const float* __restrict__ data, int w, int h, int pitch, size_t ps,
                                                      VVK k, const double* __restrict__ state, float* __restrict__ dst,
                                                      int w2, int h2, int dpitch, size_t dps, ZeroTiles zt,
                                                      const double* __restrict__ resume, double* __restrict__ state_out, int wh, int wh2) {
    // wh, wh2: the heights the decimation's overlap weights are taken from (CImg.h:29557-29575 weights by the LEVEL's
    // heights): h, h2 for a whole level; for a row band of a split pair the level's, not the band's (the quotient is the same
    // number, but (x*32 + y*32)/64 and (x*96 + y*96)/192 round differently)
    __shared__ __attribute__((aligned(16))) float ring[2][YCH][YCOLS];
    // the wave id is wave-uniform, but anything derived from threadIdx is a lane value to the compiler: readfirstlane
    // keeps the producer/consumer role branches scalar
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nchunks = (h + YCH - 1) / YCH;
    const int x = (blockIdx.x * WAVE + lane) * 2;  // first of this lane's two columns
    const bool col_live = x < pitch;
    const float* p = data + blockIdx.y * ps + (col_live ? x : 0);
    Y2 s{};
    f2 first{};
    f2 buf[YST][YCH];
    // zero-tile flags of this lane's 64-column tile, one bit per 64-row band, gathered once before the walk starts (the
    // host enables the flags only up to 256 bands); the row loop then only tests a bit
    unsigned long long zm[4] = {0, 0, 0, 0};
    const bool zt_on = zt.flags != nullptr;
    if (zt_on) {
        // a wavefront's 128 columns are two tiles (lanes 0..31 / 32..63): lane i fetches the flag of band g*64+i of either tile
        // (the bands of a tile column are contiguous bytes) and a ballot turns 64 flags into the mask
        const int tA = blockIdx.x * 2;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g * 64 < zt.NR) {  // wave-uniform
                const int band = g * 64 + lane;
                const bool inb = band < zt.NR;
                const unsigned char fa = inb ? zt.flags[zt.index(blockIdx.y, band, tA)] : 0;
                const unsigned char fb = inb && tA + 1 < zt.NC ? zt.flags[zt.index(blockIdx.y, band, tA + 1)] : 0;
                const unsigned long long mA = __ballot(fa != 0), mB = __ballot(fb != 0);
                zm[g] = lane < 32 ? mA : mB;
            }
        }
    }
    // A chunk of YCH rows never straddles a 64-row band (heights with flags are multiples of 64), so the flag is looked up
    // once per chunk.  The loads stay branch-free (the prefetch stream must remain one straight run of loads): a lane whose
    // tile is flagged re-reads one fixed row of its column -- a cache hit after the first time, no HBM traffic -- and
    // discards it.
    auto chunk_zero = [&](int chunk) {
        if (!zt_on) return false;
        const int rc = chunk * YCH, r = rc < h ? rc : h - 1, b = (h - 1 - r) >> 6, g = b >> 6;
        const unsigned long long m = g == 0 ? zm[0] : g == 1 ? zm[1] : g == 2 ? zm[2] : zm[3];
        return ((m >> (b & 63)) & 1) != 0;
    };
    auto ld = [&](int rc, bool zero) {
        const int r = rc < h ? rc : h - 1, y = h - 1 - r;
        const f2 v = *reinterpret_cast<const f2*>(p + (size_t)(zero ? h - 1 : y) * pitch);
        return zero ? f2{0.f, 0.f} : v;
    };
    if (wave == 0) {
        if (col_live) {
            if (resume)
                y2_load(resume, (size_t)gridDim.y * pitch, ystate_index(blockIdx.y, pitch, x), s);
            else
                y2_triggs(k, state, (size_t)gridDim.y * pitch, ystate_index(blockIdx.y, pitch, x), s, first);
        }
#pragma unroll
        for (int st = 0; st < YST - 1; ++st) {
            const bool z = chunk_zero(st);
#pragma unroll
            for (int u = 0; u < YCH; ++u) buf[st][u] = ld(st * YCH + u, z);
        }
    }
    // consumer state: this lane's output column t_x = x/2
    const float fsx = (float)(unsigned)w2, fw = (float)(unsigned)w, fh = (float)(unsigned)wh, fsy = (float)(unsigned)wh2;
    const bool h_odd = (h & 1) != 0;
    const int tx = blockIdx.x * WAVE + lane;
    float* dcol = dst + blockIdx.y * dps + tx;
    const bool dst_live = tx < w2;
    float Xp1 = 0.f, Xp2 = 0.f;  // x-decimated rows y+1 and y+2 (previous two rows in processing order)

    // The two roles run separate loops (same trip count, one barrier per chunk each): in one loop body the producer's 64
    // registers of prefetched rows would stay allocated across the consumer's code.
    if (wave == 0) {
        for (int j0 = 0; j0 <= nchunks; j0 += YST) {
#pragma unroll
            for (int st = 0; st < YST; ++st) {
                const int j = j0 + st;
                const bool z = chunk_zero(j + YST - 1);
#pragma unroll
                for (int u = 0; u < YCH; ++u) buf[(st + YST - 1) % YST][u] = ld((j + YST - 1) * YCH + u, z);
                if (j < nchunks) {
#pragma unroll
                    for (int u = 0; u < YCH; ++u) {
                        const int rc = j * YCH + u;
                        // sample h-1 takes the Triggs boundary value (CImg.h:34920); rows rc >= h are never read
                        const f2 o = (rc == 0 && !resume) ? first : y2_step_bwd(s, k, buf[st][u]);
                        *reinterpret_cast<f2*>(&ring[st & 1][u][2 * lane]) = o;
                    }
                }
                __syncthreads();
            }
        }
    } else {
        for (int j0 = 0; j0 <= nchunks; j0 += YST) {
#pragma unroll
            for (int st = 0; st < YST; ++st) {
                const int j = j0 + st;
                if (j >= 1 && j - 1 < nchunks) {
                    // The rows of a chunk are independent until the y step, and a row is one dependent chain (LDS read, two
                    // multiply-adds, an IEEE divide): all YCH chains are laid side by side, free of branches, so that they overlap --
                    // row after row the consumer was the slower wavefront of the two (264 cycles per row against the producer's
                    // 173).  Rows past the image (last chunk) are computed from whatever the ring holds and never stored.
                    const int rc0 = (j - 1) * YCH;
                    float E[YCH + 2];  // x-decimated rows: E[0], E[1] = the two rows before this chunk, E[2 + u] = row rc0 + u
                    E[0] = Xp2;
                    E[1] = Xp1;
#pragma unroll
                    for (int u = 0; u < YCH; ++u) {
                        const f2 v = *reinterpret_cast<const f2*>(&ring[(st + 1) & 1][u][2 * lane]);
                        float X = 0.f;
                        X += v.x * fsx;
                        X += v.y * fsx;
                        X /= fw;
                        E[2 + u] = X;
                        if (u == YCH / 2 - 1) __builtin_amdgcn_sched_barrier(0);  // two groups of four: eight divides side by side need 140 VGPRs
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // row y = 2t completes output row t; y = h-1-rc0-u and rc0 is a multiple of YCH (even): for an even height the
                    // odd u, for an odd height the even u
                    if (h_odd) {
#pragma unroll
                        for (int u = 0; u < YCH; u += 2) {
                            const int y = h - 1 - rc0 - u, t = y >> 1;
                            float a2 = 0.f;
                            a2 += E[2 + u] * (float)(unsigned)(h2 - t);
                            a2 += E[1 + u] * fsy;
                            a2 += E[u] * (float)(unsigned)(t + 1);
                            a2 /= fh;
                            if (y >= 0 && t < h2 && dst_live) dcol[(size_t)t * dpitch] = a2;
                        }
                    } else {
#pragma unroll
                        for (int u = 1; u < YCH; u += 2) {
                            const int y = h - 1 - rc0 - u, t = y >> 1;
                            float a2 = 0.f;
                            a2 += E[2 + u] * fsy;
                            a2 += E[1 + u] * fsy;
                            a2 /= fh;
                            if (y >= 0 && t < h2 && dst_live) dcol[(size_t)t * dpitch] = a2;
                        }
                    }
                    Xp2 = E[YCH];
                    Xp1 = E[YCH + 1];
                }
                __syncthreads();
            }
        }
    }
    if (state_out && wave == 0 && col_live) y2_store(state_out, (size_t)gridDim.y * pitch, ystate_index(blockIdx.y, pitch, x), s);
}

// ---- fused anticausal-x + causal-y sweep: row bands as pipeline stages ---------------------------------------
// The anticausal x sweep walks a row right-to-left, the causal y sweep walks a column top-down.  One wavefront owns
// a 64-row band of one plane and walks it right-to-left in 64x64 tiles: each tile is fetched to LDS, swept along x
// (lane = row; the x state never leaves the registers), then along y (lane = column) and written back once -- the
// level is read and written ONCE instead of twice.  The y sweep of tile (R,C) needs the y state of the 64 columns
// as band R-1 left it, so band R simply runs a constant lag (one y sweep + one hand-off) behind band R-1: all
// bands of all planes are in flight at once, a 64-deep software pipeline per plane.  The per-sample arithmetic and
// its order are those of k_vv_x_bwd / k_vv_y_fwd.
//
// Workgroups are persistent and claim bands in increasing R from one atomic counter, so the band a claimed band
// depends on has always been claimed earlier by a workgroup that is running (no residency assumption, no deadlock).
// The y state travels as 8-byte {tag, word} granules (the data is the flag; relaxed agent-scope atomics = sc1
// write-through stores / L1-bypassing loads; guide 6, Guideline 16, form R2): six granules per lane for three
// doubles.  tag = (epoch << 12) | (producer band + 1), so the slot of a column block is reused band after band
// within a launch.  Between launches the host clears the slots with k_clear_words and passes a constant epoch: a
// per-launch epoch would be frozen by a HIP-graph capture and stale tags of the previous replay would match at
// once.  Every spin is bounded; a timeout raises `abort` for all workgroups.
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

struct Wavefront {
    u64* yg;            // [planes][NC][7][64] granules: y state leaving band R towards band R+1 (+ one spare word)
    int mask_l0;        // level-0 implicit mask (see MaskL0): planes p%7==6 carry their x-blurred row in the 7th granule
    unsigned* counter;  // band queue head (zeroed by the host before the launch)
    unsigned* abort;    // set when a spin timed out (cleared in front of every launch sequence)
    unsigned* sticky;   // count of timed-out waits since the plan was created: NEVER cleared by the launch sequence, so
                        // a bail-out in any earlier queued call is still visible to stitch_plan_status_at
    int NR, NC, NP;
    unsigned epoch;
    unsigned spin_limit;      // polls before a wait gives up (STITCH_XBYF_SPIN_LIMIT; tests force the bail-out path with 0)
    ZeroTiles zt;             // zero-tile flags of T at this level (see ZeroTiles)
    int early_read;           // read the hand-off state ahead of the tile prefetch (granules_issue)
    unsigned long long* dbg;  // diagnostic build only: [workgroups][8] cycle sums per segment
};

constexpr int WF_GRAN = 7;  // granules per lane and slot: 3 doubles = 6 words, + 1 spare word
__device__ __forceinline__ void granules_publish(u64* base, int lane, unsigned tag, double a, double b, double c, unsigned extra) {
    const u64 w[3] = {(u64)__double_as_longlong(a), (u64)__double_as_longlong(b), (u64)__double_as_longlong(c)};
    __hip_atomic_store((gu64*)(base + 6 * WAVE + lane), ((u64)tag << 32) | extra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        __hip_atomic_store((gu64*)(base + (2 * i) * WAVE + lane), ((u64)tag << 32) | (w[i] & 0xffffffffu), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((gu64*)(base + (2 * i + 1) * WAVE + lane), ((u64)tag << 32) | (w[i] >> 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Wait for a state.  Pollers cost the streaming wavefronts bandwidth (guide, "polling-cost"), and most bands wait
// long for their first tile (band R starts R hand-offs after band 0), so the wait is two-phase: ONE lane re-reads
// ONE granule with a long sleep between polls; once its tag matches, the whole wavefront reads its six granules,
// repeating that (rarely) until every tag matches -- the stores of one publish may become visible in any order.
// Wave-uniform exit; false on timeout / abort.
// The same read, split in two so that it can be issued early: the seven loads go out BEFORE the next tile's prefetch
// (loads return in order, so a poll issued behind 16 tile loads would wait for all of them), the check happens after the
// x sweep.  In the steady state of the pipeline the band above is ahead and the early read already holds the state.
// (A second tile of prefetch was tried and is slower: more bytes in flight lengthen every queue the hand-off sits in.)
__device__ __forceinline__ void granules_issue(const u64* base, int lane, u64 g[WF_GRAN]) {
#pragma unroll
    for (int i = 0; i < WF_GRAN; ++i) g[i] = __hip_atomic_load((gu64*)(base + i * WAVE + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool granules_accept(const u64 g[WF_GRAN], unsigned tag, double& a, double& b, double& c, unsigned& extra) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < WF_GRAN; ++i) ok &= (unsigned)(g[i] >> 32) == tag;
    if (!__all(ok)) return false;
    extra = (unsigned)g[6];
    a = __longlong_as_double((long long)((g[0] & 0xffffffffu) | (g[1] << 32)));
    b = __longlong_as_double((long long)((g[2] & 0xffffffffu) | (g[3] << 32)));
    c = __longlong_as_double((long long)((g[4] & 0xffffffffu) | (g[5] << 32)));
    return true;
}
__device__ __forceinline__ bool granules_consume(const u64* base, int lane, unsigned tag, unsigned* abort, unsigned* sticky,
                                                 unsigned spin_limit, double& a, double& b, double& c, unsigned& extra) {
    for (unsigned spins = 0;; ++spins) {
        // two-phase poll: one lane watches the tag of one granule, the whole wavefront reads the seven only once it has
        // appeared (polling all seven, or sleeping less, measured within +-3 %)
        unsigned seen = 0;
        if (lane == 0)
            seen = (unsigned)(__hip_atomic_load((gu64*)(base + 5 * WAVE + (WAVE - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
        seen = __shfl(seen, 0, 64);
        if (seen == tag) {
            u64 g[WF_GRAN];
            bool ok = true;
#pragma unroll
            for (int i = 0; i < WF_GRAN; ++i) {
                g[i] = __hip_atomic_load((gu64*)(base + i * WAVE + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok &= (unsigned)(g[i] >> 32) == tag;
            }
            if (__all(ok)) {
                extra = (unsigned)g[6];
                a = __longlong_as_double((long long)((g[0] & 0xffffffffu) | (g[1] << 32)));
                b = __longlong_as_double((long long)((g[2] & 0xffffffffu) | (g[3] << 32)));
                c = __longlong_as_double((long long)((g[4] & 0xffffffffu) | (g[5] << 32)));
                return true;
            }
        }
        {
            if ((spins & 31) == 31) {
                unsigned ab = 0;
                if (lane == 0) ab = __hip_atomic_load((gu32*)abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ab = __shfl(ab, 0, 64);
                if (ab != 0 || spins > spin_limit) {
                    if (lane == 0) {
                        __hip_atomic_store((gu32*)abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicAdd(sticky, 1u);
                    }
                    return false;
                }
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
}

//
// MODE 0: the samples the causal x sweep left are read from `data` and swept in place.  MODE 1 / 2: the causal x sweep ran
// as k_vv_x_fwd<.., CKPT> and kept only its state in front of every tile; this kernel fetches the level's INPUT tile (MODE 1:
// the planes `in`; MODE 2, source-fused level 0: the pair's frames through the index plane, as k_vv_x_fwd<PX, true> does),
// re-runs the 64 causal steps from the checkpoint in LDS -- the same operations on the same operands, so the same bits --
// and carries on as MODE 0.  The level's x-swept samples never travel to HBM and back.
struct Recompute {
    const float* in;       // MODE 1: level input planes; MODE 2: the plan's level-0 planes (slot 0 of a pair = its index plane)
    const double* ckpt;    // [3][tiles][lines]
    const SeamDev* seam;   // implicit level-0 mask: band 0 of a mask plane generates the step
    ZeroTiles zi;          // MODE 2: index tiles that lie outside the frame
};
template <bool STAMP, typename PX = float, int MODE = 0>
__global__ __launch_bounds__(64) void k_vv_xbyf(float* __restrict__ data, int w, int h, int pitch, VVK k,
                                                 const double* __restrict__ state_x, long lines, double* __restrict__ state_y,
                                                 Wavefront wf, Recompute rc, typename CollapseSrc<PX, MODE == 2>::type pa) {
    __shared__ __attribute__((aligned(16))) float tile[TS * TP];
    const int lane = threadIdx.x;
    const unsigned nbands = (unsigned)wf.NP * wf.NR;
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    auto stamp = [&](int i) {
        if (STAMP) {
            t1 = __builtin_amdgcn_s_memtime();
            seg[i] += t1 - t0;
            t0 = t1;
        }
    };
    if (STAMP) t0 = __builtin_amdgcn_s_memtime();
    bool dead = false;
    while (!dead) {
        unsigned q = 0;
        if (lane == 0) q = atomicAdd(wf.counter, 1u);
        q = __builtin_amdgcn_readfirstlane(__shfl(q, 0, 64));  // provably wave-uniform: band, plane and every branch on them stay scalar
        if (q >= nbands) break;
        const int R = (int)(q / wf.NP), p = (int)(q % wf.NP);  // every plane's band R before any band R+1
        const int r0 = R * TS, nrows = min(TS, h - r0);
        float* base = data + ((size_t)p * h + r0) * pitch;
        // level-0 implicit mask plane below its first band: every row equals the x-blurred row that band 0 passes down
        // with the y state, so there is nothing to fetch and no x sweep to run (k_vv_x_fwd skipped these rows too)
        const bool const_rows = wf.mask_l0 && (p % 7 == 6) && R > 0;
        const bool pass_row = wf.mask_l0 && (p % 7 == 6);
        stamp(0);  // claim

        // x state of this band's rows: where the causal x sweep left it, through the Triggs boundary (CImg.h:34911-34922)
        double v1 = 0, v2 = 0, v3 = 0;
        float first;
        {
            const long line = (long)p * h + r0 + lane;
            double iplus = 0;
            if (lane < nrows) {
                v1 = state_x[line];
                v2 = state_x[lines + line];
                v3 = state_x[2 * lines + line];
                iplus = state_x[3 * lines + line];
            }
            triggs(k, iplus, v1, v2, v3, first);
        }
        f4 pre[16];
        // MODE 1/2: where this band's input comes from (all wave-uniform)
        const int src_q = p % 7, src_pr = p / 7;
        const bool gen_mask = MODE != 0 && pass_row;  // band 0 of an implicit mask plane: the step itself
        const bool gen_frame = MODE == 2 && src_q < 3, gen_mosaic = MODE == 2 && src_q >= 3 && src_q < 6;
        const float* in_base = MODE == 1 ? rc.in + ((size_t)p * h + r0) * pitch : nullptr;
        const float* idx_rows = MODE == 2 ? rc.in + ((size_t)src_pr * 7 * h + r0) * pitch : nullptr;  // rows r0.. of the pair's index plane
        const __amdgpu_buffer_rsrc_t rs = x_rsrc<PX>(pa, src_pr, src_q);
        const XShift ms = x_shift<PX>(pa, src_pr);
        SeamDev sd{};
        if (gen_mask) sd = rc.seam[src_pr];
        // checkpoint of tile C (the causal state in front of its first sample); tile 0 starts from the boundary value instead
        const long cline = (long)p * h + r0 + lane;
        double n1 = 0, n2 = 0, n3 = 0;
        auto fetch_ckpt = [&](int C) {
            if (MODE != 0 && C > 0 && lane < nrows) {
                const double* c = rc.ckpt + (size_t)(3 * C) * lines + cline;
                n1 = c[0];
                n2 = c[lines];
                n3 = c[2 * lines];
            }
        };
        // Tile C of this band into `pre`: from HBM, or zeros when the causal x sweep recorded it as all +0.  MODE 2 fetches in
        // two steps through the SAME registers: fetch_tile puts the tile's element indices into `pre` (index plane, frame
        // channels), gather_tile -- issued one sweep later -- replaces them by the samples (a second register tile for the
        // indices would halve the wavefronts per SIMD).
        auto fetch_tile = [&](int C) {  // true: the tile is all +0
            const bool zero = wf.zt.flags && wf.zt.flags[wf.zt.index(p, R, C)];
            if (zero) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pre[i] = f4{0.f, 0.f, 0.f, 0.f};
            } else if (MODE == 0)
                tile_load(base, pitch, C * TS, lane, pre);
            else if (gen_mask) {
                f4 v;
                const int c = C * TS + ((lane & 15) << 2);
                v.x = mask_step(sd, c);
                v.y = mask_step(sd, c + 1);
                v.z = mask_step(sd, c + 2);
                v.w = mask_step(sd, c + 3);
#pragma unroll
                for (int i = 0; i < 16; ++i) pre[i] = v;
            } else if (MODE == 1)
                tile_load(in_base, pitch, C * TS, lane, pre);
            else if (gen_frame) {
                if (rc.zi.flags && rc.zi.flags[rc.zi.index(src_pr, R, C)]) {
                    const float o = __uint_as_float(off_outside<PX>());
#pragma unroll
                    for (int i = 0; i < 16; ++i) pre[i] = f4{o, o, o, o};
                } else
                    tile_load(idx_rows, pitch, C * TS, lane, pre);
            }
            if (MODE != 0 && !zero) fetch_ckpt(C);
            return zero;
        };
        auto gather_tile = [&](int C, bool zero) {
            if constexpr (MODE == 2) {
                if (zero || gen_mask) return;
                if (gen_mosaic) mosaic_indices<PX>(ms, C * TS, lane, r0, w, pre);
                src_gather<PX>(rs, pre, pre);
            }
        };
        bool next_zero = false, spec = true;
        if (!const_rows) {
            next_zero = fetch_tile(wf.NC - 1);
            gather_tile(wf.NC - 1, next_zero);
        }
        for (int C = wf.NC - 1; C >= 0; --C) {
            const int c0 = C * TS, ncols = min(TS, w - c0);
            const bool tile_zero = next_zero;  // the tile now going to LDS holds +0 only
            u64* slot = wf.yg + ((size_t)p * wf.NC + C) * WF_GRAN * WAVE;
            u64 early[WF_GRAN];
            // speculation is dropped while it fails (the band above is not ahead: the seven loads would only be repeated
            // by the poll) and probed again every fourth tile
            const bool early_on = R > 0 && wf.early_read && (spec || (C & 3) == 0);
            if (early_on) granules_issue(slot, lane, early);  // ahead of the prefetch below (in-order return)
            double f1 = n1, f2 = n2, f3 = n3;  // this tile's checkpoint (MODE 1/2)
            if (!const_rows) {
                if constexpr (MODE == 2)
                    if (!tile_zero && !gen_mask) src_finish<PX>(gen_frame, pre);
                tile_to_lds(tile, lane, pre);
                if (C > 0) next_zero = fetch_tile(C - 1);  // next tile of the band, in flight during both sweeps
            }
            __syncthreads();
            stamp(1);  // tile fetch
            // ---- MODE 1/2: the causal x sweep of this tile again, lane = row r0+lane (k_vv_x_fwd's loop) -------------
            if (MODE != 0 && !const_rows && !tile_zero) {
                float* row = tile + lane * TP;
                if (C == 0) f1 = f2 = f3 = (double)row[0] / k.sumsq;  // CImg.h:34909
                const int jfull = ncols & ~15;
                for (int jb = 0; jb < jfull; jb += 16) {
                    float xs[16];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f4*>(xs + 4 * qq) = *reinterpret_cast<const f4*>(row + jb + 4 * qq);
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        double v0 = (double)xs[u];
                        v0 += f1 * k.f1;
                        v0 += f2 * k.f2;
                        v0 += f3 * k.f3;
                        xs[u] = (float)v0;
                        f3 = f2;
                        f2 = f1;
                        f1 = v0;
                    }
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f4*>(row + jb + 4 * qq) = *reinterpret_cast<const f4*>(xs + 4 * qq);
                }
                for (int j = jfull; j < ncols; ++j) {
                    double v0 = (double)row[j];
                    v0 += f1 * k.f1;
                    v0 += f2 * k.f2;
                    v0 += f3 * k.f3;
                    row[j] = (float)v0;
                    f3 = f2;
                    f2 = f1;
                    f1 = v0;
                }
            }
            if (!const_rows && C > 0) gather_tile(C - 1, next_zero);  // MODE 2: the indices have arrived during the causal sweep
            // ---- anticausal x sweep, lane = row r0+lane ---------------------------------------------------------
            // Zeros in, zero state: every product and sum of the recurrence is +0 again (x*sum = +0, and +0 plus a zero of
            // either sign is +0), so the sweep would rewrite the zeros it found and leave the state as it is: skipped.
            const bool x_idle = tile_zero && !const_rows && C != wf.NC - 1 &&
                                __ballot((__double_as_longlong(v1) | __double_as_longlong(v2) | __double_as_longlong(v3)) != 0) == 0;
            if (!const_rows && !x_idle) {
                float* row = tile + lane * TP;
                int jtop = ncols;
                if (C == wf.NC - 1) {
                    row[jtop - 1] = first;  // sample w-1 takes the boundary value (CImg.h:34920)
                    --jtop;
                }
                const int jfull = jtop & ~15;
                for (int j = jtop - 1; j >= jfull; --j) {
                    double v0 = (double)row[j];
                    v0 *= k.sum;
                    v0 += v1 * k.f1;
                    v0 += v2 * k.f2;
                    v0 += v3 * k.f3;
                    row[j] = (float)v0;
                    v3 = v2;
                    v2 = v1;
                    v1 = v0;
                }
                for (int jb = jfull - 16; jb >= 0; jb -= 16) {
                    float xs[16];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f4*>(xs + 4 * qq) = *reinterpret_cast<const f4*>(row + jb + 4 * qq);
#pragma unroll
                    for (int u = 15; u >= 0; --u) {
                        double v0 = (double)xs[u];
                        v0 *= k.sum;
                        v0 += v1 * k.f1;
                        v0 += v2 * k.f2;
                        v0 += v3 * k.f3;
                        xs[u] = (float)v0;
                        v3 = v2;
                        v2 = v1;
                        v1 = v0;
                    }
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f4*>(row + jb + 4 * qq) = *reinterpret_cast<const f4*>(xs + 4 * qq);
                }
            }
            __syncthreads();
            stamp(2);  // x sweep
            // ---- causal y sweep, lane = column c0+lane -----------------------------------------------------------
            float* colp = tile + lane;
            double u1, u2, u3;
            unsigned rowbits = 0;
            if (R == 0) {
                u1 = u2 = u3 = (double)colp[0] / k.sumsq;  // CImg.h:34909
                rowbits = __float_as_uint(colp[0]);     // the x-blurred row (mask plane: identical for every y)
            } else {
                const bool hit = early_on && granules_accept(early, (wf.epoch << 12) | (unsigned)R, u1, u2, u3, rowbits);
                if (early_on) spec = hit;
                if (!hit && !granules_consume(slot, lane, (wf.epoch << 12) | (unsigned)R, wf.abort, wf.sticky, wf.spin_limit, u1, u2, u3, rowbits)) {
                    dead = true;
                    break;
                }
            }
            stamp(3);  // y state wait
            const float rowv = __uint_as_float(rowbits);
            unsigned ynz = 0;  // OR of the bit patterns of this lane's column after the sweep
            const double iplus_y = const_rows ? (double)rowv : (double)colp[(nrows - 1) * TP];  // last band only (CImg.h:34906)
            // the same for the y sweep: a tile of +0 under a +0 state stays +0 (only the columns that exist are asked)
            const bool y_idle = x_idle && __ballot(lane < ncols && (__double_as_longlong(u1) | __double_as_longlong(u2) |
                                                                   __double_as_longlong(u3)) != 0) == 0;
            for (int j0 = 0; j0 < (y_idle ? 0 : nrows); j0 += 16) {
                if (j0 + 16 <= nrows) {
                    // all 16 column samples are read before the chain starts and written after it ends: a read placed
                    // between the writes would be kept in program order (the compiler cannot tell the rows apart) and
                    // put one LDS round trip per sample on the critical path
                    float ys[16];
                    if (const_rows) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) ys[u] = rowv;
                    } else {
#pragma unroll
                        for (int u = 0; u < 16; ++u) ys[u] = colp[(j0 + u) * TP];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        double v0 = (double)ys[u];
                        v0 += u1 * k.f1;
                        v0 += u2 * k.f2;
                        v0 += u3 * k.f3;
                        ys[u] = (float)v0;
                        ynz |= __float_as_uint(ys[u]);
                        u3 = u2;
                        u2 = u1;
                        u1 = v0;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) colp[(j0 + u) * TP] = ys[u];
                } else {
                    for (int j = j0; j < nrows; ++j) {
                        double v0 = const_rows ? (double)rowv : (double)colp[j * TP];
                        v0 += u1 * k.f1;
                        v0 += u2 * k.f2;
                        v0 += u3 * k.f3;
                        colp[j * TP] = (float)v0;
                        ynz |= __float_as_uint(colp[j * TP]);
                        u3 = u2;
                        u2 = u1;
                        u1 = v0;
                    }
                }
            }
            if (R < wf.NR - 1)
                granules_publish(slot, lane, (wf.epoch << 12) | (unsigned)(R + 1), u1, u2, u3, pass_row ? rowbits : 0u);
            else {  // last band: the state the anticausal y sweep starts from ([4][planes][pitch], as k_vv_y_fwd leaves it)
                const size_t n = (size_t)wf.NP * pitch, i = (size_t)p * pitch + c0 + lane;
                state_y[i] = u1;
                state_y[n + i] = u2;
                state_y[2 * n + i] = u3;
                state_y[3 * n + i] = iplus_y;
            }
            __syncthreads();
            stamp(4);  // y sweep + publish
            bool out_zero = false;
            if (wf.zt.flags) {  // the y sweep ran over every column of the tile; columns >= w hold whatever the fetch left: not counted
                out_zero = __ballot(lane < ncols && ynz != 0) == 0;
                if (lane == 0) wf.zt.flags[wf.zt.index(p, R, C)] = out_zero ? 1 : 0;
            }
            if (!out_zero) tile_store_rows(base, pitch, c0, lane, tile, nrows);  // a partial last band must not touch the next plane's rows
            __syncthreads();  // the tile buffer is refilled next
            stamp(5);  // store
        }
    }
    if (STAMP && lane == 0)
        for (int i = 0; i < 8; ++i) wf.dbg[(size_t)blockIdx.x * 8 + i] = seg[i];
}

// ---- B3': Deriche order 0, CImg.h:34779-34797 (all float).  The causal pass needs a line of temporaries Y; the
// anticausal pass adds Y back.  One line per work-item, y pass only coalesced; the x pass reuses the y kernel on
// a transposed view is NOT possible in place, so x runs one row per work-item straight from global memory
// (ex6 variant; not on the benchmarked path).
__global__ __launch_bounds__(64) void k_deriche(float* __restrict__ data, float* __restrict__ Y, int N, size_t off,
                                                size_t line_stride, long lines_per_plane, size_t ps, long lines, DRK k) {
    const long line = (long)blockIdx.x * WAVE + threadIdx.x;
    if (line >= lines) return;
    const long plane = line / lines_per_plane, li = line % lines_per_plane;
    float* ptrX = data + plane * ps + li * line_stride;
    float* ptrY = Y + plane * ps + li * line_stride;
    float xp = *ptrX, yb, yp;
    yb = yp = (float)(k.coefp * xp);
    for (int m = 0; m < N; ++m) {
        const float xc = ptrX[(size_t)m * off];
        const float yc = k.a0 * xc + k.a1 * xp - k.b1 * yp - k.b2 * yb;
        ptrY[(size_t)m * off] = yc;
        xp = xc;
        yb = yp;
        yp = yc;
    }
    float xn, xa, yn, ya;
    xn = xa = ptrX[(size_t)(N - 1) * off];
    yn = ya = k.coefn * xn;
    for (int n = N - 1; n >= 0; --n) {
        const float xc = ptrX[(size_t)n * off];
        const float yc = k.a2 * xn + k.a3 * xa - k.b1 * yn - k.b2 * ya;
        xa = xn;
        xn = xc;
        ya = yn;
        yn = yc;
        ptrX[(size_t)n * off] = ptrY[(size_t)n * off] + yc;
    }
}

// ---- B4: moving-average decimation, CImg.h:29539-29575 -------------------------------------------------------
// Output t of an axis accumulates, in increasing s, src[s]*(float)d into a float that starts at 0, where d is
// the overlap of [t*n_src,(t+1)*n_src) with [s*n_dst,(s+1)*n_dst), then divides once by (float)n_src.  x first
// (result rounded to float), then y -- both inside one work-item, which owns one output sample of one plane.
// yoff / soff: level row of the destination's / source's row 0 (0 for a whole level; a row band of a split pair passes its
// first rows, and the LEVEL's heights as h, h2, so that the overlap weights are the level's)
__global__ __launch_bounds__(256) void k_decimate(const float* __restrict__ src, int w, int h, int spitch, size_t sps,
                                                  float* __restrict__ dst, int w2, int h2, int dpitch, size_t dps, int yoff, int soff) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, pl = blockIdx.z;
    if (x >= dpitch) return;
    float out = 0.f;
    if (x < w2) {
        const Taps tx = make_taps(x, w, w2), ty = make_taps(y + yoff, h, h2);
        const float* p = src + pl * sps + tx.s0;
        const float fw = (float)(unsigned)w, fh = (float)(unsigned)h;
        float acc_y = 0.f;
        for (int j = 0; j < ty.n; ++j) {
            const float* row = p + (size_t)(ty.s0 - soff + j) * spitch;
            float acc = 0.f;
            for (int i = 0; i < tx.n; ++i) acc += row[i] * tx.d[i];
            acc /= fw;
            acc_y += acc * ty.d[j];
        }
        out = acc_y / fh;
    }
    dst[pl * dps + (size_t)y * dpitch + x] = out;
}

// ---- B5/B6: expand, Laplacian, per-level blend, collapse -----------------------------------------------------
// CImg.h:29618-29690 linear interpolation: out = (float)((1-alpha)*v1 + alpha*v2) in double with
// v2 = v1 at the last source sample; the x pass is rounded to float before the y pass.  Index/alpha tables are
// the reference's serial `curr = min(n_src-1, curr+f)` walk, computed on the host.  A source axis of length 1
// is nearest-neighbour (:29620, :29657).
struct ExpandTab {
    const int32_t* ix;
    const double* ax;
    const int32_t* iy;
    const double* ay;
};

__device__ __forceinline__ float lerp_ref(double alpha, float v1, float v2) {
    return (float)((1 - alpha) * (double)v1 + alpha * (double)v2);
}

struct ExpandPos {  // everything about (x,y) that does not depend on the plane
    int o11, o12, o21, o22;
    double ax, ay;
    bool x_nearest, y_nearest;
};
__device__ __forceinline__ ExpandPos expand_pos(const ExpandTab& tb, int x, int y, int sw, int sh, int spitch) {
    ExpandPos e;
    const int ix = tb.ix[x], iy = tb.iy[y];
    const int ix2 = ix < sw - 1 ? ix + 1 : ix, iy2 = iy < sh - 1 ? iy + 1 : iy;
    e.ax = tb.ax[x];
    e.ay = tb.ay[y];
    e.o11 = iy * spitch + ix;
    e.o12 = iy * spitch + ix2;
    e.o21 = iy2 * spitch + ix;
    e.o22 = iy2 * spitch + ix2;
    e.x_nearest = (sw == 1);
    e.y_nearest = (sh == 1);
    return e;
}
__device__ __forceinline__ float expand_at(const float* __restrict__ pl, const ExpandPos& e) {
    float r1, r2;
    if (e.x_nearest) {
        r1 = pl[e.o11];
        r2 = pl[e.o21];
    } else {
        r1 = lerp_ref(e.ax, pl[e.o11], pl[e.o12]);
        r2 = lerp_ref(e.ax, pl[e.o21], pl[e.o22]);
    }
    return e.y_nearest ? r1 : lerp_ref(e.ay, r1, r2);
}

// blend of one sample, ImageProcess.cpp:749-751: a*m is a float product, b*(1.0-m) and the sum are double.
__device__ __forceinline__ float blend_ref(float a, float b, float m) {
    const float am = a * m;
    return (float)((double)am + (double)b * (1.0 - (double)m));
}

// top level: E = a*m + b*(1-m) on the Gaussian top (no Laplacian, no clamp); E has 3 pitched planes.
__global__ __launch_bounds__(256) void k_blend_top(const float* __restrict__ g_all, int pitch, int h, size_t ps,
                                                   float* __restrict__ e_all) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= pitch) return;
    const float* g = g_all + (size_t)blockIdx.z * 7 * ps;
    float* e = e_all + (size_t)blockIdx.z * 3 * ps;
    const size_t o = (size_t)y * pitch + x;
    const float m = g[o + 6 * ps];
#pragma unroll
    for (int c = 0; c < 3; ++c) e[o + c * ps] = blend_ref(g[o + c * ps], g[o + (3 + c) * ps], m);
}

// level l < L-1:  La = Ga_l - EXPAND(Ga_{l+1}), Lb likewise (float subtract, CImg.h:12096-12107);
// S = blend(La, Lb, m_l);  E_l = clamp(S + EXPAND(E_{l+1}), 0, 255) (ImageProcess.cpp:766-769).
// OUT = float planes (pitched, next collapse input) or the final dense canvas (float, or uint8_t by truncation).
template <typename OUT>
struct OutPtrs {
    OUT* p[MAXB];
    uint8_t* q[MAXB];  // optional second copy of a float mosaic as unsigned char (the reference's own output type,
                       // CImg<unsigned char>(CImg<float>): C-cast truncation), written by the level-0 collapse; nullptr = none
};
// One work-item owns one column of a strip of CROWS output rows.  EXPAND's x pass depends only on the source row,
// and consecutive output rows share their source rows (iy advances by at most one per output row when
// up-sampling), so the x-interpolated values of the two current source rows are kept in registers and only a newly
// entered source row is interpolated: the double-precision work per pixel halves, the values are the same floats.
#ifndef STITCH_CROWS
#define STITCH_CROWS 8
#endif
constexpr int CROWS = STITCH_CROWS;
// G_0 of a source-fused plan: evaluated from the pair's frames; otherwise read from the level's planes.
template <typename OUT>
__device__ __forceinline__ void level_ab(const PairArgs<OUT>& pa, int pr, int use_src, const float* g, size_t o, size_t ps, int x, int y,
                                         float a[3], float b[3]) {
    if (use_src) {  // branch-free: clamped addresses, zero selected afterwards
        const PairSrc<OUT> src(pa, pr);
        const unsigned fo = reinterpret_cast<const unsigned*>(g)[o];  // k_src_index's byte offset in the slot of plane 0
        const bool fin = fo != off_outside<OUT>();
        size_t mo = 0;
        const bool min_ = src.mosaic_at(x, y, mo);
        mo = min_ ? mo : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float fa = src.frame_val((fin ? fo : 0u) / sizeof(OUT), c), fb = src.mosaic_val(mo, c);
            a[c] = fin ? fa : 0.f;
            b[c] = min_ ? fb : 0.f;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = g[o + c * ps];
            b[c] = g[o + (3 + c) * ps];
        }
    }
}
__device__ __forceinline__ void level_ab(const NoPairArgs&, int, int, const float* g, size_t o, size_t ps, int, int, float a[3], float b[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a[c] = g[o + c * ps];
        b[c] = g[o + (3 + c) * ps];
    }
}
// everything a collapse launch of one level needs (kernel argument)
template <typename OUT, bool DENSE>
struct CollapseArgs {
    const float* g_all;  // level l: 7 planes per pair (plane 0 = the index plane when source-fused)
    int w, h, pitch;
    size_t ps;
    const float *gn_all, *en_all;  // level l+1: G (7 planes per pair) and E (3 planes per pair)
    int sw, sh, spitch;
    size_t sps;
    ExpandTab tb;
    OutPtrs<OUT> outs;
    int opitch;
    size_t ops;
    const SeamDev* seam_l0;  // level 0: the mask is the seam's step function
    typename CollapseSrc<OUT, DENSE>::type pa;
    int use_src, crows;
    int xa, xb;  // columns [xa, xb) are done four per work-item (collapse_cols4), the rest one per work-item
};

// one column x, rows [y0, y1) of pair pr
template <typename OUT, bool DENSE>
__device__ __forceinline__ void collapse_cols1(const CollapseArgs<OUT, DENSE>& A, int x, int y0, int y1, int pr) {
    const int w = A.w, pitch = A.pitch, sw = A.sw, sh = A.sh, spitch = A.spitch, opitch = A.opitch;
    const size_t ps = A.ps, sps = A.sps, ops = A.ops;
    const ExpandTab& tb = A.tb;
    const SeamDev* __restrict__ seam_l0 = A.seam_l0;
    const float* g = A.g_all + (size_t)pr * 7 * ps;
    const float* gn = A.gn_all + (size_t)pr * 7 * sps;
    const float* en = A.en_all + (size_t)pr * 3 * sps;
    OUT* __restrict__ out = DENSE ? A.outs.p[pr] : A.outs.p[0] + (size_t)pr * 3 * ops;
    if (!DENSE && x >= w) {
        for (int y = y0; y < y1; ++y)
#pragma unroll
            for (int c = 0; c < 3; ++c) out[(size_t)y * opitch + x + c * ops] = OUT(0);
        return;
    }
    const int ix = tb.ix[x], ix2 = ix < sw - 1 ? ix + 1 : ix;
    const double ax = tb.ax[x];
    const bool x_nearest = (sw == 1), y_nearest = (sh == 1);
    // x pass of one source row for the nine planes that are expanded: a0..a2, b0..b2 of G_{l+1}, then E_{l+1}
    auto xrow = [&](int row, float X[9]) {
        const size_t o1 = (size_t)row * spitch + ix, o2 = (size_t)row * spitch + ix2;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float* pl = q < 6 ? gn + q * sps : en + (q - 6) * sps;
            X[q] = x_nearest ? pl[o1] : lerp_ref(ax, pl[o1], pl[o2]);
        }
    };
    float X1[9], X2[9];
    int cur1 = -1, cur2 = -1;
    const float m_step = seam_l0 ? mask_step(seam_l0[pr], x) : 0.f;  // level 0: the mask is the step itself
    for (int y = y0; y < y1; ++y) {
        const int iy = tb.iy[y], iy2 = iy < sh - 1 ? iy + 1 : iy;
        const double ay = tb.ay[y];
        if (iy != cur1) {
            if (iy == cur2) {
#pragma unroll
                for (int q = 0; q < 9; ++q) X1[q] = X2[q];
            } else
                xrow(iy, X1);
            cur1 = iy;
        }
        if (iy2 != cur2) {
            if (iy2 == cur1) {
#pragma unroll
                for (int q = 0; q < 9; ++q) X2[q] = X1[q];
            } else
                xrow(iy2, X2);
            cur2 = iy2;
        }
        const size_t o = (size_t)y * pitch + x;
        const float m = seam_l0 ? m_step : g[o + 6 * ps];
        float ga[3], gb[3];
        level_ab(A.pa, pr, A.use_src, g, o, ps, x, y, ga, gb);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ea = y_nearest ? X1[c] : lerp_ref(ay, X1[c], X2[c]);
            const float eb = y_nearest ? X1[3 + c] : lerp_ref(ay, X1[3 + c], X2[3 + c]);
            const float ee = y_nearest ? X1[6 + c] : lerp_ref(ay, X1[6 + c], X2[6 + c]);
            const float la = ga[c] - ea;
            const float lb = gb[c] - eb;
            const float s_ = blend_ref(la, lb, m);
            float v = s_ + ee;
            if (v > 255.f)
                v = 255.f;
            else if (v < 0.f)
                v = 0.f;
            if (DENSE) {  // the finished mosaic is not read again by this sequence
                __builtin_nontemporal_store(px_store<OUT>(v), &out[(size_t)y * opitch + x + c * ops]);
                if (sizeof(OUT) == 4 && A.outs.q[pr]) A.outs.q[pr][(size_t)y * opitch + x + c * ops] = px_store<uint8_t>(v);
            } else
                out[(size_t)y * opitch + x + c * ops] = px_store<OUT>(v);
        }
    }
}

// one column per work-item over the whole level (xa == xb) -- levels too small or too irregular for k_collapse4
template <typename OUT, bool DENSE>
__global__ __launch_bounds__(256) void k_collapse(CollapseArgs<OUT, DENSE> A) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * A.crows, pr = blockIdx.z;
    if (x >= (DENSE ? A.w : A.pitch)) return;
    collapse_cols1<OUT, DENSE>(A, x, y0, min(y0 + A.crows, A.h), pr);
}

// ---- the same collapse, four columns and ONE channel per work-item ------------------------------------------------
// k_collapse moves 4 bytes per lane and access and spends as many instructions on addresses and table look-ups as on
// the arithmetic.  Here a work-item owns FOUR adjacent columns x0..x0+3 (x0 a multiple of 4) of a strip of rows of ONE
// colour channel; the three wavefronts of a workgroup are the three channels of the same 256 columns.  The level's own
// planes (G_l, the mask, the index plane of a source-fused level 0, the output) are moved 16 bytes per lane, table
// entries are read once per strip, and -- the point of the channel split -- the x-interpolated source rows a work-item
// keeps are 2 x 3 planes x 4 columns = 24 registers instead of 72, so that 7-8 wavefronts per SIMD are resident instead
// of 2-3: the kernel is a chain of dependent loads per row (index -> gather -> arithmetic -> store) and lives on the
// number of such chains in flight.  The channels of a tile share its index and mask loads through the CU's L1.
// The up-sampling step is (sw-1)/(w-1) < 1/2, and away from the ends of a row the x taps follow one pattern:
// ix[x0..x0+3] = s, s+1, s+1, s+2 (columns 2m, 2m+1 interpolate between samples m-1, m and m, m+1 while the accumulated
// step stays within half a sample of x/2), so the taps of the four columns are fixed elements of ONE 4-byte-aligned
// 16-byte load at s.  The host knows the tables and hands this path only the column range [xa, xb) where every group of
// four has that pattern (6144 -> all but the first 256 columns); the other columns run collapse_cols1 in the same launch.
// Arithmetic and its order per sample are k_collapse's (lerp_ref, blend_ref, clamp).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at 4-byte alignment (dword-aligned dwordx4)

template <typename OUT>
__device__ __forceinline__ void store4(OUT* p, const float v[4]);
template <>
__device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
    __builtin_nontemporal_store(f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4*>(p));
}
template <>
__device__ __forceinline__ void store4<uint8_t>(uint8_t* p, const float v[4]) {
    const unsigned w = (unsigned)(uint8_t)(int)v[0] | ((unsigned)(uint8_t)(int)v[1] << 8) | ((unsigned)(uint8_t)(int)v[2] << 16) |
                       ((unsigned)(uint8_t)(int)v[3] << 24);
    __builtin_nontemporal_store(w, reinterpret_cast<unsigned*>(p));
}

// level-0 samples of channel c at four adjacent canvas columns of a source-fused plan (what k_compose would have stored)
template <typename PX>
struct SrcRow4 {
    __amdgpu_buffer_rsrc_t fr, mo;
    int mw, mh, ox, oy;
    bool m_all, m_col[4];  // x part of the mosaic range test for this lane's columns
    unsigned m_x0;         // byte offset of column x0 within a mosaic row (valid when m_col[0])
    __device__ __forceinline__ SrcRow4(const PairArgs<PX>& pa, int pr, int c, int x0, int w) {
        const size_t fe = (size_t)pa.fw[pr] * pa.fh[pr], me = (size_t)pa.mw[pr] * pa.mh[pr];
        fr = plane_rsrc(pa.frame[pr] + c * fe, fe);
        mo = plane_rsrc(pa.mosaic[pr] + c * me, me);
        mw = pa.mw[pr];
        mh = pa.mh[pr];
        ox = pa.ox[pr];
        oy = pa.oy[pr];
        m_all = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long mx = (long long)(x0 + j) + ox;
            m_col[j] = x0 + j < w && mx >= 0 && mx < mw;
            m_all = m_all && m_col[j];
        }
        m_x0 = (unsigned)((long long)x0 + ox) * (unsigned)sizeof(PX);
    }
    // raw bits of a[j], b[j] of canvas row y (loads only: nothing here waits for them); idx = the four byte offsets
    // k_src_index left (or "outside")
    __device__ __forceinline__ void issue(const u4 idx, int y, float a[4], float b[4]) const {
        const long long my = (long long)y + oy;
        const bool row_ok = my >= 0 && my < mh;  // wave-uniform
        const unsigned rowb = (unsigned)my * (unsigned)mw * (unsigned)sizeof(PX);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = buf_raw<PX>(fr, idx[j]);  // "outside" is beyond the plane: reads 0
        if (row_ok) {
            if (sizeof(PX) == 4 && __all(m_all)) {  // the common case: four consecutive floats, dword-aligned
                const u4 v = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(mo, rowb + m_x0, 0, 0));
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = __uint_as_float(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = buf_raw<PX>(mo, m_col[j] ? rowb + m_x0 + (unsigned)(j * sizeof(PX)) : off_outside<PX>());
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = 0.f;  // raw 0 = sample 0 in either pixel type
        }
    }
    // raw bits -> the values k_compose would have stored
    __device__ __forceinline__ void finish(float a[4], float b[4]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = warped_px<PX>(raw_to_px<PX>(a[j]));
            b[j] = raw_to_px<PX>(b[j]);
        }
    }
};
struct NoSrcRow4 {
    __device__ __forceinline__ NoSrcRow4(const NoPairArgs&, int, int, int, int) {}
    __device__ __forceinline__ void issue(const u4, int, float[4], float[4]) const {}
    __device__ __forceinline__ void finish(float[4], float[4]) const {}
};
template <typename OUT, bool DENSE>
struct SrcRow4Of {
    typedef NoSrcRow4 type;
};
template <typename OUT>
struct SrcRow4Of<OUT, true> {
    typedef SrcRow4<OUT> type;
};

#ifndef STITCH_C4_WAVES
#define STITCH_C4_WAVES 7  // <= 72 registers: 7 wavefronts per SIMD (measured, level 0 of 8 config-2 pairs: 4 -> 2.39 ms, 6 -> 1.75, 7 -> 1.67, 8 with spills -> 1.70)
#endif
#ifndef STITCH_C4_PREFETCH
#define STITCH_C4_PREFETCH 0
#endif
#ifndef STITCH_C4_OPAQUE_ALPHA
#define STITCH_C4_OPAQUE_ALPHA 1
#endif
// channel c of columns x0..x0+3, rows [y0, y1) of pair pr
template <typename OUT, bool DENSE>
__device__ __forceinline__ void collapse_cols4(const CollapseArgs<OUT, DENSE>& A, int c, int x0, int y0, int y1, int pr) {
    const int pitch = A.pitch, sh = A.sh, spitch = A.spitch, opitch = A.opitch;
    const size_t ps = A.ps, sps = A.sps, ops = A.ops;
    const ExpandTab& tb = A.tb;
    const SeamDev* __restrict__ seam_l0 = A.seam_l0;
    const float* g = A.g_all + (size_t)pr * 7 * ps;
    // the three planes of level l+1 this channel expands: a_c, b_c of G_{l+1} and channel c of E_{l+1}
    const float* sa = A.gn_all + ((size_t)pr * 7 + c) * sps;
    const float* sb = sa + 3 * sps;
    const float* se = A.en_all + ((size_t)pr * 3 + c) * sps;
    OUT* __restrict__ out = (DENSE ? A.outs.p[pr] : A.outs.p[0] + (size_t)pr * 3 * ops) + c * ops;
    uint8_t* __restrict__ out8 = DENSE && A.outs.q[pr] ? A.outs.q[pr] + c * ops : nullptr;
    const bool use_src = DENSE && A.use_src;
    // table entries of the four columns; their taps are samples s0 .. s0+3 of a source row
    const int s0 = tb.ix[x0];
    double axs[4];
    {
        const double2 a01 = *reinterpret_cast<const double2*>(tb.ax + x0), a23 = *reinterpret_cast<const double2*>(tb.ax + x0 + 2);
        axs[0] = a01.x, axs[1] = a01.y, axs[2] = a23.x, axs[3] = a23.y;
    }
    // x pass of one source row for the three planes
    auto xrow = [&](int row, float X[3][4]) {
        const unsigned o = (unsigned)row * (unsigned)spitch + (unsigned)s0;
        const f4 v[3] = {*reinterpret_cast<const f4u*>(sa + o), *reinterpret_cast<const f4u*>(sb + o), *reinterpret_cast<const f4u*>(se + o)};
#if STITCH_C4_OPAQUE_ALPHA
        // registers are what bounds the number of resident wavefronts here: keep the four alphas, not also the four
        // (1 - alpha) the compiler would hoist out of the row loop (recomputed per source row: four v_add_f64)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(axs[j]));
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            X[q][0] = lerp_ref(axs[0], v[q].x, v[q].y);
            X[q][1] = lerp_ref(axs[1], v[q].y, v[q].z);
            X[q][2] = lerp_ref(axs[2], v[q].y, v[q].z);
            X[q][3] = lerp_ref(axs[3], v[q].z, v[q].w);
        }
    };
    float X1[3][4], X2[3][4];
    int cur1 = -1, cur2 = -1;
    float m4[4] = {0.f, 0.f, 0.f, 0.f};
    if (seam_l0) {  // level 0: the mask is the step itself
        const SeamDev sd = seam_l0[pr];
#pragma unroll
        for (int j = 0; j < 4; ++j) m4[j] = mask_step(sd, x0 + j);
    }
    const typename SrcRow4Of<OUT, DENSE>::type src(A.pa, pr, c, x0, A.w);
    // the level's own samples of one row, as raw bits: two planes (or the gathers of a source-fused level 0) + the mask
    auto row_issue = [&](int y, float ga[4], float gb[4], f4& vm) {
        const unsigned o = (unsigned)y * (unsigned)pitch + (unsigned)x0;
        if (use_src)
            src.issue(*reinterpret_cast<const u4*>(g + o), y, ga, gb);
        else {
            const f4 va = *reinterpret_cast<const f4*>(g + o + c * ps), vb = *reinterpret_cast<const f4*>(g + o + (3 + c) * ps);
#pragma unroll
            for (int j = 0; j < 4; ++j) ga[j] = va[j], gb[j] = vb[j];
        }
        if (!seam_l0) vm = *reinterpret_cast<const f4*>(g + o + 6 * ps);
    };
    float ga[4], gb[4];
    f4 vm = {0.f, 0.f, 0.f, 0.f};
#if STITCH_C4_PREFETCH
    // the row's loads are issued one row ahead of their use
    float gan[4], gbn[4];
    f4 vmn = {0.f, 0.f, 0.f, 0.f};
    row_issue(y0, gan, gbn, vmn);
#endif
    for (int y = y0; y < y1; ++y) {
        const int iy = tb.iy[y], iy2 = iy < sh - 1 ? iy + 1 : iy;
        const double ay = tb.ay[y];
        if (iy != cur1) {
            if (iy == cur2) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) X1[q][j] = X2[q][j];
            } else
                xrow(iy, X1);
            cur1 = iy;
        }
        if (iy2 != cur2) {
            if (iy2 == cur1) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) X2[q][j] = X1[q][j];
            } else
                xrow(iy2, X2);
            cur2 = iy2;
        }
#if STITCH_C4_PREFETCH
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[j] = gan[j], gb[j] = gbn[j];
        vm = vmn;
        if (y + 1 < y1) row_issue(y + 1, gan, gbn, vmn);  // behind the x pass: its loads are waited for at once
#else
        row_issue(y, ga, gb, vm);
#endif
        if (use_src) src.finish(ga, gb);
        if (!seam_l0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) m4[j] = vm[j];
        }
        float v4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ea = lerp_ref(ay, X1[0][j], X2[0][j]);
            const float eb = lerp_ref(ay, X1[1][j], X2[1][j]);
            const float ee = lerp_ref(ay, X1[2][j], X2[2][j]);
            const float la = ga[j] - ea;
            const float lb = gb[j] - eb;
            const float s_ = blend_ref(la, lb, m4[j]);
            float v = s_ + ee;
            if (v > 255.f)
                v = 255.f;
            else if (v < 0.f)
                v = 0.f;
            v4[j] = v;
        }
        if constexpr (DENSE)
        {
            store4<OUT>(&out[(size_t)y * opitch + x0], v4);
            if (sizeof(OUT) == 4 && out8) store4<uint8_t>(&out8[(size_t)y * opitch + x0], v4);
        }
        else
            *reinterpret_cast<f4*>(&out[(size_t)y * opitch + x0]) = f4{v4[0], v4[1], v4[2], v4[3]};
    }
}

// One launch per level, workgroups of three wavefronts: workgroups [0, nb4) of a strip row do columns [xa, xb), four
// columns per work-item, one channel per wavefront; the remaining workgroups do the other columns one per work-item
// (all channels), each over an eighth of the strip's rows (that part is small, and a work-item's rows are a serial
// chain of load latencies: short chains keep it off the launch's critical path).
constexpr int C4_SUB = 8, C4_THREADS = 3 * WAVE;
template <typename OUT, bool DENSE>
__global__ __launch_bounds__(C4_THREADS, STITCH_C4_WAVES) void k_collapse4(CollapseArgs<OUT, DENSE> A, int nb4, int ncb) {
    const int y0 = blockIdx.y * A.crows, pr = blockIdx.z, y1 = min(y0 + A.crows, A.h);
    if ((int)blockIdx.x < nb4) {
        const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // the channel is wave-uniform: plane bases stay scalar
        const int x0 = A.xa + (blockIdx.x * WAVE + (threadIdx.x & 63)) * 4;
        if (x0 < A.xb) collapse_cols4<OUT, DENSE>(A, c, x0, y0, y1, pr);
    } else {
        const int r = blockIdx.x - nb4, cb = r % ncb, sub = r / ncb, rows = (A.crows + C4_SUB - 1) / C4_SUB;
        int x = cb * C4_THREADS + threadIdx.x;
        if (x >= A.xa) x += A.xb - A.xa;
        const int ya = y0 + sub * rows, yb = min(ya + rows, y1);
        if (x < (DENSE ? A.w : A.pitch) && ya < yb) collapse_cols1<OUT, DENSE>(A, x, ya, yb, pr);
    }
}

// single-level pyramid (max side 2 or 3): the result is the top-level blend itself, cast to the output type
template <typename OUT>
__global__ void k_emit_top(const float* __restrict__ e_all, int w, int h, int pitch, size_t ps, OutPtrs<OUT> outs) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float* e = e_all + (size_t)blockIdx.z * 3 * ps;
    OUT* __restrict__ out = outs.p[blockIdx.z];
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(size_t)c * w * h + (size_t)y * w + x] = px_store<OUT>(e[(size_t)y * pitch + x + c * ps]);
}

// ---- E1-E3 / M1: equalisation and luminance mix --------------------------------------------------------------
__device__ __forceinline__ float clamp256(float v) { return v > 0 ? (v < 256 ? v : 255.f) : 0.f; }
// ---- the colour transforms on BYTE inputs, in integers -------------------------------------------------------------------
// equalization.cpp:78-80 / ImageProcess.cpp:242-244 evaluate   Y  = 0.299 R + 0.857 G + 0.114 B   (0.857 sic)
//                                                              Cb = 128 - 0.168736 R - 0.331264 G + 0.5 B
//                                                              Cr = 128 + 0.5 R - 0.418688 G - 0.081312 B
// in double, store to float and clamp in float.  R, G, B are bytes, so the exact values are t / 1000 and t / 10^6 with integer
// t (ycc_terms); the double evaluation is within 1e-13 of them, and no t / 10^k comes within 7e-9 of a point where the float
// rounding could go either way -- so   (float)((double)t * 1e-k)   IS the reference's float, and its truncation (the
// CImg<unsigned char> store of equalization.cpp:83-85) is the integer quotient t / 10^k (Cb and Cr lie in 0.5 .. 255.5 and all
// their t are multiples of 32, never just below an integer).  Likewise the way back from BYTE Y, Cb, Cr (equalization.cpp:93-98):
// Y + 1.402 (Cr-128) etc. are t / 1000 and t / 10^5, the clamp and the truncation are max(0, min(255, t / 10^k)).
// tests/test_oracle_golden.py::test_integer_colour_transforms_are_exact checks every one of the 2^24 inputs of either direction
// against the double / float expressions; the GPU kernels are compared with the oracle on an image that holds every colour.
// A pixel costs a few integer multiply-adds instead of ~25 double-precision operations (which occupy a SIMD for 8 cycles each).
struct YccTerms {
    unsigned ty, tcb, tcr;  // 1000 Y, 10^6 Cb, 10^6 Cr
};
__device__ __forceinline__ YccTerms ycc_terms(unsigned r, unsigned g, unsigned b) {
    YccTerms t;
    t.ty = 299u * r + 857u * g + 114u * b;
    t.tcb = 128000000u - 168736u * r - 331264u * g + 500000u * b;  // 0.5e6 .. 255.5e6
    t.tcr = 128000000u + 500000u * r - 418688u * g - 81312u * b;
    return t;
}
// the float Y, Cb, Cr the reference holds (ImageProcess.cpp:242-244: not truncated there)
__device__ __forceinline__ void rgb_to_ycc(unsigned r, unsigned g, unsigned b, float& Y, float& Cb, float& Cr) {
    const YccTerms t = ycc_terms(r, g, b);
    Y = clamp256((float)((double)t.ty * 0.001));
    Cb = clamp256((float)((double)t.tcb * 1e-6));
    Cr = clamp256((float)((double)t.tcr * 1e-6));
}
// the bytes equalization.cpp:83-85 stores
__device__ __forceinline__ void rgb_to_ycc_bins(unsigned r, unsigned g, unsigned b, unsigned& yq, unsigned& cbq, unsigned& crq) {
    const YccTerms t = ycc_terms(r, g, b);
    const unsigned q = t.ty / 1000u;
    yq = q < 255u ? q : 255u;
    cbq = t.tcb / 1000000u;
    crq = t.tcr / 1000000u;
}
__device__ __forceinline__ unsigned clamp_quot(int t, int d) { return t <= 0 ? 0u : (t >= 256 * d ? 255u : (unsigned)t / (unsigned)d); }
// equalization.cpp:93-98 on byte Y, Cb, Cr
__device__ __forceinline__ void ycc_bins_to_rgb(unsigned y, unsigned cb, unsigned cr, unsigned& r, unsigned& g, unsigned& b) {
    const int Y = (int)y, cbd = (int)cb - 128, crd = (int)cr - 128;
    r = clamp_quot(1000 * Y + 1402 * crd, 1000);
    g = clamp_quot(100000 * Y - 34414 * cbd - 71414 * crd, 100000);
    b = clamp_quot(1000 * Y + 1772 * cbd, 1000);
}
// equalization.cpp:93-98 / ImageProcess.cpp:262-267 on float inputs (the luminance mix)
__device__ __forceinline__ void ycc_to_rgb_u8(float Y, float Cb, float Cr, uint8_t& r, uint8_t& g, uint8_t& b) {
    const float R = (float)((double)Y + 1.402 * ((double)Cr - 128.0));
    const float G = (float)((double)Y - 0.34414 * ((double)Cb - 128.0) - 0.71414 * ((double)Cr - 128.0));
    const float B = (float)((double)Y + 1.772 * ((double)Cb - 128.0));
    r = (uint8_t)(int)clamp256(R);
    g = (uint8_t)(int)clamp256(G);
    b = (uint8_t)(int)clamp256(B);
}

// The histogram bin of a pixel in integers.  The reference evaluates Y = 0.299 R + 0.857 G + 0.114 B in double, rounds to
// float, clamps and truncates (equalization.cpp:78, :83-85, then `hist[Y]`).  R, G, B are integers 0..255, so the exact sum
// is t / 1000 with t = 299 R + 857 G + 114 B: either an integer or at least 0.001 away from one, while the double and float
// roundings move the value by less than 2e-5 -- the truncation therefore equals floor(t / 1000), capped at 255 by the clamp.
// tests/test_oracle_golden.py::test_integer_luma_bin_is_exact checks all 2^24 colours against the oracle.
__device__ __forceinline__ unsigned luma_bin(unsigned r, unsigned g, unsigned b) {
    const unsigned q = (299u * r + 857u * g + 114u * b) / 1000u;
    return q < 255u ? q : 255u;
}

// Y histogram (equalization.cpp:104-107): each wavefront owns a private 256-bin LDS histogram (no cross-wave
// contention), the workgroup's wavefronts are summed through LDS, and each bin is flushed with one global
// atomic per workgroup.  Four pixels per work-item per step (uchar4 loads when the plane size allows).
constexpr int HIST_WAVES = 4;
__global__ __launch_bounds__(HIST_WAVES * 64) void k_hist(const uint8_t* __restrict__ img, size_t n, int32_t* __restrict__ hist) {
    __shared__ int lh[HIST_WAVES][256];
    const int wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < HIST_WAVES * 256; i += blockDim.x) (&lh[0][0])[i] = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        atomicAdd(&lh[wid][luma_bin(img[i], img[i + n], img[i + 2 * n])], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 256; b += blockDim.x) {
        int s = 0;
#pragma unroll
        for (int k2 = 0; k2 < HIST_WAVES; ++k2) s += lh[k2][b];
        if (s) atomicAdd(&hist[b], s);
    }
}

// CDF and LUT (equalization.cpp:110-124): p_i = hist_i / total for all bins side by side (the divisions are independent),
// then the running sum as 256 sequential double additions on one lane -- its order is the reference's, so the sum is
// bit-identical -- then lut_i = round(255 cdf_i) side by side again; round() is half-away-from-zero.  (All on one lane this took
// 20 us -- a sixth of an equalisation of 25 MPix: every iteration waited for a double-precision divide.)
__global__ __launch_bounds__(256) void k_lut(const int32_t* __restrict__ hist, int w, int h, int32_t* __restrict__ lut) {
    __shared__ double pc[256];
    const double total = (double)(w * h);
    const int i = threadIdx.x;
    pc[i] = (double)hist[i] / total;
    __syncthreads();
    if (i == 0) {
        double cdf = pc[0];
        for (int j = 1; j < 256; ++j) {
            cdf = cdf + pc[j];
            pc[j] = cdf;
        }
    }
    __syncthreads();
    lut[i] = (int32_t)round(255.0 * pc[i]);
}

// apply (equalization.cpp:127-130 + :92-99), in place; FUSE_MIX additionally performs M1 so that the equalised
// copy never exists in memory (stitch_dev_finish_u8).
template <bool FUSE_MIX>
__global__ __launch_bounds__(256) void k_equalize_apply(uint8_t* __restrict__ img, size_t n, const int32_t* __restrict__ lut,
                                                        double num, double den) {
    __shared__ int slut[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) slut[i] = lut[i];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned r = img[i], g = img[i + n], b = img[i + 2 * n];
        unsigned yq, cbq, crq, er, eg, eb;
        rgb_to_ycc_bins(r, g, b, yq, cbq, crq);  // CImg<uchar> store
        ycc_bins_to_rgb((unsigned)slut[yq] & 255u, cbq, crq, er, eg, eb);
        uint8_t o0 = (uint8_t)er, o1 = (uint8_t)eg, o2 = (uint8_t)eb;
        if (FUSE_MIX) {
            float Y, Cb, Cr, Ye, Cbe, Cre;
            rgb_to_ycc(r, g, b, Y, Cb, Cr);
            rgb_to_ycc(er, eg, eb, Ye, Cbe, Cre);
            const float Ym = (float)((double)Y * num / den + (double)Ye / den);  // ImageProcess.cpp:261
            ycc_to_rgb_u8(Ym, Cb, Cr, o0, o1, o2);
        }
        img[i] = o0;
        img[i + n] = o1;
        img[i + 2 * n] = o2;
    }
}

// M1 stand-alone, ImageProcess.cpp:240-268
__global__ __launch_bounds__(256) void k_lummix(uint8_t* __restrict__ res, const uint8_t* __restrict__ eq, size_t n, double num,
                                                double den) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float Y, Cb, Cr, Ye, Cbe, Cre;
        rgb_to_ycc(res[i], res[i + n], res[i + 2 * n], Y, Cb, Cr);
        rgb_to_ycc(eq[i], eq[i + n], eq[i + 2 * n], Ye, Cbe, Cre);
        const float Ym = (float)((double)Y * num / den + (double)Ye / den);
        uint8_t r, g, b;
        ycc_to_rgb_u8(Ym, Cb, Cr, r, g, b);
        res[i] = r;
        res[i + n] = g;
        res[i + 2 * n] = b;
    }
}

// ---- histogram and apply, four pixels per work-item and step (planes of n % 4 == 0 bytes at 4-byte aligned bases) ----
// Every plane is moved as 32-bit words (4 pixels) instead of bytes; the arithmetic per pixel is unchanged (these kernels
// are bound by their double-precision conversions, so the gain is modest: equalise 0.167 -> 0.150 ms, finish 0.234 -> 0.203
// at 6144 x 4096; the luminance mix alone is no faster this way, 0.108 against 0.100, and stays a byte kernel).  The histogram keeps HIST_COPIES copies per wavefront (lane & 7 picks one; rows padded to 257 words so
// that equal bins of different copies fall into different banks): images saturate (the 0.857 coefficient pushes bright
// pixels to Y = 255) and neighbouring pixels share bins, and LDS atomics on one address serialise.
constexpr int HIST_COPIES = 8, HIST_PITCH = 257;
typedef unsigned u4a __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access at 4-byte alignment
__device__ __forceinline__ void unpack4(unsigned w, float v[4]) {
    v[0] = (float)(w & 255u), v[1] = (float)((w >> 8) & 255u), v[2] = (float)((w >> 16) & 255u), v[3] = (float)(w >> 24);
}
__global__ __launch_bounds__(HIST_WAVES * 64) void k_hist4(const uint8_t* __restrict__ img, size_t n, int32_t* __restrict__ hist) {
    __shared__ int lh[HIST_WAVES * HIST_COPIES * HIST_PITCH];
    for (int i = threadIdx.x; i < HIST_WAVES * HIST_COPIES * HIST_PITCH; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    int* mine = lh + ((threadIdx.x >> 6) * HIST_COPIES + (threadIdx.x & (HIST_COPIES - 1))) * HIST_PITCH;
    const unsigned* __restrict__ pr = reinterpret_cast<const unsigned*>(img);
    const unsigned* __restrict__ pg = reinterpret_cast<const unsigned*>(img + n);
    const unsigned* __restrict__ pb = reinterpret_cast<const unsigned*>(img + 2 * n);
    const size_t n4 = n / 4, n16 = n4 / 4, stride = (size_t)gridDim.x * blockDim.x;
    // sixteen pixels per step (dword-aligned 16-byte loads): with 512 workgroups a work-item walks dozens of steps, and a step
    // is load -> wait -> atomics; four words per load keep four times the bytes in flight (29 -> 17 us at 25 MPix)
    const u4a* __restrict__ qr = reinterpret_cast<const u4a*>(pr);
    const u4a* __restrict__ qg = reinterpret_cast<const u4a*>(pg);
    const u4a* __restrict__ qb = reinterpret_cast<const u4a*>(pb);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const u4a r4 = qr[i], g4 = qg[i], b4 = qb[i];
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const unsigned r = r4[k2], g = g4[k2], b = b4[k2];
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(&mine[luma_bin((r >> (8 * j)) & 255u, (g >> (8 * j)) & 255u, (b >> (8 * j)) & 255u)], 1);
        }
    }
    for (size_t i = n16 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {  // up to three words left over
        const unsigned r = pr[i], g = pg[i], b = pb[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&mine[luma_bin((r >> (8 * j)) & 255u, (g >> (8 * j)) & 255u, (b >> (8 * j)) & 255u)], 1);
    }
    __syncthreads();
    for (int bin = threadIdx.x; bin < 256; bin += blockDim.x) {
        int s_ = 0;
#pragma unroll
        for (int k2 = 0; k2 < HIST_WAVES * HIST_COPIES; ++k2) s_ += lh[k2 * HIST_PITCH + bin];
        if (s_) atomicAdd(&hist[bin], s_);
    }
}

template <bool FUSE_MIX>
__global__ __launch_bounds__(256) void k_equalize_apply4(uint8_t* __restrict__ img, size_t n, const int32_t* __restrict__ lut, double num, double den) {
    __shared__ int slut[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) slut[i] = lut[i];
    __syncthreads();
    unsigned* __restrict__ pr = reinterpret_cast<unsigned*>(img);
    unsigned* __restrict__ pg = reinterpret_cast<unsigned*>(img + n);
    unsigned* __restrict__ pb = reinterpret_cast<unsigned*>(img + 2 * n);
    const size_t n4 = n / 4, n16 = n4 / 4, stride = (size_t)gridDim.x * blockDim.x;
    auto word = [&](unsigned wr, unsigned wg, unsigned wb, unsigned& o_r, unsigned& o_g, unsigned& o_b) {
        o_r = o_g = o_b = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned r = (wr >> (8 * j)) & 255u, g = (wg >> (8 * j)) & 255u, b = (wb >> (8 * j)) & 255u;
            unsigned yq, cbq, crq, er, eg, eb;
            rgb_to_ycc_bins(r, g, b, yq, cbq, crq);  // CImg<uchar> store
            ycc_bins_to_rgb((unsigned)slut[yq] & 255u, cbq, crq, er, eg, eb);
            if (FUSE_MIX) {
                float Y, Cb, Cr, Ye, Cbe, Cre;
                rgb_to_ycc(r, g, b, Y, Cb, Cr);
                rgb_to_ycc(er, eg, eb, Ye, Cbe, Cre);
                const float Ym = (float)((double)Y * num / den + (double)Ye / den);  // ImageProcess.cpp:261
                uint8_t m0, m1, m2;
                ycc_to_rgb_u8(Ym, Cb, Cr, m0, m1, m2);
                er = m0, eg = m1, eb = m2;
            }
            o_r |= er << (8 * j);
            o_g |= eg << (8 * j);
            o_b |= eb << (8 * j);
        }
    };
    // sixteen pixels per step (dword-aligned 16-byte accesses), then the up to three words left over
    u4a* __restrict__ qr = reinterpret_cast<u4a*>(pr);
    u4a* __restrict__ qg = reinterpret_cast<u4a*>(pg);
    u4a* __restrict__ qb = reinterpret_cast<u4a*>(pb);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const u4a r4 = qr[i], g4 = qg[i], b4 = qb[i];
        u4a o_r, o_g, o_b;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            unsigned a, b, c;
            word(r4[k2], g4[k2], b4[k2], a, b, c);
            o_r[k2] = a, o_g[k2] = b, o_b[k2] = c;
        }
        qr[i] = o_r;
        qg[i] = o_g;
        qb[i] = o_b;
    }
    for (size_t i = n16 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        unsigned a, b, c;
        word(pr[i], pg[i], pb[i], a, b, c);
        pr[i] = a;
        pg[i] = b;
        pb[i] = c;
    }
}

// ---- synthetic frames and the CImg<uchar>(CImg<float>) cast ---------------------------------
// v = 1 + ((3x + 5y + 37c + 101f) mod 200) + (splitmix64(seed ^ key) mod 50), never 0; the float twin adds
// frac = ((hash >> 32) & 0xFFFF) / 65536.  Deterministic, so every rank and the CPU baseline see the same frames.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
template <typename PX>
__global__ __launch_bounds__(256) void k_synth(PX* __restrict__ dst, int w, int h, int f) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, c = blockIdx.z;
    if (x >= w) return;
    const uint64_t key = ((uint64_t)f << 40) | ((uint64_t)c << 36) | ((uint64_t)y << 18) | (uint64_t)x;
    const uint64_t hsh = splitmix64(0x5717C4EDULL ^ key);
    const int v = 1 + (int)((3LL * x + 5LL * y + 37LL * c + 101LL * f) % 200) + (int)(hsh % 50);
    float out = (float)v;
    if (sizeof(PX) == 4) out += (float)((hsh >> 32) & 0xFFFF) / 65536.0f;
    dst[(size_t)c * w * h + (size_t)y * w + x] = px_store<PX>(out);
}

// float -> unsigned char by C-cast truncation: what `return expand;` does at ImageProcess.cpp:772 through
// CImg<unsigned char>(const CImg<float>&) (CImg.h:11167-11182).  Values are in [0,255] after the collapse clamp.
// Zeroes `n` 64-bit words (hand-off granules, queue heads, abort flag).  A kernel rather than hipMemsetAsync so that a
// captured HIP graph orders it like every other node of the sequence.
__global__ __launch_bounds__(256) void k_fill_bytes(uint8_t* __restrict__ p, size_t n, uint8_t v) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void k_clear_words(u64* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}

__global__ __launch_bounds__(256) void k_quantize(const float* __restrict__ src, uint8_t* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n && ((reinterpret_cast<uintptr_t>(src + i) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst + i) & 3) == 0)) {
            const f4 v = *reinterpret_cast<const f4*>(src + i);
            uchar4 o;
            o.x = (uint8_t)(int)v.x;
            o.y = (uint8_t)(int)v.y;
            o.z = (uint8_t)(int)v.z;
            o.w = (uint8_t)(int)v.w;
            *reinterpret_cast<uchar4*>(dst + i) = o;
        } else {
            for (size_t j = i; j < n && j < i + 4; ++j) dst[j] = (uint8_t)(int)src[j];
        }
    }
}

}  // namespace sk
