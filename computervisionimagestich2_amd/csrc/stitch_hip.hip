// libstitch_hip.so -- host side of the C ABI in include/stitch.h: argument checks, device workspace (plans),
// launch sequencing on HIP streams, per-stage event timing.  All arithmetic is in stitch_kernels.hpp (and the k_*.inc parts it includes).
// Built for gfx950 only (hipcc --offload-arch=gfx950 -ffp-contract=off); there is no CPU path in this library.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <list>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "stitch.h"
#include "stitch_kernels.hpp"

#pragma clang fp contract(off)

using namespace sk;

namespace {

thread_local std::string g_err;

// The HIP runtime multiplexes streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and launch sequences on
// streams that share a queue serialise behind each other (four sequences in flight: 1.50 ms per pair on 4 queues, 1.31 on 8).
// The variable is read when the runtime initialises, so it has to be in the environment before the process's first HIP call:
// stitch_init() (below, C ABI) puts it there -- called by whoever owns the process start-up (the C++ adaptor's static
// initialiser, the Python package's import, bench.py), never from a library constructor: setenv() races with getenv() in a host
// that already runs threads, and a dlopen() after HIP is up would change nothing anyway.

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return fail(STITCH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                                   \
    } while (0)

int need_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(STITCH_ERR_NO_DEVICE, "no HIP device visible (hipGetDeviceCount: %s); this library has no CPU path",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    }
    return STITCH_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// Row pitch of a level's planes in floats: the width rounded up to 64, plus -- for rows of 16 KB and more -- `pad` floats
// (STITCH_PITCH_PAD, rounded up to 64) so that the 64 rows of a tile do not start a power-of-two-ish number of bytes apart.
inline int level_pitch(int w, int pad) { return round_up(w, 64) + (pad > 0 && w >= 4096 ? round_up(pad, 64) : 0); }
inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
constexpr double kPI = 3.14159265358979323846;  // cimg::PI

// Projection.cpp:24-28: flag, width/height swap, tanVal rounded to float, r = (width/2.0)/tanVal stored to float
struct ProjParams {
    int flag, width, height;
    float r;
};
ProjParams proj_params(int w, int h, float fov_deg) {
    ProjParams p;
    p.flag = w > h;
    p.width = p.flag ? h : w;
    p.height = p.flag ? w : h;
    const float tanVal = (float)std::tan((double)fov_deg * kPI / 180.0);
    p.r = (float)((p.width / 2.0) / tanVal);
    return p;
}

// CImg.h:35053-35065 + :34889-34903
VVK make_vvk(float sigma) {
    const float nsigma = sigma >= 0 ? sigma : -sigma;
    const double nnsigma = nsigma < 0.5f ? 0.5f : nsigma, m0 = 1.16680, m1 = 1.10783, m2 = 1.40586, m1sq = m1 * m1,
                 m2sq = m2 * m2,
                 q = (nnsigma < 3.556 ? -0.2568 + 0.5784 * nnsigma + 0.0561 * nnsigma * nnsigma
                                      : 2.5091 + 0.9804 * (nnsigma - 3.556)),
                 qsq = q * q, scale = (m0 + q) * (m1sq + m2sq + 2 * m1 * q + qsq),
                 b1 = -q * (2 * m0 * m1 + m1sq + m2sq + (2 * m0 + 4 * m1) * q + 3 * qsq) / scale,
                 b2 = qsq * (m0 + 2 * m1 + 3 * q) / scale, b3 = -qsq * q / scale, B = (m0 * (m1sq + m2sq)) / scale;
    const double sumsq = B, sum = sumsq * sumsq, a1 = -b1, a2 = -b2, a3 = -b3,
                 scaleM = 1.0 / ((1.0 + a1 - a2 + a3) * (1.0 - a1 - a2 - a3) * (1.0 + a2 + (a1 - a3) * a3));
    VVK k;
    k.f1 = a1;
    k.f2 = a2;
    k.f3 = a3;
    k.sumsq = sumsq;
    k.sum = sum;
    k.den = 1.0 - a1 - a2 - a3;
    k.M[0] = scaleM * (-a3 * a1 + 1.0 - a3 * a3 - a2);
    k.M[1] = scaleM * (a3 + a1) * (a2 + a3 * a1);
    k.M[2] = scaleM * a3 * (a1 + a3 * a2);
    k.M[3] = scaleM * (a1 + a3 * a2);
    k.M[4] = -scaleM * (a2 - 1.0) * (a2 + a3 * a1);
    k.M[5] = -scaleM * a3 * (a3 * a1 + a3 * a3 + a2 - 1.0);
    k.M[6] = scaleM * (a3 * a1 + a2 + a1 * a1 - a2 * a2);
    k.M[7] = scaleM * (a1 * a2 + a3 * a2 * a2 - a1 * a3 * a3 - a3 * a3 * a3 - a3 * a2 + a3);
    k.M[8] = scaleM * a3 * (a1 + a3 * a2);
    return k;
}

// CImg.h:34801-34816,34840-34841 (float throughout; std::exp(float) is the float overload)
DRK make_drk(float sigma) {
    const float nsigma = sigma >= 0 ? sigma : -sigma;
    const float nnsigma = nsigma < 0.1f ? 0.1f : nsigma, alpha = 1.695f / nnsigma, ema = std::exp(-alpha),
                ema2 = std::exp(-2 * alpha), b1 = -2 * ema, b2 = ema2;
    const float kk = (1 - ema) * (1 - ema) / (1 + 2 * alpha * ema - ema2);
    DRK k;
    k.a0 = kk;
    k.a1 = kk * (alpha - 1) * ema;
    k.a2 = kk * (alpha + 1) * ema;
    k.a3 = -kk * ema2;
    k.b1 = b1;
    k.b2 = b2;
    k.coefp = (k.a0 + k.a1) / (1 + b1 + b2);
    k.coefn = (k.a2 + k.a3) / (1 + b1 + b2);
    return k;
}

// CImg.h:29625-29637: idx/alpha of the linear up-sampling walk
void expand_table(int n_src, int n_dst, std::vector<int32_t>& idx, std::vector<double>& alpha) {
    idx.resize(n_dst);
    alpha.resize(n_dst);
    if (n_src == 1) {
        for (int x = 0; x < n_dst; ++x) idx[x] = 0, alpha[x] = 0;
        return;
    }
    const double fx = n_dst > 1 ? (n_src - 1.0) / (n_dst - 1) : 0;
    double curr = 0;
    for (int x = 0; x < n_dst; ++x) {
        idx[x] = (int32_t)(unsigned int)curr;
        alpha[x] = curr - (unsigned int)curr;
        const double nxt = curr + fx;
        curr = (n_src - 1.0) < nxt ? (n_src - 1.0) : nxt;
    }
}

// The longest run of 256-column blocks in which every group of four columns x0.. has the resize taps s, s+1, s+1, s+2 with
// s+3 inside the source row (see k_collapse4); [*xa, *xb) empty when there is none or the source level is a single row/column.
void regular_range(const std::vector<int32_t>& idx, int w, int sw, int sh, int* xa, int* xb) {
    *xa = *xb = 0;
    int best_a = 0, best_b = 0, run_a = -1;
    const int nblk = w / 256;
    for (int b = 0; b <= nblk; ++b) {
        bool reg = b < nblk && sw >= 2 && sh >= 2;
        for (int x0 = b * 256; reg && x0 < (b + 1) * 256; x0 += 4) {
            const int s_ = idx[x0];
            reg = idx[x0 + 1] == s_ + 1 && idx[x0 + 2] == s_ + 1 && idx[x0 + 3] == s_ + 2 && s_ + 3 <= sw - 1;
        }
        if (reg && run_a < 0) run_a = b;
        if (!reg && run_a >= 0) {
            if (b - run_a > best_b - best_a) best_a = run_a, best_b = b;
            run_a = -1;
        }
    }
    if (best_b > best_a) *xa = best_a * 256, *xb = best_b * 256;
}

// The columns [0, *xb) (a multiple of 256) in which every group of four columns x0.. has its taps inside ONE 16-byte load at
// ix[x0]: ix[x0 + 3] + 1 <= ix[x0] + 3 (always true for a step below 1/2; see k_collapse4, GEN).  The window may reach past the
// source row's last sample: the last group of an even width has the taps s, s+1, s+1, s+1 or s, s+1, s+1, s+2 with the row ending
// at s + 2, so element 3 of its window is the first sample of the next row (of the next plane, of whatever follows the planes in
// the arena) -- loaded, selected by no tap: a column whose first tap IS the row's last sample takes that sample twice, as the
// reference does (CImg.h:29648), whatever the window holds behind it.
void general_range(const std::vector<int32_t>& idx, int w, int sw, int sh, int* xb) {
    *xb = 0;
    if (sw < 4 || sh < 2) return;
    int x0 = 0;
    for (; x0 + 3 < w; x0 += 4) {
        const int s_ = idx[x0];
        if (!(idx[x0 + 1] >= s_ && idx[x0 + 2] >= idx[x0 + 1] && idx[x0 + 3] >= idx[x0 + 2] && idx[x0 + 3] - s_ <= 2 && s_ >= 0 && s_ + 1 <= sw - 1)) break;
    }
    *xb = x0 / 256 * 256;
}

int pyramid_levels(int w, int h, int level_rule, int* lw, int* lh) {
    if (w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "pyramid: non-positive size %dx%d", w, h);
    const int len = level_rule ? (w < h ? w : h) : (w >= h ? w : h);
    int levels = 0;
    while ((len >> (levels + 1)) > 0) ++levels;  // floor(log2(len)), ImageProcess.cpp:675-676
    if (levels < 1 || levels > 32) return fail(STITCH_ERR_PYRAMID, "pyramid: %dx%d gives %d levels", w, h, levels);
    int cw = w, ch = h;
    for (int i = 0; i < levels; ++i) {
        if (cw <= 0 || ch <= 0)
            return fail(STITCH_ERR_PYRAMID, "pyramid: level %d of %dx%d has a zero dimension (the reference degenerates here)",
                        i, w, h);
        if (lw) lw[i] = cw;
        if (lh) lh[i] = ch;
        cw /= 2;  // ImageProcess.cpp:706-707
        ch /= 2;
    }
    return levels;
}

// Tuning / A-B switches of a plan (listed in include/stitch.h; none changes a result bit): ONE snapshot of the environment,
// taken when a plan is created and again whenever a host-pointer entry point looks for an idle plan in the cache.  The
// snapshot is part of the cache key (PlanKey), so a caller who flips a switch between two calls gets a workspace built for
// the new setting, never a stale one.  All fields are ints (no padding: compared with memcmp); -1 = not set.
struct Tuning {
    int wavefront, no_fuse, no_src_fuse, no_zero_tiles, crows_l0, crows_ln, collapse4, xbyf_wgs, xbyf_spin_limit, xbyf_early, y2,
        recompute, stamp, gate64, coarse, single_fast, odd_dec, c4_gen, collapse_px, y1s, c4_lock, c4_swz, mover, src_lone, dec7, xbym, xbym_mpix, coarse_lds, pitch_pad, crows_wgs;
    static int env_int(const char* name) {
        const char* e = std::getenv(name);
        return e ? std::max(0, atoi(e)) : -1;
    }
    static Tuning from_env() {
        Tuning t{};
        t.wavefront = env_int("STITCH_WAVEFRONT");
        t.no_fuse = std::getenv("STITCH_NO_FUSE") != nullptr;
        t.no_src_fuse = env_int("STITCH_NO_SRC_FUSE") > 0;
        t.no_zero_tiles = env_int("STITCH_NO_ZERO_TILES") > 0;
        t.crows_l0 = env_int("STITCH_CROWS_L0");
        t.crows_ln = env_int("STITCH_CROWS_LN");
        t.collapse4 = env_int("STITCH_COLLAPSE4");
        t.xbyf_wgs = env_int("STITCH_XBYF_WGS");
        t.xbyf_spin_limit = env_int("STITCH_XBYF_SPIN_LIMIT");
        t.xbyf_early = env_int("STITCH_XBYF_EARLY");
        t.y2 = std::getenv("STITCH_Y2") != nullptr;
        t.recompute = env_int("STITCH_RECOMPUTE");
        t.stamp = std::getenv("STITCH_WAVEFRONT_STAMP") != nullptr;
        t.gate64 = env_int("STITCH_GATE64") > 0;
        t.coarse = env_int("STITCH_COARSE");
        t.single_fast = env_int("STITCH_SINGLE_FAST");
        t.odd_dec = env_int("STITCH_ODD_DEC");
        t.c4_gen = env_int("STITCH_C4_GEN");
        t.collapse_px = env_int("STITCH_COLLAPSE_PX");
        t.y1s = env_int("STITCH_Y1S");
        t.c4_lock = env_int("STITCH_C4_LOCKSTEP");
        t.c4_swz = env_int("STITCH_C4_SWIZZLE");
        t.mover = env_int("STITCH_MOVER");
        t.src_lone = env_int("STITCH_SRC_LONE_MPIX");
        t.dec7 = env_int("STITCH_DEC7");
        t.xbym = env_int("STITCH_XBYM");
        t.xbym_mpix = env_int("STITCH_XBYM_MPIX");
        t.coarse_lds = env_int("STITCH_COARSE_LDS");
        t.pitch_pad = env_int("STITCH_PITCH_PAD");
        t.crows_wgs = env_int("STITCH_CROWS_WGS");
        return t;
    }
    bool operator==(const Tuning& o) const { return std::memcmp(this, &o, sizeof o) == 0; }
};

struct Level {
    int w, h, pitch;
    size_t ps;       // plane stride in floats = pitch*h
    float* g;        // 7 planes [a0 a1 a2 b0 b1 b2 m] + 64 slack rows
    float* e;        // 3 planes (collapse chain), levels >= 1
    int32_t *ix, *iy;  // expand tables level+1 -> level (levels < L-1)
    double *ax, *ay;
    int c4_xa, c4_xb;  // columns [c4_xa, c4_xb) of this level go to k_collapse4 (multiples of 256; empty when equal)
    int c4_gen;        // != 0: [0, c4_xb) with per-lane tap offsets (k_collapse4 GEN) -- chosen where it covers more than the fixed pattern
};

constexpr int CO_LDS_BYTES = 144 * 1024;  // dynamic LDS k_coarse_lds may take (160 KB per CU, 15 KB of tables beside it)
constexpr int WF_CTRL_WORDS = 4 * 16 + 16;  // band-queue heads of up to 4 levels (64 bytes apart) + the abort word
constexpr int WF_STICKY_WORDS = 16;         // behind them, outside what a launch sequence clears: the plan's count of timed-out waits

struct ProfRec {
    int stage, level;
    hipEvent_t a, b;
};

}  // namespace

struct stitch_plan {
    int device = 0;
    int cw = 0, ch = 0, L = 0;
    int cap = 1;     // pairs per launch sequence this workspace can hold (planes of pair b follow pair b-1)
    int last_n = 0;  // pairs of the last call
    stitch_blend_opts opts{};
    Tuning tune{};  // the environment's switches as they were when the plan was created
    Level lv[32]{};
    void* arena = nullptr;
    size_t arena_bytes = 0;
    float* T = nullptr;   // blur scratch, 7 planes of level 0 + slack
    float* T2 = nullptr;  // Deriche temporaries (blur_kind 1 only)
    double* state = nullptr;
    // fused anticausal-x + causal-y wavefront sweep (k_vv_xbyf) for the first `wf_levels` levels
    int wf_levels = 0;
    u64* wf_yg = nullptr;
    size_t wf_yg_bytes = 0;
    unsigned* wf_ctrl = nullptr;   // per level one band-queue head (16 words apart), then the abort flag
    uint8_t* zi = nullptr;         // [cap][tiles][bands] of level 0: "every pixel of the tile lies outside the frame" (k_src_index)
    uint8_t* zt = nullptr;         // zero-tile flags of T, [7*cap][bands][tiles] of level 0 (ZeroTiles); reused level by level
    bool zero_tiles = false;
    const float* zero_page = nullptr;  // 4 KB of zeros in the arena (k_vv_xby_m's loader reads a flagged tile from here)
    int wf_max_wgs = 2304;  // persistent workgroups of the fused sweep (STITCH_XBYF_WGS)
    unsigned wf_spin_limit = 1u << 23;  // polls before a hand-off wait gives up, about 20 s (STITCH_XBYF_SPIN_LIMIT); 2^20 (2-3 s) was reached once in a warm-up
    int wf_early_read = 1;  // STITCH_XBYF_EARLY=0: poll for the hand-off only when it is needed
    // The causal x sweep of a wavefront level keeps only its state in front of every tile and the fused sweep re-runs it
    // tile by tile (k_vv_x_fwd<.., CKPT>, k_vv_xbyf MODE 1/2): the x-swept level is neither written nor read back.
    // STITCH_RECOMPUTE=0 keeps the two-pass form.
    int recompute = 0;  // 0 off, 1 every wavefront level, 2 the wavefront levels >= 1 only
    double* ckpt = nullptr;  // [3][tiles of level 0][lines of level 0]
    unsigned long long* d7_dbg = nullptr;  // STITCH_D7_STAMP=<level>: k_vv_y_bwd_dec7's per-chunk stamps of one workgroup at that level (diagnostics)
    int d7_dbg_level = -1, d7_dbg_chunks = 0;
    unsigned long long* xy_dbg = nullptr;  // STITCH_XBYM_STAMP=1: k_vv_xby_m's per-tile stamps of one band of the last level-0 launch (diagnostics)
    int xy_dbg_nc = 0;
    unsigned long long* wf_dbg = nullptr;  // STITCH_WAVEFRONT_STAMP=1: [wf_max_wgs][8] segment cycle sums (diagnostics)
    // pinned copy of the plan's sticky count of timed-out hand-off waits, refreshed at the end of every call.  The device
    // word only grows (no launch sequence clears it), so the last copy covers every earlier queued call as well.
    unsigned* h_wf_abort = nullptr;
    unsigned wf_abort_seen = 0;  // count already acknowledged by stitch_plan_clear_fault
    float* side = nullptr;  // [cap][pitch0] x-blurred level-0 mask rows (implicit level-0 mask)
    bool mask_opt = false;  // level-0 mask handled implicitly (Van Vliet, level-0 height a multiple of 64)
    // rows per work-item strip of the collapse (a strip re-uses its x-interpolated source rows from output row to output
    // row, and its rows are one serial chain of load latencies): 32 at the large levels (fewer re-reads of level l+1), h/32
    // but at least 4 at the small ones, whose launches are nothing but that chain.  STITCH_CROWS_L0 / STITCH_CROWS_LN override.
    int crows_ln = 0;  // 0 = per level, crows_of()
    int crows_l0 = 4 * CROWS;
    bool collapse4 = true;  // four columns per work-item in the collapse where possible (STITCH_COLLAPSE4=0: always k_collapse)
    unsigned long long* co_dbg = nullptr;  // STITCH_COARSE_STAMP=1: k_coarse's phase stamps of the last launch (diagnostics)
    int coarse_from = 0;  // > 0: levels coarse_from .. L-1 run in ONE launch (k_coarse: REDUCE to the top, top blend, collapse back up)
    bool src_fuse = false;  // pairs: level-0 planes evaluated from the frames by their consumers, k_compose never runs
    SeamDev* d_seam = nullptr;
    SeamDev* h_seam = nullptr;  // pinned
    hipStream_t last_stream = nullptr;
    bool pending = false;
    VVK vvk{};
    DRK drk{};
    bool blur_skip = false;
    bool no_fuse = false;  // STITCH_NO_FUSE=1: keep blur and decimation as separate kernels (A/B and tests)
    bool planes_in = false;  // level 0 (a, b AND the mask plane) is handed in by the caller (the coarse levels of a band-split pair): no seam scan, no implicit mask
    bool profiling = false;
    int prof_only = -1;  // >= 0: record events only around launches of this kernel id
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> free_events;
};

namespace {

struct StageTimer {  // records a pair of events around a group of launches when profiling is on
    stitch_plan* p;
    hipStream_t s;
    ProfRec r{};
    bool on;
    StageTimer(stitch_plan* plan, hipStream_t st, int stage, int level)
        : p(plan), s(st), on(plan->profiling && (plan->prof_only < 0 || plan->prof_only == stage)) {
        if (!on) return;
        r.stage = stage;
        r.level = level;
        auto get = [&](hipEvent_t& e) {
            if (!p->free_events.empty()) {
                e = p->free_events.back();
                p->free_events.pop_back();
            } else
                (void)hipEventCreate(&e);
        };
        get(r.a);
        get(r.b);
        (void)hipEventRecord(r.a, s);
    }
    ~StageTimer() {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        p->recs.push_back(r);
    }
};

dim3 grid_xy(int pitch_or_w, int h, int z = 1) { return dim3((pitch_or_w + 255) / 256, h, z); }

int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(STITCH_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return STITCH_OK;
}

// The fused anticausal-x + causal-y sweep of a call with n pairs (see run_reduce): a per-CALL choice, like src_fused_call.
// (Until round 4 one pair of >= 800 bands -- config 5 -- took the batch's sweep too; the five-wavefront form with zero-tile flags is 2.6 %
// faster there, 22.3 against 22.9 ms per pair.  STITCH_XBYM=0 brings the old choice back.)
bool fused_sweep_call(const stitch_plan* p, int n) {
    return p->tune.wavefront >= 0 || p->tune.single_fast > 0 || n >= 2 || (p->tune.xbym == 0 && 7L * ((p->lv[0].h + TS - 1) / TS) >= 800);
}

// One pair in flight: the anticausal x and the causal y sweep of level l as ONE launch of five-wavefront bands (k_vv_xby_m) where the
// separate launches are bound by their bytes -- from STITCH_XBYM_MPIX megapixels per plane (default 20: 6144 x 4096).  Below that a level's time is
// its chain of rows and columns either way, and the fused form adds a hand-off per 64-row band.  STITCH_XBYM=0: never; =1: the first
// four levels of every lone pair, whatever their size (tests).
bool lone_fused_level(const stitch_plan* p, int n, int l) {
    const Level& a = p->lv[l];
    const long mpix = p->tune.xbym > 0 ? 0 : p->tune.xbym_mpix >= 0 ? p->tune.xbym_mpix : 20;
    return n == 1 && p->wf_ctrl && l < 4 && p->opts.blur_kind == 0 && !p->blur_skip && p->tune.mover != 0 && p->tune.xbym != 0 && a.w > 1 && a.h > 1 &&
           (long)a.w * a.h >= mpix * 1000000L && (a.h + TS - 1) / TS < 4095 && !fused_sweep_call(p, n);
}

// REDUCE for every level (ImageProcess.cpp:705-715): blur(G_l) into T, decimate T into G_{l+1}.
template <typename PX>
int run_reduce(stitch_plan* p, int n, hipStream_t s, const PairArgs<PX>& pa, bool src, const ZeroTiles& zi) {
    const int np = 7 * n;  // planes in flight: every launch covers all pairs of the batch
    const bool any_lone_fused = lone_fused_level(p, n, 0);  // levels shrink: level 0 qualifies whenever any level does
    if (p->wf_levels > 0 || any_lone_fused) k_clear_words<<<1, 256, 0, s>>>((u64*)p->wf_ctrl, WF_CTRL_WORDS / 2);  // band-queue heads + abort flag
    const int l_end = p->coarse_from > 0 ? p->coarse_from : p->L - 1;  // levels [0, l_end) are reduced level by level
    for (int l = 0; l < l_end; ++l) {
        const Level& a = p->lv[l];
        const Level& b = p->lv[l + 1];
        const long lines = (long)np * a.h;
        const bool do_x = a.w > 1 && !p->blur_skip, do_y = a.h > 1 && !p->blur_skip;
        bool decimated = false;
        MaskL0 mk{};
        if (l == 0 && p->mask_opt) {
            mk.seam = p->d_seam;
            mk.side = p->side;
            mk.h = a.h;
            mk.enabled = 1;
        }
        // The band pipeline pays where a launch has many bands in flight to hide its fill (see stitch_plan_create_batched): decided
        // per CALL -- a batched plan asked for one pair runs the separate sweeps, which are faster for a lone pair (3.0 against
        // 3.4 ms at 6144 x 4096) -- unless STITCH_WAVEFRONT or STITCH_SINGLE_FAST pins the form.
        const bool wavefront = p->opts.blur_kind == 0 && do_x && do_y && l < p->wf_levels && fused_sweep_call(p, n);
        // odd widths decimate inside the anticausal sweep too (three overlaps per output column); a width of 1 has no next level column
        const bool odd_dec = (a.w & 1) != 0 && a.w >= 3 && !p->no_fuse && p->tune.odd_dec != 0;
        // zero-tile flags: only where all three users run (causal x sweep, fused sweep, fused anticausal-y + decimation)
        const int NR = (a.h + TS - 1) / TS;  // 64-row bands per plane, the last one possibly partial
        ZeroTiles zt{};
        // (the five-wavefront sweep of ONE pair carries the flags too where its x sweeps are the one-wavefront kernels anyway -- more bands
        // than the chain + loader + storer form takes: a pair of config 5's size)
        const int nbx_bands = np * NR;
        const bool xby_zt = !wavefront && p->zt && do_x && do_y && lone_fused_level(p, n, l) && (a.w + TS - 1) / TS <= XY_MAXNC &&
                            !(p->tune.mover != 0 && (nbx_bands <= 340 || (src && l == 0 && nbx_bands <= 1024)));
        if ((wavefront || xby_zt) && p->zero_tiles && NR <= 256 && ((a.w & 1) == 0 || odd_dec) && !p->no_fuse && !(p->tune.gate64 && (a.h % TS || (a.w & 1)))) {
            zt.flags = p->zt;
            zt.h = a.h;
            zt.NR = NR;
            zt.NC = (a.w + TS - 1) / TS;
        }
        // blocks of the x sweeps: per-plane bands wherever a block's treatment depends on its plane (implicit mask, source-fused
        // level 0, zero-tile flags), 64 stacked lines otherwise (fewer blocks when planes are shorter than 64 rows)
        Bands bd{};
        if (mk.enabled || (src && l == 0) || zt.flags) bd = Bands{NR, a.h};
        const int nbx = bd.nr ? np * bd.nr : (int)((lines + TS - 1) / TS);
        const bool rowz = zt.flags && (a.h % YCH) != 0;  // fused anticausal-y + decimation: a chunk of rows may straddle bands
        if (wavefront) {
            const int nb = nbx;
            const bool rcmp = (p->recompute == 1 || (p->recompute == 2 && l >= 1)) && !(p->wf_dbg && l == 0);
            {
                StageTimer t(p, s, src && l == 0 ? STITCH_K_VV_X_FWD_SRC : STITCH_K_VV_X_FWD, l);
                if (src && l == 0) {
                    if (rcmp)
                        k_vv_x_fwd<PX, true, true><<<nb, 64, 0, s>>>(a.g, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, pa, zt, zi, p->ckpt, bd);
                    else
                        k_vv_x_fwd<PX, true><<<nb, 64, 0, s>>>(a.g, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, pa, zt, zi, nullptr, bd);
                } else if (rcmp)
                    k_vv_x_fwd<PX, false, true><<<nb, 64, 0, s>>>(a.g, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, NoPairArgs{}, zt, ZeroTiles{}, p->ckpt, bd);
                else
                    k_vv_x_fwd<PX, false><<<nb, 64, 0, s>>>(a.g, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, NoPairArgs{}, zt, ZeroTiles{}, nullptr, bd);
            }
            Wavefront wf{};
            wf.yg = p->wf_yg;
            wf.counter = p->wf_ctrl + (size_t)l * 16;
            wf.abort = p->wf_ctrl + WF_CTRL_WORDS - 16;
            wf.sticky = p->wf_ctrl + WF_CTRL_WORDS;
            wf.spin_limit = p->wf_spin_limit;
            wf.NR = NR;
            wf.NC = (a.w + TS - 1) / TS;
            wf.NP = np;
            // every polled word is cleared in front of the launch (never told apart by a per-launch argument: a captured
            // graph replays frozen arguments, and stale tags from the previous replay would match at once)
            const size_t gran_words = (size_t)wf.NP * wf.NC * WF_GRAN * WAVE;
            k_clear_words<<<(int)std::min<size_t>((gran_words + 255) / 256, 2048), 256, 0, s>>>(p->wf_yg, gran_words);
            wf.epoch = 1;
            wf.mask_l0 = mk.enabled;
            wf.zt = zt;
            wf.early_read = p->wf_early_read;
            const long ntiles = (long)wf.NP * wf.NR;  // one persistent wavefront per row band
            // the x-sweep state lives in state[0 .. 4*lines); the y state the kernel leaves goes behind it
            double* state_y = p->state + 4 * (size_t)p->cap * 7 * (a.h + 64);
            {
                StageTimer t(p, s, STITCH_K_VV_XBYF, l);
                const int wg = (int)std::min<long>(ntiles, p->wf_max_wgs);  // at most 9 workgroups per CU by LDS
                Recompute rcv{};
                if (rcmp) {
                    rcv.in = a.g;
                    rcv.ckpt = p->ckpt;
                    rcv.seam = p->d_seam;
                    if (src && l == 0) rcv.zi = zi;
                }
                if (p->wf_dbg && l == 0) {  // diagnostic build: per-segment cycle sums of the level-0 launch
                    wf.dbg = p->wf_dbg;
                    k_vv_xbyf<true><<<wg, 64, 0, s>>>(p->T, a.w, a.h, a.pitch, p->vvk, p->state, lines, state_y, wf, rcv, NoPairArgs{});
                } else if (rcmp && src && l == 0)
                    k_vv_xbyf<false, PX, 2><<<wg, 64, 0, s>>>(p->T, a.w, a.h, a.pitch, p->vvk, p->state, lines, state_y, wf, rcv, pa);
                else if (rcmp)
                    k_vv_xbyf<false, float, 1><<<wg, 64, 0, s>>>(p->T, a.w, a.h, a.pitch, p->vvk, p->state, lines, state_y, wf, rcv, NoPairArgs{});
                else
                    k_vv_xbyf<false><<<wg, 64, 0, s>>>(p->T, a.w, a.h, a.pitch, p->vvk, p->state, lines, state_y, wf, rcv, NoPairArgs{});
            }
            StageTimer t(p, s, STITCH_K_VV_Y_BWD, l);
            dim3 g((a.pitch + YCOLS - 1) / YCOLS, np);
            if ((a.w & 1) == 0 && !p->no_fuse) {
                if (rowz)
                    k_vv_y_bwd_dec<true><<<g, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, state_y, b.g, b.w, b.h, b.pitch, b.ps, zt, nullptr, nullptr, a.h, b.h);
                else
                    k_vv_y_bwd_dec<false><<<g, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, state_y, b.g, b.w, b.h, b.pitch, b.ps, zt, nullptr, nullptr, a.h, b.h);
                decimated = true;
            } else if (odd_dec) {  // odd width: three-tap x decimation, workgroups overlap by two columns
                const dim3 go((b.w + WAVE - 2) / (WAVE - 1), np);
                if (rowz)
                    k_vv_y_bwd_dec<true, YST, true, true><<<go, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, state_y, b.g, b.w, b.h, b.pitch, b.ps, zt, nullptr, nullptr, a.h, b.h);
                else
                    k_vv_y_bwd_dec<false, YST, true, true><<<go, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, state_y, b.g, b.w, b.h, b.pitch, b.ps, zt, nullptr, nullptr, a.h, b.h);
                decimated = true;
            } else
                k_vv_y_bwd<<<g, 64, 0, s>>>(p->T, a.h, a.pitch, a.ps, p->vvk, state_y, nullptr, nullptr);
        } else if (p->opts.blur_kind == 0) {
            // few enough blocks for a workgroup of three wavefronts each (chain, loader, storer: k_sweeps1.inc) to find SIMDs of its own:
            // the recurrence alone on one wavefront, the tiles' traffic -- and, at a source-fused level 0, the gathers -- on the others
            const bool mover = p->tune.mover != 0;
            const bool src0 = src && l == 0;
            const bool xby = do_x && do_y && lone_fused_level(p, n, l);
            const double* ystate = p->state;  // where the causal y sweep leaves its state for the anticausal one
            if (do_x && mover && !zt.flags && (nbx <= 340 || (src0 && nbx <= 1024))) {
                {
                    StageTimer t(p, s, src0 ? STITCH_K_VV_X_FWD_SRC : STITCH_K_VV_X_FWD, l);
                    if (src0)
                        k_vv_x_m<true, PX, true><<<nbx, 192, 0, s>>>(a.g, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, bd, pa);
                    else
                        k_vv_x_m<true><<<nbx, 192, 0, s>>>(a.g, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, bd, NoPairArgs{});
                }
                StageTimer t(p, s, STITCH_K_VV_X_BWD, l);
                if (xby) {
                } else if (nbx <= 340)
                    k_vv_x_m<false><<<nbx, 192, 0, s>>>(p->T, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, bd, NoPairArgs{});
                else
                    k_vv_x_bwd<<<nbx, 64, 0, s>>>(p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, bd);
            } else if (do_x) {
                const int nb = nbx;
                {
                    StageTimer t(p, s, src && l == 0 ? STITCH_K_VV_X_FWD_SRC : STITCH_K_VV_X_FWD, l);
                    if (src && l == 0)
                        k_vv_x_fwd<PX, true><<<nb, 64, 0, s>>>(a.g, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, pa, zt, zi, nullptr, bd);
                    else
                        k_vv_x_fwd<PX, false><<<nb, 64, 0, s>>>(a.g, p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, NoPairArgs{}, zt, ZeroTiles{}, nullptr, bd);
                }
                if (!xby) {
                    StageTimer t(p, s, STITCH_K_VV_X_BWD, l);
                    k_vv_x_bwd<<<nb, 64, 0, s>>>(p->T, a.w, a.pitch, lines, p->vvk, p->state, mk, bd);
                }
            } else
                HIPCHK(hipMemcpyAsync(p->T, a.g, sizeof(float) * a.ps * np, hipMemcpyDeviceToDevice, s));
            if (xby) {
                Wavefront wf{};
                wf.yg = p->wf_yg;
                wf.counter = p->wf_ctrl + (size_t)l * 16;
                wf.abort = p->wf_ctrl + WF_CTRL_WORDS - 16;
                wf.sticky = p->wf_ctrl + WF_CTRL_WORDS;
                wf.spin_limit = p->wf_spin_limit;
                wf.NR = NR;
                wf.NC = (a.w + TS - 1) / TS;
                wf.NP = np;
                wf.epoch = 1;
                wf.mask_l0 = mk.enabled;
                wf.zt = zt;
                if (p->xy_dbg && l == 0) wf.dbg = p->xy_dbg, p->xy_dbg_nc = wf.NC;
                // every polled word is cleared in front of the launch (see the batch form above)
                const size_t gran_words = (size_t)wf.NP * wf.NC * WF_GRAN * WAVE;
                k_clear_words<<<(int)std::min<size_t>((gran_words + 255) / 256, 2048), 256, 0, s>>>(p->wf_yg, gran_words);
                // the x state lives in state[0 .. 4 * lines); the y state the kernel leaves goes behind it
                double* state_y = p->state + 4 * (size_t)p->cap * 7 * (a.h + 64);
                ystate = state_y;
                StageTimer t(p, s, STITCH_K_VV_XBYF, l);
                // two workgroups per CU by LDS: every band of a lone 6144 x 4096 pair (448) is resident at once
                k_vv_xby_m<<<(int)std::min<long>((long)wf.NP * wf.NR, XY_WGS), XY_THREADS, 0, s>>>(p->T, a.w, a.h, a.pitch, p->vvk, p->state, lines, state_y, wf, p->zero_page);
            }
            if (do_y) {
                dim3 g((a.pitch + YCOLS - 1) / YCOLS, np);
                // fewer than 1.5 wavefronts per SIMD: the launch's time is one wavefront's chain of rows, i.e. its instructions per
                // row (k_sweeps1.inc): one column per work-item, rows through scalar offsets, the decimation on three consumer wavefronts
                const bool lone = (long)g.x * g.y < 1536 && !p->tune.y2, small_plane = a.ps * sizeof(float) < 0x7fffffffULL;
                const bool ymover = mover && lone && (long)(a.pitch / WAVE) * np <= 340;  // chain + mover, 64 columns per workgroup
                if (!xby) {
                    StageTimer t(p, s, STITCH_K_VV_Y_FWD, l);
                    if (ymover)
                        k_vv_y_m<true><<<dim3(a.pitch / WAVE, np), 192, 0, s>>>(p->T, a.h, a.pitch, a.ps, p->vvk, p->state, mk);
                    else if (lone && small_plane && p->tune.y1s != 0 && (p->tune.y1s >= 2 || a.ps * np * sizeof(float) * 2 < (1000u << 20)))
                        // (56 rows in flight; a sweep that moves more than ~1 GB is bound by bytes and streams better on flat addresses:
                        // 4421 x 2315 level 0 129 -> 98 us, 6144 x 4096 level 0 266 -> 271)
                        k_vv_y_fwd1s<8><<<dim3(a.pitch / WAVE, np), 64, 0, s>>>(p->T, a.h, a.pitch, a.ps, p->vvk, p->state, mk, nullptr);
                    else if (lone)
                        k_vv_y_fwd1<<<dim3(a.pitch / WAVE, np), 64, 0, s>>>(p->T, a.h, a.pitch, a.ps, p->vvk, p->state, mk, nullptr);
                    else
                        k_vv_y_fwd<<<g, 64, 0, s>>>(p->T, a.h, a.pitch, a.ps, p->vvk, p->state, mk, nullptr);
                }
                StageTimer t(p, s, STITCH_K_VV_Y_BWD, l);
                // loader + two chains + four consumers, free-running (k_vv_y_bwd_dec7) where the launch leaves SIMDs idle; the two-wavefront
                // kernel otherwise
                const bool dec7 = lone && small_plane && p->tune.dec7 != 0 && !zt.flags;
                if (zt.flags && ((a.w & 1) == 0 || odd_dec) && !p->no_fuse) {  // (zero-tile flags from k_vv_xby_m: the kernels that read them)
                    if ((a.w & 1) == 0) {
                        if (rowz)
                            k_vv_y_bwd_dec<true><<<g, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, ystate, b.g, b.w, b.h, b.pitch, b.ps, zt, nullptr, nullptr, a.h, b.h);
                        else
                            k_vv_y_bwd_dec<false><<<g, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, ystate, b.g, b.w, b.h, b.pitch, b.ps, zt, nullptr, nullptr, a.h, b.h);
                    } else {
                        const dim3 go((b.w + WAVE - 2) / (WAVE - 1), np);
                        if (rowz)
                            k_vv_y_bwd_dec<true, YST, true, true><<<go, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, ystate, b.g, b.w, b.h, b.pitch, b.ps, zt, nullptr, nullptr, a.h, b.h);
                        else
                            k_vv_y_bwd_dec<false, YST, true, true><<<go, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, ystate, b.g, b.w, b.h, b.pitch, b.ps, zt, nullptr, nullptr, a.h, b.h);
                    }
                    decimated = true;
                } else if ((a.w & 1) == 0 && !p->no_fuse) {  // even width: decimation fused into the anticausal pass
                    if (dec7) {
                        unsigned long long* dbg = p->d7_dbg && l == p->d7_dbg_level ? p->d7_dbg : nullptr;
                        if (dbg) p->d7_dbg_chunks = (a.h + YCH - 1) / YCH;
                        k_vv_y_bwd_dec7<false><<<g, D7_THREADS, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, ystate, b.g, b.w, b.h, b.pitch, b.ps, dbg, dbg ? Tuning::env_int("STITCH_D7_STAMP_MODE") : 0);
                    }
                    else
                        k_vv_y_bwd_dec<false, YST, false><<<g, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, ystate, b.g, b.w, b.h, b.pitch, b.ps, ZeroTiles{}, nullptr, nullptr, a.h, b.h);
                    decimated = true;
                } else if (odd_dec) {
                    const dim3 go((b.w + WAVE - 2) / (WAVE - 1), np);
                    if (dec7)
                        k_vv_y_bwd_dec7<true><<<go, D7_THREADS, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, ystate, b.g, b.w, b.h, b.pitch, b.ps);
                    else
                        k_vv_y_bwd_dec<false, YST, false, true><<<go, 128, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, p->vvk, ystate, b.g, b.w, b.h, b.pitch, b.ps, ZeroTiles{}, nullptr, nullptr, a.h, b.h);
                    decimated = true;
                } else
                    k_vv_y_bwd<<<g, 64, 0, s>>>(p->T, a.h, a.pitch, a.ps, p->vvk, ystate, nullptr, nullptr);
            }
        } else {
            HIPCHK(hipMemcpyAsync(p->T, a.g, sizeof(float) * a.ps * np, hipMemcpyDeviceToDevice, s));
            if (do_x) {
                StageTimer t(p, s, src && l == 0 ? STITCH_K_VV_X_FWD_SRC : STITCH_K_VV_X_FWD, l);
                k_deriche<<<(int)((lines + 63) / 64), 64, 0, s>>>(p->T, p->T2, a.w, 1, a.pitch, a.h, a.ps, lines, p->drk);
            }
            if (do_y) {
                StageTimer t(p, s, STITCH_K_VV_Y_FWD, l);
                const long cols = (long)np * a.w;
                k_deriche<<<(int)((cols + 63) / 64), 64, 0, s>>>(p->T, p->T2, a.h, a.pitch, 1, a.w, a.ps, cols, p->drk);
            }
        }
        if (!decimated) {
            StageTimer t(p, s, STITCH_K_DECIMATE, l);
            k_decimate<<<grid_xy(b.pitch, b.h, np), 256, 0, s>>>(p->T, a.w, a.h, a.pitch, a.ps, b.g, b.w, b.h, b.pitch, b.ps, 0, 0);
        }
    }
    if (p->coarse_from > 0) {  // every remaining level in one launch, one workgroup per pair; leaves E of level coarse_from
        StageTimer t(p, s, STITCH_K_COARSE, p->coarse_from);
        CoarseArgs ca{};
        ca.n = p->L - p->coarse_from;
        for (int i = 0; i < ca.n; ++i) {
            const Level& v = p->lv[p->coarse_from + i];
            ca.lv[i] = CoarseLevel{v.g, v.e, v.ix, v.iy, v.ax, v.ay, v.w, v.h, v.pitch, v.ps};
        }
        ca.T = p->T;
        ca.k = p->vvk;
        ca.dbg = p->co_dbg;
        // every level in LDS where they fit (k_coarse_lds): G of all levels, the blur scratch, the collapse chain; rows pitched odd
        CoarseLds lo{};
        int fl = 0, tabs_w = 0, tabs_h = 0;
        for (int i = 0; i < ca.n; ++i) {
            lo.p[i] = ca.lv[i].w | 1;
            lo.g[i] = fl;
            fl += 7 * ca.lv[i].h * lo.p[i];
            lo.e[i] = fl;
            fl += 3 * ca.lv[i].h * lo.p[i];
            if (i + 1 < ca.n) tabs_w += ca.lv[i].w, tabs_h += ca.lv[i].h;
        }
        lo.t = fl;
        fl += 7 * ca.lv[0].h * lo.p[0];
        const size_t lds_bytes = sizeof(float) * (size_t)fl;
        if (p->tune.coarse_lds != 0 && !p->co_dbg && lds_bytes <= CO_LDS_BYTES && tabs_w <= CO_TABN && tabs_h <= CO_TABN) {
            static std::once_flag once;  // more than 64 KB of dynamic LDS is opt-in per function
            std::call_once(once, [] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_coarse_lds), hipFuncAttributeMaxDynamicSharedMemorySize, CO_LDS_BYTES); });
            k_coarse_lds<<<n, CO_THREADS, lds_bytes, s>>>(ca, lo);
        } else
            k_coarse<<<n, CO_THREADS, 0, s>>>(ca);
    }
    if (p->wf_levels > 0 || any_lone_fused) HIPCHK(hipMemcpyAsync(p->h_wf_abort, p->wf_ctrl + WF_CTRL_WORDS, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    return launch_check("reduce");
}

// Rows per strip of the collapse of level l in a call with n pairs.  A strip is one serial chain of rows per work-item (load -> arithmetic ->
// store, one row after the other) and re-uses the x-interpolated source rows best when it is long: 32 rows wherever the launch has
// workgroups enough to hide that chain behind each other -- every batch.  A launch with few workgroups lives on the chains alone: the
// strip of a ONE-pair call is halved until the launch has STITCH_CROWS_WGS workgroups (default 8192; 0 = round 3's rule: by the level's height alone) or
// is 4 rows high (one pair in flight, 6144 x 4096: level 0 32 -> 8 rows, levels >= 1 32 -> 4: 2.41 -> 2.33 ms; 4421 x 2315 1.30 -> 1.25;
// the rows two strips share are read once more per halving, which a lone pair does not notice).  STITCH_CROWS_L0 / _LN pin the height.
int crows_of(const stitch_plan* p, int l, int n) {
    const Level& a = p->lv[l];
    int c;
    if (l == 0) {
        if (p->tune.crows_l0 >= 0) return p->crows_l0;
        c = a.h >= 2048 ? p->crows_l0 : a.h >= 1024 ? 16 : 8;
    } else {
        if (p->crows_ln > 0) return p->crows_ln;
        c = std::max(4, std::min(4 * CROWS, a.h / 32));
    }
    const long want = p->tune.crows_wgs >= 0 ? p->tune.crows_wgs : 8192;
    const long blocks = std::max(1, (a.c4_xb - a.c4_xa) / 256);
    // (one pair per call only: a batch has other workgroups to run meanwhile, and with four 4-pair sequences in flight shorter strips
    // measured 5 % slower, scripts/experiments/r4_run36.sh)
    while (n == 1 && c > 4 && blocks * ((a.h + c - 1) / c) < want) c /= 2;
    return c;
}

template <typename OUT>
int run_collapse(stitch_plan* p, int n, const OutPtrs<OUT>& outs, hipStream_t s, const PairArgs<OUT>& pa, bool src) {
    const int L = p->L;
    if (p->coarse_from == 0) {
        const Level& t = p->lv[L - 1];
        StageTimer tm(p, s, STITCH_K_COLLAPSE_TOP, L - 1);
        k_blend_top<<<grid_xy(t.pitch, t.h, n), 256, 0, s>>>(t.g, t.pitch, t.h, t.ps, t.e);
        if (L == 1) k_emit_top<OUT><<<grid_xy(t.w, t.h, n), 256, 0, s>>>(t.e, t.w, t.h, t.pitch, t.ps, outs);  // writes out_u8 too
    }
    // with k_coarse the chain arrives collapsed down to level coarse_from
    for (int l = (p->coarse_from > 0 ? p->coarse_from - 1 : L - 2); l >= 0; --l) {
        const Level& a = p->lv[l];
        const Level& nx = p->lv[l + 1];
        StageTimer tm(p, s, l == 0 ? STITCH_K_COLLAPSE_L0 : STITCH_K_COLLAPSE, l);
        // four columns per work-item on the column range whose taps follow the regular pattern (Level::c4_xa, c4_xb), one
        // column per work-item on the rest, in ONE launch (k_collapse4); levels without such a range run k_collapse
        int xa = 0, xb = 0;
        if (p->collapse4) xa = a.c4_xa, xb = a.c4_xb;
        int u8_words = 1;  // level 0: unsigned char rows start on 32-bit boundaries
        if (l == 0) {
            u8_words = (a.w % 4) == 0;
            for (int i = 0; i < n; ++i) {
                if (sizeof(OUT) == 1) u8_words = u8_words && (reinterpret_cast<uintptr_t>(outs.p[i]) % 4) == 0;
                u8_words = u8_words && (reinterpret_cast<uintptr_t>(outs.q[i]) % 4) == 0;
                if (sizeof(OUT) == 4 && (reinterpret_cast<uintptr_t>(outs.p[i]) % 4) != 0) xa = xb = 0;  // a float canvas that is not even 4-byte aligned
            }
        }
        const int cols = l == 0 ? a.w : a.pitch, rest = cols - (xb - xa);
        const int nb4 = ((xb - xa) / 4 + WAVE - 1) / WAVE, ncb = (rest + C4_THREADS - 1) / C4_THREADS;
        if (l == 0) {  // level 0: the mask is the seam's step function itself (never read from memory)
            CollapseArgs<OUT, true> A{a.g, a.w, a.h, a.pitch, a.ps, nx.g, nx.e, nx.w, nx.h, nx.pitch, nx.ps, {a.ix, a.ax, a.iy, a.ay}, outs,
                                      a.w, (size_t)a.w * a.h, p->planes_in ? nullptr : p->d_seam, pa, src ? 1 : 0, crows_of(p, 0, n), xa, xb, u8_words,
                                      std::max(0, p->tune.c4_lock), p->tune.c4_swz < 0 ? 1 : p->tune.c4_swz};
            const int strips = (a.h + A.crows - 1) / A.crows;
            const dim3 g4(c4_padded_blocks(nb4, A.swizzle) + ncb * C4_SUB, strips, n);
            const bool gen = a.c4_gen && p->collapse4;
            if (xb > xa && src && pa.a_dense && gen)
                k_collapse4<OUT, true, true, true><<<g4, C4_THREADS, 0, s>>>(A, nb4, ncb);
            else if (xb > xa && src && pa.a_dense)
                k_collapse4<OUT, true, true><<<g4, C4_THREADS, 0, s>>>(A, nb4, ncb);
            else if (xb > xa && gen)
                k_collapse4<OUT, true, false, true><<<g4, C4_THREADS, 0, s>>>(A, nb4, ncb);
            else if (xb > xa)
                k_collapse4<OUT, true><<<g4, C4_THREADS, 0, s>>>(A, nb4, ncb);
            else
                k_collapse<OUT, true><<<grid_xy(cols, strips, n), 256, 0, s>>>(A);
        } else {
            OutPtrs<float> eo{};
            eo.p[0] = a.e;
            CollapseArgs<float, false> A{a.g, a.w, a.h, a.pitch, a.ps, nx.g, nx.e, nx.w, nx.h, nx.pitch, nx.ps, {a.ix, a.ax, a.iy, a.ay}, eo,
                                         a.pitch, a.ps, nullptr, NoPairArgs{}, 0, crows_of(p, l, n), xa, xb, 1, std::max(0, p->tune.c4_lock), p->tune.c4_swz < 0 ? 1 : p->tune.c4_swz};
            const int strips = (a.h + A.crows - 1) / A.crows;
            const dim3 g4(c4_padded_blocks(nb4, A.swizzle) + ncb * C4_SUB, strips, n);
            if (xb > xa && a.c4_gen && p->collapse4)
                k_collapse4<float, false, false, true><<<g4, C4_THREADS, 0, s>>>(A, nb4, ncb);
            else if (xb > xa)
                k_collapse4<float, false><<<g4, C4_THREADS, 0, s>>>(A, nb4, ncb);
            else if ((long)n * a.pitch * a.h <= 4096L * 64 * 2 && p->tune.collapse_px != 0)  // a launch that leaves most SIMDs with at most a wavefront or two
                k_collapse_px<<<grid_xy(a.pitch, a.h, n), 256, 0, s>>>(A);
            else
                k_collapse<float, false><<<grid_xy(cols, strips, n), 256, 0, s>>>(A);
        }
    }
    return launch_check("collapse");
}

template <typename PX>
int run_seam_mask(stitch_plan* p, int n, hipStream_t s, const PairArgs<PX>& pa, bool src) {
    const Level& a = p->lv[0];
    {
        StageTimer t(p, s, STITCH_K_SEAM, 0);
        // 256 work-items, not 1024: a 16-wavefront workgroup waits for a CU with that much room when other batches fill the chip
        // (1.7 ms per launch with four batches in flight)
        // (one pair per call: nothing else competes for the CU, and the scan is a chain of dependent map evaluations and gathers per
        // work-item -- 6 of them instead of 24 at 6144 columns)
        k_seam<PX><<<n, n == 1 ? 1024 : 256, 0, s>>>(a.g, a.w, a.h, a.pitch, a.ps, p->opts.seam_rule, p->d_seam, pa, src ? 1 : 0);
    }
    if (!p->mask_opt) {
        StageTimer t(p, s, STITCH_K_MASK, 0);
        k_mask<<<grid_xy(a.pitch, a.h, n), 256, 0, s>>>(a.g, a.w, a.pitch, a.ps, p->d_seam);
    }
    HIPCHK(hipMemcpyAsync(p->h_seam, p->d_seam, sizeof(SeamDev) * n, hipMemcpyDeviceToHost, s));
    return launch_check("seam/mask");
}

int check_plan_call(stitch_plan* p, const void* a, const void* b, const void* out) {
    if (!p || !a || !b || !out) return fail(STITCH_ERR_ARG, "null plan or buffer");
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    if (dev != p->device) return fail(STITCH_ERR_ARG, "plan belongs to device %d, current device is %d", p->device, dev);
    return STITCH_OK;
}

// Source fusion trades bytes for instructions: the consumers of level 0 gather their samples (64 four-byte gathers per lane and
// tile in the causal x sweep) instead of streaming planes.  A launch sequence over several pairs is bound by the bytes it moves
// and gains (12 % at config 2); ONE pair in flight is bound by the length of its recurrence chains, its sweeps have a SIMD to
// themselves, and the gathers land on the critical path: measured at 4421 x 2315 / 1081 x 527, materialised S1 2.14 / 0.81 ms,
// source-fused 2.27 / 0.84 ms per pair.  So the form is chosen per CALL: source-fused for launch sequences with enough canvas
// in them, materialised (k_compose / k_load_canvases; the implicit mask stays) otherwise.  STITCH_SINGLE_FAST=1 pins the fused
// forms everywhere (A/B runs, tests).
// Measured per launch sequence of n pairs, 1 and 4 sequences in flight (scripts/experiments/exp_srcfuse_n.py, ms per pair, fused /
// materialised): 1081 x 527: n = 2 0.255 / 0.239, n = 8 0.053 / 0.047 (4 in flight), n = 16 0.040 / 0.046; 4421 x 2315: n = 2
// 0.597 / 0.858 (4 in flight), n = 8 0.500 / 0.588.  The fused form pays from about 8 MPix of canvas per launch sequence.
// (A lone pair that is large enough to fill the chip by itself -- 7 planes x 64-row bands >= 800, the fused sweep's own criterion:
// config 5's 24576 x 16384 -- is a throughput case too: 23.5 ms source-fused against 25.9 ms materialised.)
bool src_fused_call(const stitch_plan* p, int n) {
    if (p->tune.single_fast > 0) return true;
    if (7L * ((p->lv[0].h + TS - 1) / TS) >= 800) return true;
    // (round 4) one pair too: with the gathers on a loader wavefront (k_vv_x_m<.., SRC>) they are off the recurrence's issue stream,
    // and level 0 of a canvas this size is bound by bytes even alone.  STITCH_SRC_LONE_MPIX moves the threshold (0 = never).
    const long long lone_px = p->tune.src_lone >= 0 ? 1000000LL * p->tune.src_lone : 8000000LL;
    if (n == 1) return p->tune.mover != 0 && lone_px > 0 && (long long)p->lv[0].w * p->lv[0].h >= lone_px && p->opts.blur_kind == 0;
    return (long long)n * p->lv[0].w * p->lv[0].h >= 8000000LL;
}

// The launch sequence of n pairs (or of one dense-canvas blend, pa.a_dense): S1, seam scan, REDUCE, collapse.
template <typename PX>
int run_pairs(stitch_plan* p, const PairArgs<PX>& pa, const OutPtrs<PX>& outs, int n, hipStream_t s, bool src) {
    const Level& a = p->lv[0];
    int rc;
    // source-fused: level 0 is a function of the inputs, evaluated by its three consumers (a warped frame through an index
    // plane, dense images as pure shifts)
    ZeroTiles zi{};
    if (src && p->zero_tiles && !pa.a_dense) {
        zi.flags = p->zi;
        zi.h = a.h;
        zi.NR = (a.h + TS - 1) / TS;
        zi.NC = (a.w + TS - 1) / TS;
    }
    {
        StageTimer t(p, s, STITCH_K_COMPOSE, 0);
        if (src) {
            if (!pa.a_dense) {
                if (zi.flags) k_fill_bytes<<<64, 256, 0, s>>>(zi.flags, (size_t)n * zi.NR * zi.NC, (uint8_t)1);
                k_src_index<PX><<<grid_xy(a.pitch, (a.h + SI_ROWS - 1) / SI_ROWS, n), 256, 0, s>>>(pa, a.g, a.w, a.h, a.pitch, a.ps, zi);
            }
        } else if (pa.a_dense)
            k_load_canvases<PX><<<grid_xy(a.pitch, a.h), 256, 0, s>>>(pa.frame[0], pa.mosaic[0], a.g, a.w, a.h, a.pitch, a.ps);
        else
            k_compose<PX><<<grid_xy(a.pitch, a.h, n), 256, 0, s>>>(pa, a.g, a.w, a.h, a.pitch, a.ps, 0);
    }
    if ((rc = run_seam_mask<PX>(p, n, s, pa, src))) return rc;
    if ((rc = run_reduce<PX>(p, n, s, pa, src, zi))) return rc;
    if ((rc = run_collapse<PX>(p, n, outs, s, pa, src))) return rc;
    p->last_stream = s;
    p->pending = true;
    p->last_n = n;
    return STITCH_OK;
}

inline bool ranges_overlap(const void* a, size_t na, const void* b, size_t nb) {
    const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
    return x < y + nb && y < x + na;
}

// blendTwoImages on two dense canvases: a pair whose "frame" is canvas a as it stands and whose "mosaic" is canvas b with a
// zero shift (PairArgs::a_dense).  Source-fused plans read both canvases where level 0 is needed -- no level-0 planes, no mask
// plane, no index plane; otherwise k_load_canvases materialises level 0.
template <typename PX>
int dev_blend(stitch_plan* p, const PX* d_a, const PX* d_b, PX* d_out, void* stream) {
    int rc = check_plan_call(p, d_a, d_b, d_out);
    if (rc) return rc;
    PairArgs<PX> pa{};
    pa.frame[0] = d_a;
    pa.mosaic[0] = d_b;
    pa.out[0] = d_out;
    pa.fw[0] = pa.mw[0] = p->cw;
    pa.fh[0] = pa.mh[0] = p->ch;
    pa.a_dense = 1;
    OutPtrs<PX> outs{};
    outs.p[0] = d_out;
    const size_t bytes = sizeof(PX) * (size_t)3 * p->cw * p->ch;
    // source-fused: a and b are read again by the causal x sweep and by the level-0 collapse WHILE d_out is written, so an output
    // that overlaps an input takes the materialised form (k_load_canvases copies both canvases before anything is written)
    const bool src = p->src_fuse && src_fused_call(p, 1) && (unsigned long long)p->cw * p->ch * sizeof(PX) < 0xfffffff0ULL &&  // 32-bit byte offsets into a channel plane
                     !ranges_overlap(d_out, bytes, d_a, bytes) && !ranges_overlap(d_out, bytes, d_b, bytes);
    return run_pairs<PX>(p, pa, outs, 1, as_stream(stream), src);
}

// n independent pairs (n <= plan capacity) through one launch sequence: every kernel covers all n pairs.
template <typename PX>
int dev_pairs(stitch_plan* p, const stitch_pair_desc* d, int n, void* stream) {
    if (!p || !d) return fail(STITCH_ERR_ARG, "null plan or descriptor array");
    if (n < 1 || n > p->cap) return fail(STITCH_ERR_ARG, "pairs: n=%d outside 1..%d (plan capacity)", n, p->cap);
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    if (dev != p->device) return fail(STITCH_ERR_ARG, "plan belongs to device %d, current device is %d", p->device, dev);
    PairArgs<PX> pa{};
    OutPtrs<PX> outs{};
    for (int i = 0; i < n; ++i) {
        if (!d[i].frame || !d[i].mosaic || !d[i].out || d[i].fw <= 0 || d[i].fh <= 0 || d[i].mw <= 0 || d[i].mh <= 0)
            return fail(STITCH_ERR_ARG, "pairs: descriptor %d has a null buffer or a non-positive size", i);
        pa.frame[i] = static_cast<const PX*>(d[i].frame);
        pa.mosaic[i] = static_cast<const PX*>(d[i].mosaic);
        pa.out[i] = static_cast<PX*>(d[i].out);
        outs.p[i] = pa.out[i];
        outs.q[i] = sizeof(PX) == 4 ? static_cast<uint8_t*>(d[i].out_u8) : nullptr;
        if (sizeof(PX) == 1 && d[i].out_u8) return fail(STITCH_ERR_ARG, "pairs: out_u8 is for float frames (descriptor %d)", i);
        std::memcpy(pa.map[i].p, d[i].p, sizeof pa.map[i].p);
        pa.fw[i] = d[i].fw;
        pa.fh[i] = d[i].fh;
        pa.mw[i] = d[i].mw;
        pa.mh[i] = d[i].mh;
        pa.ox[i] = d[i].ox;
        pa.oy[i] = d[i].oy;
        pa.offx[i] = d[i].offx;
        pa.offy[i] = d[i].offy;
    }
    bool src = p->src_fuse && src_fused_call(p, n);
    for (int i = 0; i < n; ++i)  // 32-bit element indices and byte offsets into one channel plane
        src = src && (unsigned long long)d[i].fw * d[i].fh * sizeof(PX) < 0xfffffff0ULL && (unsigned long long)d[i].mw * d[i].mh * sizeof(PX) < 0xfffffff0ULL;
    // source-fused: the frames are read again while the mosaics are written (see dev_blend): no output of the call may overlap an input of it
    const size_t ob = sizeof(PX) * (size_t)3 * p->cw * p->ch;
    for (int i = 0; src && i < n; ++i)
        for (int j = 0; src && j < n; ++j)
            src = !ranges_overlap(d[i].out, ob, d[j].frame, sizeof(PX) * (size_t)3 * d[j].fw * d[j].fh) &&
                  !ranges_overlap(d[i].out, ob, d[j].mosaic, sizeof(PX) * (size_t)3 * d[j].mw * d[j].mh);
    return run_pairs<PX>(p, pa, outs, n, as_stream(stream), src);
}

template <typename PX>
int dev_pair(stitch_plan* p, const PX* d_frame, int fw, int fh, const double pm[8], float offx, float offy,
             const PX* d_mosaic, int mw, int mh, int ox, int oy, PX* d_out, void* stream) {
    if (!pm) return fail(STITCH_ERR_ARG, "pair: null map");
    stitch_pair_desc d{};
    d.frame = d_frame;
    d.fw = fw;
    d.fh = fh;
    std::memcpy(d.p, pm, sizeof d.p);
    d.offx = offx;
    d.offy = offy;
    d.mosaic = d_mosaic;
    d.mw = mw;
    d.mh = mh;
    d.ox = ox;
    d.oy = oy;
    d.out = d_out;
    return dev_pairs<PX>(p, &d, 1, stream);
}

void seam_to_public(const SeamDev& d, stitch_seam* o) {
    if (!o) return;
    o->sum_a_x = d.sum_a_x;
    o->n_a = d.n_a;
    o->sum_ov_x = d.sum_ov_x;
    o->n_ov = d.n_ov;
    o->ratio = d.ratio;
    o->ov = d.ov;
    o->branch = d.branch;
    o->start = d.start;
}

template <typename PX>
int dev_project(const PX* d_src, int w, int h, float fov_deg, PX* d_dst, void* stream, uint8_t* d_gray = nullptr,
                float* d_gray_f32 = nullptr) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_src || !d_dst || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "project: null buffer or bad size %dx%d", w, h);
    const ProjParams pp = proj_params(w, h, fov_deg);
    // source rows tiled into LDS (k_project_lds; k_project_lds_t for landscape frames, where the axes swap roles) when the
    // largest source box of a tile fits the budget; otherwise (widths that are not a multiple of 4, STITCH_PROJECT1=1): k_project
    constexpr int TW = sizeof(PX) == 1 ? PJ_TW_U8 : PJ_TW_F32, TH = sizeof(PX) == 1 ? PJ_TH_U8 : PJ_TH_F32, CPX = PJ_CHUNK / (int)sizeof(PX);
    size_t lds = 0;
    // (float landscape frames stay with k_project: there a wavefront's taps of a row already share their cache lines -- the
    // source row depends on the output row alone -- and it runs at 0.52 of the roofline, 0.073 ms at 4096 x 3072 against 0.092
    // through LDS; unsigned char: 0.063 -> 0.050)
    bool tiled = (w % 4) == 0 && (unsigned long long)w * h * 3 * sizeof(PX) < 0xfffffff0ULL && (reinterpret_cast<uintptr_t>(d_src) % 4) == 0 &&
                 !(pp.flag && sizeof(PX) == 4) && !std::getenv("STITCH_PROJECT1");
    if (tiled) {
        // the box of a tile, as the kernel derives it, over the tiles that can have the largest one (those farthest from the
        // axis and from the middle: the four corner tiles), plus a margin of two rows and two chunks.  `along` = the axis k
        // depends on (columns for portrait frames, rows for landscape ones), `across` the other.
        const int n_al = pp.flag ? h : w, n_ac = pp.flag ? w : h;
        auto k_of = [&](int i) {
            const float d = (float)(i - n_al / 2);
            const double rd = (double)pp.r, dd = (double)d;
            return (float)(rd / std::sqrt(rd * rd + dd * dd));
        };
        auto u_of = [&](int i) { return (float)(i - n_al / 2) / k_of(i) + (float)(n_al / 2); };
        auto v_of = [&](int j, float k) { return (float)(j - n_ac / 2) / k + (float)(n_ac / 2); };
        const int t_al = pp.flag ? TH : TW, t_ac = pp.flag ? TW : TH;  // tile extent along / across
        const int nt_al = (n_al + t_al - 1) / t_al, nt_ac = (n_ac + t_ac - 1) / t_ac;
        for (int bc : {0, nt_ac - 1})
            for (int ba : {0, nt_al - 1}) {
                const int ia = ba * t_al, ib = std::min(ia + t_al, n_al) - 1, ja = bc * t_ac, jb = std::min(ja + t_ac, n_ac) - 1;
                const int mid = n_al / 2, inear = ia <= mid && mid <= ib ? mid : (std::abs(ia - mid) < std::abs(ib - mid) ? ia : ib),
                          ifar = std::abs(ia - mid) > std::abs(ib - mid) ? ia : ib;
                const float kn = k_of(inear), kf = k_of(ifar);
                const float vs[4] = {v_of(ja, kn), v_of(ja, kf), v_of(jb, kn), v_of(jb, kf)};
                const float vmin = std::min(std::min(vs[0], vs[1]), std::min(vs[2], vs[3])), vmax = std::max(std::max(vs[0], vs[1]), std::max(vs[2], vs[3]));
                const long n_across = (long)std::ceil(vmax) - (long)std::floor(vmin) + 1, n_along = (long)std::ceil(u_of(ib)) - (long)std::floor(u_of(ia)) + 1;
                const long rows = (pp.flag ? n_along : n_across) + 2;
                const long cols = ((pp.flag ? n_across : n_along) + 2 * CPX + CPX - 1) / CPX * CPX;
                lds = std::max(lds, pj_box_bytes<PX>((size_t)rows * cols));
            }
        tiled = lds <= 60 * 1024;
    }
    const dim3 tgrid((w + TW - 1) / TW, (h + TH - 1) / TH);
    if (tiled && !pp.flag)
        k_project_lds<PX, TW, TH><<<tgrid, 256, lds, as_stream(stream)>>>(d_src, d_dst, w, h, pp.r, d_gray, d_gray_f32, (int)lds);
    else if (tiled)
        k_project_lds_t<PX, TW, TH><<<tgrid, 256, lds, as_stream(stream)>>>(d_src, d_dst, w, h, pp.r, d_gray, d_gray_f32, (int)lds);
    else
        k_project<PX><<<grid_xy(w, h), 256, 0, as_stream(stream)>>>(d_src, d_dst, w, h, pp.flag, pp.width, pp.height, pp.r, d_gray,
                                                                    d_gray_f32);
    return launch_check("k_project");
}

template <typename PX>
int dev_warp(const PX* d_src, int sw, int sh, const double pm[8], float offx, float offy, PX* d_canvas, int cw, int ch,
             void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_src || !d_canvas || !pm || sw <= 0 || sh <= 0 || cw <= 0 || ch <= 0) return fail(STITCH_ERR_ARG, "warp: bad argument");
    MapP m;
    std::memcpy(m.p, pm, sizeof m.p);
    k_warp<PX><<<grid_xy(cw, ch), 256, 0, as_stream(stream)>>>(d_src, sw, sh, m, offx, offy, d_canvas, cw, ch);
    return launch_check("k_warp");
}

template <typename PX>
int dev_move(const PX* d_src, int sw, int sh, int ox, int oy, PX* d_canvas, int cw, int ch, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_src || !d_canvas || sw <= 0 || sh <= 0 || cw <= 0 || ch <= 0) return fail(STITCH_ERR_ARG, "move: bad argument");
    k_move<PX><<<grid_xy(cw, ch), 256, 0, as_stream(stream)>>>(d_src, sw, sh, ox, oy, d_canvas, cw, ch);
    return launch_check("k_move");
}

// RAII device buffer for the host-pointer entry points.  Blocks come from the device's stream-ordered memory pool with its
// release threshold lifted (once per device), so that a caller that stitches frame after frame -- the reference's matching()
// loop through the C++ adaptor -- re-uses the same device memory instead of paying hipMalloc + hipFree (both synchronise
// the device) on every call; stitch_trim() hands the pool's memory back.
void keep_pool_memory() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> g(mu);
    if (done[dev]) return;
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
        uint64_t keep = UINT64_MAX;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    (void)hipGetLastError();
    done[dev] = true;
}
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFreeAsync(p, nullptr);
    }
    int alloc(size_t bytes) {
        keep_pool_memory();
        HIPCHK(hipMallocAsync(&p, bytes, nullptr));
        return STITCH_OK;
    }
    template <typename T>
    T* as() {
        return static_cast<T*>(p);
    }
};

// Host <-> device copies of the host-pointer entry points.  The caller's buffers are ordinary pageable memory (CImg's
// new[]), which the runtime moves at about 10 GB/s through one staging thread -- 70 ms for the 704 MB of a 4096 x 4096
// float pair, fifty times the blend itself.  Large copies are therefore cut into chunks and spread over a few persistent
// worker threads, each with its own pair of pinned staging buffers and its own stream: the workers' memcpy()s run side by
// side and overlap the DMA of earlier chunks.  Small copies go straight through hipMemcpy.
class HostCopier {
  public:
    static HostCopier& get() {
        static HostCopier* c = new HostCopier();  // never destroyed: its threads are parked on its condition variable until the process ends
        return *c;
    }
    // returns a HIP error code (0 = ok); both calls return when the data has arrived
    int h2d(void* dev, const void* host, size_t bytes) { return run(static_cast<char*>(dev), const_cast<char*>(static_cast<const char*>(host)), bytes, true); }
    int d2h(void* host, const void* dev, size_t bytes) { return run(const_cast<char*>(static_cast<const char*>(dev)), static_cast<char*>(host), bytes, false); }
    void release() {  // stitch_trim(): pinned buffers are given back, the threads stay parked
        std::lock_guard<std::mutex> g(call_mu_);
        for (auto& w : workers_)
            for (auto& b : w->buf)
                if (b) {
                    (void)hipHostFree(b);
                    b = nullptr;
                }
    }

  private:
    static constexpr size_t CHUNK = 8u << 20, DIRECT_BELOW = 4u << 20;
    struct Worker {
        std::thread th;
        hipStream_t stream = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        char* buf[2] = {nullptr, nullptr};
    };
    std::vector<Worker*> workers_;
    std::mutex call_mu_, mu_;  // one copy at a time; job hand-off
    std::condition_variable cv_, done_cv_;
    // the current job
    char *dev_ = nullptr, *host_ = nullptr;
    size_t bytes_ = 0;
    bool to_dev_ = true;
    int device_ = 0, generation_ = 0, pending_ = 0;
    std::atomic<size_t> next_{0};
    std::atomic<int> err_{0};

    HostCopier() {
        int n = 4;
        if (const char* e = std::getenv("STITCH_COPY_THREADS")) n = std::max(0, std::min(16, atoi(e)));
        for (int i = 0; i < n; ++i) {
            Worker* w = new Worker();
            workers_.push_back(w);
            w->th = std::thread([this, w] { loop(w); });
            w->th.detach();  // parked on the condition variable for the life of the process
        }
    }
    int run(char* dev, char* host, size_t bytes, bool to_dev) {
        if (bytes < DIRECT_BELOW || workers_.empty())
            return (int)(to_dev ? hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) : hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
        std::lock_guard<std::mutex> call(call_mu_);
        hipError_t e = hipStreamSynchronize(nullptr);  // the kernels that produced / will consume the device buffer run on the null stream
        if (e != hipSuccess) return (int)e;
        if ((e = hipGetDevice(&device_)) != hipSuccess) return (int)e;
        {
            std::lock_guard<std::mutex> g(mu_);
            dev_ = dev, host_ = host, bytes_ = bytes, to_dev_ = to_dev;
            next_ = 0;
            err_ = 0;
            pending_ = (int)workers_.size();
            ++generation_;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> g(mu_);
        done_cv_.wait(g, [this] { return pending_ == 0; });
        return err_.load();
    }
    void loop(Worker* w) {
        int seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
            }
            work(w);
            {
                std::lock_guard<std::mutex> g(mu_);
                --pending_;
            }
            done_cv_.notify_all();
        }
    }
    void fail(hipError_t e) {
        int z = 0;
        err_.compare_exchange_strong(z, (int)e);
    }
    void work(Worker* w) {
        hipError_t e = hipSetDevice(device_);
        if (e == hipSuccess && !w->stream) e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; ++i) {
            if (!w->ev[i]) e = hipEventCreateWithFlags(&w->ev[i], hipEventDisableTiming);
            if (e == hipSuccess && !w->buf[i]) e = hipHostMalloc((void**)&w->buf[i], CHUNK);
        }
        if (e != hipSuccess) return fail(e);
        bool busy[2] = {false, false};
        size_t off_of[2] = {0, 0}, len_of[2] = {0, 0};
        int slot = 0;
        auto drain = [&](int s_) {  // the DMA that last used slot s_ has finished; D2H: its bytes go on to the caller's buffer
            if (!busy[s_]) return;
            if ((e = hipEventSynchronize(w->ev[s_])) != hipSuccess) return fail(e);
            if (!to_dev_) std::memcpy(host_ + off_of[s_], w->buf[s_], len_of[s_]);
            busy[s_] = false;
        };
        for (;;) {
            const size_t off = next_.fetch_add(CHUNK);
            if (off >= bytes_ || err_.load()) break;
            const size_t len = std::min(CHUNK, bytes_ - off);
            drain(slot);
            if (to_dev_) {
                std::memcpy(w->buf[slot], host_ + off, len);
                e = hipMemcpyAsync(dev_ + off, w->buf[slot], len, hipMemcpyHostToDevice, w->stream);
            } else
                e = hipMemcpyAsync(w->buf[slot], dev_ + off, len, hipMemcpyDeviceToHost, w->stream);
            if (e == hipSuccess) e = hipEventRecord(w->ev[slot], w->stream);
            if (e != hipSuccess) return fail(e);
            busy[slot] = true, off_of[slot] = off, len_of[slot] = len;
            slot ^= 1;
        }
        drain(slot);
        drain(slot ^ 1);
    }
};
#define H2D(dst, src, bytes)                                                                                                   \
    do {                                                                                                                       \
        int e_ = HostCopier::get().h2d(dst, src, bytes);                                                                       \
        if (e_) return fail(STITCH_ERR_HIP, "host-to-device copy of %zu bytes failed: %s", (size_t)(bytes), hipGetErrorString((hipError_t)e_)); \
    } while (0)
#define D2H(dst, src, bytes)                                                                                                   \
    do {                                                                                                                       \
        int e_ = HostCopier::get().d2h(dst, src, bytes);                                                                       \
        if (e_) return fail(STITCH_ERR_HIP, "device-to-host copy of %zu bytes failed: %s", (size_t)(bytes), hipGetErrorString((hipError_t)e_)); \
    } while (0)

template <typename PX>
int host_project(const PX* src, int w, int h, float fov_deg, PX* dst) {
    int rc = need_device();
    if (rc) return rc;
    if (!src || !dst || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "project: null buffer or bad size %dx%d", w, h);
    const size_t bytes = sizeof(PX) * (size_t)w * h * 3;
    DevBuf s, d;
    if ((rc = s.alloc(bytes)) || (rc = d.alloc(bytes))) return rc;
    H2D(s.p, src, bytes);
    if ((rc = dev_project<PX>(s.as<PX>(), w, h, fov_deg, d.as<PX>(), nullptr))) return rc;
    D2H(dst, d.p, bytes);
    return STITCH_OK;
}

template <typename PX>
int host_warp(const PX* src, int sw, int sh, const double pm[8], float offx, float offy, PX* canvas, int cw, int ch) {
    int rc = need_device();
    if (rc) return rc;
    if (!src || !canvas || !pm || sw <= 0 || sh <= 0 || cw <= 0 || ch <= 0) return fail(STITCH_ERR_ARG, "warp: bad argument");
    const size_t sb = sizeof(PX) * (size_t)sw * sh * 3, cb = sizeof(PX) * (size_t)cw * ch * 3;
    DevBuf s, c;
    if ((rc = s.alloc(sb)) || (rc = c.alloc(cb))) return rc;
    H2D(s.p, src, sb);
    H2D(c.p, canvas, cb);
    if ((rc = dev_warp<PX>(s.as<PX>(), sw, sh, pm, offx, offy, c.as<PX>(), cw, ch, nullptr))) return rc;
    D2H(canvas, c.p, cb);
    return STITCH_OK;
}

template <typename PX>
int host_move(const PX* src, int sw, int sh, int ox, int oy, PX* canvas, int cw, int ch) {
    int rc = need_device();
    if (rc) return rc;
    if (!src || !canvas || sw <= 0 || sh <= 0 || cw <= 0 || ch <= 0) return fail(STITCH_ERR_ARG, "move: bad argument");
    const size_t sb = sizeof(PX) * (size_t)sw * sh * 3, cb = sizeof(PX) * (size_t)cw * ch * 3;
    DevBuf s, c;
    if ((rc = s.alloc(sb)) || (rc = c.alloc(cb))) return rc;
    H2D(s.p, src, sb);
    H2D(c.p, canvas, cb);
    if ((rc = dev_move<PX>(s.as<PX>(), sw, sh, ox, oy, c.as<PX>(), cw, ch, nullptr))) return rc;
    D2H(canvas, c.p, cb);
    return STITCH_OK;
}

// Workspaces of the host-pointer entry points are kept between calls: creating one costs a device allocation of the whole
// pyramid (1.9 GB at 6144 x 4096), its clearing, the resize tables and two pinned allocations -- more than the blend itself
// below about 2000 x 2000.  A small LRU keyed by (device, canvas, options) holds idle plans; a call takes one out (so that
// concurrent callers never share a workspace) and puts it back when it is done.  STITCH_PLAN_CACHE=<n> sets the number of
// idle plans kept (default 8, 0 = none); stitch_trim() destroys them.
struct PlanKey {
    int dev, cw, ch;
    stitch_blend_opts o;
    Tuning t;  // the tuning switches the plan was built under (Tuning): a changed environment never meets a stale plan
    bool operator==(const PlanKey& k) const {
        return dev == k.dev && cw == k.cw && ch == k.ch && o.sigma == k.o.sigma && o.blur_kind == k.o.blur_kind && o.level_rule == k.o.level_rule &&
               o.seam_rule == k.o.seam_rule && t == k.t;
    }
};
std::mutex g_plan_mu;
std::list<std::pair<PlanKey, stitch_plan*>> g_idle_plans;  // most recently used first
size_t plan_cache_limit() {
    static const size_t n = [] {
        const char* e = std::getenv("STITCH_PLAN_CACHE");
        return e ? (size_t)std::max(0, atoi(e)) : (size_t)8;
    }();
    return n;
}
struct PlanLease {
    stitch_plan* p = nullptr;
    PlanKey key{};
    bool healthy = false;  // the leased call got as far as a status the plan survives (finish()); otherwise the plan is destroyed
    int acquire(int cw, int ch, const stitch_blend_opts* opts) {
        stitch_blend_opts o;
        stitch_blend_opts_default(&o);
        if (opts) o = *opts;
        int dev = 0;
        HIPCHK(hipGetDevice(&dev));
        key = PlanKey{dev, cw, ch, o, Tuning::from_env()};
        {
            std::lock_guard<std::mutex> g(g_plan_mu);
            for (auto it = g_idle_plans.begin(); it != g_idle_plans.end(); ++it)
                if (it->first == key) {
                    p = it->second;
                    g_idle_plans.erase(it);
                    return STITCH_OK;
                }
        }
        return stitch_plan_create(cw, ch, opts, &p);
    }
    // outcome of the leased call's stitch_plan_status: a failed seam scan leaves the workspace as good as before
    int finish(int rc) {
        healthy = rc == STITCH_OK || rc == STITCH_ERR_EMPTY_MIDROW || rc == STITCH_ERR_ZERO_OVERLAP;
        return rc;
    }
    ~PlanLease() {
        if (!p) return;
        // wait for the plan's last call and acknowledge its faults BEFORE the cache mutex is taken: other threads' host-pointer
        // calls only ever block on the list operations below, never on this plan's GPU work
        const bool keep = healthy && plan_cache_limit() > 0 && stitch_plan_clear_fault(p) == STITCH_OK;
        std::vector<stitch_plan*> evict;
        if (keep) {
            std::lock_guard<std::mutex> g(g_plan_mu);
            g_idle_plans.emplace_front(key, p);
            p = nullptr;
            while (g_idle_plans.size() > plan_cache_limit()) {
                evict.push_back(g_idle_plans.back().second);
                g_idle_plans.pop_back();
            }
        }
        if (p) stitch_plan_destroy(p);  // a call that failed part-way (or a timed-out hand-off) never goes back into the cache
        for (auto* e : evict) stitch_plan_destroy(e);
    }
};

template <typename PX>
int host_blend(const PX* a, const PX* b, int w, int h, const stitch_blend_opts* opts, PX* out, stitch_seam* seam_out) {
    int rc = need_device();
    if (rc) return rc;
    if (!a || !b || !out || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "blend: null buffer or bad size %dx%d", w, h);
    PlanLease pg;
    if ((rc = pg.acquire(w, h, opts))) return rc;
    const size_t bytes = sizeof(PX) * (size_t)w * h * 3;
    DevBuf da, db, dout;
    if ((rc = da.alloc(bytes)) || (rc = db.alloc(bytes)) || (rc = dout.alloc(bytes))) return rc;
    H2D(da.p, a, bytes);
    H2D(db.p, b, bytes);
    if ((rc = dev_blend<PX>(pg.p, da.as<PX>(), db.as<PX>(), dout.as<PX>(), nullptr))) return rc;
    if ((rc = pg.finish(stitch_plan_status(pg.p, seam_out)))) return rc;
    D2H(out, dout.p, bytes);
    return STITCH_OK;
}

template <typename PX>
int host_pair(const PX* frame, int fw, int fh, const double pm[8], float offx, float offy, const PX* mosaic, int mw, int mh,
              int ox, int oy, int cw, int ch, const stitch_blend_opts* opts, PX* out, stitch_seam* seam_out) {
    int rc = need_device();
    if (rc) return rc;
    if (!frame || !mosaic || !out || !pm || fw <= 0 || fh <= 0 || mw <= 0 || mh <= 0 || cw <= 0 || ch <= 0)
        return fail(STITCH_ERR_ARG, "pair: bad argument");
    PlanLease pg;
    if ((rc = pg.acquire(cw, ch, opts))) return rc;
    const size_t fb = sizeof(PX) * (size_t)fw * fh * 3, mb = sizeof(PX) * (size_t)mw * mh * 3,
                 ob = sizeof(PX) * (size_t)cw * ch * 3;
    DevBuf df, dm, dout;
    if ((rc = df.alloc(fb)) || (rc = dm.alloc(mb)) || (rc = dout.alloc(ob))) return rc;
    H2D(df.p, frame, fb);
    H2D(dm.p, mosaic, mb);
    if ((rc = dev_pair<PX>(pg.p, df.as<PX>(), fw, fh, pm, offx, offy, dm.as<PX>(), mw, mh, ox, oy, dout.as<PX>(), nullptr)))
        return rc;
    if ((rc = pg.finish(stitch_plan_status(pg.p, seam_out)))) return rc;
    D2H(out, dout.p, ob);
    return STITCH_OK;
}

// planes of n bytes that can be moved as 32-bit words (the four-pixels-per-work-item kernels); STITCH_BYTE_KERNELS=1: never
bool words_ok(const void* base, size_t n) {
    return (n % 4) == 0 && (reinterpret_cast<uintptr_t>(base) % 4) == 0 && !std::getenv("STITCH_BYTE_KERNELS");
}

// The mix's divides by `den` as multiplications by its reciprocal plus one fma correction -- only where that is PROVEN equal for
// every operand the kernels can meet (k_equalize.inc, MixK): Y and Yeq are floats (float)(t/1000) clamped, t = 0 .. 323850.
// Both quotients of ImageProcess.cpp:261, a = Y*num and a = Yeq, are compared in both forms for all of them (~2 ms, once per
// (num, den); the last few verdicts are kept).  STITCH_NO_FASTDIV=1: always divide.
MixK mix_params(double num, double den) {
    MixK k{num, den, 0.0};
    static const bool off = std::getenv("STITCH_NO_FASTDIV") != nullptr;
    if (off || !(den == den) || den == 0.0 || std::isinf(den) || !(num == num) || std::isinf(num)) return k;
    struct Verdict {
        double num, den;
        bool ok;
    };
    static std::mutex mu;
    static std::vector<Verdict> seen;
    {
        std::lock_guard<std::mutex> g(mu);
        for (const Verdict& v : seen)
            if (std::memcmp(&v.num, &num, sizeof num) == 0 && std::memcmp(&v.den, &den, sizeof den) == 0) {
                if (v.ok) k.rden = 1.0 / den;
                return k;
            }
    }
    const double rden = 1.0 / den;
    auto quot = [&](double a) {
        const double q = a * rden;
        return std::fma(std::fma(-q, den, a), rden, q);
    };
    bool ok = std::isfinite(rden) && rden != 0.0;
    for (unsigned t = 0; ok && t <= 323850u; ++t) {
        float y = (float)((double)t * 0.001);
        y = y < 256.f ? y : 255.f;
        const double a1 = (double)y * num, a2 = (double)y;
        const double d1 = a1 / den, d2 = a2 / den, f1 = quot(a1), f2 = quot(a2);
        ok = std::memcmp(&d1, &f1, sizeof d1) == 0 && std::memcmp(&d2, &f2, sizeof d2) == 0;
    }
    {
        std::lock_guard<std::mutex> g(mu);
        if (seen.size() >= 16) seen.erase(seen.begin());
        seen.push_back(Verdict{num, den, ok});
    }
    if (ok) k.rden = rden;
    return k;
}

int eq_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 2048 ? (g ? g : 1) : 2048);  // memory-bound: cap the grid and stride (guide 6, guideline 11)
}

// E1-E3 (+ optional fused M1).  Scratch (256-bin histogram + LUT) comes from the stream-ordered allocator, so
// concurrent calls on different streams do not share state.
int dev_equalize_impl(uint8_t* d_img, int w, int h, int32_t* d_hist_out, bool fuse_mix, double num, double den, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_img || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "equalize: null buffer or bad size %dx%d", w, h);
    if ((long long)w * h > 0x7fffffffLL) return fail(STITCH_ERR_ARG, "equalize: w*h overflows int (the reference's int product)");
    hipStream_t s = as_stream(stream);
    int32_t* scratch = nullptr;
    HIPCHK(hipMallocAsync((void**)&scratch, sizeof(int32_t) * 512, s));
    int32_t *hist = scratch, *lut = scratch + 256;
    HIPCHK(hipMemsetAsync(hist, 0, sizeof(int32_t) * 256, s));
    const size_t n = (size_t)w * h;
    const bool v4 = words_ok(d_img, n);
    if (v4)
        // every workgroup ends with 256 global atomics on the same 256 words: 512 workgroups (measured at 25 MPix: 2048 -> 50 us,
        // 1024 -> 31, 512 -> 29, 256 -> 38)
        k_hist4<<<std::min(eq_grid(n / 4), 512), HIST_WAVES * 64, 0, s>>>(d_img, n, hist);
    else
        k_hist<<<eq_grid(n), HIST_WAVES * 64, 0, s>>>(d_img, n, hist);
    k_lut<<<1, 256, 0, s>>>(hist, w, h, lut);
    if (fuse_mix) {
        if (v4)
            k_equalize_apply4<true><<<eq_grid(n / 4), 256, 0, s>>>(d_img, n, lut, mix_params(num, den));
        else
            k_equalize_apply<true><<<eq_grid(n), 256, 0, s>>>(d_img, n, lut, mix_params(num, den));
    } else {
        if (v4)
            k_equalize_apply4<false><<<eq_grid(n / 4), 256, 0, s>>>(d_img, n, lut, MixK{0.0, 1.0, 0.0});
        else
            k_equalize_apply<false><<<eq_grid(n), 256, 0, s>>>(d_img, n, lut, MixK{0.0, 1.0, 0.0});
    }
    if (d_hist_out) HIPCHK(hipMemcpyAsync(d_hist_out, hist, sizeof(int32_t) * 256, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipFreeAsync(scratch, s));
    return launch_check("equalize");
}

template <typename PX>
int dev_step(const PX* d_frame, int fw, int fh, const double p_fwd[8], const double p_bwd[8], const PX* d_mosaic, int mw, int mh,
                    const stitch_blend_opts* opts, PX* d_out, size_t out_capacity, stitch_step_geom* geom_out, stitch_seam* seam_out,
                    void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_frame || !d_mosaic || !d_out || !p_fwd || !p_bwd) return fail(STITCH_ERR_ARG, "step: null argument");
    stitch_step_geom g;
    if ((rc = stitch_step_geometry(fw, fh, p_fwd, mw, mh, &g))) return rc;
    if (geom_out) *geom_out = g;
    if (out_capacity < (size_t)3 * g.cw * g.ch)
        return fail(STITCH_ERR_ARG, "step: the new mosaic is %d x %d x 3 = %zu samples, the output buffer holds %zu", g.cw, g.ch,
                    (size_t)3 * g.cw * g.ch, out_capacity);
    PlanLease lease;  // idle workspace of this canvas size, or a new one; goes back to the cache when the step is done
    if ((rc = lease.acquire(g.cw, g.ch, opts))) return rc;
    if ((rc = dev_pair<PX>(lease.p, d_frame, fw, fh, p_bwd, g.min_x, g.min_y, d_mosaic, mw, mh, g.ox, g.oy, d_out, stream))) return rc;
    return lease.finish(stitch_plan_status(lease.p, seam_out));
}

}  // namespace

// =========================================== C ABI ==========================================================
extern "C" {

int stitch_abi_version(void) { return STITCH_ABI_VERSION; }
int stitch_init(void) {
    if (std::getenv("GPU_MAX_HW_QUEUES")) return 0;
    return setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0) == 0 ? 1 : 0;
}
const char* stitch_last_error(void) { return g_err.c_str(); }

int stitch_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int stitch_set_device(int ordinal) {
    int rc = need_device();
    if (rc) return rc;
    HIPCHK(hipSetDevice(ordinal));
    return STITCH_OK;
}

void stitch_trim(void) {
    std::vector<stitch_plan*> all;
    {
        std::lock_guard<std::mutex> g(g_plan_mu);
        for (auto& e : g_idle_plans) all.push_back(e.second);
        g_idle_plans.clear();
    }
    for (auto* p : all) stitch_plan_destroy(p);
    HostCopier::get().release();
    int dev = 0;
    hipMemPool_t pool;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess)
        (void)hipMemPoolTrimTo(pool, 0);
    (void)hipGetLastError();
}

int stitch_plan_cache_query(int cw, int ch, const stitch_blend_opts* opts, int* fused_sweep_levels) {
    stitch_blend_opts o;
    stitch_blend_opts_default(&o);
    if (opts) o = *opts;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    const PlanKey key{dev, cw, ch, o, Tuning::from_env()};
    int n = 0;
    std::lock_guard<std::mutex> g(g_plan_mu);
    for (auto& e : g_idle_plans)
        if (e.first == key) {
            if (n == 0 && fused_sweep_levels) *fused_sweep_levels = e.second->wf_levels;
            ++n;
        }
    return n;
}

void stitch_blend_opts_default(stitch_blend_opts* o) {
    if (!o) return;
    o->sigma = 2.0f;
    o->blur_kind = 0;
    o->level_rule = 0;
    o->seam_rule = 0;
}

int stitch_pyramid_levels(int w, int h, int level_rule, int* level_w, int* level_h) {
    return pyramid_levels(w, h, level_rule, level_w, level_h);
}

int stitch_project_u8(const uint8_t* src, int w, int h, float fov_deg, uint8_t* dst) { return host_project(src, w, h, fov_deg, dst); }
int stitch_project_f32(const float* src, int w, int h, float fov_deg, float* dst) { return host_project(src, w, h, fov_deg, dst); }
int stitch_warp_u8(const uint8_t* src, int sw, int sh, const double p[8], float offx, float offy, uint8_t* canvas, int cw, int ch) {
    return host_warp(src, sw, sh, p, offx, offy, canvas, cw, ch);
}
int stitch_warp_f32(const float* src, int sw, int sh, const double p[8], float offx, float offy, float* canvas, int cw, int ch) {
    return host_warp(src, sw, sh, p, offx, offy, canvas, cw, ch);
}
int stitch_move_u8(const uint8_t* src, int sw, int sh, int ox, int oy, uint8_t* canvas, int cw, int ch) {
    return host_move(src, sw, sh, ox, oy, canvas, cw, ch);
}
int stitch_move_f32(const float* src, int sw, int sh, int ox, int oy, float* canvas, int cw, int ch) {
    return host_move(src, sw, sh, ox, oy, canvas, cw, ch);
}
int stitch_blend_u8(const uint8_t* a, const uint8_t* b, int w, int h, const stitch_blend_opts* opts, uint8_t* out,
                    stitch_seam* seam_out) {
    return host_blend(a, b, w, h, opts, out, seam_out);
}
int stitch_blend_f32(const float* a, const float* b, int w, int h, const stitch_blend_opts* opts, float* out, stitch_seam* seam_out) {
    return host_blend(a, b, w, h, opts, out, seam_out);
}
int stitch_pair_u8(const uint8_t* frame, int fw, int fh, const double p[8], float offx, float offy, const uint8_t* mosaic, int mw,
                   int mh, int ox, int oy, int cw, int ch, const stitch_blend_opts* opts, uint8_t* out, stitch_seam* seam_out) {
    return host_pair(frame, fw, fh, p, offx, offy, mosaic, mw, mh, ox, oy, cw, ch, opts, out, seam_out);
}
int stitch_pair_f32(const float* frame, int fw, int fh, const double p[8], float offx, float offy, const float* mosaic, int mw,
                    int mh, int ox, int oy, int cw, int ch, const stitch_blend_opts* opts, float* out, stitch_seam* seam_out) {
    return host_pair(frame, fw, fh, p, offx, offy, mosaic, mw, mh, ox, oy, cw, ch, opts, out, seam_out);
}

int stitch_equalize_u8(uint8_t* img, int w, int h, int32_t hist_out[256]) {
    int rc = need_device();
    if (rc) return rc;
    if (!img || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "equalize: null buffer or bad size %dx%d", w, h);
    const size_t bytes = (size_t)w * h * 3;
    DevBuf d, dh;
    if ((rc = d.alloc(bytes)) || (rc = dh.alloc(sizeof(int32_t) * 256))) return rc;
    H2D(d.p, img, bytes);
    if ((rc = dev_equalize_impl(d.as<uint8_t>(), w, h, dh.as<int32_t>(), false, 0, 1, nullptr))) return rc;
    D2H(img, d.p, bytes);
    if (hist_out) D2H(hist_out, dh.p, sizeof(int32_t) * 256);
    return STITCH_OK;
}

int stitch_lummix_u8(uint8_t* result, const uint8_t* equalized, int w, int h, double num, double den) {
    int rc = need_device();
    if (rc) return rc;
    if (!result || !equalized || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "lummix: null buffer or bad size %dx%d", w, h);
    const size_t bytes = (size_t)w * h * 3;
    DevBuf r, e;
    if ((rc = r.alloc(bytes)) || (rc = e.alloc(bytes))) return rc;
    H2D(r.p, result, bytes);
    H2D(e.p, equalized, bytes);
    if ((rc = stitch_dev_lummix_u8(r.as<uint8_t>(), e.as<uint8_t>(), w, h, num, den, nullptr))) return rc;
    D2H(result, r.p, bytes);
    return STITCH_OK;
}

int stitch_finish_u8(uint8_t* result, int w, int h, double num, double den, int32_t hist_out[256]) {
    int rc = need_device();
    if (rc) return rc;
    if (!result || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "finish: null buffer or bad size %dx%d", w, h);
    const size_t bytes = (size_t)w * h * 3;
    DevBuf d, dh;
    if ((rc = d.alloc(bytes)) || (rc = dh.alloc(sizeof(int32_t) * 256))) return rc;
    H2D(d.p, result, bytes);
    if ((rc = dev_equalize_impl(d.as<uint8_t>(), w, h, dh.as<int32_t>(), true, num, den, nullptr))) return rc;
    D2H(result, d.p, bytes);
    if (hist_out) D2H(hist_out, dh.p, sizeof(int32_t) * 256);
    return STITCH_OK;
}

int stitch_dev_project_u8(const uint8_t* d_src, int w, int h, float fov_deg, uint8_t* d_dst, void* stream) {
    return dev_project(d_src, w, h, fov_deg, d_dst, stream);
}
int stitch_dev_project_f32(const float* d_src, int w, int h, float fov_deg, float* d_dst, void* stream) {
    return dev_project(d_src, w, h, fov_deg, d_dst, stream);
}
int stitch_dev_warp_u8(const uint8_t* d_src, int sw, int sh, const double p[8], float offx, float offy, uint8_t* d_canvas, int cw,
                       int ch, void* stream) {
    return dev_warp(d_src, sw, sh, p, offx, offy, d_canvas, cw, ch, stream);
}
int stitch_dev_warp_f32(const float* d_src, int sw, int sh, const double p[8], float offx, float offy, float* d_canvas, int cw,
                        int ch, void* stream) {
    return dev_warp(d_src, sw, sh, p, offx, offy, d_canvas, cw, ch, stream);
}
int stitch_dev_move_u8(const uint8_t* d_src, int sw, int sh, int ox, int oy, uint8_t* d_canvas, int cw, int ch, void* stream) {
    return dev_move(d_src, sw, sh, ox, oy, d_canvas, cw, ch, stream);
}
int stitch_dev_move_f32(const float* d_src, int sw, int sh, int ox, int oy, float* d_canvas, int cw, int ch, void* stream) {
    return dev_move(d_src, sw, sh, ox, oy, d_canvas, cw, ch, stream);
}

int stitch_plan_create(int cw, int ch, const stitch_blend_opts* opts, stitch_plan** plan_out) {
    return stitch_plan_create_batched(cw, ch, opts, 1, plan_out);
}

int stitch_plan_create_batched(int cw, int ch, const stitch_blend_opts* opts, int max_pairs, stitch_plan** plan_out) {
    if (!plan_out) return fail(STITCH_ERR_ARG, "plan_create: null plan_out");
    if (max_pairs < 1 || max_pairs > MAXB) return fail(STITCH_ERR_ARG, "plan_create: max_pairs must be 1..%d", MAXB);
    *plan_out = nullptr;
    int rc = need_device();
    if (rc) return rc;
    stitch_blend_opts o;
    stitch_blend_opts_default(&o);
    if (opts) o = *opts;
    if (o.blur_kind < 0 || o.blur_kind > 1 || o.level_rule < 0 || o.level_rule > 1 || o.seam_rule < 0 || o.seam_rule > 1 ||
        !(o.sigma >= 0))
        return fail(STITCH_ERR_ARG, "plan_create: bad blend options");
    int lw[32], lh[32];
    const int L = pyramid_levels(cw, ch, o.level_rule, lw, lh);
    if (L < 0) return L;

    stitch_plan* p = new stitch_plan();
    p->cw = cw;
    p->ch = ch;
    p->L = L;
    p->cap = max_pairs;
    const size_t B = (size_t)max_pairs;
    p->opts = o;
    p->vvk = make_vvk(o.sigma);
    p->drk = make_drk(o.sigma);
    const Tuning tn = Tuning::from_env();
    p->tune = tn;
    p->no_fuse = tn.no_fuse != 0;
    p->blur_skip = o.blur_kind == 0 ? (o.sigma < 0.5f) : (o.sigma < 0.1f);  // CImg.h:35051 / :34800
    if (hipGetDevice(&p->device) != hipSuccess) {
        delete p;
        return fail(STITCH_ERR_HIP, "hipGetDevice failed");
    }
    // carve one arena
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o2 = off;
        off += align256(bytes);
        return o2;
    };
    size_t g_off[32], e_off[32], ix_off[32], ax_off[32], iy_off[32], ay_off[32];
    for (int l = 0; l < L; ++l) {
        Level& v = p->lv[l];
        v.w = lw[l];
        v.h = lh[l];
        v.pitch = level_pitch(v.w, tn.pitch_pad);
        v.ps = (size_t)v.pitch * v.h;
        g_off[l] = take(sizeof(float) * (v.ps * 7 * B + (size_t)v.pitch * 64));
        e_off[l] = l >= 1 || L == 1 ? take(sizeof(float) * v.ps * 3 * B) : 0;
        if (l + 1 < L) {
            ix_off[l] = take(sizeof(int32_t) * v.w);
            ax_off[l] = take(sizeof(double) * v.w);
            iy_off[l] = take(sizeof(int32_t) * v.h);
            ay_off[l] = take(sizeof(double) * v.h);
        }
    }
    const Level& v0 = p->lv[0];
    const size_t t_bytes = sizeof(float) * (v0.ps * 7 * B + (size_t)v0.pitch * 64);
    const size_t t_off = take(t_bytes);
    const size_t t2_off = o.blur_kind == 1 ? take(t_bytes) : 0;
    // wavefront sweep: enabled with STITCH_WAVEFRONT=<levels> (Van Vliet only); granule buffers sized for level 0
    int wf_levels = 0;
    if (tn.wavefront >= 0)
        wf_levels = std::min(4, tn.wavefront);
    else if (max_pairs >= 2 || 7L * ((lh[0] + TS - 1) / TS) >= 800)
        // auto: the band pipeline pays where a launch has many bands in flight (planes x 64-row bands of level 0: 896 for two
        // 6144x4096 pairs, 1792 for one 24576x16384 pair) to hide its fill (bands x hand-off latency) and the levels have
        // enough tiles to stream; a lone 6144x4096 pair (448 bands) is faster unfused
        while (wf_levels < 2 && wf_levels < L - 1 && lw[wf_levels] >= 1024 && lh[wf_levels] >= 1024) ++wf_levels;
    if (o.blur_kind != 0 || tn.no_fuse || o.sigma < 0.5f) wf_levels = 0;
    wf_levels = std::min(wf_levels, L - 1);
    while (wf_levels > 0 && (lw[wf_levels - 1] < 2 || lh[wf_levels - 1] < 2)) --wf_levels;
    const int NC0 = (v0.w + TS - 1) / TS;
    // (a plan without fused-sweep levels keeps them for k_vv_xby_m: one pair in flight, from STITCH_XBYM_MPIX megapixels per level)
    const bool wf_alloc = wf_levels > 0 || (o.blur_kind == 0 && !tn.no_fuse && o.sigma >= 0.5f && L >= 2);
    const size_t yg_off = wf_alloc ? take(sizeof(u64) * B * 7 * NC0 * WF_GRAN * WAVE) : 0;
    const size_t wfc_off = wf_alloc ? take(sizeof(unsigned) * (WF_CTRL_WORDS + WF_STICKY_WORDS)) : 0;
    const size_t zi_off = take((size_t)B * ((v0.h + TS - 1) / TS) * ((v0.w + TS - 1) / TS));
    const size_t zt_off = wf_levels ? take((size_t)7 * B * ((v0.h + TS - 1) / TS) * ((v0.w + TS - 1) / TS)) : 0;
    int recompute = 0;  // opt-in: level 0 re-run from the frames measured 3 % slower, the plane levels within noise
    if (tn.recompute >= 0) recompute = wf_levels > 0 ? std::min(2, tn.recompute) : 0;
    const size_t ck_off = recompute ? take(sizeof(double) * 3 * NC0 * 7 * B * (size_t)(v0.h + 64)) : 0;
    // x-sweep state [4][lines] followed by the y state the wavefront kernel leaves [4][planes][pitch]
    const size_t state_n = 4 * 7 * B * (size_t)(v0.h + 64) + 4 * 7 * B * (size_t)std::max(v0.h + 64, v0.pitch);
    const size_t st_off = take(sizeof(double) * state_n);
    const size_t seam_off = take(sizeof(SeamDev) * B);
    const size_t side_off = take(sizeof(float) * B * v0.pitch);
    const size_t zp_off = take(4096);  // zeros for good (k_vv_xby_m's loader reads a flagged tile from here)
    p->arena_bytes = off;
    if (hipMalloc(&p->arena, off) != hipSuccess) {
        (void)hipGetLastError();
        delete p;
        return fail(STITCH_ERR_HIP, "plan_create: hipMalloc of %zu bytes failed", off);
    }
    char* base = static_cast<char*>(p->arena);
    for (int l = 0; l < L; ++l) {
        Level& v = p->lv[l];
        v.g = reinterpret_cast<float*>(base + g_off[l]);
        v.e = (l >= 1 || L == 1) ? reinterpret_cast<float*>(base + e_off[l]) : nullptr;
        if (l + 1 < L) {
            v.ix = reinterpret_cast<int32_t*>(base + ix_off[l]);
            v.ax = reinterpret_cast<double*>(base + ax_off[l]);
            v.iy = reinterpret_cast<int32_t*>(base + iy_off[l]);
            v.ay = reinterpret_cast<double*>(base + ay_off[l]);
        }
    }
    p->T = reinterpret_cast<float*>(base + t_off);
    p->T2 = o.blur_kind == 1 ? reinterpret_cast<float*>(base + t2_off) : nullptr;
    p->state = reinterpret_cast<double*>(base + st_off);
    p->d_seam = reinterpret_cast<SeamDev*>(base + seam_off);
    p->wf_levels = wf_levels;
    if (wf_alloc) {
        p->wf_yg = reinterpret_cast<u64*>(base + yg_off);
        p->wf_yg_bytes = sizeof(u64) * B * 7 * NC0 * WF_GRAN * WAVE;
        p->wf_ctrl = reinterpret_cast<unsigned*>(base + wfc_off);
        if (tn.xbyf_spin_limit >= 0) p->wf_spin_limit = (unsigned)tn.xbyf_spin_limit;
    }
    if (std::getenv("STITCH_D7_STAMP")) {
        p->d7_dbg_level = atoi(std::getenv("STITCH_D7_STAMP"));
        const size_t nb = sizeof(unsigned long long) * (7 * ((size_t)(v0.h + YCH - 1) / YCH + 1) + 8);
        if (hipMalloc((void**)&p->d7_dbg, nb) != hipSuccess || hipMemset(p->d7_dbg, 0, nb) != hipSuccess) {
            (void)hipGetLastError();
            p->d7_dbg = nullptr;
        }
    }
    if (wf_alloc && std::getenv("STITCH_XBYM_STAMP") &&
        (hipMalloc((void**)&p->xy_dbg, sizeof(unsigned long long) * 5 * NC0) != hipSuccess || hipMemset(p->xy_dbg, 0, sizeof(unsigned long long) * 5 * NC0) != hipSuccess)) {
        (void)hipGetLastError();
        p->xy_dbg = nullptr;
    }
    if (wf_levels) {
        p->zt = reinterpret_cast<uint8_t*>(base + zt_off);
        p->zi = reinterpret_cast<uint8_t*>(base + zi_off);
        p->recompute = recompute;
        if (recompute) p->ckpt = reinterpret_cast<double*>(base + ck_off);
        if (tn.xbyf_wgs >= 0) p->wf_max_wgs = std::max(1, tn.xbyf_wgs);
        if (tn.xbyf_spin_limit >= 0) p->wf_spin_limit = (unsigned)tn.xbyf_spin_limit;
        // diagnostic build: one record per persistent workgroup, sized from the workgroup count actually used
        if (tn.stamp &&
            (hipMalloc((void**)&p->wf_dbg, sizeof(unsigned long long) * (size_t)p->wf_max_wgs * 8) != hipSuccess ||
             hipMemset(p->wf_dbg, 0, sizeof(unsigned long long) * (size_t)p->wf_max_wgs * 8) != hipSuccess)) {
            (void)hipGetLastError();
            p->wf_dbg = nullptr;
        }
        if (tn.xbyf_early >= 0) p->wf_early_read = tn.xbyf_early != 0;
        p->zero_tiles = !tn.no_zero_tiles;  // A/B and tests: move the zeros like any other sample
    }
    p->side = reinterpret_cast<float*>(base + side_off);
    p->zero_page = reinterpret_cast<const float*>(base + zp_off);
    // implicit level-0 mask: needs both Van Vliet sweeps at level 0 (the x sweeps then run per-plane bands, Bands, at any height;
    // STITCH_GATE64=1 restores the old restriction to heights that are multiples of 64 for A/B runs)
    p->mask_opt = !p->no_fuse && o.blur_kind == 0 && !p->blur_skip && L >= 2 && v0.w > 1 && v0.h > 1 && !(tn.gate64 && v0.h % 64);
    {
        if (tn.crows_l0 >= 0) p->crows_l0 = std::max(1, tn.crows_l0);
        if (tn.crows_ln >= 0) p->crows_ln = std::max(1, tn.crows_ln);
        if (tn.collapse4 >= 0) p->collapse4 = tn.collapse4 != 0;
        p->src_fuse = p->mask_opt && !tn.no_src_fuse;  // STITCH_NO_SRC_FUSE: A/B and tests, keep S1 as its own kernel (k_compose)
    }
    // coarse levels in one launch (k_coarse): from the first level l >= 1 whose sides are both at most STITCH_COARSE (default 40;
    // 0 = never), when at least two levels qualify.  One workgroup -- one CU -- per pair: measured on the reference's 1081 x 527
    // panorama, the levels from 33 x 16 down take 65 us in k_coarse against 124 us as 31 launches of their own (4.5 us of
    // start-up each, 15 us for a collapse level), but 67 x 32 already costs more in one CU (the collapse's ~220 double-precision
    // operations per pixel) than spread over the chip.  Van Vliet only.
    {
        const int thr = std::min(tn.coarse >= 0 ? tn.coarse : 40, CO_MAXSIDE);
        int lc = 1;
        while (lc < L && std::max(lw[lc], lh[lc]) > thr) ++lc;
        if (thr > 0 && o.blur_kind == 0 && !p->blur_skip && lc <= L - 2 && L - lc <= CO_MAXL) p->coarse_from = lc;
        if (p->coarse_from && std::getenv("STITCH_COARSE_STAMP") &&
            (hipMalloc((void**)&p->co_dbg, sizeof(unsigned long long) * CO_MAXL * 8) != hipSuccess ||
             hipMemset(p->co_dbg, 0, sizeof(unsigned long long) * CO_MAXL * 8) != hipSuccess)) {
            (void)hipGetLastError();
            p->co_dbg = nullptr;
        }
    }
    // the slack rows and pitch padding are read by partial tiles: give them defined (zero) contents once
    if (hipMemset(p->arena, 0, off) != hipSuccess || hipHostMalloc((void**)&p->h_seam, sizeof(SeamDev) * B) != hipSuccess) {
        stitch_plan_destroy(p);
        return fail(STITCH_ERR_HIP, "plan_create: workspace initialisation failed");
    }
#ifdef STITCH_ZP_POISON  // (defined only to prove that a test reads flagged tiles from here: the page is then anything but zero)
    (void)hipMemset(base + zp_off, 0x3f, 4096);
#endif
    // resize tables (CImg.h:29625-29637), computed on the host once per plan
    for (int l = 0; l + 1 < L; ++l) {
        Level& v = p->lv[l];
        std::vector<int32_t> idx;
        std::vector<double> al;
        expand_table(lw[l + 1], v.w, idx, al);
        regular_range(idx, v.w, lw[l + 1], lh[l + 1], &v.c4_xa, &v.c4_xb);
        {
            int gxb = 0;
            general_range(idx, v.w, lw[l + 1], lh[l + 1], &gxb);
            // STITCH_C4_GEN: 0 never, 1 where it gains two blocks or more, 2 / unset wherever it gains a block
            v.c4_gen = tn.c4_gen != 0 && gxb >= (v.c4_xb - v.c4_xa) + (tn.c4_gen == 1 ? 512 : 256);
            if (v.c4_gen) v.c4_xa = 0, v.c4_xb = gxb;
        }
        (void)hipMemcpy(v.ix, idx.data(), sizeof(int32_t) * v.w, hipMemcpyHostToDevice);
        (void)hipMemcpy(v.ax, al.data(), sizeof(double) * v.w, hipMemcpyHostToDevice);
        expand_table(lh[l + 1], v.h, idx, al);
        (void)hipMemcpy(v.iy, idx.data(), sizeof(int32_t) * v.h, hipMemcpyHostToDevice);
        if (hipMemcpy(v.ay, al.data(), sizeof(double) * v.h, hipMemcpyHostToDevice) != hipSuccess) {
            stitch_plan_destroy(p);
            return fail(STITCH_ERR_HIP, "plan_create: table upload failed");
        }
    }
    std::memset(p->h_seam, 0, sizeof(SeamDev) * B);
    if (wf_alloc) {
        if (hipHostMalloc((void**)&p->h_wf_abort, sizeof(unsigned)) != hipSuccess) {
            stitch_plan_destroy(p);
            return fail(STITCH_ERR_HIP, "plan_create: pinned allocation failed");
        }
        *p->h_wf_abort = 0;
    }
    *plan_out = p;
    return STITCH_OK;
}

void stitch_plan_destroy(stitch_plan* p) {
    if (!p) return;
    if (p->pending && p->last_stream != nullptr) (void)hipStreamSynchronize(p->last_stream);
    else if (p->pending) (void)hipDeviceSynchronize();
    for (auto& r : p->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : p->free_events) (void)hipEventDestroy(e);
    if (p->arena) (void)hipFree(p->arena);
    if (p->h_seam) (void)hipHostFree(p->h_seam);
    if (p->h_wf_abort) (void)hipHostFree(p->h_wf_abort);
    if (p->d7_dbg) {
        const int nc = p->d7_dbg_chunks;
        std::vector<unsigned long long> h((size_t)7 * std::max(nc, 1) + 8);
        if (nc > 8 && hipMemcpy(h.data(), p->d7_dbg, sizeof(unsigned long long) * (7 * nc + 6), hipMemcpyDeviceToHost) == hipSuccess) {
            static const char* nm[7] = {"loader", "chain 0", "chain 1", "cons 0", "cons 1", "cons 2", "cons 3"};
            const unsigned long long t0 = h[0];
            std::fprintf(stderr, "[k_vv_y_bwd_dec7 stamps, level %d, %d chunks, middle workgroup of plane 0: ticks since the loader's first chunk, every %d-th chunk; then ticks per chunk over the middle half]\n",
                         p->d7_dbg_level, nc, std::max(1, nc / 16));
            for (int wv = 0; wv < 7; ++wv) {
                std::fprintf(stderr, "  %-8s", nm[wv]);
                const int c0 = wv >= 3 ? wv - 3 : 0, step = wv >= 3 ? 4 : 1;  // consumer c stamps chunks c, c+4, ...
                for (int j = 0; j < nc; j += std::max(1, nc / 16)) {
                    const int jj = c0 + (j / step) * step;
                    if (jj < nc) std::fprintf(stderr, " %7.0f", (double)h[(size_t)wv * nc + jj] - (double)t0);
                }
                const int ja = c0 + (nc / 4 / step) * step, jb = c0 + (3 * nc / 4 / step) * step;
                if (jb > ja && jb < nc) std::fprintf(stderr, " | %.1f per chunk", ((double)h[(size_t)wv * nc + jb] - (double)h[(size_t)wv * nc + ja]) / (jb - ja));
                std::fprintf(stderr, "\n");
            }
            if (h[(size_t)7 * nc + 1])
                std::fprintf(stderr, "  chain 0, ticks per chunk by segment: top + prefetch %.0f | rows 0-3 %.0f | release %.0f | rows 4-7 %.0f | write back %.0f | tail + back edge %.0f\n",
                             (double)h[(size_t)7 * nc + 0] / nc, (double)h[(size_t)7 * nc + 1] / nc, (double)h[(size_t)7 * nc + 2] / nc, (double)h[(size_t)7 * nc + 3] / nc,
                             (double)h[(size_t)7 * nc + 4] / nc, (double)h[(size_t)7 * nc + 5] / nc);
        }
        (void)hipFree(p->d7_dbg);
    }
    if (p->xy_dbg) {
        const int nc = p->xy_dbg_nc;
        std::vector<unsigned long long> h((size_t)5 * std::max(nc, 1));
        if (nc > 0 && hipMemcpy(h.data(), p->xy_dbg, sizeof(unsigned long long) * 5 * nc, hipMemcpyDeviceToHost) == hipSuccess) {
            static const char* nm[5] = {"x chain", "y chain", "loader", "storer", "courier"};
            const unsigned long long t0 = h[2 * (size_t)nc];  // the loader's first tile
            std::fprintf(stderr, "[k_vv_xby_m stamps, middle band of plane 0, last level-0 launch: ticks since the loader's first tile / 1000, per tile]\n");
            for (int wv = 0; wv < 5; ++wv) {
                std::fprintf(stderr, "  %-8s", nm[wv]);
                for (int j = 0; j < nc; j += std::max(1, nc / 24)) std::fprintf(stderr, " %6.1f", ((double)h[(size_t)wv * nc + j] - (double)t0) * 1e-3);
                std::fprintf(stderr, " | last %6.1f\n", ((double)h[(size_t)wv * nc + nc - 1] - (double)t0) * 1e-3);
            }
        }
        (void)hipFree(p->xy_dbg);
    }
    if (p->co_dbg) {
        unsigned long long h[CO_MAXL * 8];
        if (hipMemcpy(h, p->co_dbg, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            // s_memtime counts at 100 MHz: 10 ns per tick
            const int nl = p->L - p->coarse_from;
            std::fprintf(stderr, "[k_coarse stamps, pair 0 of the last launch, us] levels %d..%d:", p->coarse_from, p->L - 1);
            for (int i = 0; i + 1 < nl; ++i) {
                const unsigned long long t0 = i == 0 ? h[0] : h[(i - 1) * 8 + 3];
                std::fprintf(stderr, " | %dx%d x %.1f y %.1f dec %.1f", p->lv[p->coarse_from + i].w, p->lv[p->coarse_from + i].h, (h[i * 8 + 1] - t0) * 0.01,
                             (h[i * 8 + 2] - h[i * 8 + 1]) * 0.01, (h[i * 8 + 3] - h[i * 8 + 2]) * 0.01);
            }
            std::fprintf(stderr, " | top %.1f | collapse", (h[(nl - 1) * 8 + 4] - h[(nl - 2) * 8 + 3]) * 0.01);
            unsigned long long prev = h[(nl - 1) * 8 + 4];
            for (int i = nl - 2; i >= 0; --i) {
                std::fprintf(stderr, " %.1f", (h[i * 8 + 5] - prev) * 0.01);
                prev = h[i * 8 + 5];
            }
            std::fprintf(stderr, " | total %.1f\n", (prev - h[0]) * 0.01);
        }
        (void)hipFree(p->co_dbg);
    }
    if (p->wf_dbg) {
        const size_t nwg = (size_t)p->wf_max_wgs;
        std::vector<unsigned long long> hsum(nwg * 8);
        if (hipMemcpy(hsum.data(), p->wf_dbg, sizeof(unsigned long long) * nwg * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double tot[8] = {0};
            for (size_t i = 0; i < nwg; ++i)
                for (int j = 0; j < 8; ++j) tot[j] += (double)hsum[i * 8 + j];
            static const char* nm[8] = {"claim", "tile-fetch", "x-sweep", "y-wait", "y-sweep", "store", "-", "-"};
            double all = 0;
            for (int j = 0; j < 6; ++j) all += tot[j];
            std::fprintf(stderr, "[wavefront stamps, last level-0 launch, share of workgroup time]");
            for (int j = 0; j < 6; ++j) std::fprintf(stderr, " %s %.1f%%", nm[j], 100.0 * tot[j] / all);
            std::fprintf(stderr, " | mean cycles per workgroup %.0f\n", all / (double)nwg);
        }
        (void)hipFree(p->wf_dbg);
    }
    delete p;
}

size_t stitch_plan_workspace_bytes(const stitch_plan* p) { return p ? p->arena_bytes : 0; }
const void* stitch_plan_workspace_base(const stitch_plan* p) { return p ? p->arena : nullptr; }

int stitch_plan_levels(const stitch_plan* p, int* level_w, int* level_h) {
    if (!p) return fail(STITCH_ERR_ARG, "null plan");
    for (int l = 0; l < p->L; ++l) {
        if (level_w) level_w[l] = p->lv[l].w;
        if (level_h) level_h[l] = p->lv[l].h;
    }
    return p->L;
}

int stitch_dev_blend_u8(stitch_plan* plan, const uint8_t* d_a, const uint8_t* d_b, uint8_t* d_out, void* stream) {
    return dev_blend(plan, d_a, d_b, d_out, stream);
}
int stitch_dev_blend_f32(stitch_plan* plan, const float* d_a, const float* d_b, float* d_out, void* stream) {
    return dev_blend(plan, d_a, d_b, d_out, stream);
}
int stitch_dev_pair_u8(stitch_plan* plan, const uint8_t* d_frame, int fw, int fh, const double p[8], float offx, float offy,
                       const uint8_t* d_mosaic, int mw, int mh, int ox, int oy, uint8_t* d_out, void* stream) {
    return dev_pair(plan, d_frame, fw, fh, p, offx, offy, d_mosaic, mw, mh, ox, oy, d_out, stream);
}
int stitch_dev_pair_f32(stitch_plan* plan, const float* d_frame, int fw, int fh, const double p[8], float offx, float offy,
                        const float* d_mosaic, int mw, int mh, int ox, int oy, float* d_out, void* stream) {
    return dev_pair(plan, d_frame, fw, fh, p, offx, offy, d_mosaic, mw, mh, ox, oy, d_out, stream);
}

int stitch_plan_status_at(stitch_plan* p, int index, stitch_seam* seam_out) {
    if (!p) return fail(STITCH_ERR_ARG, "null plan");
    if (index < 0 || index >= std::max(1, p->last_n))
        return fail(STITCH_ERR_ARG, "status: index %d outside the %d pair(s) of the plan's last call", index, p->last_n);
    if (p->pending) {
        HIPCHK(hipStreamSynchronize(p->last_stream));
        p->pending = false;
    }
    if (p->h_wf_abort && *p->h_wf_abort != p->wf_abort_seen)
        return fail(STITCH_ERR_HIP,
                    "fused sweep: %u hand-off wait(s) timed out in this or an earlier queued call on the plan (every result since the "
                    "last stitch_plan_clear_fault is invalid)",
                    *p->h_wf_abort - p->wf_abort_seen);
    const SeamDev& sd = p->h_seam[index];
    seam_to_public(sd, seam_out);
    if (sd.status == -2) return fail(STITCH_ERR_EMPTY_MIDROW, "blend: channel 0 of a's middle row is empty (pair %d)", index);
    if (sd.status == -3) return fail(STITCH_ERR_ZERO_OVERLAP, "blend: a and b do not overlap on the middle row (pair %d)", index);
    return STITCH_OK;
}

int stitch_plan_status(stitch_plan* p, stitch_seam* seam_out) { return stitch_plan_status_at(p, 0, seam_out); }

int stitch_plan_set_handoff_spin_limit(stitch_plan* p, unsigned polls) {
    if (!p) return fail(STITCH_ERR_ARG, "null plan");
    p->wf_spin_limit = polls;
    return STITCH_OK;
}

int stitch_plan_clear_fault(stitch_plan* p) {
    if (!p) return fail(STITCH_ERR_ARG, "null plan");
    if (p->pending) {
        HIPCHK(hipStreamSynchronize(p->last_stream));
        p->pending = false;
    }
    if (p->h_wf_abort) p->wf_abort_seen = *p->h_wf_abort;
    return STITCH_OK;
}

int stitch_plan_capacity(const stitch_plan* p) { return p ? p->cap : 0; }
int stitch_plan_coarse_from(const stitch_plan* p) { return p ? p->coarse_from : 0; }
int stitch_plan_fast_paths(const stitch_plan* p) {
    if (!p) return 0;
    int f = 0;
    if (p->mask_opt) f |= STITCH_FAST_IMPLICIT_MASK;
    if (p->src_fuse) f |= STITCH_FAST_SOURCE_FUSED;
    if (p->wf_levels > 0) f |= STITCH_FAST_FUSED_SWEEP;
    if (p->wf_levels > 0 && p->zero_tiles && !p->no_fuse && ((p->lv[0].w & 1) == 0 || (p->lv[0].w >= 3 && p->tune.odd_dec != 0)) &&
        (p->lv[0].h + TS - 1) / TS <= 256 && !(p->tune.gate64 && (p->lv[0].h % TS || (p->lv[0].w & 1))))
        f |= STITCH_FAST_ZERO_TILES;
    if (!p->no_fuse && p->opts.blur_kind == 0 && !p->blur_skip && p->L >= 2 && ((p->lv[0].w & 1) == 0 || (p->lv[0].w >= 3 && p->tune.odd_dec != 0)) && p->lv[0].h > 1)
        f |= STITCH_FAST_FUSED_DECIMATE;
    if (p->coarse_from > 0) f |= STITCH_FAST_COARSE_LEVELS;
    return f;
}
int stitch_plan_fused_sweep_levels(const stitch_plan* p) { return p ? p->wf_levels : 0; }
int stitch_plan_collapse_range(const stitch_plan* p, int level, int* xa, int* xb, int* per_lane_taps) {
    if (!p || level < 0 || level + 1 >= p->L || !xa || !xb || !per_lane_taps) return fail(STITCH_ERR_ARG, "plan_collapse_range: bad argument");
    const Level& v = p->lv[level];
    *xa = p->collapse4 ? v.c4_xa : 0;
    *xb = p->collapse4 ? v.c4_xb : 0;
    *per_lane_taps = p->collapse4 ? v.c4_gen : 0;
    return STITCH_OK;
}
int stitch_plan_call_forms(const stitch_plan* p, int n_pairs) {
    if (!p || n_pairs < 1 || n_pairs > p->cap) return 0;
    int f = 0;
    if (p->src_fuse && src_fused_call(p, n_pairs)) f |= STITCH_FAST_SOURCE_FUSED;
    if (p->wf_levels > 0 && p->opts.blur_kind == 0 && !p->blur_skip && fused_sweep_call(p, n_pairs)) f |= STITCH_FAST_FUSED_SWEEP;
    if (lone_fused_level(p, n_pairs, 0)) f |= STITCH_FAST_FUSED_SWEEP;  // one pair in flight: k_vv_xby_m
    return f;
}

int stitch_dev_pairs_u8(stitch_plan* plan, const stitch_pair_desc* pairs, int n, void* stream) {
    return dev_pairs<uint8_t>(plan, pairs, n, stream);
}
int stitch_dev_pairs_f32(stitch_plan* plan, const stitch_pair_desc* pairs, int n, void* stream) {
    return dev_pairs<float>(plan, pairs, n, stream);
}

int stitch_plan_set_profiling(stitch_plan* p, int enabled) {
    if (!p) return fail(STITCH_ERR_ARG, "null plan");
    p->profiling = enabled != 0;
    p->prof_only = -1;
    return STITCH_OK;
}

int stitch_plan_set_profiling_kernel(stitch_plan* p, int kernel_id) {
    if (!p) return fail(STITCH_ERR_ARG, "null plan");
    if (kernel_id < 0 || kernel_id >= STITCH_K_COUNT) return fail(STITCH_ERR_ARG, "bad kernel id %d", kernel_id);
    p->profiling = true;
    p->prof_only = kernel_id;
    return STITCH_OK;
}

int stitch_plan_read_profile(stitch_plan* p, double stage_ms[STITCH_K_COUNT], int stage_launches[STITCH_K_COUNT],
                             double level0_ms[STITCH_K_COUNT]) {
    if (!p) return fail(STITCH_ERR_ARG, "null plan");
    if (p->pending) {
        HIPCHK(hipStreamSynchronize(p->last_stream));
        p->pending = false;
    }
    for (int i = 0; i < STITCH_K_COUNT; ++i) {
        if (stage_ms) stage_ms[i] = 0;
        if (stage_launches) stage_launches[i] = 0;
        if (level0_ms) level0_ms[i] = 0;
    }
    for (auto& r : p->recs) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        if (stage_ms) stage_ms[r.stage] += ms;
        if (stage_launches) stage_launches[r.stage] += 1;
        if (level0_ms && r.level == 0) level0_ms[r.stage] += ms;
        p->free_events.push_back(r.a);
        p->free_events.push_back(r.b);
    }
    p->recs.clear();
    return STITCH_OK;
}

int stitch_dev_equalize_u8(uint8_t* d_img, int w, int h, int32_t* d_hist256, void* stream) {
    return dev_equalize_impl(d_img, w, h, d_hist256, false, 0, 1, stream);
}

int stitch_dev_lummix_u8(uint8_t* d_result, const uint8_t* d_equalized, int w, int h, double num, double den, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_result || !d_equalized || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "lummix: null buffer or bad size %dx%d", w, h);
    const size_t n = (size_t)w * h;
    k_lummix<<<eq_grid(n), 256, 0, as_stream(stream)>>>(d_result, d_equalized, n, mix_params(num, den));
    return launch_check("k_lummix");
}

int stitch_dev_finish_u8(uint8_t* d_result, int w, int h, double num, double den, int32_t* d_hist256, void* stream) {
    return dev_equalize_impl(d_result, w, h, d_hist256, true, num, den, stream);
}

// ---- SURVEY.md 8(f) rows 1 and 2 -------------------------------------------------------------------------------
int stitch_dev_gray_u8(const uint8_t* d_rgb, int w, int h, uint8_t* d_gray, float* d_gray_f32, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_rgb || (!d_gray && !d_gray_f32) || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "gray: bad argument");
    const size_t n = (size_t)w * h;
    k_gray<<<eq_grid(n), 256, 0, as_stream(stream)>>>(d_rgb, n, d_gray, d_gray_f32);
    return launch_check("k_gray");
}
// ---- l-alpha-beta colour transfer ---------------------------------------------------------------------------------
int stitch_dev_transfer_u8(const uint8_t* d_src, int sw, int sh, const uint8_t* d_tem, int tw, int th, uint8_t* d_out, float* d_stats12,
                           void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_src || !d_tem || !d_out || sw <= 0 || sh <= 0 || tw <= 0 || th <= 0) return fail(STITCH_ERR_ARG, "transfer: null buffer or bad size");
    if ((long long)sw * sh > 0x7fffffffLL || (long long)tw * th > 0x7fffffffLL)
        return fail(STITCH_ERR_ARG, "transfer: w*h overflows int (the reference's int product)");
    const size_t ns = (size_t)sw * sh, nt = (size_t)tw * th;
    TrK k{};
    k.a1 = (float)(1.0 / std::sqrt(3.0));
    k.b1 = (float)(1.0 / std::sqrt(6.0));
    k.c1 = (float)(1.0 / std::sqrt(2.0));
    k.a2 = (float)(std::sqrt(3.0) / 3.0);
    k.b2 = (float)(std::sqrt(6.0) / 6.0);
    k.c2 = (float)(std::sqrt(2.0) / 2.0);
    k.ln10 = 2.302585092994046;  // log(10)
    hipStream_t s = as_stream(stream);
    float* scratch = nullptr;
    HIPCHK(hipMallocAsync((void**)&scratch, sizeof(float) * (3 * (ns + nt) + 16), s));
    float *lab_s = scratch, *lab_t = scratch + 3 * ns, *stats = scratch + 3 * (ns + nt);
    k_tr_to_lab<<<eq_grid(ns), 256, 0, s>>>(d_src, ns, k, lab_s);
    k_tr_to_lab<<<eq_grid(nt), 256, 0, s>>>(d_tem, nt, k, lab_t);
    k_tr_stats<<<6, 64, 0, s>>>(lab_s, ns, (float)(sw * sh), lab_t, nt, (float)(tw * th), stats);
    k_tr_apply<<<eq_grid(ns), 256, 0, s>>>(lab_s, ns, stats, k, d_out);
    if (d_stats12) HIPCHK(hipMemcpyAsync(d_stats12, stats, sizeof(float) * 12, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipFreeAsync(scratch, s));
    return launch_check("transfer");
}
int stitch_transfer_u8(const uint8_t* src, int sw, int sh, const uint8_t* tem, int tw, int th, uint8_t* out, float stats_out[12]) {
    int rc = need_device();
    if (rc) return rc;
    if (!src || !tem || !out || sw <= 0 || sh <= 0 || tw <= 0 || th <= 0) return fail(STITCH_ERR_ARG, "transfer: null buffer or bad size");
    const size_t bs = (size_t)3 * sw * sh, bt = (size_t)3 * tw * th;
    DevBuf a, b, o, st;
    if ((rc = a.alloc(bs)) || (rc = b.alloc(bt)) || (rc = o.alloc(bs)) || (rc = st.alloc(sizeof(float) * 12))) return rc;
    HIPCHK(hipMemcpy(a.p, src, bs, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b.p, tem, bt, hipMemcpyHostToDevice));
    if ((rc = stitch_dev_transfer_u8(a.as<uint8_t>(), sw, sh, b.as<uint8_t>(), tw, th, o.as<uint8_t>(), st.as<float>(), nullptr))) return rc;
    HIPCHK(hipMemcpy(out, o.p, bs, hipMemcpyDeviceToHost));
    if (stats_out) HIPCHK(hipMemcpy(stats_out, st.p, sizeof(float) * 12, hipMemcpyDeviceToHost));
    return STITCH_OK;
}

// ---- BMP <-> planar RGB ------------------------------------------------------------------------------------------
static int le32(const uint8_t* p) { return (int)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
int stitch_bmp_parse(const uint8_t* f, size_t n, stitch_bmp_info* info) {
    // CImg.h:48401-48441, for the layouts this path meets; every quantity as the reference derives it
    if (!f || !info) return fail(STITCH_ERR_ARG, "bmp_parse: null argument");
    if (n < 54 || f[0] != 'B' || f[1] != 'M') return fail(STITCH_ERR_ARG, "bmp_parse: not a BMP file");
    int file_size = le32(f + 0x02);
    const int offset = le32(f + 0x0A), header_size = le32(f + 0x0E), dx = le32(f + 0x12), dy = le32(f + 0x16), compression = le32(f + 0x1E),
              bpp = f[0x1C] + (f[0x1D] << 8);
    if (!file_size || file_size == offset) file_size = (int)n;
    if (compression) return fail(STITCH_ERR_ARG, "bmp_parse: compressed BMP (compression=%d) is not supported", compression);
    if (bpp != 24 && bpp != 32) return fail(STITCH_ERR_ARG, "bmp_parse: only 24- and 32-bit BMP is supported (bpp=%d)", bpp);
    if (dx <= 0 || dy == 0 || dy == INT32_MIN) return fail(STITCH_ERR_ARG, "bmp_parse: bad size %d x %d", dx, dy);
    const int h = dy < 0 ? -dy : dy;
    const long long dx_bytes = (long long)dx * bpp / 8;
    const int align_bytes = (int)((4 - dx_bytes % 4) % 4);
    const unsigned long long want = (unsigned long long)h * (unsigned long long)(dx_bytes + align_bytes);
    const unsigned long long avail_hdr = (unsigned long long)(long long)file_size - (unsigned long long)(long long)offset;
    unsigned long long buf = std::min(want, avail_hdr);
    long long pos = 54;
    if (header_size > 40) pos += header_size - 40;
    const long long xoffset = (long long)offset - 14 - header_size;
    if (xoffset > 0) pos += xoffset;
    if (pos < 0 || (unsigned long long)pos > n) return fail(STITCH_ERR_ARG, "bmp_parse: pixel data starts beyond the file");
    buf = std::min(buf, (unsigned long long)n - (unsigned long long)pos);
    info->width = dx;
    info->height = h;
    info->bpp = bpp;
    info->top_down = dy < 0;
    info->data_pos = (uint64_t)pos;
    info->stride = (uint64_t)(dx_bytes + align_bytes);
    info->data_bytes = buf;
    return STITCH_OK;
}
size_t stitch_bmp_file_bytes(int w, int h) {
    if (w <= 0 || h <= 0) return 0;
    return 54 + (((size_t)3 * w + 3) & ~(size_t)3) * (size_t)h;
}
static int bmp_align(const void* p, unsigned long long pos) {
    const unsigned long long a = (unsigned long long)reinterpret_cast<uintptr_t>(p) + pos;
    return (a & 3) == 0 ? 4 : (a & 1) == 0 ? 2 : 1;
}
int stitch_dev_bmp_decode_u8(const uint8_t* d_file, size_t n, const stitch_bmp_info* bi, uint8_t* d_planar, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_file || !bi || !d_planar) return fail(STITCH_ERR_ARG, "bmp_decode: null argument");
    if (bi->width <= 0 || bi->height <= 0 || (bi->bpp != 24 && bi->bpp != 32) || bi->stride < (uint64_t)bi->width * (bi->bpp / 8) ||
        (bi->stride & 3) || bi->data_pos > n || bi->data_bytes > n - bi->data_pos)
        return fail(STITCH_ERR_ARG, "bmp_decode: inconsistent stitch_bmp_info (use stitch_bmp_parse)");
    BmpGeom g{};
    g.w = bi->width;
    g.h = bi->height;
    g.bp = bi->bpp / 8;
    g.top_down = bi->top_down;
    g.data_pos = bi->data_pos;
    g.stride = bi->stride;
    g.data_bytes = bi->data_bytes;
    g.planar_vec = (g.w % 4 == 0) && (reinterpret_cast<uintptr_t>(d_planar) % 4 == 0);
    const dim3 grid((g.w + BMP_SEG - 1) / BMP_SEG, g.h);
    hipStream_t s = as_stream(stream);
    switch (bmp_align(d_file, g.data_pos)) {  // stride and a segment's byte offset are multiples of 4
        case 4: k_bmp_decode<4><<<grid, 256, 0, s>>>(d_file, g, d_planar); break;
        case 2: k_bmp_decode<2><<<grid, 256, 0, s>>>(d_file, g, d_planar); break;
        default: k_bmp_decode<1><<<grid, 256, 0, s>>>(d_file, g, d_planar); break;
    }
    return launch_check("k_bmp_decode");
}
int stitch_dev_bmp_encode_u8(const uint8_t* d_planar, int w, int h, uint8_t* d_file, size_t cap, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    const size_t total = stitch_bmp_file_bytes(w, h);
    if (!d_planar || !d_file || !total) return fail(STITCH_ERR_ARG, "bmp_encode: bad argument");
    if (total > 0xffffffffULL) return fail(STITCH_ERR_ARG, "bmp_encode: %d x %d does not fit the format's 32-bit file size", w, h);
    if (cap < total) return fail(STITCH_ERR_ARG, "bmp_encode: buffer of %zu bytes, file needs %zu", cap, total);
    BmpGeom g{};
    g.w = w;
    g.h = h;
    g.bp = 3;
    g.stride = ((unsigned long long)3 * w + 3) & ~3ULL;
    g.planar_vec = (w % 4 == 0) && (reinterpret_cast<uintptr_t>(d_planar) % 4 == 0);
    BmpHeader hd{};  // CImg.h:52633-52664
    const unsigned buf_size = (unsigned)(total - 54), file_size = (unsigned)total;
    hd.b[0] = 'B';
    hd.b[1] = 'M';
    for (int i = 0; i < 4; ++i) {
        hd.b[0x02 + i] = (uint8_t)(file_size >> (8 * i));
        hd.b[0x12 + i] = (uint8_t)((unsigned)w >> (8 * i));
        hd.b[0x16 + i] = (uint8_t)((unsigned)h >> (8 * i));
        hd.b[0x22 + i] = (uint8_t)(buf_size >> (8 * i));
    }
    hd.b[0x0A] = 0x36;
    hd.b[0x0E] = 0x28;
    hd.b[0x1A] = 1;
    hd.b[0x1C] = 24;
    hd.b[0x27] = 0x1;
    hd.b[0x2B] = 0x1;
    const dim3 grid((unsigned)((g.stride + BMP_SEG * 3 - 1) / (BMP_SEG * 3)), h);
    hipStream_t s = as_stream(stream);
    switch (bmp_align(d_file, 54)) {
        case 4: k_bmp_encode<4><<<grid, 256, 0, s>>>(d_planar, g, hd, d_file); break;
        case 2: k_bmp_encode<2><<<grid, 256, 0, s>>>(d_planar, g, hd, d_file); break;
        default: k_bmp_encode<1><<<grid, 256, 0, s>>>(d_planar, g, hd, d_file); break;
    }
    return launch_check("k_bmp_encode");
}
int stitch_bmp_decode_u8(const uint8_t* file, size_t n, uint8_t* planar) {
    int rc = need_device();
    if (rc) return rc;
    stitch_bmp_info bi;
    if ((rc = stitch_bmp_parse(file, n, &bi))) return rc;
    if (!planar) return fail(STITCH_ERR_ARG, "bmp_decode: null output");
    const size_t out = (size_t)3 * bi.width * bi.height;
    DevBuf f, p;
    if ((rc = f.alloc(n)) || (rc = p.alloc(out))) return rc;
    HIPCHK(hipMemcpy(f.p, file, n, hipMemcpyHostToDevice));
    if ((rc = stitch_dev_bmp_decode_u8(f.as<uint8_t>(), n, &bi, p.as<uint8_t>(), nullptr))) return rc;
    HIPCHK(hipMemcpy(planar, p.p, out, hipMemcpyDeviceToHost));
    return STITCH_OK;
}
int stitch_bmp_encode_u8(const uint8_t* planar, int w, int h, uint8_t* file, size_t cap) {
    int rc = need_device();
    if (rc) return rc;
    const size_t total = stitch_bmp_file_bytes(w, h);
    if (!planar || !file || !total) return fail(STITCH_ERR_ARG, "bmp_encode: bad argument");
    if (cap < total) return fail(STITCH_ERR_ARG, "bmp_encode: buffer of %zu bytes, file needs %zu", cap, total);
    DevBuf f, p;
    if ((rc = f.alloc(total)) || (rc = p.alloc((size_t)3 * w * h))) return rc;
    HIPCHK(hipMemcpy(p.p, planar, (size_t)3 * w * h, hipMemcpyHostToDevice));
    if ((rc = stitch_dev_bmp_encode_u8(p.as<uint8_t>(), w, h, f.as<uint8_t>(), total, nullptr))) return rc;
    HIPCHK(hipMemcpy(file, f.p, total, hipMemcpyDeviceToHost));
    return STITCH_OK;
}
int stitch_dev_project_gray_u8(const uint8_t* d_src, int w, int h, float fov_deg, uint8_t* d_projected, uint8_t* d_gray,
                               float* d_gray_f32, void* stream) {
    return dev_project<uint8_t>(d_src, w, h, fov_deg, d_projected, stream, d_gray, d_gray_f32);
}
int stitch_gray_u8(const uint8_t* rgb, int w, int h, uint8_t* gray, float* gray_f32) {
    int rc = need_device();
    if (rc) return rc;
    if (!rgb || (!gray && !gray_f32) || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "gray: bad argument");
    const size_t n = (size_t)w * h;
    DevBuf d, g, f;
    if ((rc = d.alloc(n * 3)) || (rc = g.alloc(n)) || (rc = f.alloc(n * sizeof(float)))) return rc;
    HIPCHK(hipMemcpy(d.p, rgb, n * 3, hipMemcpyHostToDevice));
    if ((rc = stitch_dev_gray_u8(d.as<uint8_t>(), w, h, g.as<uint8_t>(), f.as<float>(), nullptr))) return rc;
    if (gray) HIPCHK(hipMemcpy(gray, g.p, n, hipMemcpyDeviceToHost));
    if (gray_f32) HIPCHK(hipMemcpy(gray_f32, f.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return STITCH_OK;
}
int stitch_project_gray_u8(const uint8_t* src, int w, int h, float fov_deg, uint8_t* projected, uint8_t* gray, float* gray_f32) {
    int rc = need_device();
    if (rc) return rc;
    if (!src || !projected || w <= 0 || h <= 0) return fail(STITCH_ERR_ARG, "project_gray: bad argument");
    const size_t n = (size_t)w * h;
    DevBuf s, d, g, f;
    if ((rc = s.alloc(n * 3)) || (rc = d.alloc(n * 3)) || (rc = g.alloc(n)) || (rc = f.alloc(n * sizeof(float)))) return rc;
    HIPCHK(hipMemcpy(s.p, src, n * 3, hipMemcpyHostToDevice));
    if ((rc = stitch_dev_project_gray_u8(s.as<uint8_t>(), w, h, fov_deg, d.as<uint8_t>(), g.as<uint8_t>(), f.as<float>(), nullptr)))
        return rc;
    HIPCHK(hipMemcpy(projected, d.p, n * 3, hipMemcpyDeviceToHost));
    if (gray) HIPCHK(hipMemcpy(gray, g.p, n, hipMemcpyDeviceToHost));
    if (gray_f32) HIPCHK(hipMemcpy(gray_f32, f.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return STITCH_OK;
}

// Canvas sizing of one stitch step, ImageProcess.cpp:206-216 with getMin/MaxX/YAfterWarping (:532-594): float
// arithmetic exactly as written there (the map itself in double, rounded to float per corner).  Host only.
static float map_x(const double p[8], float x, float y) { return (float)(p[0] * (double)x + p[1] * (double)y + p[2] * (double)x * (double)y + p[3]); }
static float map_y(const double p[8], float x, float y) { return (float)(p[4] * (double)x + p[5] * (double)y + p[6] * (double)x * (double)y + p[7]); }
int stitch_canvas_bbox(int fw, int fh, const double p_fwd[8], int result_w, int result_h, float* min_x, float* min_y, int* new_w,
                       int* new_h) {
    if (!p_fwd || !min_x || !min_y || !new_w || !new_h || fw <= 0 || fh <= 0 || result_w <= 0 || result_h <= 0)
        return fail(STITCH_ERR_ARG, "canvas_bbox: bad argument");
    const float cx[4] = {0.f, (float)(fw - 1), 0.f, (float)(fw - 1)}, cy[4] = {0.f, 0.f, (float)(fh - 1), (float)(fh - 1)};
    float mnx = map_x(p_fwd, cx[0], cy[0]), mxx = mnx, mny = map_y(p_fwd, cx[0], cy[0]), mxy = mny;
    for (int i = 1; i < 4; ++i) {  // strict comparisons in the reference's corner order (0,0) (w-1,0) (0,h-1) (w-1,h-1)
        const float X = map_x(p_fwd, cx[i], cy[i]), Y = map_y(p_fwd, cx[i], cy[i]);
        if (X < mnx) mnx = X;
        if (X > mxx) mxx = X;
        if (Y < mny) mny = Y;
        if (Y > mxy) mxy = Y;
    }
    mnx = (mnx < 0) ? mnx : 0;                                  // :207
    mny = (mny < 0) ? mny : 0;                                  // :209
    mxx = (mxx >= (float)result_w) ? mxx : (float)result_w;    // :211
    mxy = (mxy >= (float)result_h) ? mxy : (float)result_h;    // :213
    *min_x = mnx;
    *min_y = mny;
    *new_w = (int)std::ceil(mxx - mnx);  // :215 (float subtraction)
    *new_h = (int)std::ceil(mxy - mny);
    return STITCH_OK;
}
int stitch_step_geometry(int fw, int fh, const double p_fwd[8], int mw, int mh, stitch_step_geom* g) {
    if (!g) return fail(STITCH_ERR_ARG, "step_geometry: null output");
    int rc = stitch_canvas_bbox(fw, fh, p_fwd, mw, mh, &g->min_x, &g->min_y, &g->cw, &g->ch);
    if (rc) return rc;
    g->ox = (int)g->min_x;  // movingImageByOffset(result, b, min_x, min_y): float arguments to int parameters (ImageProcess.cpp:224)
    g->oy = (int)g->min_y;
    if (g->cw <= 0 || g->ch <= 0) return fail(STITCH_ERR_ARG, "step_geometry: empty canvas %d x %d", g->cw, g->ch);
    return STITCH_OK;
}
int stitch_dev_step_u8(const uint8_t* d_frame, int fw, int fh, const double p_fwd[8], const double p_bwd[8], const uint8_t* d_mosaic, int mw,
                       int mh, const stitch_blend_opts* opts, uint8_t* d_out, size_t out_capacity, stitch_step_geom* geom_out,
                       stitch_seam* seam_out, void* stream) {
    return dev_step<uint8_t>(d_frame, fw, fh, p_fwd, p_bwd, d_mosaic, mw, mh, opts, d_out, out_capacity, geom_out, seam_out, stream);
}
int stitch_dev_step_f32(const float* d_frame, int fw, int fh, const double p_fwd[8], const double p_bwd[8], const float* d_mosaic, int mw, int mh,
                        const stitch_blend_opts* opts, float* d_out, size_t out_capacity, stitch_step_geom* geom_out, stitch_seam* seam_out,
                        void* stream) {
    return dev_step<float>(d_frame, fw, fh, p_fwd, p_bwd, d_mosaic, mw, mh, opts, d_out, out_capacity, geom_out, seam_out, stream);
}
// updateFeaturesByHomography / updateFeaturesByOffset, ImageProcess.cpp:622-640, on arrays of keypoint coordinates
int stitch_map_points(float* x, float* y, int32_t* ix, int32_t* iy, int n, const double p_fwd[8], float offx, float offy) {
    if (!x || !y || !p_fwd || n < 0) return fail(STITCH_ERR_ARG, "map_points: bad argument");
    for (int i = 0; i < n; ++i) {
        const float cx = x[i], cy = y[i];
        x[i] = map_x(p_fwd, cx, cy) - offx;
        y[i] = map_y(p_fwd, cx, cy) - offy;
        if (ix) ix[i] = (int32_t)x[i];
        if (iy) iy[i] = (int32_t)y[i];
    }
    return STITCH_OK;
}
int stitch_shift_points(float* x, float* y, int32_t* ix, int32_t* iy, int n, int ox, int oy) {
    if (!x || !y || n < 0) return fail(STITCH_ERR_ARG, "shift_points: bad argument");
    for (int i = 0; i < n; ++i) {
        x[i] -= (float)ox;
        y[i] -= (float)oy;
        if (ix) ix[i] = (int32_t)x[i];
        if (iy) iy[i] = (int32_t)y[i];
    }
    return STITCH_OK;
}

int stitch_dev_synth_u8(uint8_t* d_dst, int w, int h, int frame_id, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_dst || w <= 0 || h <= 0 || w >= (1 << 18) || h >= (1 << 18)) return fail(STITCH_ERR_ARG, "synth: bad argument");
    k_synth<uint8_t><<<grid_xy(w, h, 3), 256, 0, as_stream(stream)>>>(d_dst, w, h, frame_id);
    return launch_check("k_synth");
}
int stitch_dev_synth_f32(float* d_dst, int w, int h, int frame_id, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_dst || w <= 0 || h <= 0 || w >= (1 << 18) || h >= (1 << 18)) return fail(STITCH_ERR_ARG, "synth: bad argument");
    k_synth<float><<<grid_xy(w, h, 3), 256, 0, as_stream(stream)>>>(d_dst, w, h, frame_id);
    return launch_check("k_synth");
}
int stitch_dev_check_fastdiv(float w, unsigned long long* tested, unsigned long long* mismatches) {
    int rc = need_device();
    if (rc) return rc;
    if (!(w >= 2.0f && w < 16777216.0f) || w != std::floor(w) || !tested || !mismatches) return fail(STITCH_ERR_ARG, "check_fastdiv: w must be an integer value in [2, 2^24)");
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, 2 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(d, 0, 2 * sizeof(unsigned long long)));
    k_check_fastdiv<<<4096, 256>>>(w, d, d + 1);
    unsigned long long h[2] = {0, 0};
    const hipError_t e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(STITCH_ERR_HIP, "check_fastdiv: %s", hipGetErrorString(e));
    *mismatches = h[0];
    *tested = h[1];
    return STITCH_OK;
}
int stitch_dev_check_collapse_taps(int w, int h, int probe, unsigned long long* compared, unsigned long long* mismatches) {
    int rc = need_device();
    if (rc) return rc;
    if (w < 256 || (w & 1) || h < 4 || (h & 1) || w > 16384 || h > 16384 || probe < 0 || probe > 1 || !compared || !mismatches)
        return fail(STITCH_ERR_ARG, "check_collapse_taps: even w >= 256, even h >= 4 (both <= 16384), probe 0 or 1");
    const int sw = w / 2, sh = h / 2, pitch = round_up(w, 64), spitch = round_up(sw, 64);
    const size_t ps = (size_t)pitch * h, sps = (size_t)spitch * sh;
    std::vector<int32_t> ix, iy;
    std::vector<double> ax, ay;
    expand_table(sw, w, ix, ax);
    expand_table(sh, h, iy, ay);
    int xa = 0, xb = 0, gxb = 0;
    regular_range(ix, w, sw, sh, &xa, &xb);
    general_range(ix, w, sw, sh, &gxb);
    if (gxb < 256) return fail(STITCH_ERR_ARG, "check_collapse_taps: no block of this width takes the four-column path");
    // level l: a, b and the mask; level l+1: G (a, b) and E.  probe 0: pseudo-random finite samples of both signs.  probe 1, the signed-zero
    // probe: G_l = -0, G_{l+1} = +0 (every Laplacian sample -0), E_{l+1} = -0 except the FIRST sample of every row = 5: a column whose first tap
    // is a row's last sample must come out -0 (the reference takes that sample twice: -0 + 0 * -0); had it taken the sample the 16-byte
    // window holds behind it (the next row's 5) the sum would be +0.
    std::vector<float> g(7 * ps + (size_t)pitch * 64, 0.f), gn(7 * sps + (size_t)spitch * 64, 0.f), en(3 * sps + (size_t)spitch * 64, 0.f);
    uint32_t lcg = 0x5717C4EDu ^ (uint32_t)(w * 31 + h);
    auto rnd = [&]() {
        lcg = lcg * 1664525u + 1013904223u;
        return ((int)(lcg >> 8) % 60001 - 30000) / 117.0f;
    };
    for (size_t q = 0; q < 7; ++q)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) g[q * ps + (size_t)y * pitch + x] = q == 6 ? (probe ? 0.5f : (float)((x * 7 + y * 3) % 11) / 10.f) : probe ? -0.0f : rnd();
    for (size_t q = 0; q < 7; ++q)
        for (int y = 0; y < sh; ++y)
            for (int x = 0; x < sw; ++x) gn[q * sps + (size_t)y * spitch + x] = probe ? 0.0f : rnd();
    for (size_t q = 0; q < 3; ++q)
        for (int y = 0; y < sh; ++y)
            for (int x = 0; x < spitch; ++x) en[q * sps + (size_t)y * spitch + x] = probe ? (x == 0 ? 5.0f : -0.0f) : (x < sw ? rnd() : 0.f);
    const size_t bytes[] = {g.size() * 4, gn.size() * 4, en.size() * 4, 3 * ps * 4, 3 * ps * 4, (size_t)w * 4, (size_t)w * 8, (size_t)h * 4, (size_t)h * 8};
    size_t off[10] = {0};
    for (int i = 0; i < 9; ++i) off[i + 1] = off[i] + align256(bytes[i]);
    char* base = nullptr;
    HIPCHK(hipMalloc((void**)&base, off[9] + 256));
    auto up = [&](int i, const void* src) { return hipMemcpy(base + off[i], src, bytes[i], hipMemcpyHostToDevice) == hipSuccess; };
    bool ok = up(0, g.data()) && up(1, gn.data()) && up(2, en.data()) && up(5, ix.data()) && up(6, ax.data()) && up(7, iy.data()) && up(8, ay.data()) &&
              hipMemset(base + off[3], 0xff, 2 * align256(bytes[3])) == hipSuccess;
    std::vector<float> o1(3 * ps), o2(3 * ps);
    if (ok) {
        const ExpandTab tb{(const int32_t*)(base + off[5]), (const double*)(base + off[6]), (const int32_t*)(base + off[7]), (const double*)(base + off[8])};
        const int crows = 8, strips = (h + crows - 1) / crows;
        for (int v = 0; v < 2; ++v) {  // v = 0: k_collapse4 with per-lane taps on [0, gxb), one column per work-item behind; v = 1: k_collapse everywhere
            OutPtrs<float> eo{};
            eo.p[0] = (float*)(base + off[3 + v]);
            CollapseArgs<float, false> A{(const float*)(base + off[0]), w, h, pitch, ps, (const float*)(base + off[1]), (const float*)(base + off[2]), sw, sh, spitch, sps,
                                         tb, eo, pitch, ps, nullptr, NoPairArgs{}, 0, crows, 0, v == 0 ? gxb : 0, 1, 0, 1};
            if (v == 0) {
                const int rest = pitch - gxb, nb4 = (gxb / 4 + WAVE - 1) / WAVE, ncb = (rest + C4_THREADS - 1) / C4_THREADS;
                k_collapse4<float, false, false, true><<<dim3(c4_padded_blocks(nb4, A.swizzle) + ncb * C4_SUB, strips, 1), C4_THREADS>>>(A, nb4, ncb);
            } else
                k_collapse<float, false><<<grid_xy(pitch, strips, 1), 256>>>(A);
        }
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(o1.data(), base + off[3], bytes[3], hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(o2.data(), base + off[4], bytes[4], hipMemcpyDeviceToHost) == hipSuccess;
    }
    const hipError_t e = ok ? hipSuccess : hipGetLastError();
    (void)hipFree(base);
    if (!ok) return fail(STITCH_ERR_HIP, "check_collapse_taps: %s", hipGetErrorString(e));
    unsigned long long n = 0, bad = 0;
    for (size_t q = 0; q < 3; ++q)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x, ++n) {
                uint32_t a, b;
                std::memcpy(&a, &o1[q * ps + (size_t)y * pitch + x], 4);
                std::memcpy(&b, &o2[q * ps + (size_t)y * pitch + x], 4);
                bad += a != b;
            }
    *compared = n;
    *mismatches = bad;
    return STITCH_OK;
}
int stitch_dev_quantize_u8(const float* d_src, uint8_t* d_dst, size_t n, void* stream) {
    int rc = need_device();
    if (rc) return rc;
    if (!d_src || !d_dst || n == 0) return fail(STITCH_ERR_ARG, "quantize: bad argument");
    k_quantize<<<eq_grid((n + 3) / 4), 256, 0, as_stream(stream)>>>(d_src, d_dst, n);
    return launch_check("k_quantize");
}

}  // extern "C"

#include "stitch_band.inc"
