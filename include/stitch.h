/* stitch.h -- C ABI of the MI355X-native stitching hot path (libstitch_hip.so).
 *
 * This is the drop-in boundary for the per-pixel path of chensh236/ComputerVisionImageStich2.  The reference
 * has no FFI: the path sits behind ordinary C++ functions on CImg<unsigned char>.  Each entry point below
 * names the reference interface it replaces (file:line under /root/reference); the C++ adaptor that
 * re-exposes the reference's own symbols on top of this ABI is
 * computervisionimagestich2_amd/adaptor/cimg_dropin.cpp.
 *
 * Conventions
 *   - Images are planar, channel-major buffers exactly as CImg lays them out (CImg.h:11787-11793):
 *     offset = x + y*W + c*W*H, 3 channels, so an adaptor passes img._data / _width / _height with no repack.
 *   - `_u8` entry points are the reference's own contract (CImg<unsigned char> in/out, results bit-identical
 *     to the reference).  `_f32` twins run the same arithmetic on float frames and skip the final
 *     float->uchar truncation (the metric's 4096x4096x3 f32 frames).
 *   - Plain entry points take HOST pointers, run on the current HIP device and return when the result is in
 *     the output buffer.  `stitch_dev_*` entry points take DEVICE pointers, enqueue on `stream`
 *     (a hipStream_t passed as void*) and return without synchronising.
 *   - Return value: STITCH_OK or a negative stitch_status; stitch_last_error() gives the text (thread local).
 *   - Every kernel is hand-written HIP for gfx950.  There is no CPU fallback: without a HIP device every
 *     compute entry point fails with STITCH_ERR_NO_DEVICE.
 */
#ifndef STITCH_H
#define STITCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STITCH_ABI_VERSION 5

typedef enum stitch_status {
    STITCH_OK = 0,
    STITCH_ERR_ARG = -1,          /* null pointer / non-positive size / bad option                              */
    STITCH_ERR_EMPTY_MIDROW = -2, /* channel 0 of a's middle row is all zero: the reference never terminates
                                     (ImageProcess.cpp:661)                                                      */
    STITCH_ERR_ZERO_OVERLAP = -3, /* a and b do not overlap on the middle row: the reference divides 0/0
                                     (ImageProcess.cpp:687)                                                      */
    STITCH_ERR_PYRAMID = -4,      /* a pyramid level would have a zero dimension (min side < 2^(levels-1))       */
    STITCH_ERR_HIP = -5,          /* a HIP runtime call failed; text in stitch_last_error()                      */
    STITCH_ERR_NO_DEVICE = -6     /* no HIP device visible                                                       */
} stitch_status;

/* Parameters of the multi-band blend.  stitch_blend_opts_default() = the root variant the oracle follows. */
typedef struct stitch_blend_opts {
    float sigma;    /* REDUCE blur sigma; reference: 2 (ImageProcess.cpp:709)                                    */
    int blur_kind;  /* 0 = Van Vliet, get_blur(2,true,true) (ImageProcess.cpp:709-714, CImg.h:35045-35091);
                       1 = Deriche, get_blur(2) (src/ex6/ImageProcess.cpp:702-705, CImg.h:34777-34869)           */
    int level_rule; /* 0 = floor(log2(max(w,h))) (ImageProcess.cpp:675-676); 1 = min (src/ex6 :662-665)          */
    int seam_rule;  /* 0 = channel 0 decides "non-empty", float ratios (ImageProcess.cpp:659-698);
                       1 = all three channels, double ratios (src/ex6/ImageProcess.cpp:641-698)                  */
} stitch_blend_opts;

/* What the seam scan found (ImageProcess.cpp:659-671,686-698). */
typedef struct stitch_seam {
    int32_t sum_a_x, n_a, sum_ov_x, n_ov; /* the four mid-row integers                                           */
    float ratio, ov;                      /* sum_a_x/n_a and sum_ov_x/n_ov                                       */
    int32_t branch;                       /* 0: mask = 1 where x < ov; 1: mask = 1 where x >= start              */
    int32_t start;                        /* (int)(ov + 1)                                                       */
} stitch_seam;

typedef struct stitch_plan stitch_plan; /* device workspace of one canvas size: pyramids, tables, scratch */

/* ---- library ------------------------------------------------------------------------------------------- */
int stitch_abi_version(void);
/* Process start-up (optional): puts GPU_MAX_HW_QUEUES=8 into the environment unless the caller has set it -- the HIP runtime
 * multiplexes streams onto 4 hardware queues by default and launch sequences that share a queue serialise (four sequences in
 * flight: 1.50 -> 1.31 ms per pair).  The runtime reads the variable when it initialises, so call this BEFORE the process's
 * first HIP call and before other threads exist (setenv is not thread-safe): the C++ adaptor does it in a static initialiser,
 * the Python package at import.  Returns 1 if it set the variable, 0 if it was set already.  (Up to ABI 4 a constructor of
 * the library did this at load time.) */
int stitch_init(void);
const char *stitch_last_error(void);
int stitch_device_count(void);       /* number of HIP devices, 0 if none */
int stitch_set_device(int ordinal);  /* hipSetDevice for the calling thread */
void stitch_blend_opts_default(stitch_blend_opts *o);
/* The host-buffer entry points below keep what they allocate between calls: idle workspaces (plans) in an LRU keyed by
 * device, canvas size and options (STITCH_PLAN_CACHE=<n> idle plans, default 8, 0 = none) and their device staging
 * buffers in the device's stream-ordered memory pool.  stitch_trim() releases all of it (current device). */
void stitch_trim(void);
/* Number of idle cached workspaces that a host-buffer call for this canvas and these options would take NOW -- the key is
 * (device, canvas, options, the tuning switches of the environment as they are at this moment: see "Tuning / A-B switches"
 * below), so a caller who changes a switch between two calls never meets a workspace built under the old setting.
 * fused_sweep_levels (optional) receives stitch_plan_fused_sweep_levels of the first match.  Negative on error. */
int stitch_plan_cache_query(int cw, int ch, const stitch_blend_opts *opts, int *fused_sweep_levels);
/* Pyramid shape for a canvas (ImageProcess.cpp:675-676,705-708).  Returns the level count (>0) or a status;
 * level_w/level_h (capacity 32) may be NULL. */
int stitch_pyramid_levels(int w, int h, int level_rule, int *level_w, int *level_h);

/* ---- host-buffer entry points (what the C++ adaptor binds) ---------------------------------------------- */
/* Projection::imageProjection + bilinearInterpolation, Projection.cpp:3-73.  fov_deg = ANGLE (Projection.h:13). */
int stitch_project_u8(const uint8_t *src, int w, int h, float fov_deg, uint8_t *dst);
int stitch_project_f32(const float *src, int w, int h, float fov_deg, float *dst);
/* ImageProcess::warpingImageByHomography (+ getX/YAfterWarping), ImageProcess.cpp:465-471,596-606.
 * p = {H00,H01,H02,H10,H11,H12,H20,H21} of `Homography` (ImageProcess.h:58-73).  The canvas is read-modify-
 * write: pixels whose source falls outside `src` keep the caller's value (the reference pre-zeroes it, :218). */
int stitch_warp_u8(const uint8_t *src, int sw, int sh, const double p[8], float offx, float offy, uint8_t *canvas,
                   int cw, int ch);
int stitch_warp_f32(const float *src, int sw, int sh, const double p[8], float offx, float offy, float *canvas,
                    int cw, int ch);
/* ImageProcess::movingImageByOffset, ImageProcess.cpp:608-620 (same read-modify-write rule). */
int stitch_move_u8(const uint8_t *src, int sw, int sh, int ox, int oy, uint8_t *canvas, int cw, int ch);
int stitch_move_f32(const float *src, int sw, int sh, int ox, int oy, float *canvas, int cw, int ch);
/* ImageProcess::blendTwoImages, ImageProcess.cpp:648-773 (opts NULL = defaults).  seam_out may be NULL. */
int stitch_blend_u8(const uint8_t *a, const uint8_t *b, int w, int h, const stitch_blend_opts *opts, uint8_t *out,
                    stitch_seam *seam_out);
int stitch_blend_f32(const float *a, const float *b, int w, int h, const stitch_blend_opts *opts, float *out,
                     stitch_seam *seam_out);
/* One stitch step, ImageProcess.cpp:218-230: zero canvases a,b; warp `frame` into a; move `mosaic` into b;
 * out = blendTwoImages(a,b).  The canvases never leave the device. */
int stitch_pair_u8(const uint8_t *frame, int fw, int fh, const double p[8], float offx, float offy,
                   const uint8_t *mosaic, int mw, int mh, int ox, int oy, int cw, int ch,
                   const stitch_blend_opts *opts, uint8_t *out, stitch_seam *seam_out);
int stitch_pair_f32(const float *frame, int fw, int fh, const double p[8], float offx, float offy,
                    const float *mosaic, int mw, int mh, int ox, int oy, int cw, int ch,
                    const stitch_blend_opts *opts, float *out, stitch_seam *seam_out);
/* equalization::equalization(src, 1), equalization.cpp:4-25,74-131: in place.  hist_out (optional) receives
 * the 256 Y-histogram bins of equalizationStep (:104-107). */
int stitch_equalize_u8(uint8_t *img, int w, int h, int32_t hist_out[256]);
/* Luminance mix inlined in ImageProcess::matching, ImageProcess.cpp:240-268:
 * Y = Y*num/den + Yeq/den (reference 19,20; src/ex6 5,6); in place on `result`. */
int stitch_lummix_u8(uint8_t *result, const uint8_t *equalized, int w, int h, double num, double den);
/* The tail of matching() as one call, ImageProcess.cpp:237-268: tmp = result; equalization(tmp,1); mix. */
int stitch_finish_u8(uint8_t *result, int w, int h, double num, double den, int32_t hist_out[256]);

/* ---- device-resident entry points (asynchronous on `stream`) -------------------------------------------- */
int stitch_dev_project_u8(const uint8_t *d_src, int w, int h, float fov_deg, uint8_t *d_dst, void *stream);
int stitch_dev_project_f32(const float *d_src, int w, int h, float fov_deg, float *d_dst, void *stream);
int stitch_dev_warp_u8(const uint8_t *d_src, int sw, int sh, const double p[8], float offx, float offy,
                       uint8_t *d_canvas, int cw, int ch, void *stream);
int stitch_dev_warp_f32(const float *d_src, int sw, int sh, const double p[8], float offx, float offy,
                        float *d_canvas, int cw, int ch, void *stream);
int stitch_dev_move_u8(const uint8_t *d_src, int sw, int sh, int ox, int oy, uint8_t *d_canvas, int cw, int ch,
                       void *stream);
int stitch_dev_move_f32(const float *d_src, int sw, int sh, int ox, int oy, float *d_canvas, int cw, int ch,
                        void *stream);

/* A plan owns every device buffer a cw x ch blend needs (Gaussian/Laplacian pyramids of a, b and the mask,
 * the blur scratch, the collapse chain, the resize tables, the seam record).  Create it once per canvas size
 * on the device that will run it; calls on one plan must be serialised by the caller (one stream at a time). */
int stitch_plan_create(int cw, int ch, const stitch_blend_opts *opts, stitch_plan **plan_out);
/* Workspace for up to max_pairs (1..16) independent pairs of the same canvas size processed by ONE launch sequence
 * (every kernel covers all pairs: the batch configs of BASELINE.json -- more independent lines per launch for the
 * recursive filters, fewer launches per pair). */
int stitch_plan_create_batched(int cw, int ch, const stitch_blend_opts *opts, int max_pairs, stitch_plan **plan_out);
int stitch_plan_capacity(const stitch_plan *plan);
void stitch_plan_destroy(stitch_plan *plan);
size_t stitch_plan_workspace_bytes(const stitch_plan *plan);
/* Device address of the plan's workspace (diagnostics: placement experiments, scripts/experiments/exp_placement.py). */
const void *stitch_plan_workspace_base(const stitch_plan *plan);
int stitch_plan_levels(const stitch_plan *plan, int *level_w, int *level_h);
/* Number of finest pyramid levels whose anticausal-x and causal-y sweeps run fused (k_vv_xbyf).  Chosen at plan
 * creation: for batched plans (max_pairs >= 2) and for single pairs at least 7360 rows high (many row bands in flight),
 * the levels of at least 1024 x 1024, at most two -- or exactly
 * STITCH_WAVEFRONT=<n> levels when that environment variable is set (0 = always separate sweeps).  Results are
 * identical either way. */
int stitch_plan_fused_sweep_levels(const stitch_plan *plan);
/* How the collapse of `level` (0 .. levels - 2) is launched: columns [*xa, *xb) run four per work-item (k_collapse4), the rest one per
 * work-item; *per_lane_taps != 0: with per-lane tap offsets (whole rows of an even-width level), 0: only where the fixed tap pattern holds.
 * Introspection for tests: a level of 256 columns and more whose [*xa, *xb) does not cover its rows fell back to the one-column path,
 * which is what the collapse over-fetched through in round 3.  STITCH_ERR_ARG for a level without a collapse. */
int stitch_plan_collapse_range(const stitch_plan *plan, int level, int *xa, int *xb, int *per_lane_taps);
/* Which of the forms that avoid HBM round trips this plan's level 0 runs with (a bit mask; results never depend on it):
 *   IMPLICIT_MASK   the level-0 mask (a vertical step, ImageProcess.cpp:690-698) is generated where it is needed, never stored
 *   SOURCE_FUSED    level 0 of a and b is read from the caller's frames / canvases by its consumers, never materialised
 *                   (pairs: through an index plane of the warp; stitch_dev_blend_*: the dense canvases themselves)
 *   FUSED_SWEEP     anticausal-x and causal-y sweeps of the finest levels run as one kernel (stitch_plan_fused_sweep_levels)
 *   ZERO_TILES      all-zero 64x64 tiles of the blur scratch are flagged instead of stored (level 0; needs FUSED_SWEEP)
 *   FUSED_DECIMATE  level 0's anticausal y sweep writes the decimated level directly (any canvas width >= 2 rows high)
 *   COARSE_LEVELS   every level from the first one with both sides <= 40 on runs in ONE launch (k_coarse: REDUCE to the top,
 *                   top blend, collapse back up; one workgroup per pair) instead of about six launches per level
 * IMPLICIT_MASK and SOURCE_FUSED hold for every canvas size with the Van Vliet blur (blur_kind 0, sigma >= 0.5) and at least
 * two pyramid levels. */
enum {
    STITCH_FAST_IMPLICIT_MASK = 1,
    STITCH_FAST_SOURCE_FUSED = 2,
    STITCH_FAST_FUSED_SWEEP = 4,
    STITCH_FAST_ZERO_TILES = 8,
    STITCH_FAST_FUSED_DECIMATE = 16,
    STITCH_FAST_COARSE_LEVELS = 32
};
int stitch_plan_fast_paths(const stitch_plan *plan);
/* stitch_plan_fast_paths is a CAPABILITY mask: what the workspace was built for.  Which of SOURCE_FUSED / FUSED_SWEEP a call
 * actually runs is decided per call from its number of pairs (a launch sequence with at least two pairs and 8 MPix of canvas,
 * or a canvas of 800 row bands, is bound by bytes and takes the fused forms; a lone pair is bound by its recurrence chains and
 * takes the materialised level 0 and the separate sweeps; STITCH_SINGLE_FAST=1 pins the fused forms).  This returns that
 * choice for a call with n_pairs pairs (0 for a bad argument).  An output that overlaps an input of the same call always takes
 * the materialised level 0 (see stitch_dev_blend_*). */
int stitch_plan_call_forms(const stitch_plan *plan, int n_pairs);
/* First pyramid level of the COARSE_LEVELS launch (0 = none: every level has its own launch sequence). */
int stitch_plan_coarse_from(const stitch_plan *plan);
/* Tuning / A-B switches, read from the environment when a plan is created (none of them changes a result bit).  The
 * host-buffer entry points read them again on every call and key their workspace cache on the values, so a switch that
 * changes between two calls takes effect at once:
 *   STITCH_WAVEFRONT=<n>      fused sweep on exactly n finest levels (0 = never)
 *   STITCH_NO_FUSE=1          blur and decimation as separate kernels, level-0 mask materialised
 *   STITCH_NO_SRC_FUSE=1      materialise level 0 (k_compose) instead of gathering it from the frames where it is needed
 *   STITCH_NO_ZERO_TILES=1    store and re-read all-zero tiles of the blur scratch like any other tile
 *   STITCH_CROWS_L0=<n>       rows per work-item strip of the level-0 collapse (default 32)
 *   STITCH_CROWS_LN=<n>       the same for the levels above (default: per level, h/32 clamped to 4..32)
 *   STITCH_CROWS_WGS=<n>      one pair per call: the strips of the collapse are halved (down to 4 rows) until a launch has n workgroups
 *                             (default 8192; 0 = heights by the level's size alone, as in a batch)
 *   STITCH_COLLAPSE4=0        collapse with one column and three channels per work-item everywhere (k_collapse) instead of
 *                             four columns of one channel where the resize taps are regular (k_collapse4)
 *   STITCH_XBYF_WGS=<n>       persistent workgroups of the fused sweep (default 2304)
 *   STITCH_XBYF_SPIN_LIMIT=<n> polls before a hand-off wait of the fused sweep gives up (default 2^23, about 20 s; 0 forces the
 *                             bail-out path: tests of the sticky time-out report)
 *   STITCH_XBYF_EARLY=0       fused sweep: poll for the hand-off only when it is needed (default: read it ahead of the prefetch)
 *   STITCH_COARSE=<n>         side length from which the coarse levels run in one launch (default 40; 0 = one launch sequence
 *                             per level everywhere)
 *   STITCH_COLLAPSE_PX=0      collapse of small middle levels in strips (k_collapse) instead of one pixel per work-item (k_collapse_px)
 *   STITCH_C4_GEN=0|1         collapse: 0 = four columns per work-item only where the resize taps follow the fixed pattern s, s+1, s+1, s+2
 *                             (the first and the last 256 columns of an even width then take the one-column path: 8 % of level 0 at
 *                             6144 columns, a sixth of level 1); 1 = per-lane tap offsets only where they cover two blocks more;
 *                             default: per-lane tap offsets wherever they cover a block more (every level, whole rows)
 *   STITCH_ODD_DEC=0          odd level widths: anticausal y sweep and decimation as two kernels (default: fused, as for even widths)
 *   STITCH_SINGLE_FAST=1      one pair per call: run the throughput forms too (source-fused level 0, fused sweep on batched plans);
 *                             default: a lone pair, whose time is the length of its recurrence chains, not its bytes, takes the
 *                             materialised level 0 and the separate sweeps, which have the shorter chains
 *   STITCH_MOVER=0            one pair in flight: x sweeps and causal y sweep on one wavefront per block (k_vv_x_fwd / k_vv_x_bwd /
 *                             k_vv_y_fwd1(s)) instead of chain + loader + storer wavefronts (k_vv_x_m, k_vv_y_m); also keeps a lone
 *                             pair's level 0 materialised (the gathers of a source-fused level 0 need the loader wavefront)
 *   STITCH_SRC_LONE_MPIX=<n>  canvas size in MPix from which ONE pair runs a source-fused level 0 (default 8; 0 = never)
 *   STITCH_DEC7=0             one pair in flight: anticausal y sweep + decimation on two wavefronts (k_vv_y_bwd_dec) instead of the
 *                             seven-wavefront pipeline with the short divide (k_vv_y_bwd_dec7)
 *   STITCH_Y1S=0|2            causal y sweep on one wavefront: flat addresses (k_vv_y_fwd1) instead of scalar row offsets through a
 *                             buffer descriptor (k_vv_y_fwd1s); 2 = the descriptor form on launches above 1 GB too
 *   STITCH_C4_LOCKSTEP=1|2    k_collapse4: 1 = the three channel wavefronts of a workgroup meet at a barrier every row (measured: the index /
 *                             mask lines they share are NOT what the kernel over-fetches -- 0.7 % fewer bytes, no time); 2 = levels >= 1:
 *                             channel 0 reads the mask samples of a row and hands them to the other two through LDS (2 % fewer bytes,
 *                             no time); default off
 *   STITCH_C4_SWIZZLE=0|2     k_collapse4: 0 = column blocks in launch order (default 1: within eight strips every XCD gets one whole
 *                             strip, so that the source lines neighbouring blocks share are fetched into one L2); 2 = every XCD walks
 *                             its own run of adjacent strips top to bottom (the rows two strips share meet in one L2 too)
 *   STITCH_PITCH_PAD=<n>      floats added to the row pitch of levels of 4096 columns and more (A/B: measured no effect, default 0)
 *   STITCH_D7_STAMP=<level>   diagnostics: per-chunk time stamps of the seven wavefronts of one workgroup of k_vv_y_bwd_dec7 at that level
 *   STITCH_XBYM=0|1           one pair in flight: anticausal x + causal y sweep of a level as ONE launch of five-wavefront bands
 *                             (k_vv_xby_m; with zero-tile flags where the plan owns them and the level has more bands than the chain + loader +
 *                             storer sweeps take): 0 = never (a pair of >= 800 bands then runs the batch's fused sweep, as until round 4),
 *                             1 = at the first four levels whatever their size (tests); default: where the
 *                             separate sweeps are bound by their bytes, from STITCH_XBYM_MPIX megapixels per plane (default 20)
 *   STITCH_XBYM_STAMP=1       diagnostics: per-tile time stamps of the five wavefronts of one band of k_vv_xby_m, printed at plan destruction
 *   STITCH_COARSE_LDS=0       coarse levels in global memory (k_coarse) instead of LDS (k_coarse_lds, where the levels fit into 144 KB)
 *   STITCH_GATE64=1           implicit level-0 mask, source fusion and zero-tile flags only for level heights that are multiples
 *                             of 64 (the round-2 behaviour; A/B runs)
 *   STITCH_NO_FASTDIV=1       luminance mix: always the IEEE divide (default: reciprocal + fma correction where the host has
 *                             shown it equal for every operand the mix can meet, once per (num, den))
 *   STITCH_Y2=1               causal y sweep always with two columns per work-item (default: one column where a launch has
 *                             fewer than 1.5 wavefronts per SIMD)
 *   STITCH_RECOMPUTE=<0|1|2>  fused levels: 0 (default) = the causal x sweep writes its samples and the fused sweep reads
 *                             them back; 1 = it keeps only its state in front of every 64-sample tile and the fused sweep
 *                             re-runs it from there (level 0 straight from the frames); 2 = that form at the fused levels >= 1
 *                             only.  Fewer bytes, more dependent arithmetic per tile: measured slower (1) / equal (2)
 * The stitch_dev_pairs_* launch sequence contains no host synchronisation and no per-launch state in kernel
 * arguments: it may be captured into a HIP graph and replayed on new contents of the same buffers.
 * Buffers of stitch_dev_blend_* / stitch_dev_pair(s)_*: the inputs must stay unchanged until the call has completed on its
 * stream (the source-fused form reads them again while the output is written).  An output may alias or overlap an input of
 * the same call -- d_out == d_b is the reference's own `result = blendTwoImages(a, result)` -- the library detects it and
 * takes the materialised level 0 (inputs copied before anything is written), at that form's cost. */

int stitch_dev_blend_u8(stitch_plan *plan, const uint8_t *d_a, const uint8_t *d_b, uint8_t *d_out, void *stream);
int stitch_dev_blend_f32(stitch_plan *plan, const float *d_a, const float *d_b, float *d_out, void *stream);
int stitch_dev_pair_u8(stitch_plan *plan, const uint8_t *d_frame, int fw, int fh, const double p[8], float offx,
                       float offy, const uint8_t *d_mosaic, int mw, int mh, int ox, int oy, uint8_t *d_out,
                       void *stream);
int stitch_dev_pair_f32(stitch_plan *plan, const float *d_frame, int fw, int fh, const double p[8], float offx,
                        float offy, const float *d_mosaic, int mw, int mh, int ox, int oy, float *d_out, void *stream);
/* One independent stitch step (ImageProcess.cpp:218-230) of a batch; buffers are device pointers of the call's
 * pixel type (uint8_t for _u8, float for _f32); `out` is the dense cw x ch x 3 mosaic. */
typedef struct stitch_pair_desc {
    const void *frame;  /* image to warp (the reference's imgs[dst].projectedSrc)                              */
    int fw, fh;
    double p[8];        /* backward map {H00,H01,H02,H10,H11,H12,H20,H21}                                       */
    float offx, offy;   /* min_x, min_y as passed to warpingImageByHomography                                   */
    const void *mosaic; /* running mosaic to move (the reference's `result`)                                    */
    int mw, mh;
    int ox, oy;         /* integer offsets as passed to movingImageByOffset                                     */
    void *out;
    void *out_u8;       /* _f32 calls only, optional (NULL = none): a second copy of the mosaic as unsigned char --
                           the reference's own output type, CImg<unsigned char>(CImg<float>), C-cast truncation
                           (ImageProcess.cpp:772) -- written by the same kernel (what a sharded batch gathers); single-level
                           pyramids included                                                                            */
} stitch_pair_desc;
int stitch_dev_pairs_u8(stitch_plan *plan, const stitch_pair_desc *pairs, int n, void *stream);
int stitch_dev_pairs_f32(stitch_plan *plan, const stitch_pair_desc *pairs, int n, void *stream);
/* Waits for the plan's last call to finish and reports its outcome: STITCH_OK, STITCH_ERR_EMPTY_MIDROW or
 * STITCH_ERR_ZERO_OVERLAP for the seam scan of pair `index` of the LAST call (in these cases the output buffer holds
 * unspecified finite values), or STITCH_ERR_HIP when a hand-off wait of the fused sweep (k_vv_xbyf) timed out (index must be below the
 * last call's n: STITCH_ERR_ARG otherwise).  The
 * time-out is STICKY: the plan counts timed-out waits in a device word that no launch sequence clears, so a bail-out in
 * ANY call queued on the plan since the last stitch_plan_clear_fault is reported here (the outputs of every call since
 * then are invalid), not just one in the last call. */
int stitch_plan_status(stitch_plan *plan, stitch_seam *seam_out);                   /* pair 0 */
int stitch_plan_status_at(stitch_plan *plan, int index, stitch_seam *seam_out);    /* pair `index` of a batch */
/* Acknowledges the time-outs reported so far (after the caller has discarded the affected outputs). */
int stitch_plan_clear_fault(stitch_plan *plan);
/* Polls before a hand-off wait of the fused sweep gives up, for the calls enqueued from now on (default 2^20, about a
 * second; 0 = give up at the first check, which is how the tests reach the time-out report). */
int stitch_plan_set_handoff_spin_limit(stitch_plan *plan, unsigned polls);

/* Per-kernel device timing of a plan's calls, with HIP events recorded on the call's stream around every
 * launch.  Kernel ids: */
enum {
    STITCH_K_COMPOSE = 0,      /* k_compose (pairs) / k_load_canvases (blend): warp + move + value cast -> level 0 */
    STITCH_K_SEAM = 1,         /* k_seam: mid-row scan                                                           */
    STITCH_K_MASK = 2,         /* k_mask: level-0 mask plane (only when it is not handled implicitly)            */
    STITCH_K_VV_X_FWD = 3,     /* k_vv_x_fwd: recursive Gaussian along rows, causal (Deriche: whole x pass)      */
    STITCH_K_VV_X_BWD = 4,     /* k_vv_x_bwd: anticausal                                                         */
    STITCH_K_VV_Y_FWD = 5,     /* k_vv_y_fwd: along columns, causal (Deriche: whole y pass)                      */
    STITCH_K_VV_Y_BWD = 6,     /* k_vv_y_bwd_dec (fused with the decimation) / k_vv_y_bwd: anticausal            */
    STITCH_K_DECIMATE = 7,     /* k_decimate: stand-alone moving-average halving                                 */
    STITCH_K_COLLAPSE_TOP = 8, /* k_blend_top: blend of the coarsest level                                       */
    STITCH_K_COLLAPSE = 9,     /* k_collapse<float,false>: expand + Laplacian + blend + collapse, levels 1..L-2  */
    STITCH_K_COLLAPSE_L0 = 10, /* k_collapse<T,true>: the same at level 0, writing the dense output canvas       */
    STITCH_K_VV_XBYF = 11,     /* k_vv_xbyf: anticausal-x + causal-y sweeps fused (row-band pipeline), finest levels */
    STITCH_K_VV_X_FWD_SRC = 12, /* k_vv_x_fwd<T,true>: the causal x sweep of a source-fused level 0 (reads the frames)  */
    STITCH_K_COARSE = 13,      /* k_coarse: all coarse levels in one launch (REDUCE, top blend, collapse)              */
    STITCH_K_COUNT = 14
};
int stitch_plan_set_profiling(stitch_plan *plan, int enabled);
/* Record events only around launches of one kernel id (near-zero overhead inside a timed region). */
int stitch_plan_set_profiling_kernel(stitch_plan *plan, int kernel_id);
/* Sums since profiling was enabled or last read; synchronises the plan's stream.  total_ms[k] / launches[k] is
 * the average launch duration of kernel k over all pyramid levels; level0_ms[k] is the finest level's share. */
int stitch_plan_read_profile(stitch_plan *plan, double total_ms[STITCH_K_COUNT], int launches[STITCH_K_COUNT],
                             double level0_ms[STITCH_K_COUNT]);

int stitch_dev_equalize_u8(uint8_t *d_img, int w, int h, int32_t *d_hist256, void *stream);
int stitch_dev_lummix_u8(uint8_t *d_result, const uint8_t *d_equalized, int w, int h, double num, double den,
                         void *stream);
int stitch_dev_finish_u8(uint8_t *d_result, int w, int h, double num, double den, int32_t *d_hist256, void *stream);

/* ---- the callers either side of the path ----------------------------------------------------- */
/* ImageProcess::toGrayScale (ImageProcess.cpp:27-40) and the float staging of siftAlgorithm (:47-51):
 * gray = (uchar)(0.299 R + 0.587 G + 0.114 B) in double; gray_f32 = (float)gray = VLFeat's vl_sift_pix input.
 * Either output may be NULL. */
int stitch_gray_u8(const uint8_t *rgb, int w, int h, uint8_t *gray, float *gray_f32);
int stitch_dev_gray_u8(const uint8_t *d_rgb, int w, int h, uint8_t *d_gray, float *d_gray_f32, void *stream);
/* The on-disk format either side of the path: CImg<T>::load_bmp (CImg.h:48395-48566) reads the
 * Input/ frames, save_bmp (CImg.h:52614-52700) writes the panorama.  Uncompressed 24- and 32-bit files (the reference's
 * are 24-bit); palette, 16-bit and compressed files are refused with STITCH_ERR_ARG.  A file is one byte array.
 * stitch_bmp_parse is host arithmetic on the 54-byte header (n = size of the whole file); decode writes planar RGB
 * (3*width*height bytes), bytes the file lacks read as 0 exactly as the reference's zero-filled buffer does; encode
 * writes the byte-identical file save_bmp would (stitch_bmp_file_bytes(w,h) bytes). */
typedef struct stitch_bmp_info {
    int32_t width, height, bpp, top_down;
    uint64_t data_pos, stride, data_bytes; /* first pixel byte, bytes per file row, pixel bytes present */
} stitch_bmp_info;
int stitch_bmp_parse(const uint8_t *file_header54, size_t n, stitch_bmp_info *info);
size_t stitch_bmp_file_bytes(int w, int h);
int stitch_bmp_decode_u8(const uint8_t *file, size_t n, uint8_t *planar);
int stitch_bmp_encode_u8(const uint8_t *planar, int w, int h, uint8_t *file, size_t cap);
int stitch_dev_bmp_decode_u8(const uint8_t *d_file, size_t n, const stitch_bmp_info *info, uint8_t *d_planar, void *stream);
int stitch_dev_bmp_encode_u8(const uint8_t *d_planar, int w, int h, uint8_t *d_file, size_t cap, void *stream);
/* transfer::transfer (transfer.cpp:3-13, :125-225): Reinhard's l-alpha-beta colour transfer of
 * `tem`'s statistics onto `src`.  Dead code in the reference (ImageProcess.cpp:180-182) and not buildable outside
 * Windows, so parity is against the CPU restatement only ("parity unpinned").  The float running sums of
 * transfer.cpp:128-164 are kept in the reference's serial order; std::log(float) / std::pow(10, float) are the
 * specified functions of include/stitch_elem.h.  stats (optional, 12 floats): mean[3], sd[3] of the source, then of the
 * template, in l-alpha-beta.  out may alias src. */
int stitch_transfer_u8(const uint8_t *src, int sw, int sh, const uint8_t *tem, int tw, int th, uint8_t *out, float stats[12]);
int stitch_dev_transfer_u8(const uint8_t *d_src, int sw, int sh, const uint8_t *d_tem, int tw, int th, uint8_t *d_out,
                           float *d_stats12, void *stream);
/* readFile's per-image chain in one kernel (ImageProcess.cpp:18-20): projection + gray + float staging. */
int stitch_project_gray_u8(const uint8_t *src, int w, int h, float fov_deg, uint8_t *projected, uint8_t *gray,
                           float *gray_f32);
int stitch_dev_project_gray_u8(const uint8_t *d_src, int w, int h, float fov_deg, uint8_t *d_projected, uint8_t *d_gray,
                               float *d_gray_f32, void *stream);
/* Canvas of one stitch step from the FORWARD map (ImageProcess.cpp:206-216 with :532-594): min_x/min_y (<= 0) are
 * the offsets handed to warp (as floats) and move (truncated); new_w x new_h is the canvas.  Host arithmetic. */
int stitch_canvas_bbox(int fw, int fh, const double p_fwd[8], int result_w, int result_h, float *min_x, float *min_y,
                       int *new_w, int *new_h);
/* One whole stitch step of matching() (ImageProcess.cpp:206-230) from the FORWARD map, device resident.
 * stitch_step_geometry is the host arithmetic of :206-216 plus the float -> int truncation of the offsets at :224:
 * it tells the caller how large the new mosaic is.  stitch_dev_step_* then runs canvas zeroing (:218-219, implicit), warp with
 * the BACKWARD map and (min_x, min_y) (:222), move by (ox, oy) (:224) and the blend (:230) as one launch sequence on a
 * workspace taken from the library's plan cache, waits for it and reports the seam outcome like stitch_plan_status.
 * d_out holds 3 * cw * ch samples (out_capacity = samples available).  The feature updates of :226-227 are
 * stitch_map_points(p_fwd, min_x, min_y) and stitch_shift_points(ox, oy) below. */
typedef struct stitch_step_geom {
    float min_x, min_y; /* <= 0: offsets handed to the warp                                                     */
    int cw, ch;         /* canvas = size of the new mosaic                                                      */
    int ox, oy;         /* (int)min_x, (int)min_y: offsets handed to the move                                   */
} stitch_step_geom;
int stitch_step_geometry(int fw, int fh, const double p_fwd[8], int mw, int mh, stitch_step_geom *geom);
int stitch_dev_step_u8(const uint8_t *d_frame, int fw, int fh, const double p_fwd[8], const double p_bwd[8],
                       const uint8_t *d_mosaic, int mw, int mh, const stitch_blend_opts *opts, uint8_t *d_out,
                       size_t out_capacity, stitch_step_geom *geom_out, stitch_seam *seam_out, void *stream);
int stitch_dev_step_f32(const float *d_frame, int fw, int fh, const double p_fwd[8], const double p_bwd[8],
                        const float *d_mosaic, int mw, int mh, const stitch_blend_opts *opts, float *d_out,
                        size_t out_capacity, stitch_step_geom *geom_out, stitch_seam *seam_out, void *stream);
/* updateFeaturesByHomography / updateFeaturesByOffset (ImageProcess.cpp:622-640) on keypoint coordinate arrays
 * (ix/iy = the truncated integer coordinates, optional).  Host arithmetic. */
int stitch_map_points(float *x, float *y, int32_t *ix, int32_t *iy, int n, const double p_fwd[8], float offx, float offy);
int stitch_shift_points(float *x, float *y, int32_t *ix, int32_t *iy, int n, int ox, int oy);

/* ---- one pair split into row bands over several GPUs (BASELINE.json configs[4]; SURVEY.md 8(e), state hand-off) ----------
 * Rank r of nranks owns rows [r*h_l/nranks, (r+1)*h_l/nranks) of every SPLIT level l < split_levels of the blend's pyramids
 * (band heights must be even); the levels from split_levels up are replicated on every rank.  This library does the
 * per-band computation; the caller moves what crosses ranks (computervisionimagestich2_amd/pipeline.py: BandStitcher over
 * torch.distributed -- RCCL send/recv over xGMI, or gloo staged through the host in the one-GPU test):
 *   compose (S1 + seam + mask, every rank from the full input frames)
 *   per split level l:  reduce_x;  per plane 0..6: reduce_y_fwd (resume state from the rank above, leaves the state for the
 *                       rank below), then per plane: reduce_y_bwd (resume from the rank below; writes this rank's band of level l+1)
 *   all-gather of the bands of level split_levels (stitch_band_rows kind 2) -> stitch_band_top (replicated coarse levels)
 *   per split level, coarse to fine: halo rows of G and E of level l+1 from both neighbours (stitch_band_rows), collapse.
 * Results equal the single-GPU path (and the oracle) bit for bit.  Root variant options only. */
typedef struct stitch_band stitch_band;
int stitch_band_create(int cw, int ch, int rank, int nranks, int split_levels, const stitch_blend_opts *opts, stitch_band **band_out);
void stitch_band_destroy(stitch_band *band);
/* out[6] = {w, rows, row0, pitch, rows of the whole level, halo} of level 0..split_levels */
int stitch_band_geometry(const stitch_band *band, int level, int out[6]);
/* level 0 source-fused (1, the default: never stored, gathered from the frames; implicit mask) or materialised (0: needed by the
 * one-plane-at-a-time forms of reduce_y_fwd / reduce_y_bwd); takes effect with the next compose.  STITCH_BAND_PLANES=1 in the
 * environment at band creation selects 0. */
int stitch_band_set_level0(stitch_band *band, int source_fused);
int stitch_band_levels(const stitch_band *band, int *total_levels); /* returns split_levels */
/* compose: with the source-fused level 0 (the default) d_frame and d_mosaic are READ AGAIN by reduce_x / reduce_xy_fwd of level 0 and
 * by collapse of level 0 -- they must stay valid and unchanged until that collapse has run (stream order is enough); the pixel type
 * of that collapse's output must be the composed inputs'. */
int stitch_band_compose_u8(stitch_band *band, const uint8_t *d_frame, int fw, int fh, const double p[8], float offx, float offy,
                           const uint8_t *d_mosaic, int mw, int mh, int ox, int oy, void *stream);
int stitch_band_compose_f32(stitch_band *band, const float *d_frame, int fw, int fh, const double p[8], float offx, float offy,
                            const float *d_mosaic, int mw, int mh, int ox, int oy, void *stream);
int stitch_band_reduce_x(stitch_band *band, int level, void *stream);
/* reduce_x + reduce_y_fwd(plane -1) with the anticausal x sweep and the causal y sweep fused into one pass over the level;
 * resume: [3][7][pitch] from the rank above (NULL exactly on rank 0); state_out: [4][7][pitch] */
int stitch_band_reduce_xy_fwd(stitch_band *band, int level, const double *d_resume, double *d_state_out, void *stream);
/* resume: [3][pitch] doubles from the rank above (NULL exactly on rank 0); state_out: [4][pitch] doubles */
int stitch_band_reduce_y_fwd(stitch_band *band, int level, int plane, const double *d_resume, double *d_state_out, void *stream);
/* fwd_state: this plane's reduce_y_fwd state_out; resume: [3][pitch] from the rank below (NULL exactly on the last rank);
 * state_out: [3][pitch] for the rank above */
int stitch_band_reduce_y_bwd(stitch_band *band, int level, int plane, const double *d_fwd_state, const double *d_resume,
                             double *d_state_out, void *stream);
/* The two y sweeps over the COLUMN RANGE [x0, x1) of all seven planes (x0, x1 multiples of 128; x1 may be the level's pitch), with the
 * chunk's own state arrays: resume [3][7][x1-x0], state_out [4][7][x1-x0] (causal) / [3][7][x1-x0] (anticausal); fwd_state: the chunk's
 * causal state_out.  A rank hands a chunk's state on as soon as the chunk is swept, so the ranks pipeline over the chunks: the chain
 * of a level is ranks + chunks - 1 chunk-sweeps long instead of `ranks` whole sweeps.  The anticausal form
 * needs an even level width. */
int stitch_band_reduce_y_fwd_cols(stitch_band *band, int level, int x0, int x1, const double *d_resume, double *d_state_out, void *stream);
int stitch_band_reduce_y_bwd_cols(stitch_band *band, int level, int x0, int x1, const double *d_fwd_state, const double *d_resume,
                                  double *d_state_out, void *stream);
/* band rows <-> dense buffer [planes][nrows][w]; kind 0: G a,b (6 planes), 1: E (3), 2: G + mask (7); first_row band-local,
 * -halo .. rows+halo-1; to_buffer != 0 packs, 0 unpacks */
int stitch_band_rows(stitch_band *band, int level, int kind, int first_row, int nrows, float *d_buf, int to_buffer, void *stream);
/* d_g7: level split_levels, all rows, 7 dense planes [a0 a1 a2 b0 b1 b2 m][h][w] */
int stitch_band_top(stitch_band *band, const float *d_g7, void *stream);
/* level 0 writes this rank's band of the mosaic, dense [3][rows][cw]; other levels ignore d_out_band */
int stitch_band_collapse_u8(stitch_band *band, int level, uint8_t *d_out_band, void *stream);
int stitch_band_collapse_f32(stitch_band *band, int level, float *d_out_band, void *stream);
int stitch_band_status(stitch_band *band, stitch_seam *seam_out);

/* Deterministic synthetic frames of the benchmark configs: values 1..250, never 0 ("empty" to
 * the seam scan); the float twin adds a 16-bit fraction.  Same generator as oracle_synth_* (tests compare). */
int stitch_dev_synth_u8(uint8_t *d_dst, int w, int h, int frame_id, void *stream);
int stitch_dev_synth_f32(float *d_dst, int w, int h, int frame_id, void *stream);
/* CImg<unsigned char>(const CImg<float>&) (CImg.h:11167-11182), the cast behind `return expand;`
 * (ImageProcess.cpp:772): the uchar mosaic of a float-frame pair, e.g. for the all-gather of finished mosaics. */
int stitch_dev_quantize_u8(const float *d_src, uint8_t *d_dst, size_t n, void *stream);
/* Verification hook.  The decimation divides by a level's width / height (CImg.h:29555, 29575); where the numerator lies in a
 * range in which the IEEE sequence of gfx950 never rescales, the kernels run that sequence's multiply-adds alone with the
 * reciprocal's refinement hoisted out (k_sweeps1.inc, fastdiv).  This call compares the short form with `numerator / w` for EVERY
 * float bit pattern inside the range (about 2^31.4 numerators, a fraction of a second) for one denominator w (an integer value,
 * 2 <= w < 2^24) and returns the number of numerators tested and the number of quotients that differ -- which must be 0. */
int stitch_dev_check_fastdiv(float w, unsigned long long *tested, unsigned long long *mismatches);
/* Verification hook.  One collapse level (CImg.h:29618-29690 + ImageProcess.cpp:756-770) of w x h samples on synthetic planes, run twice:
 * by k_collapse4 with per-lane tap offsets over whole rows -- the form every plan takes -- and by k_collapse, which forms every tap pair
 * (ix, min(ix + 1, sw - 1)) the way the reference does; the two results are compared bit for bit.  probe 0: pseudo-random finite samples of
 * both signs; probe 1: planes of signed zeros built so that a column whose first tap is the source row's last sample comes out -0.0f only
 * if that sample is taken twice (CImg.h:29648) and +0.0f if the sample behind it in the 16-byte window (the next row's first) were used.
 * w, h even, 256 <= w, 4 <= h, both <= 16384.  Returns the samples compared and the number that differ -- which must be 0. */
int stitch_dev_check_collapse_taps(int w, int h, int probe, unsigned long long *compared, unsigned long long *mismatches);

#ifdef __cplusplus
}
#endif
#endif /* STITCH_H */
