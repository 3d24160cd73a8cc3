/* TEST INFRASTRUCTURE ONLY -- the parity oracle.
 *
 * A plain-C, CPU restatement of the reference's per-pixel stitching hot path (cylindrical projection,
 * bilinear-map backward warp + canvas move, Laplacian-pyramid blend, Y-histogram equalisation and
 * luminance mix).  It is NOT part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it, and only as the checker.  The product path is
 * computervisionimagestich2_amd/csrc (HIP) behind include/stitch.h and never calls into this file.
 *
 * Parity pin: every function here is checked bit-for-bit against the reference itself
 * (oracle/_ref/libref_hotpath.so, built from /root/reference by oracle/Makefile) in
 * tests/test_oracle_vs_reference.py (runs where the reference build exists) and against the golden
 * vectors that build produced, committed under tests/golden/ (runs everywhere).
 *
 * All images are planar, channel-major, exactly as CImg lays them out (CImg.h:11787-11793):
 * offset = x + y*W + c*W*H, three channels.  Citations are file:line under /root/reference.
 */
#ifndef STITCH_ORACLE_H
#define STITCH_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout and meaning as stitch_blend_opts in include/stitch.h. */
typedef struct oracle_blend_opts {
    float sigma;     /* REDUCE blur sigma; the reference uses 2 (ImageProcess.cpp:709)                   */
    int blur_kind;   /* 0 = Van Vliet (root variant, get_blur(2,true,true)); 1 = Deriche (src/ex6)        */
    int level_rule;  /* 0 = floor(log2(max(w,h))) (ImageProcess.cpp:675-676); 1 = min (src/ex6 :662-665)  */
    int seam_rule;   /* 0 = channel 0 only (ImageProcess.cpp:661-670); 1 = all three channels (src/ex6)   */
} oracle_blend_opts;

typedef struct oracle_seam {
    int sum_a_x, n_a, sum_ov_x, n_ov; /* the four mid-row integers (ImageProcess.cpp:659-671)            */
    float ratio, ov;                  /* ImageProcess.cpp:686-687                                          */
    int branch;                       /* 0: mask=1 for (float)x < ov ; 1: mask=1 for x >= start           */
    int start;                        /* (int)(ov + 1) for branch 1                                        */
} oracle_seam;

enum {
    ORACLE_OK = 0,
    ORACLE_ERR_ARG = -1,
    ORACLE_ERR_EMPTY_MIDROW = -2, /* reference: unbounded while loop (ImageProcess.cpp:661)              */
    ORACLE_ERR_ZERO_OVERLAP = -3, /* reference: 0/0 -> NaN seam (ImageProcess.cpp:687)                   */
    ORACLE_ERR_PYRAMID = -4       /* a pyramid level would have a zero dimension                          */
};

/* P1+P2  Projection.cpp:3-73 */
int oracle_project_u8(const uint8_t *src, int w, int h, float fov_deg, uint8_t *dst);
int oracle_project_f32(const float *src, int w, int h, float fov_deg, float *dst);
/* W1     ImageProcess.cpp:465-471 */
void oracle_map_xy(float x, float y, const double p[8], float *X, float *Y);
/* W2     ImageProcess.cpp:596-606 (canvas pre-zeroed by the caller, untouched where out of range) */
int oracle_warp_u8(const uint8_t *src, int sw, int sh, const double p[8], float offx, float offy,
                   uint8_t *canvas, int cw, int ch);
int oracle_warp_f32(const float *src, int sw, int sh, const double p[8], float offx, float offy,
                    float *canvas, int cw, int ch);
/* W3     ImageProcess.cpp:608-620 */
int oracle_move_u8(const uint8_t *src, int sw, int sh, int ox, int oy, uint8_t *canvas, int cw, int ch);
int oracle_move_f32(const float *src, int sw, int sh, int ox, int oy, float *canvas, int cw, int ch);
/* B1     ImageProcess.cpp:650-671,686-698 */
int oracle_seam_u8(const uint8_t *a, const uint8_t *b, int w, int h, int seam_rule, oracle_seam *out);
int oracle_seam_f32(const float *a, const float *b, int w, int h, int seam_rule, oracle_seam *out);
/* B2     ImageProcess.cpp:675-676,705-708: returns the level count, fills lw/lh (capacity 32) */
int oracle_pyramid_levels(int w, int h, int level_rule, int *lw, int *lh);
/* B3/B3' CImg.h:35045-35091,34887-34932 (Van Vliet), CImg.h:34777-34869 (Deriche); in place, c planes */
void oracle_blur_f32(float *img, int w, int h, int c, float sigma, int blur_kind);
void oracle_vanvliet_coeffs(float sigma, double filter[4]);
/* B4     CImg.h:29539-29575 through case 3's dispatch at :29618-29626,29656-29659 */
void oracle_decimate_f32(const float *src, int w, int h, int c, float *dst, int w2, int h2);
/* B5     CImg.h:29618-29690 */
void oracle_expand_f32(const float *src, int w, int h, int c, float *dst, int w2, int h2);
void oracle_expand_table(int n_src, int n_dst, int32_t *idx, double *alpha);
/* B1-B6  ImageProcess.cpp:648-773.  out_f32 (optional, 3*w*h) receives the float image before the
 * final truncation; seam_out optional. */
int oracle_blend_u8(const uint8_t *a, const uint8_t *b, int w, int h, const oracle_blend_opts *opts,
                    uint8_t *out, float *out_f32, oracle_seam *seam_out);
int oracle_blend_f32(const float *a, const float *b, int w, int h, const oracle_blend_opts *opts, float *out,
                     oracle_seam *seam_out);
/* warp + move + blend of one pair (ImageProcess.cpp:218-230) */
int oracle_pair_u8(const uint8_t *frame, int fw, int fh, const double p[8], float offx, float offy,
                   const uint8_t *mosaic, int mw, int mh, int ox, int oy, int cw, int ch,
                   const oracle_blend_opts *opts, uint8_t *out);
int oracle_pair_f32(const float *frame, int fw, int fh, const double p[8], float offx, float offy,
                    const float *mosaic, int mw, int mh, int ox, int oy, int cw, int ch,
                    const oracle_blend_opts *opts, float *out);
/* E1-E3  equalization.cpp:74-131 (mode 1); in place; hist/lut optional */
int oracle_equalize_u8(uint8_t *img, int w, int h, int32_t hist[256], int32_t lut[256]);
/* M1     ImageProcess.cpp:240-268: Y = Y*num/den + Yeq/den (root 19,20; src/ex6 5,6); in place on result */
int oracle_lummix_u8(uint8_t *result, const uint8_t *equalized, int w, int h, double num, double den);
/* ImageProcess::toGrayScale (ImageProcess.cpp:27-40) + float staging (:47-51);
 * canvas sizing (ImageProcess.cpp:206-216, :532-594); feature updates (:622-640) */
int oracle_gray_u8(const uint8_t *rgb, int w, int h, uint8_t *gray, float *gray_f32);
int oracle_canvas_bbox(int fw, int fh, const double p_fwd[8], int result_w, int result_h, float *min_x, float *min_y,
                       int *new_w, int *new_h);
void oracle_map_points(float *x, float *y, int32_t *ix, int32_t *iy, int n, const double p_fwd[8], float offx, float offy);
void oracle_shift_points(float *x, float *y, int32_t *ix, int32_t *iy, int n, int ox, int oy);
/* BMP <-> planar RGB, CImg.h:48395-48566 (_load_bmp) and :52614-52700 (_save_bmp);
 * uncompressed 24/32-bit only.  A file is one byte array. */
typedef struct oracle_bmp_info {
    int32_t width, height, bpp, top_down;
    uint64_t data_pos, stride, data_bytes;
} oracle_bmp_info;
int oracle_bmp_parse(const uint8_t *file, size_t n, oracle_bmp_info *info);
int oracle_bmp_decode_u8(const uint8_t *file, size_t n, uint8_t *planar);
size_t oracle_bmp_file_bytes(int w, int h);
int oracle_bmp_encode_u8(const uint8_t *planar, int w, int h, uint8_t *file, size_t cap);
/* transfer.cpp:3-13,125-225 (l-alpha-beta colour transfer; dead code in the reference).
 * PARITY UNPINNED (transfer.cpp needs windows.h; no output of it exists in the reference).  stats (optional) receives
 * mean[3], sd[3] of the source and mean[3], sd[3] of the template in l-alpha-beta.  use_libm = 0: the specified
 * log / pow10 of include/stitch_elem.h (what the product computes); 1: this platform's logf / pow. */
int oracle_transfer_u8(const uint8_t *src, int sw, int sh, const uint8_t *tem, int tw, int th, uint8_t *out, float stats[12],
                       int use_libm);
/* synthetic frame generator of SURVEY.md 8(d) */
void oracle_synth_u8(uint8_t *dst, int w, int h, int frame_id);
void oracle_synth_f32(float *dst, int w, int h, int frame_id);
/* number of OpenMP threads the restatement will use (1 when built without OpenMP) */
int oracle_threads(void);
void oracle_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
