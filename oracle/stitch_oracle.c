/* TEST INFRASTRUCTURE ONLY -- the parity oracle (see stitch_oracle.h for the contract).
 *
 * Plain C99 restatement of the reference's hot path.  Built with -O2 -ffp-contract=off (no FMA
 * contraction: the reference semantics are x86-64 SSE2, strict IEEE, left-to-right).  Loop NESTING is
 * free (the y passes here run rows-outer / columns-inner so that they stream through memory, and rows or
 * lines may be spread over OpenMP threads); the arithmetic applied to each sample -- operand order,
 * float/double promotion points, where values are rounded back to float -- is the reference's, cited
 * line by line.  Citations are file:line under /root/reference.
 */
#include "stitch_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_PI 3.14159265358979323846 /* cimg::PI, CImg.h:4588 region (const double PI) */
#define ORACLE_MAX_LEVELS 32

int oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* float -> int the way the reference's `int newX = <float>` does for every value that can pass the
 * following range test.  Out-of-int-range and NaN inputs give INT_MIN on x86-64 (cvttss2si), which then
 * fails `newX >= 0`; report them as "skip" so that no undefined conversion is executed. */
static inline int oracle_f2i(float v, int *out) {
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0;
    *out = (int)v;
    return 1;
}

/* W1  ImageProcess::getXAfterWarping / getYAfterWarping, ImageProcess.cpp:465-471.
 * H is double[3][3] filled in the order H00,H01,H02,H10 / H11,H12,H20,H21 (ImageProcess.h:58-73) = p[0..7].
 * Evaluated in double: ((p0*x + p1*y) + (p2*x)*y) + p3, rounded to float on return. */
void oracle_map_xy(float x, float y, const double p[8], float *X, float *Y) {
    const double dx = (double)x, dy = (double)y;
    *X = (float)(p[0] * dx + p[1] * dy + p[2] * dx * dy + p[3]);
    *Y = (float)(p[4] * dx + p[5] * dy + p[6] * dx * dy + p[7]);
}

/* B1 tail: ImageProcess.cpp:686-698 (rule 0, float) / src/ex6/ImageProcess.cpp:678-698 (rule 1, double) */
static int oracle_seam_finish_rule(int sum_a_x, int n_a, int sum_ov_x, int n_ov, int rule, oracle_seam *out) {
    out->sum_a_x = sum_a_x;
    out->n_a = n_a;
    out->sum_ov_x = sum_ov_x;
    out->n_ov = n_ov;
    out->ratio = out->ov = 0.f;
    out->branch = out->start = 0;
    if (n_a == 0) return ORACLE_ERR_EMPTY_MIDROW;
    if (n_ov == 0) return ORACLE_ERR_ZERO_OVERLAP;
    if (rule == 0) {
        const float ratio = (float)(1.0 * sum_a_x / n_a);
        const float ov = (float)(1.0 * sum_ov_x / n_ov);
        out->ratio = ratio;
        out->ov = ov;
        out->branch = (ratio < ov) ? 0 : 1;
        out->start = (int)(ov + 1); /* float add, then truncation (:696) */
    } else {
        const double ratio = (double)sum_a_x / n_a, ov = (double)sum_ov_x / n_ov;
        out->ratio = (float)ratio; /* informational only in this mode */
        out->ov = (float)ov;
        out->branch = (ratio < ov) ? 0 : 1;
        out->start = (int)(ov + 1);
    }
    return ORACLE_OK;
}
/* threshold used by branch 0: `x < overlap_ratio` with int x.  Rule 0 compares in float (x is exactly
 * representable), rule 1 in double; both are reproduced by comparing (double)x with this value. */
static double oracle_seam_thr(const oracle_seam *s, int rule) {
    if (rule == 0) return (double)s->ov;
    return (double)s->sum_ov_x / s->n_ov;
}

/* ---- pixel-type-generic functions, instantiated for uint8_t and float -------------------------------- */
#define PX uint8_t
#define SUF(n) n##_u8
#define PX_STORE(f) ((uint8_t)(f))
#include "stitch_oracle_px.inc"
#undef PX
#undef SUF
#undef PX_STORE

#define PX float
#define SUF(n) n##_f32
#define PX_STORE(f) (f)
#include "stitch_oracle_px.inc"
#undef PX
#undef SUF
#undef PX_STORE

/* B2  ImageProcess.cpp:675-676 (max) / src/ex6/ImageProcess.cpp:662-665 (min); sizes :706-707 */
int oracle_pyramid_levels(int w, int h, int level_rule, int *lw, int *lh) {
    if (w <= 0 || h <= 0) return ORACLE_ERR_ARG;
    const int len = level_rule ? (w < h ? w : h) : (w >= h ? w : h);
    int levels = 0;
    while ((len >> (levels + 1)) > 0) ++levels; /* floor(log2(len)) for len >= 1 */
    if (levels < 1 || levels > ORACLE_MAX_LEVELS) return ORACLE_ERR_PYRAMID;
    int cw = w, ch = h;
    for (int i = 0; i < levels; ++i) {
        if (cw <= 0 || ch <= 0) return ORACLE_ERR_PYRAMID;
        if (lw) lw[i] = cw;
        if (lh) lh[i] = ch;
        cw /= 2;
        ch /= 2;
    }
    return levels;
}

/* B3  CImg<T>::vanvliet coefficient block, CImg.h:35049-35065 (sigma >= 0.5 path; order 0) */
void oracle_vanvliet_coeffs(float sigma, double filter[4]) {
    const float nsigma = sigma >= 0 ? sigma : -sigma; /* callers pass sigma >= 0; percent form not used */
    const double nnsigma = nsigma < 0.5f ? 0.5f : nsigma, m0 = 1.16680, m1 = 1.10783, m2 = 1.40586, m1sq = m1 * m1,
                 m2sq = m2 * m2,
                 q = (nnsigma < 3.556 ? -0.2568 + 0.5784 * nnsigma + 0.0561 * nnsigma * nnsigma
                                      : 2.5091 + 0.9804 * (nnsigma - 3.556)),
                 qsq = q * q, scale = (m0 + q) * (m1sq + m2sq + 2 * m1 * q + qsq),
                 b1 = -q * (2 * m0 * m1 + m1sq + m2sq + (2 * m0 + 4 * m1) * q + 3 * qsq) / scale,
                 b2 = qsq * (m0 + 2 * m1 + 3 * q) / scale, b3 = -qsq * q / scale, B = (m0 * (m1sq + m2sq)) / scale;
    filter[0] = B;
    filter[1] = -b1;
    filter[2] = -b2;
    filter[3] = -b3;
}

/* Triggs boundary matrix and scalars shared by every line: CImg.h:34889-34903 */
typedef struct vv_consts {
    double f1, f2, f3, sumsq, sum, M[9], den; /* den = 1 - a1 - a2 - a3 */
} vv_consts;
static void vv_prepare(const double filter[4], vv_consts *k) {
    const double sumsq = filter[0], sum = sumsq * sumsq, a1 = filter[1], a2 = filter[2], a3 = filter[3],
                 scaleM = 1.0 / ((1.0 + a1 - a2 + a3) * (1.0 - a1 - a2 - a3) * (1.0 + a2 + (a1 - a3) * a3));
    k->f1 = a1;
    k->f2 = a2;
    k->f3 = a3;
    k->sumsq = sumsq;
    k->sum = sum;
    k->M[0] = scaleM * (-a3 * a1 + 1.0 - a3 * a3 - a2);
    k->M[1] = scaleM * (a3 + a1) * (a2 + a3 * a1);
    k->M[2] = scaleM * a3 * (a1 + a3 * a2);
    k->M[3] = scaleM * (a1 + a3 * a2);
    k->M[4] = -scaleM * (a2 - 1.0) * (a2 + a3 * a1);
    k->M[5] = -scaleM * a3 * (a3 * a1 + a3 * a3 + a2 - 1.0);
    k->M[6] = scaleM * (a3 * a1 + a2 + a1 * a1 - a2 * a2);
    k->M[7] = scaleM * (a1 * a2 + a3 * a2 * a2 - a1 * a3 * a3 - a3 * a3 * a3 - a3 * a2 + a3);
    k->M[8] = scaleM * a3 * (a1 + a3 * a2);
    k->den = 1.0 - a1 - a2 - a3;
}

/* One line, order 0, Neumann boundaries: CImg<T>::_cimg_recursive_apply, CImg.h:34887-34932.
 * Accumulators are double; every output sample is stored to float; the recurrence state keeps the
 * UNROUNDED doubles (val[k] = val[k-1]). */
static void vv_line(float *data, int N, size_t off, const vv_consts *k) {
    double v1, v2, v3;
    const double iplus = (double)data[(size_t)(N - 1) * off]; /* :34906, read before the forward pass */
    v1 = v2 = v3 = (double)data[0] / k->sumsq;                /* :34909 */
    float *d = data;
    for (int n = 0; n < N; ++n) { /* forward, :34924-34931 with pass==0 */
        double v0 = (double)*d;
        v0 += v1 * k->f1;
        v0 += v2 * k->f2;
        v0 += v3 * k->f3;
        *d = (float)v0;
        d += off;
        v3 = v2;
        v2 = v1;
        v1 = v0;
    }
    d -= off; /* :34932 */
    {         /* Triggs boundary, :34911-34922 */
        const double uplus = iplus / k->den, vplus = uplus / k->den, unp = v1 - uplus, unp1 = v2 - uplus,
                     unp2 = v3 - uplus;
        const double n0 = (k->M[0] * unp + k->M[1] * unp1 + k->M[2] * unp2 + vplus) * k->sum;
        const double n1 = (k->M[3] * unp + k->M[4] * unp1 + k->M[5] * unp2 + vplus) * k->sum;
        const double n2 = (k->M[6] * unp + k->M[7] * unp1 + k->M[8] * unp2 + vplus) * k->sum;
        *d = (float)n0;
        d -= off;
        /* for (k = 3; k>0; --k) val[k] = val[k-1] with val[0..2] = n0,n1,n2 */
        v3 = n2;
        v2 = n1;
        v1 = n0;
    }
    for (int n = 1; n < N; ++n) { /* backward, pass==1 */
        double v0 = (double)*d;
        v0 *= k->sum;
        v0 += v1 * k->f1;
        v0 += v2 * k->f2;
        v0 += v3 * k->f3;
        *d = (float)v0;
        d -= off;
        v3 = v2;
        v2 = v1;
        v1 = v0;
    }
}

/* The same recurrence applied to `nc` adjacent columns at once (rows outer, columns inner) so that the y pass
 * streams through memory; per sample the operations and their order are exactly those of vv_line. */
#define VV_CB 256
static void vv_cols(float *base, int nc, int N, size_t stride, const vv_consts *k) {
    double v1[VV_CB], v2[VV_CB], v3[VV_CB], ip[VV_CB];
    for (int j = 0; j < nc; ++j) {
        ip[j] = (double)base[(size_t)(N - 1) * stride + j];
        v1[j] = v2[j] = v3[j] = (double)base[j] / k->sumsq;
    }
    for (int n = 0; n < N; ++n) {
        float *row = base + (size_t)n * stride;
        for (int j = 0; j < nc; ++j) {
            double v0 = (double)row[j];
            v0 += v1[j] * k->f1;
            v0 += v2[j] * k->f2;
            v0 += v3[j] * k->f3;
            row[j] = (float)v0;
            v3[j] = v2[j];
            v2[j] = v1[j];
            v1[j] = v0;
        }
    }
    {
        float *row = base + (size_t)(N - 1) * stride;
        for (int j = 0; j < nc; ++j) {
            const double uplus = ip[j] / k->den, vplus = uplus / k->den, unp = v1[j] - uplus, unp1 = v2[j] - uplus,
                         unp2 = v3[j] - uplus;
            const double n0 = (k->M[0] * unp + k->M[1] * unp1 + k->M[2] * unp2 + vplus) * k->sum;
            const double n1 = (k->M[3] * unp + k->M[4] * unp1 + k->M[5] * unp2 + vplus) * k->sum;
            const double n2 = (k->M[6] * unp + k->M[7] * unp1 + k->M[8] * unp2 + vplus) * k->sum;
            row[j] = (float)n0;
            v3[j] = n2;
            v2[j] = n1;
            v1[j] = n0;
        }
    }
    for (int n = N - 2; n >= 0; --n) {
        float *row = base + (size_t)n * stride;
        for (int j = 0; j < nc; ++j) {
            double v0 = (double)row[j];
            v0 *= k->sum;
            v0 += v1[j] * k->f1;
            v0 += v2[j] * k->f2;
            v0 += v3[j] * k->f3;
            row[j] = (float)v0;
            v3[j] = v2[j];
            v2[j] = v1[j];
            v1[j] = v0;
        }
    }
}

/* Deriche order 0 coefficients, CImg.h:34801-34816,34840-34841 (all float, exp through the float
 * overload of std::exp) */
typedef struct dr_consts {
    float a0, a1, a2, a3, b1, b2, coefp, coefn;
} dr_consts;
static void dr_prepare(float sigma, dr_consts *k) {
    const float nsigma = sigma >= 0 ? sigma : -sigma;
    const float nnsigma = nsigma < 0.1f ? 0.1f : nsigma, alpha = 1.695f / nnsigma, ema = expf(-alpha),
                ema2 = expf(-2 * alpha), b1 = -2 * ema, b2 = ema2;
    const float kk = (1 - ema) * (1 - ema) / (1 + 2 * alpha * ema - ema2);
    k->a0 = kk;
    k->a1 = kk * (alpha - 1) * ema;
    k->a2 = kk * (alpha + 1) * ema;
    k->a3 = -kk * ema2;
    k->b1 = b1;
    k->b2 = b2;
    k->coefp = (k->a0 + k->a1) / (1 + b1 + b2);
    k->coefn = (k->a2 + k->a3) / (1 + b1 + b2);
}
/* One line: the _cimg_deriche_apply macro, CImg.h:34779-34797 (T = Tfloat = float) */
static void dr_line(float *data, int N, size_t off, const dr_consts *k, float *Y) {
    float *ptrX = data, *ptrY = Y, yb, yp, xp;
    xp = *ptrX;
    yb = yp = (float)(k->coefp * xp);
    for (int m = 0; m < N; ++m) {
        const float xc = *ptrX;
        ptrX += off;
        const float yc = *(ptrY++) = (float)(k->a0 * xc + k->a1 * xp - k->b1 * yp - k->b2 * yb);
        xp = xc;
        yb = yp;
        yp = yc;
    }
    float xn, xa, yn, ya;
    xn = xa = *(ptrX - off);
    yn = ya = (float)k->coefn * xn;
    for (int n = N - 1; n >= 0; --n) {
        const float xc = *(ptrX -= off);
        const float yc = (float)(k->a2 * xn + k->a3 * xa - k->b1 * yn - k->b2 * ya);
        xa = xn;
        xn = xc;
        ya = yn;
        yn = yc;
        *ptrX = (float)(*(--ptrY) + yc);
    }
}

/* B3  CImg<T>::blur(sigma,sigma,sigma,neumann,is_gaussian), CImg.h:35111-35124: x pass over every row if
 * W>1, then y pass over every column if H>1, per channel. */
void oracle_blur_f32(float *img, int w, int h, int c, float sigma, int blur_kind) {
    if (blur_kind == 0) {
        if (sigma < 0.5f && sigma >= 0) { /* vanvliet(): nsigma<0.5 && !order -> return (CImg.h:35051) */
            return;
        }
        double filter[4];
        vv_consts k;
        oracle_vanvliet_coeffs(sigma, filter);
        vv_prepare(filter, &k);
        if (w > 1) {
#pragma omp parallel for schedule(static)
            for (long long line = 0; line < (long long)h * c; ++line) vv_line(img + (size_t)line * w, w, 1, &k);
        }
        if (h > 1) {
            const int nblk = (w + VV_CB - 1) / VV_CB;
#pragma omp parallel for schedule(dynamic)
            for (long long job = 0; job < (long long)nblk * c; ++job) {
                const int ch_i = (int)(job / nblk), x0 = (int)(job % nblk) * VV_CB;
                vv_cols(img + (size_t)ch_i * w * h + x0, (w - x0) < VV_CB ? (w - x0) : VV_CB, h, (size_t)w, &k);
            }
        }
    } else {
        if (sigma < 0.1f && sigma >= 0) return; /* deriche(): CImg.h:34800 */
        dr_consts k;
        dr_prepare(sigma, &k);
        if (w > 1) {
#pragma omp parallel
            {
                float *Y = (float *)malloc(sizeof(float) * (size_t)w);
#pragma omp for schedule(static)
                for (long long line = 0; line < (long long)h * c; ++line) dr_line(img + (size_t)line * w, w, 1, &k, Y);
                free(Y);
            }
        }
        if (h > 1) {
#pragma omp parallel
            {
                float *Y = (float *)malloc(sizeof(float) * (size_t)h);
#pragma omp for schedule(static)
                for (long long col = 0; col < (long long)w * c; ++col) {
                    const int ch_i = (int)(col / w), x = (int)(col % w);
                    dr_line(img + (size_t)ch_i * w * h + x, h, (size_t)w, &k, Y);
                }
                free(Y);
            }
        }
    }
}

/* B4  moving-average resize along one axis, CImg.h:29542-29555 (x) / :29557-29575 (y).
 * Output t accumulates, in increasing s, src[s]*d (float multiply by the integer overlap d converted to
 * float, float add into an accumulator that starts at 0), then is divided once by the source length.
 * The (a,b,c) counter walk of the reference emits exactly the non-empty overlaps of
 * [t*n_src,(t+1)*n_src) with [s*n_dst,(s+1)*n_dst) in increasing s. */
static void decimate_axis(const float *src, int n_src, size_t src_stride, float *dst, int n_dst, size_t dst_stride) {
    unsigned int b = (unsigned)n_src, cc = (unsigned)n_dst, s = 0, t = 0;
    unsigned long long a = (unsigned long long)n_src * (unsigned)n_dst;
    float acc = 0.f;
    while (a) {
        const unsigned int d = b < cc ? b : cc;
        a -= d;
        b -= d;
        cc -= d;
        acc += src[(size_t)s * src_stride] * (float)d;
        if (!b) {
            acc /= (float)(unsigned)n_src;
            dst[(size_t)t * dst_stride] = acc;
            acc = 0.f;
            ++t;
            b = (unsigned)n_src;
        }
        if (!cc) {
            ++s;
            cc = (unsigned)n_dst;
        }
    }
}
void oracle_decimate_f32(const float *src, int w, int h, int c, float *dst, int w2, int h2) {
    /* get_resize(w2,h2,1,c,3) with w2<w, h2<h: case 3 dispatches each shrinking axis to case 2
     * (CImg.h:29621, :29659): x first into a float temporary, then y.  An unchanged axis is skipped. */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w2 * h * c);
    if (w2 != w) {
#pragma omp parallel for schedule(static)
        for (long long line = 0; line < (long long)h * c; ++line)
            decimate_axis(src + (size_t)line * w, w, 1, tmp + (size_t)line * w2, w2, 1);
    } else
        memcpy(tmp, src, sizeof(float) * (size_t)w * h * c);
    if (h2 != h) {
#pragma omp parallel for schedule(static)
        for (long long col = 0; col < (long long)w2 * c; ++col) {
            const int ch_i = (int)(col / w2), x = (int)(col % w2);
            decimate_axis(tmp + (size_t)ch_i * w2 * h + x, h, (size_t)w2, dst + (size_t)ch_i * w2 * h2 + x, h2, (size_t)w2);
        }
    } else
        memcpy(dst, tmp, sizeof(float) * (size_t)w2 * h * c);
    free(tmp);
}

/* B5  linear-interpolation tables, CImg.h:29625-29637: fx = (n_src-1)/(n_dst-1) in double
 * (boundary_conditions = 0, n_dst > n_src), serial walk curr = min(n_src-1, curr+fx);
 * idx[x] = (unsigned)curr_x (the running sum of the reference's off[] increments),
 * alpha[x] = curr_x - (unsigned)curr_x. */
void oracle_expand_table(int n_src, int n_dst, int32_t *idx, double *alpha) {
    const double fx = n_dst > 1 ? (n_src - 1.0) / (n_dst - 1) : 0;
    double curr = 0;
    for (int x = 0; x < n_dst; ++x) {
        idx[x] = (int32_t)(unsigned int)curr;
        alpha[x] = curr - (unsigned int)curr;
        const double nxt = curr + fx;
        curr = (n_src - 1.0) < nxt ? (n_src - 1.0) : nxt; /* std::min(width()-1.0, curr+fx) */
    }
}
void oracle_expand_f32(const float *src, int w, int h, int c, float *dst, int w2, int h2) {
    /* get_resize(w2,h2,1,c,3) with w2>w, h2>h (CImg.h:29618-29690).  A source axis of length 1 goes through
     * nearest-neighbour (:29620, :29657) = replication.  x pass is stored to float before the y pass. */
    int32_t *ix = (int32_t *)malloc(sizeof(int32_t) * (size_t)w2), *iy = (int32_t *)malloc(sizeof(int32_t) * (size_t)h2);
    double *ax = (double *)malloc(sizeof(double) * (size_t)w2), *ay = (double *)malloc(sizeof(double) * (size_t)h2);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w2 * h * c);
    if (w == 1) {
        for (int x = 0; x < w2; ++x) ix[x] = 0, ax[x] = 0;
    } else
        oracle_expand_table(w, w2, ix, ax);
    if (h == 1) {
        for (int y = 0; y < h2; ++y) iy[y] = 0, ay[y] = 0;
    } else
        oracle_expand_table(h, h2, iy, ay);
#pragma omp parallel for schedule(static)
    for (long long line = 0; line < (long long)h * c; ++line) {
        const float *s = src + (size_t)line * w;
        float *d = tmp + (size_t)line * w2;
        if (w == w2)
            memcpy(d, s, sizeof(float) * (size_t)w);
        else if (w == 1)
            for (int x = 0; x < w2; ++x) d[x] = s[0];
        else
            for (int x = 0; x < w2; ++x) {
                const double alpha = ax[x];
                const float v1 = s[ix[x]], v2 = ix[x] < w - 1 ? s[ix[x] + 1] : v1;
                d[x] = (float)((1 - alpha) * v1 + alpha * v2); /* :29647 */
            }
    }
#pragma omp parallel for schedule(static)
    for (long long row = 0; row < (long long)h2 * c; ++row) {
        const int ch_i = (int)(row / h2), y = (int)(row % h2);
        const float *pl = tmp + (size_t)ch_i * w2 * h;
        float *d = dst + (size_t)ch_i * w2 * h2 + (size_t)y * w2;
        if (h == h2)
            memcpy(d, pl + (size_t)y * w2, sizeof(float) * (size_t)w2);
        else if (h == 1)
            memcpy(d, pl, sizeof(float) * (size_t)w2);
        else {
            const double alpha = ay[y];
            const float *r1 = pl + (size_t)iy[y] * w2, *r2 = iy[y] < h - 1 ? r1 + w2 : r1;
            for (int x = 0; x < w2; ++x) d[x] = (float)((1 - alpha) * r1[x] + alpha * r2[x]); /* :29680 */
        }
    }
    free(ix);
    free(iy);
    free(ax);
    free(ay);
    free(tmp);
}

/* B1-B6  ImageProcess::blendTwoImages, ImageProcess.cpp:648-773, on float planes a0,b0 (level 0 = the
 * value-cast inputs, :680-681).  Writes the collapsed float image E (before the final cast) to `out`. */
static int blend_core(float *a0, float *b0, int w, int h, const oracle_blend_opts *opts, const oracle_seam *seam,
                      float *out) {
    int lw[ORACLE_MAX_LEVELS], lh[ORACLE_MAX_LEVELS];
    const int L = oracle_pyramid_levels(w, h, opts->level_rule, lw, lh);
    if (L < 0) return L;
    float *A[ORACLE_MAX_LEVELS], *B[ORACLE_MAX_LEVELS], *M[ORACLE_MAX_LEVELS];
    A[0] = a0;
    B[0] = b0;
    /* mask level 0: a vertical step, :682,:690-698.  One channel is carried: from level 1 on the reference's
     * mask has three identical channels (sp at :713) of which only channel 0 is read (:749-750). */
    M[0] = (float *)malloc(sizeof(float) * (size_t)w * h);
    {
        const double thr = oracle_seam_thr(seam, opts->seam_rule);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                M[0][(size_t)y * w + x] = seam->branch == 0 ? ((double)x < thr ? 1.f : 0.f) : (x >= seam->start ? 1.f : 0.f);
    }
    /* REDUCE, :705-715 */
    for (int i = 1; i < L; ++i) {
        const size_t np = (size_t)lw[i - 1] * lh[i - 1], nn = (size_t)lw[i] * lh[i];
        float *t3 = (float *)malloc(sizeof(float) * np * 3);
        A[i] = (float *)malloc(sizeof(float) * nn * 3);
        B[i] = (float *)malloc(sizeof(float) * nn * 3);
        M[i] = (float *)malloc(sizeof(float) * nn);
        memcpy(t3, A[i - 1], sizeof(float) * np * 3);
        oracle_blur_f32(t3, lw[i - 1], lh[i - 1], 3, opts->sigma, opts->blur_kind);
        oracle_decimate_f32(t3, lw[i - 1], lh[i - 1], 3, A[i], lw[i], lh[i]);
        memcpy(t3, B[i - 1], sizeof(float) * np * 3);
        oracle_blur_f32(t3, lw[i - 1], lh[i - 1], 3, opts->sigma, opts->blur_kind);
        oracle_decimate_f32(t3, lw[i - 1], lh[i - 1], 3, B[i], lw[i], lh[i]);
        memcpy(t3, M[i - 1], sizeof(float) * np);
        oracle_blur_f32(t3, lw[i - 1], lh[i - 1], 1, opts->sigma, opts->blur_kind);
        oracle_decimate_f32(t3, lw[i - 1], lh[i - 1], 1, M[i], lw[i], lh[i]);
        free(t3);
    }
    /* Laplacian, :727-733: G_i -= EXPAND(G_{i+1}) in float (operator-=, CImg.h:12096-12107), finest first so
     * that G_{i+1} is still Gaussian when it is expanded. */
    for (int i = 0; i < L - 1; ++i) {
        const size_t np = (size_t)lw[i] * lh[i] * 3;
        float *e = (float *)malloc(sizeof(float) * np);
        oracle_expand_f32(A[i + 1], lw[i + 1], lh[i + 1], 3, e, lw[i], lh[i]);
        for (size_t j = 0; j < np; ++j) A[i][j] = A[i][j] - e[j];
        oracle_expand_f32(B[i + 1], lw[i + 1], lh[i + 1], 3, e, lw[i], lh[i]);
        for (size_t j = 0; j < np; ++j) B[i][j] = B[i][j] - e[j];
        free(e);
    }
    /* per-level blend, :744-753: a*m is a float product; b*(1.0-m) and the sum are double; stored float.
     * The blended level overwrites A[i]. */
    for (int i = 0; i < L; ++i) {
        const size_t np = (size_t)lw[i] * lh[i];
        for (int c = 0; c < 3; ++c)
            for (size_t j = 0; j < np; ++j) {
                const float m = M[i][j];
                const float am = A[i][c * np + j] * m;
                A[i][c * np + j] = (float)((double)am + (double)B[i][c * np + j] * (1.0 - (double)m));
            }
    }
    /* collapse, :762-772 */
    float *E = (float *)malloc(sizeof(float) * (size_t)lw[L - 1] * lh[L - 1] * 3);
    memcpy(E, A[L - 1], sizeof(float) * (size_t)lw[L - 1] * lh[L - 1] * 3);
    for (int i = L - 2; i >= 0; --i) {
        const size_t np = (size_t)lw[i] * lh[i] * 3;
        float *e = (float *)malloc(sizeof(float) * np);
        oracle_expand_f32(E, lw[i + 1], lh[i + 1], 3, e, lw[i], lh[i]);
        free(E);
        E = e;
        for (size_t j = 0; j < np; ++j) {
            float v = A[i][j] + E[j];
            if (v > 255)
                v = 255;
            else if (v < 0)
                v = 0;
            E[j] = v;
        }
    }
    memcpy(out, E, sizeof(float) * (size_t)w * h * 3);
    free(E);
    free(M[0]);
    for (int i = 1; i < L; ++i) {
        free(A[i]);
        free(B[i]);
        free(M[i]);
    }
    return ORACLE_OK;
}

int oracle_blend_u8(const uint8_t *a, const uint8_t *b, int w, int h, const oracle_blend_opts *opts, uint8_t *out,
                    float *out_f32, oracle_seam *seam_out) {
    if (!a || !b || !out || !opts || w <= 0 || h <= 0) return ORACLE_ERR_ARG;
    int rc = oracle_pyramid_levels(w, h, opts->level_rule, NULL, NULL);
    if (rc < 0) return rc;
    oracle_seam seam;
    rc = oracle_seam_u8(a, b, w, h, opts->seam_rule, &seam);
    if (seam_out) *seam_out = seam;
    if (rc) return rc;
    const size_t n = (size_t)w * h * 3;
    float *a0 = (float *)malloc(sizeof(float) * n), *b0 = (float *)malloc(sizeof(float) * n),
          *e = (float *)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; ++i) a0[i] = (float)a[i], b0[i] = (float)b[i]; /* :680-681 value cast */
    rc = blend_core(a0, b0, w, h, opts, &seam, e);
    if (rc == ORACLE_OK) {
        for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)e[i]; /* return expand; -> CImg<uchar>, truncation */
        if (out_f32) memcpy(out_f32, e, sizeof(float) * n);
    }
    free(a0);
    free(b0);
    free(e);
    return rc;
}

int oracle_blend_f32(const float *a, const float *b, int w, int h, const oracle_blend_opts *opts, float *out,
                     oracle_seam *seam_out) {
    if (!a || !b || !out || !opts || w <= 0 || h <= 0) return ORACLE_ERR_ARG;
    int rc = oracle_pyramid_levels(w, h, opts->level_rule, NULL, NULL);
    if (rc < 0) return rc;
    oracle_seam seam;
    rc = oracle_seam_f32(a, b, w, h, opts->seam_rule, &seam);
    if (seam_out) *seam_out = seam;
    if (rc) return rc;
    const size_t n = (size_t)w * h * 3;
    float *a0 = (float *)malloc(sizeof(float) * n), *b0 = (float *)malloc(sizeof(float) * n);
    memcpy(a0, a, sizeof(float) * n);
    memcpy(b0, b, sizeof(float) * n);
    rc = blend_core(a0, b0, w, h, opts, &seam, out);
    free(a0);
    free(b0);
    return rc;
}

/* ImageProcess.cpp:218-230: a,b zero canvases; warp the new frame into a, move the mosaic into b, blend. */
int oracle_pair_u8(const uint8_t *frame, int fw, int fh, const double p[8], float offx, float offy,
                   const uint8_t *mosaic, int mw, int mh, int ox, int oy, int cw, int ch, const oracle_blend_opts *opts,
                   uint8_t *out) {
    const size_t n = (size_t)cw * ch * 3;
    uint8_t *a = (uint8_t *)calloc(n, 1), *b = (uint8_t *)calloc(n, 1);
    int rc = oracle_warp_u8(frame, fw, fh, p, offx, offy, a, cw, ch);
    if (!rc) rc = oracle_move_u8(mosaic, mw, mh, ox, oy, b, cw, ch);
    if (!rc) rc = oracle_blend_u8(a, b, cw, ch, opts, out, NULL, NULL);
    free(a);
    free(b);
    return rc;
}
int oracle_pair_f32(const float *frame, int fw, int fh, const double p[8], float offx, float offy, const float *mosaic,
                    int mw, int mh, int ox, int oy, int cw, int ch, const oracle_blend_opts *opts, float *out) {
    const size_t n = (size_t)cw * ch * 3;
    float *a = (float *)calloc(n, sizeof(float)), *b = (float *)calloc(n, sizeof(float));
    int rc = oracle_warp_f32(frame, fw, fh, p, offx, offy, a, cw, ch);
    if (!rc) rc = oracle_move_f32(mosaic, mw, mh, ox, oy, b, cw, ch);
    if (!rc) rc = oracle_blend_f32(a, b, cw, ch, opts, out, NULL);
    free(a);
    free(b);
    return rc;
}

/* E1/M1 forward transform: equalization.cpp:78-81 and ImageProcess.cpp:242-244 / :252-254.  Each is a
 * double expression (double literals times float-cast pixels) stored to float, then clamped with
 * `v > 0 ? (v < 256 ? v : 255) : 0` in float. */
static inline float clamp256(float v) { return v > 0 ? (v < 256 ? v : 255.f) : 0.f; }
static inline void rgb_to_ycbcr(float r, float g, float b, float *Y, float *Cb, float *Cr) {
    const float y = (float)(0.299 * r + 0.857 * g + 0.114 * b); /* 0.857 (sic), equalization.cpp:78 */
    const float cb = (float)(128.0 - 0.168736 * r - 0.331264 * g + 0.5 * b);
    const float cr = (float)(128.0 + 0.5 * r - 0.418688 * g - 0.081312 * b);
    *Y = clamp256(y);
    *Cb = clamp256(cb);
    *Cr = clamp256(cr);
}
/* E3/M1 inverse transform: equalization.cpp:93-98, ImageProcess.cpp:262-267 */
static inline void ycbcr_to_rgb_u8(float Y, float Cb, float Cr, uint8_t *r, uint8_t *g, uint8_t *b) {
    const float R = (float)((double)Y + 1.402 * ((double)Cr - 128.0));
    const float G = (float)((double)Y - 0.34414 * ((double)Cb - 128.0) - 0.71414 * ((double)Cr - 128.0));
    const float B = (float)((double)Y + 1.772 * ((double)Cb - 128.0));
    *r = (uint8_t)clamp256(R);
    *g = (uint8_t)clamp256(G);
    *b = (uint8_t)clamp256(B);
}

/* E1-E3  equalization::equalization(src, 1): equalization.cpp:4-25 -> colorHistogramEqualization :74-100 ->
 * equalizationStep :102-131 */
int oracle_equalize_u8(uint8_t *img, int w, int h, int32_t hist_out[256], int32_t lut_out[256]) {
    if (!img || w <= 0 || h <= 0) return ORACLE_ERR_ARG;
    const size_t n = (size_t)w * h;
    uint8_t *ycc = (uint8_t *)malloc(n * 3);
    int32_t hist[256] = {0};
    for (size_t i = 0; i < n; ++i) { /* :77-85, stored into CImg<unsigned char>: truncation */
        float Y, Cb, Cr;
        rgb_to_ycbcr((float)img[i], (float)img[i + n], (float)img[i + 2 * n], &Y, &Cb, &Cr);
        ycc[i] = (uint8_t)Y;
        ycc[i + n] = (uint8_t)Cb;
        ycc[i + 2 * n] = (uint8_t)Cr;
        hist[ycc[i]] += 1; /* :104-107 */
    }
    double prob[256], cdf[256];
    int32_t lut[256];
    const double total = (double)(w * h); /* int product, :114 */
    for (int i = 0; i < 256; ++i) prob[i] = (double)hist[i] / total;
    cdf[0] = prob[0];
    lut[0] = (int32_t)round(255.0 * cdf[0]);
    for (int i = 1; i < 256; ++i) {
        cdf[i] = cdf[i - 1] + prob[i];
        lut[i] = (int32_t)round(255.0 * cdf[i]); /* half away from zero */
    }
    for (size_t i = 0; i < n; ++i) {
        const uint8_t yeq = (uint8_t)lut[ycc[i]]; /* :127-130 */
        ycbcr_to_rgb_u8((float)yeq, (float)ycc[i + n], (float)ycc[i + 2 * n], &img[i], &img[i + n], &img[i + 2 * n]);
    }
    if (hist_out) memcpy(hist_out, hist, sizeof(hist));
    if (lut_out) memcpy(lut_out, lut, sizeof(lut));
    free(ycc);
    return ORACLE_OK;
}

/* M1  ImageProcess.cpp:240-268: float (untruncated) YCbCr of result and of the equalised copy;
 * Y = Y*num/den + Yeq/den evaluated in double and stored to float; back to RGB; truncated into result. */
int oracle_lummix_u8(uint8_t *result, const uint8_t *equalized, int w, int h, double num, double den) {
    if (!result || !equalized || w <= 0 || h <= 0) return ORACLE_ERR_ARG;
    const size_t n = (size_t)w * h;
#pragma omp parallel for schedule(static)
    for (long long ii = 0; ii < (long long)n; ++ii) {
        const size_t i = (size_t)ii;
        float Y, Cb, Cr, Ye, Cbe, Cre;
        rgb_to_ycbcr((float)result[i], (float)result[i + n], (float)result[i + 2 * n], &Y, &Cb, &Cr);
        rgb_to_ycbcr((float)equalized[i], (float)equalized[i + n], (float)equalized[i + 2 * n], &Ye, &Cbe, &Cre);
        const float Ym = (float)((double)Y * num / den + (double)Ye / den); /* :261 */
        ycbcr_to_rgb_u8(Ym, Cb, Cr, &result[i], &result[i + n], &result[i + 2 * n]);
    }
    return ORACLE_OK;
}

/* ImageProcess::toGrayScale, ImageProcess.cpp:27-40 (double expression on float-cast pixels, truncated into an
 * unsigned char image) and siftAlgorithm's float staging, :47-51 */
int oracle_gray_u8(const uint8_t *rgb, int w, int h, uint8_t *gray, float *gray_f32) {
    if (!rgb || w <= 0 || h <= 0) return ORACLE_ERR_ARG;
    const size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t v = (uint8_t)(0.299 * (float)rgb[i] + 0.587 * (float)rgb[i + n] + 0.114 * (float)rgb[i + 2 * n]);
        if (gray) gray[i] = v;
        if (gray_f32) gray_f32[i] = (float)v;
    }
    return ORACLE_OK;
}

/* ImageProcess.cpp:206-216 with getMin/MaxX/YAfterWarping (:532-594): corners visited in the order
 * (0,0) (w-1,0) (0,h-1) (w-1,h-1) with strict comparisons; clamps and the ceil of a FLOAT difference as written. */
int oracle_canvas_bbox(int fw, int fh, const double p[8], int result_w, int result_h, float *min_x, float *min_y, int *new_w,
                       int *new_h) {
    if (!p || fw <= 0 || fh <= 0) return ORACLE_ERR_ARG;
    const float cx[4] = {0.f, (float)(fw - 1), 0.f, (float)(fw - 1)}, cy[4] = {0.f, 0.f, (float)(fh - 1), (float)(fh - 1)};
    float X, Y;
    oracle_map_xy(cx[0], cy[0], p, &X, &Y);
    float mnx = X, mxx = X, mny = Y, mxy = Y;
    for (int i = 1; i < 4; ++i) {
        oracle_map_xy(cx[i], cy[i], p, &X, &Y);
        if (X < mnx) mnx = X;
        if (X > mxx) mxx = X;
        if (Y < mny) mny = Y;
        if (Y > mxy) mxy = Y;
    }
    mnx = (mnx < 0) ? mnx : 0;
    mny = (mny < 0) ? mny : 0;
    mxx = (mxx >= result_w) ? mxx : result_w;
    mxy = (mxy >= result_h) ? mxy : result_h;
    *min_x = mnx;
    *min_y = mny;
    *new_w = (int)ceilf(mxx - mnx);
    *new_h = (int)ceilf(mxy - mny);
    return ORACLE_OK;
}

/* updateFeaturesByHomography, ImageProcess.cpp:622-631; updateFeaturesByOffset, :633-640 */
void oracle_map_points(float *x, float *y, int32_t *ix, int32_t *iy, int n, const double p[8], float offx, float offy) {
    for (int i = 0; i < n; ++i) {
        float X, Y;
        oracle_map_xy(x[i], y[i], p, &X, &Y);
        x[i] = X - offx;
        y[i] = Y - offy;
        if (ix) ix[i] = (int32_t)x[i];
        if (iy) iy[i] = (int32_t)y[i];
    }
}
void oracle_shift_points(float *x, float *y, int32_t *ix, int32_t *iy, int n, int ox, int oy) {
    for (int i = 0; i < n; ++i) {
        x[i] -= ox;
        y[i] -= oy;
        if (ix) ix[i] = (int32_t)x[i];
        if (iy) iy[i] = (int32_t)y[i];
    }
}

/* Synthetic frames, SURVEY.md 8(d): v = 1 + ((3x+5y+37c+101f) mod 200) + (splitmix64(seed ^ key) mod 50),
 * never 0.  The float twin adds frac = ((hash>>32)&0xFFFF)/65536. */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
static inline uint64_t synth_hash(int x, int y, int c, int f) {
    const uint64_t key = ((uint64_t)f << 40) | ((uint64_t)c << 36) | ((uint64_t)y << 18) | (uint64_t)x;
    return splitmix64(0x5717C4EDULL ^ key);
}
void oracle_synth_u8(uint8_t *dst, int w, int h, int f) {
#pragma omp parallel for schedule(static)
    for (int row = 0; row < 3 * h; ++row) {
        const int c = row / h, y = row % h;
        for (int x = 0; x < w; ++x) {
            const uint64_t hsh = synth_hash(x, y, c, f);
            dst[(size_t)c * w * h + (size_t)y * w + x] =
                (uint8_t)(1 + ((3LL * x + 5LL * y + 37LL * c + 101LL * f) % 200) + (int)(hsh % 50));
        }
    }
}
void oracle_synth_f32(float *dst, int w, int h, int f) {
#pragma omp parallel for schedule(static)
    for (int row = 0; row < 3 * h; ++row) {
        const int c = row / h, y = row % h;
        for (int x = 0; x < w; ++x) {
            const uint64_t hsh = synth_hash(x, y, c, f);
            const int v = 1 + (int)((3LL * x + 5LL * y + 37LL * c + 101LL * f) % 200) + (int)(hsh % 50);
            dst[(size_t)c * w * h + (size_t)y * w + x] = (float)v + (float)((hsh >> 32) & 0xFFFF) / 65536.0f;
        }
    }
}

/* ---- the on-disk format either side of the path -------------------------------------
 * CImg<T>::_load_bmp (CImg.h:48395-48566) and _save_bmp (CImg.h:52614-52700), for the uncompressed 24- and
 * 32-bit layouts (the reference's Input/ files and its result file are 24-bit).  File image = one byte array. */
static int le32(const uint8_t *p) { return (int)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }

int oracle_bmp_parse(const uint8_t *file, size_t n, oracle_bmp_info *info) {
    if (!file || !info || n < 54) return ORACLE_ERR_ARG;
    if (file[0] != 'B' || file[1] != 'M') return ORACLE_ERR_ARG; /* CImg.h:48404 */
    int file_size = le32(file + 0x02);                             /* CImg.h:48413-48421 */
    const int offset = le32(file + 0x0A), header_size = le32(file + 0x0E), dx = le32(file + 0x12), dy = le32(file + 0x16),
              compression = le32(file + 0x1E), bpp = file[0x1C] + (file[0x1D] << 8);
    if (!file_size || file_size == offset) file_size = (int)n;     /* CImg.h:48423-48427 */
    if (compression) return ORACLE_ERR_ARG;                        /* CImg.h:48452-48462: external converter */
    if (bpp != 24 && bpp != 32) return ORACLE_ERR_ARG;             /* palette / 16-bit layouts are not on this path */
    if (dx <= 0 || dy == 0 || dy == INT32_MIN) return ORACLE_ERR_ARG;
    const int h = dy < 0 ? -dy : dy;
    const long long dx_bytes = (long long)dx * bpp / 8;            /* CImg.h:48431 */
    const int align_bytes = (int)((4 - dx_bytes % 4) % 4);         /* CImg.h:48432 */
    const unsigned long long want = (unsigned long long)h * (unsigned long long)(dx_bytes + align_bytes);
    /* CImg.h:48435: min(rows * stride, (ulongT)file_size - offset); the subtraction is unsigned */
    const unsigned long long avail_hdr = (unsigned long long)(long long)file_size - (unsigned long long)(long long)offset;
    unsigned long long buf = want < avail_hdr ? want : avail_hdr;
    /* position of the first pixel byte: 54, + header_size-40 if larger (CImg.h:48428), + xoffset if positive (:48440-41) */
    long long pos = 54;
    if (header_size > 40) pos += header_size - 40;
    const long long xoffset = (long long)offset - 14 - header_size;
    if (xoffset > 0) pos += xoffset;
    if (pos < 0 || (unsigned long long)pos > n) return ORACLE_ERR_ARG;
    const unsigned long long in_file = n - (unsigned long long)pos; /* fread stops at the end of the file */
    if (buf > in_file) buf = in_file;
    info->width = dx;
    info->height = h;
    info->bpp = bpp;
    info->top_down = dy < 0;
    info->data_pos = (uint64_t)pos;
    info->stride = (uint64_t)(dx_bytes + align_bytes);
    info->data_bytes = buf; /* bytes of pixel data actually present; the rest of the buffer reads as 0 (CImg.h:48445) */
    return ORACLE_OK;
}

int oracle_bmp_decode_u8(const uint8_t *file, size_t n, uint8_t *planar) {
    oracle_bmp_info bi;
    int rc = oracle_bmp_parse(file, n, &bi);
    if (rc) return rc;
    const int w = bi.width, h = bi.height, bpp_bytes = bi.bpp / 8;
    const size_t pl = (size_t)w * h;
    for (int y = 0; y < h; ++y) {
        /* the file's first row is the image's last (CImg.h:48536 walks y = height-1 .. 0); dy < 0 mirrors (:48563) */
        const int img_y = bi.top_down ? y : h - 1 - y;
        for (int x = 0; x < w; ++x)
            for (int k = 0; k < 3; ++k) { /* bytes B, G, R -> channels 2, 1, 0 (CImg.h:48540-48542) */
                const uint64_t o = (uint64_t)y * bi.stride + (uint64_t)x * bpp_bytes + k;
                planar[(size_t)(2 - k) * pl + (size_t)img_y * w + x] = o < bi.data_bytes ? file[bi.data_pos + o] : 0;
            }
    }
    return ORACLE_OK;
}

size_t oracle_bmp_file_bytes(int w, int h) {
    if (w <= 0 || h <= 0) return 0;
    const unsigned align = (4 - (3u * (unsigned)w) % 4) % 4; /* CImg.h:52635 */
    return 54 + ((size_t)3 * w + align) * (size_t)h;
}

int oracle_bmp_encode_u8(const uint8_t *planar, int w, int h, uint8_t *file, size_t cap) {
    const size_t total = oracle_bmp_file_bytes(w, h);
    if (!planar || !file || !total || cap < total) return ORACLE_ERR_ARG;
    const unsigned align = (4 - (3u * (unsigned)w) % 4) % 4;
    const unsigned buf_size = (unsigned)(total - 54), file_size = (unsigned)total; /* CImg.h:52636-52637 */
    memset(file, 0, 54);
    file[0] = 'B';
    file[1] = 'M';
    for (int i = 0; i < 4; ++i) {
        file[0x02 + i] = (uint8_t)(file_size >> (8 * i));
        file[0x12 + i] = (uint8_t)((unsigned)w >> (8 * i));
        file[0x16 + i] = (uint8_t)((unsigned)h >> (8 * i));
        file[0x22 + i] = (uint8_t)(buf_size >> (8 * i));
    }
    file[0x0A] = 0x36; /* CImg.h:52643-52664 */
    file[0x0E] = 0x28;
    file[0x1A] = 1;
    file[0x1C] = 24;
    file[0x27] = 0x1;
    file[0x2B] = 0x1;
    const size_t pl = (size_t)w * h, stride = (size_t)3 * w + align;
    for (int r = 0; r < h; ++r) { /* rows bottom-up, bytes B G R, zero padding (CImg.h:52690-52699) */
        uint8_t *row = file + 54 + (size_t)r * stride;
        const size_t src = (size_t)(h - 1 - r) * w;
        for (int x = 0; x < w; ++x) {
            row[3 * x] = planar[2 * pl + src + x];
            row[3 * x + 1] = planar[pl + src + x];
            row[3 * x + 2] = planar[src + x];
        }
        for (unsigned a = 0; a < align; ++a) row[(size_t)3 * w + a] = 0;
    }
    return ORACLE_OK;
}

/* ---- transfer.cpp:3-13,125-225, the l-alpha-beta colour transfer (dead code in the reference:
 * ImageProcess.cpp:180-182 are commented out).  Restatement of the per-pixel arithmetic, which is the same in the
 * serial branch (transfer.cpp:71-78,113-121) and in the Win32-threaded one (:15-41, rows are independent).
 * PARITY UNPINNED: transfer.cpp does not compile here (windows.h threads) and the reference holds no output of it.
 * The two libm calls (std::log(float), std::pow(10, float) -> pow(double,double)) are the specified functions of
 * include/stitch_elem.h unless use_libm is set (then logf / pow of this platform's libm, for measuring the distance). */
#include "../include/stitch_elem.h"
static float tr_log(float v, int use_libm) { return use_libm ? logf(v) : stitch_elem_logf(v); }
static double tr_pow10(double v, int use_libm) { return use_libm ? pow(10.0, v) : stitch_elem_pow10(v); }

static void tr_rgb_to_lab(float R, float G, float B, float *L, float *a, float *b, int use_libm) { /* transfer.cpp:176-199 */
    float l = (float)(0.3811 * (double)R + 0.5783 * (double)G + 0.0402 * (double)B);
    float m = (float)(0.1967 * (double)R + 0.7244 * (double)G + 0.0782 * (double)B);
    float s = (float)(0.0241 * (double)R + 0.1288 * (double)G + 0.8444 * (double)B);
    if (l == 0) l = 1;
    if (m == 0) m = 1;
    if (s == 0) s = 1;
    const double ln10 = 2.302585092994046; /* log(10): the double nearest ln 10, what any libm returns for this constant */
    l = (float)((double)tr_log(l, use_libm) / ln10);
    m = (float)((double)tr_log(m, use_libm) / ln10);
    s = (float)((double)tr_log(s, use_libm) / ln10);
    const float paraA = (float)(1.0 / sqrt(3.0)), paraB = (float)(1.0 / sqrt(6.0)), paraC = (float)(1.0 / sqrt(2.0));
    *L = paraA * ((l + m) + s);
    *a = (float)((double)(paraB * l + paraB * m) - (2.0 * (double)paraB) * (double)s);
    *b = paraC * l - paraC * m;
}
static void tr_lab_to_rgb(float L, float a, float b, float *R, float *G, float *B, int use_libm) { /* transfer.cpp:201-225 */
    const float paraA = (float)(sqrt(3.0) / 3.0), paraB = (float)(sqrt(6.0) / 6.0), paraC = (float)(sqrt(2.0) / 2.0);
    float l = (paraA * L + paraB * a) + paraC * b;
    float m = (paraA * L + paraB * a) - paraC * b;
    float s = (float)((double)(paraA * L) - (2.0 * (double)paraB) * (double)a);
    l = (float)tr_pow10((double)l, use_libm);
    m = (float)tr_pow10((double)m, use_libm);
    s = (float)tr_pow10((double)s, use_libm);
    float r = (float)((4.4679 * (double)l - 3.5873 * (double)m) + 0.1193 * (double)s);
    float g = (float)(((-1.2186) * (double)l + 2.3809 * (double)m) - 0.1624 * (double)s);
    float bb = (float)((0.0497 * (double)l - 0.2439 * (double)m) + 1.2045 * (double)s);
    *R = r > 0.0f ? (r < 255.0f ? r : 255.0f) : 0.0f;
    *G = g > 0.0f ? (g < 255.0f ? g : 255.0f) : 0.0f;
    *B = bb > 0.0f ? (bb < 255.0f ? bb : 255.0f) : 0.0f;
}
static float *tr_lab_image(const uint8_t *rgb, int w, int h, int use_libm) { /* transfer.cpp:4-9, :83-123 */
    const size_t n = (size_t)w * h;
    float *lab = (float *)malloc(sizeof(float) * 3 * n);
    if (!lab) return NULL;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i)
        tr_rgb_to_lab((float)rgb[i], (float)rgb[n + i], (float)rgb[2 * n + i], &lab[i], &lab[n + i], &lab[2 * n + i], use_libm);
    return lab;
}
/* mean and standard deviation exactly as transfer.cpp:128-164: float accumulators, raster order, serial */
static void tr_stats(const float *lab, size_t n, int w, int h, float mean[3], float sd[3]) {
    for (int c = 0; c < 3; ++c) {
        const float *p = lab + (size_t)c * n;
        float acc = 0;
        for (size_t i = 0; i < n; ++i) acc += p[i];
        mean[c] = acc / (float)(w * h);
        float var = 0;
        for (size_t i = 0; i < n; ++i) var += (p[i] - mean[c]) * (p[i] - mean[c]);
        sd[c] = sqrtf(var / (float)(w * h));
    }
}
int oracle_transfer_u8(const uint8_t *src, int sw, int sh, const uint8_t *tem, int tw, int th, uint8_t *out, float stats[12],
                       int use_libm) {
    if (!src || !tem || !out || sw <= 0 || sh <= 0 || tw <= 0 || th <= 0) return ORACLE_ERR_ARG;
    const size_t n = (size_t)sw * sh, nt = (size_t)tw * th;
    float *ls = tr_lab_image(src, sw, sh, use_libm), *lt = tr_lab_image(tem, tw, th, use_libm);
    if (!ls || !lt) {
        free(ls);
        free(lt);
        return ORACLE_ERR_ARG;
    }
    float ms[3], ss[3], mt[3], st[3];
    tr_stats(ls, n, sw, sh, ms, ss);
    tr_stats(lt, nt, tw, th, mt, st);
    if (stats)
        for (int c = 0; c < 3; ++c) {
            stats[c] = ms[c];
            stats[3 + c] = ss[c];
            stats[6 + c] = mt[c];
            stats[9 + c] = st[c];
        }
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) { /* transfer.cpp:165-171, then :43-81, then the cast of :12 */
        float v[3], R, G, B;
        for (int c = 0; c < 3; ++c) v[c] = (ls[(size_t)c * n + i] - ms[c]) * st[c] / ss[c] + mt[c];
        tr_lab_to_rgb(v[0], v[1], v[2], &R, &G, &B, use_libm);
        out[i] = (uint8_t)R;
        out[n + i] = (uint8_t)G;
        out[2 * n + i] = (uint8_t)B;
    }
    free(ls);
    free(lt);
    return ORACLE_OK;
}
