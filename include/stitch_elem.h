/* Specified elementary functions for stitch_transfer_u8 (the lαβ colour transfer, transfer.cpp:176-225).
 *
 * The reference calls its platform's libm: std::log(float) and std::pow(double, double) (transfer.cpp:188-190,
 * :212-214; the authors linked the MSVC runtime).  No two libms agree bit for bit, so the result of the reference on
 * this path is platform-defined in its last bit.  To make the CPU restatement (oracle/) and the HIP kernels agree
 * exactly, both evaluate the two functions below: plain IEEE-754 double +,-,*,/ in a fixed order, no FMA contraction
 * (both sides are compiled with -ffp-contract=off), accurate to a few 1e-16 relative, so that the float the reference
 * rounds to is reproduced except within ~1e-8 ulp of a rounding boundary.  tests/test_oracle_golden.py bounds the
 * distance to glibc's logf / pow.
 */
#ifndef STITCH_ELEM_H
#define STITCH_ELEM_H
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define STITCH_HD __host__ __device__ __forceinline__
#else
#define STITCH_HD static inline
#endif

STITCH_HD double stitch_elem_bits_to_double(uint64_t u) {
    double d;
    memcpy(&d, &u, sizeof d);
    return d;
}
STITCH_HD uint64_t stitch_elem_double_to_bits(double d) {
    uint64_t u;
    memcpy(&u, &d, sizeof u);
    return u;
}

/* natural logarithm of a positive, finite, normal float, rounded to float (stands for std::log(float)).
 * x = m * 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716: 13 odd terms. */
STITCH_HD float stitch_elem_logf(float x) {
    const double xd = (double)x;
    uint64_t u = stitch_elem_double_to_bits(xd);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; /* mantissa in [1, 2) */
    double m = stitch_elem_bits_to_double(u);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 25.0;
    p = p * z + 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    const double logm = 2.0 * s * p;
    /* ln 2 split so that e * LN2_HI is exact for |e| < 2^10 */
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double y = (double)e * LN2_HI + (logm + (double)e * LN2_LO);
    return (float)y;
}

/* 10^y in double (stands for std::pow(10, float) = pow(10.0, (double)y)); |y| below ~300, otherwise 0 / inf.
 * 10^y = 2^n * e^r with t = y*log2(10) carried as hi + lo, n = nearest integer of t, r = (t - n) ln 2. */
STITCH_HD double stitch_elem_pow10(double y) {
    if (!(y == y)) return y;
    if (y > 308.3) return stitch_elem_bits_to_double(0x7ff0000000000000ULL);
    if (y < -323.4) return 0.0;
    const double L2T_HI = 3.32192809488736218171e+00; /* log2(10) rounded to double */
    const double L2T_LO = 1.66184682231674719e-16;    /* log2(10) - L2T_HI */
    /* product y * L2T_HI as hi + lo by Dekker splitting (exact without FMA) */
    const double SPLIT = 134217729.0; /* 2^27 + 1 */
    double c = SPLIT * y, yh = c - (c - y), yl = y - yh;
    c = SPLIT * L2T_HI;
    const double lh = c - (c - L2T_HI), ll = L2T_HI - lh;
    const double hi = y * L2T_HI;
    const double lo = ((yh * lh - hi) + yh * ll + yl * lh) + yl * ll + y * L2T_LO;
    const double nd = (double)(long long)(hi + (hi < 0 ? -0.5 : 0.5));
    const double f = (hi - nd) + lo; /* |f| <= 0.5 + tiny */
    const double LN2 = 6.93147180559945286227e-01;
    const double r = f * LN2;
    /* Taylor series of e^r, |r| <= 0.3466, Horner from degree 20: q_k = 1 + (r/k) q_{k+1} */
    double q = 1.0;
    for (int k = 20; k >= 1; --k) q = 1.0 + (r / (double)k) * q;
    /* scale by 2^n in two steps so that subnormal results round once */
    long long n = (long long)nd;
    double scale1 = 1.0, scale2 = 1.0;
    if (n > 1000) {
        scale2 = stitch_elem_bits_to_double((uint64_t)(1023 + 1000) << 52);
        n -= 1000;
    } else if (n < -1000) {
        scale2 = stitch_elem_bits_to_double((uint64_t)(1023 - 1000) << 52);
        n += 1000;
    }
    scale1 = stitch_elem_bits_to_double((uint64_t)(1023 + n) << 52);
    return q * scale1 * scale2;
}

#endif
