/* TEST INFRASTRUCTURE ONLY -- sanitizer run of the oracle ("compile CPU restatement tests with
 * -fsanitize=address,undefined").  Exercises every oracle entry point on small synthetic inputs, odd and even sizes,
 * and prints one checksum; oracle/Makefile target `sanitize-check` builds it twice (plain and ASan+UBSan) and
 * requires identical output and no sanitizer report. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "stitch_oracle.h"

static unsigned long long fnv(const void *p, size_t n, unsigned long long h) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ULL;
    return h;
}

int main(void) {
    unsigned long long h = 1469598103934665603ULL;
    const int sizes[][2] = {{67, 33}, {100, 64}, {33, 67}, {4, 2}, {131, 128}, {3, 3}, {2, 2}};
    oracle_blend_opts root = {2.0f, 0, 0, 0}, ex6 = {2.0f, 1, 1, 1};
    for (unsigned s = 0; s < sizeof sizes / sizeof sizes[0]; ++s) {
        const int w = sizes[s][0], hh = sizes[s][1];
        const size_t n = (size_t)w * hh * 3;
        uint8_t *a = malloc(n), *b = malloc(n), *o = malloc(n), *p = malloc(n);
        float *af = malloc(n * 4), *bf = malloc(n * 4), *of = malloc(n * 4);
        oracle_synth_u8(a, w, hh, 1 + s);
        oracle_synth_u8(b, w, hh, 20 + s);
        oracle_synth_f32(af, w, hh, 1 + s);
        oracle_synth_f32(bf, w, hh, 20 + s);
        for (int c = 0; c < 3; ++c)
            for (int y = 0; y < hh; ++y)
                for (int x = 0; x < w; ++x) {
                    const size_t i = (size_t)c * w * hh + (size_t)y * w + x;
                    if (x >= (2 * w) / 3) a[i] = 0, af[i] = 0;
                    if (x < w / 3) b[i] = 0, bf[i] = 0;
                }
        oracle_seam sm;
        int rc = oracle_blend_u8(a, b, w, hh, &root, o, NULL, &sm);
        h = fnv(&rc, sizeof rc, h);
        if (!rc) h = fnv(o, n, h);
        rc = oracle_blend_u8(a, b, w, hh, &ex6, o, NULL, &sm);
        h = fnv(&rc, sizeof rc, h);
        if (!rc) h = fnv(o, n, h);
        rc = oracle_blend_f32(af, bf, w, hh, &root, of, &sm);
        h = fnv(&rc, sizeof rc, h);
        if (!rc) h = fnv(of, n * 4, h);
        oracle_project_u8(a, w, hh, 15.0f, p);
        h = fnv(p, n, h);
        oracle_project_f32(af, w, hh, 15.0f, of);
        h = fnv(of, n * 4, h);
        const double m[8] = {1.01, 0.01, -1e-5, -w / 3.0, 0.002, 0.99, 1e-6, 1.25};
        memset(p, 0, n);
        oracle_warp_u8(a, w, hh, m, -1.5f, 0.75f, p, w, hh);
        oracle_move_u8(b, w, hh, -2, 1, p, w, hh);
        h = fnv(p, n, h);
        rc = oracle_pair_u8(b, w, hh, m, 0.f, 0.f, a, w, hh, 0, 0, w, hh, &root, o);
        h = fnv(&rc, sizeof rc, h);
        int32_t hist[256], lut[256];
        oracle_equalize_u8(p, w, hh, hist, lut);
        h = fnv(hist, sizeof hist, h);
        oracle_lummix_u8(a, p, w, hh, 19.0, 20.0);
        h = fnv(a, n, h);
        free(a); free(b); free(o); free(p); free(af); free(bf); free(of);
    }
    printf("%016llx\n", h);
    return 0;
}
