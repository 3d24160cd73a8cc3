// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// C-ABI wrappers around the REFERENCE's own hot-path functions, compiled from the sources where
// they lie under /root/reference (nothing is copied into this repo).  The result,
// oracle/_ref/libref_hotpath.so, exists only to (1) pin oracle/stitch_oracle.c and (2) generate
// the golden vectors under tests/golden/ (tests/golden/make_golden.py).  It is built by
// oracle/Makefile (target `ref`), only in the container that has /root/reference.
//
// Reference entry points wrapped (file:line in /root/reference):
//   Projection::imageProjection            Projection.cpp:20-73
//   Projection::bilinearInterpolation      Projection.cpp:3-18
//   ImageProcess::getX/YAfterWarping       ImageProcess.cpp:465-471
//   ImageProcess::warpingImageByHomography ImageProcess.cpp:596-606
//   ImageProcess::movingImageByOffset      ImageProcess.cpp:608-620
//   ImageProcess::blendTwoImages           ImageProcess.cpp:648-773
//   equalization::equalization(mode 1)     equalization.cpp:4-25,74-131
//   luminance mix (inline in matching())   ImageProcess.cpp:237-268  (reached through ref_pipeline)
//   CImg<float>::get_blur / get_resize     CImg.h:35145, CImg.h:29344  (the blend's arithmetic)
//
// The three ImageProcess members are private and the constructor runs the whole SIFT pipeline, so
// the reference TU is compiled through this wrapper TU with `private` opened up and with
// `result.display()` (which throws when cimg_display==0, CImg.h:7228-7232) renamed to a harmless
// const accessor.  The member functions used here never touch `this`, so they are invoked on raw
// storage without running the constructor.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <queue>
#include <set>
#include <string>
#include <vector>
#include <time.h>

#include "CImg.h"

#define private public
#define display is_empty
#include "ImageProcess.cpp"
#undef display
#undef private

#define REF_API extern "C" __attribute__((visibility("default")))

typedef CImg<unsigned char> U8Img;

static ImageProcess *fake_ip() {
    // warping/moving/blend/getX/getY are stateless members: any suitably aligned storage will do.
    static std::aligned_storage<sizeof(ImageProcess), alignof(ImageProcess)>::type buf;
    return reinterpret_cast<ImageProcess *>(&buf);
}

static Homography make_h(const double p[8]) {
    return Homography(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
}

REF_API int ref_project_u8(const uint8_t *src, int w, int h, uint8_t *dst) {
    U8Img s(src, w, h, 1, 3);
    U8Img r = Projection::imageProjection(s);
    if (r.width() != w || r.height() != h || r.spectrum() != 3) return -1;
    std::memcpy(dst, r.data(), (size_t)w * h * 3);
    return 0;
}

REF_API int ref_bilinear_u8(const uint8_t *src, int w, int h, float x, float y, int c) {
    U8Img s(src, w, h, 1, 3, true);
    return Projection::bilinearInterpolation(s, x, y, c);
}

REF_API void ref_map_xy(float x, float y, const double p[8], float *X, float *Y) {
    Homography H = make_h(p);
    *X = fake_ip()->getXAfterWarping(x, y, H);
    *Y = fake_ip()->getYAfterWarping(x, y, H);
}

REF_API void ref_bbox(int w, int h, const double p[8], float out[4]) {
    // min_x, min_y, max_x, max_y of the four warped corners (ImageProcess.cpp:532-594)
    Homography H = make_h(p);
    U8Img s(w, h, 1, 3, 0);
    out[0] = fake_ip()->getMinXAfterWarping(s, H);
    out[1] = fake_ip()->getMinYAfterWarping(s, H);
    out[2] = fake_ip()->getMaxXAfterWarping(s, H);
    out[3] = fake_ip()->getMaxYAfterWarping(s, H);
}

REF_API int ref_gray_u8(const uint8_t *rgb, int w, int h, uint8_t *gray) {
    // ImageProcess::toGrayScale, ImageProcess.cpp:27-40
    U8Img s(rgb, w, h, 1, 3, true);
    U8Img g = fake_ip()->toGrayScale(s);
    if (g.width() != w || g.height() != h || g.spectrum() != 1) return -1;
    std::memcpy(gray, g.data(), (size_t)w * h);
    return 0;
}

REF_API void ref_update_features(float *x, float *y, int *ix, int *iy, int n, const double p[8], float offx, float offy, int ox,
                                 int oy, int by_offset) {
    // updateFeaturesByHomography / updateFeaturesByOffset, ImageProcess.cpp:622-640, on a feature map built here
    std::map<std::vector<float>, VlSiftKeypoint> feat;
    for (int i = 0; i < n; ++i) {
        VlSiftKeypoint k;
        std::memset(&k, 0, sizeof k);
        k.x = x[i];
        k.y = y[i];
        feat[std::vector<float>(1, (float)i)] = k;  // keys keep the insertion order 0..n-1
    }
    Homography H = make_h(p);
    if (by_offset)
        fake_ip()->updateFeaturesByOffset(feat, ox, oy);
    else
        fake_ip()->updateFeaturesByHomography(feat, H, offx, offy);
    int i = 0;
    for (auto it = feat.begin(); it != feat.end(); ++it, ++i) {
        x[i] = it->second.x;
        y[i] = it->second.y;
        ix[i] = it->second.ix;
        iy[i] = it->second.iy;
    }
}

REF_API void ref_canvas(int fw, int fh, const double p[8], int result_w, int result_h, float *min_x, float *min_y, int *new_w,
                        int *new_h) {
    // the canvas sizing statements of matching(), ImageProcess.cpp:206-216, on the reference's own corner functions
    Homography H = make_h(p);
    U8Img s(fw, fh, 1, 3, 0);
    float mnx = fake_ip()->getMinXAfterWarping(s, H);
    mnx = (mnx < 0) ? mnx : 0;
    float mny = fake_ip()->getMinYAfterWarping(s, H);
    mny = (mny < 0) ? mny : 0;
    float mxx = fake_ip()->getMaxXAfterWarping(s, H);
    mxx = (mxx >= result_w) ? mxx : result_w;
    float mxy = fake_ip()->getMaxYAfterWarping(s, H);
    mxy = (mxy >= result_h) ? mxy : result_h;
    *min_x = mnx;
    *min_y = mny;
    *new_w = ceil(mxx - mnx);
    *new_h = ceil(mxy - mny);
}

REF_API int ref_warp_u8(const uint8_t *src, int sw, int sh, const double p[8], float offx, float offy,
                        uint8_t *canvas, int cw, int ch) {
    U8Img s(src, sw, sh, 1, 3, true);
    U8Img d(canvas, cw, ch, 1, 3, true);  // shared: written in place, untouched where out of range
    Homography H = make_h(p);
    fake_ip()->warpingImageByHomography(s, d, H, offx, offy);
    return 0;
}

REF_API int ref_move_u8(const uint8_t *src, int sw, int sh, int ox, int oy, uint8_t *canvas, int cw, int ch) {
    U8Img s(src, sw, sh, 1, 3, true);
    U8Img d(canvas, cw, ch, 1, 3, true);
    fake_ip()->movingImageByOffset(s, d, ox, oy);
    return 0;
}

REF_API int ref_blend_u8(const uint8_t *a, const uint8_t *b, int w, int h, uint8_t *out) {
    U8Img A(a, w, h, 1, 3, true), B(b, w, h, 1, 3, true);
    U8Img r = fake_ip()->blendTwoImages(A, B);
    if (r.width() != w || r.height() != h || r.spectrum() != 3) return -1;
    std::memcpy(out, r.data(), (size_t)w * h * 3);
    return 0;
}

REF_API int ref_equalize_u8(uint8_t *img, int w, int h) {
    U8Img s(img, w, h, 1, 3);
    equalization eq(s, 1);
    std::memcpy(img, s.data(), (size_t)w * h * 3);
    return 0;
}

// CImg<float> primitives the blend is made of.
REF_API void ref_cimg_blur_f32(float *img, int w, int h, int c, float sigma, int is_gaussian) {
    CImg<float> I(img, w, h, 1, c, true);
    CImg<float> r = I.get_blur(sigma, true, is_gaussian != 0);
    std::memcpy(img, r.data(), sizeof(float) * (size_t)w * h * c);
}

REF_API int ref_cimg_resize_f32(const float *img, int w, int h, int c, float *out, int w2, int h2, int c2,
                                int interp) {
    CImg<float> I(img, w, h, 1, c, true);
    CImg<float> r = I.get_resize(w2, h2, 1, c2, interp);
    if (r.width() != w2 || r.height() != h2 || r.spectrum() != c2) return -1;
    std::memcpy(out, r.data(), sizeof(float) * (size_t)w2 * h2 * c2);
    return 0;
}

// Whole program (L3 control flow + SIFT + RANSAC untouched): ImageProcess(dir, n), result copied out.
// Returns 0 and fills w/h; `out` may be NULL to query the size (the run is repeated, it is deterministic:
// srand(666666), ImageProcess.cpp:397).
REF_API int ref_pipeline(const char *dir, int n, uint8_t *out, int cap, int *w, int *h) {
    ImageProcess ip(std::string(dir), n);
    *w = ip.result.width();
    *h = ip.result.height();
    size_t sz = (size_t)ip.result.width() * ip.result.height() * 3;
    if (out && (size_t)cap >= sz) std::memcpy(out, ip.result.data(), sz);
    return 0;
}

REF_API int ref_load_bmp(const char *path, uint8_t *out, int cap, int *w, int *h) {
    U8Img s(path);
    *w = s.width();
    *h = s.height();
    if (s.spectrum() != 3) return -2;
    size_t sz = (size_t)s.width() * s.height() * 3;
    if (out && (size_t)cap >= sz) std::memcpy(out, s.data(), sz);
    return 0;
}

// CImg<unsigned char>::save_bmp (CImg.h:52605) on a planar RGB image
REF_API int ref_save_bmp(const uint8_t *planar, int w, int h, const char *path) {
    U8Img s(planar, w, h, 1, 3, true);
    s.save_bmp(path);
    return 0;
}
