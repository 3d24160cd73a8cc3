// TEST INFRASTRUCTURE ONLY -- records what the reference's control flow passes to its hot-path functions.
//
// Built into oracle/_ref/libref_record.so (oracle/Makefile, target `ref`).  Loaded with RTLD_GLOBAL *before*
// libref_hotpath.so, its definitions of the three ImageProcess members interpose the reference's own (which
// libref_hotpath.so calls through its PLT): each hook appends the call's arguments to a log, optionally dumps
// the images involved, and forwards to the reference's original implementation found with dlsym().  Used only
// by tests/golden/make_golden.py to capture the transforms / canvas sizes of the reference's Input/ runs
// (the values SURVEY.md 8(c) lists), which then become committed fixtures.
//
// Hooked (file:line under /root/reference):
//   ImageProcess::warpingImageByHomography  ImageProcess.cpp:596-606   called at :222
//   ImageProcess::movingImageByOffset       ImageProcess.cpp:608-620   called at :224
//   ImageProcess::blendTwoImages            ImageProcess.cpp:648-773   called at :230
//   ImageProcess::updateFeaturesByHomography ImageProcess.cpp:622-631  called at :226 (logs the FORWARD map of the step,
//                                            which sizes the canvas at :206-216 and is not seen by the three hooks above)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <string>
#include <vector>

#include "CImg.h"
#define private public
#include "ImageProcess.h"
#undef private

typedef CImg<unsigned char> U8Img;

static void *g_ref = nullptr;
static FILE *g_log = nullptr;
static std::string g_dump_dir;
static int g_step = 0;

typedef void (*warp_fn)(ImageProcess *, const U8Img &, U8Img &, Homography &, float, float);
typedef void (*move_fn)(ImageProcess *, const U8Img &, U8Img &, int, int);
typedef U8Img (*blend_fn)(ImageProcess *, const U8Img &, const U8Img &);
typedef std::map<std::vector<float>, VlSiftKeypoint> FeatMap;
typedef void (*updh_fn)(ImageProcess *, FeatMap &, Homography &, float, float);
static updh_fn o_updh;
static warp_fn o_warp;
static move_fn o_move;
static blend_fn o_blend;

extern "C" __attribute__((visibility("default"))) int rec_init(const char *ref_so, const char *log_path,
                                                               const char *dump_dir) {
    g_ref = dlopen(ref_so, RTLD_NOW | RTLD_LOCAL);
    if (!g_ref) return -1;
    o_warp = (warp_fn)dlsym(g_ref, "_ZN12ImageProcess24warpingImageByHomographyERKN12cimg_library4CImgIhEERS2_R10Homographyff");
    o_move = (move_fn)dlsym(g_ref, "_ZN12ImageProcess19movingImageByOffsetERKN12cimg_library4CImgIhEERS2_ii");
    o_blend = (blend_fn)dlsym(g_ref, "_ZN12ImageProcess14blendTwoImagesERKN12cimg_library4CImgIhEES4_");
    o_updh = (updh_fn)dlsym(g_ref, "_ZN12ImageProcess26updateFeaturesByHomographyERSt3mapISt6vectorIfSaIfEE15_VlSiftKeypointSt4lessIS3_ESaISt4pairIKS3_S4_EEER10Homographyff");
    if (!o_warp || !o_move || !o_blend || !o_updh) return -2;
    if (g_log) fclose(g_log);
    g_log = fopen(log_path, "w");
    g_dump_dir = dump_dir ? dump_dir : "";
    g_step = 0;
    return g_log ? 0 : -3;
}

extern "C" __attribute__((visibility("default"))) void rec_close() {
    if (g_log) fclose(g_log);
    g_log = nullptr;
}

static void dump(const char *tag, const U8Img &img) {
    if (g_dump_dir.empty()) return;
    char name[512];
    snprintf(name, sizeof name, "%s/step%d_%s_%dx%d.raw", g_dump_dir.c_str(), g_step, tag, img.width(), img.height());
    FILE *f = fopen(name, "wb");
    if (!f) return;
    fwrite(img.data(), 1, (size_t)img.width() * img.height() * img.spectrum(), f);
    fclose(f);
}

void ImageProcess::warpingImageByHomography(const U8Img &src, U8Img &dst, Homography &H, float offx, float offy) {
    if (g_log) {
        fprintf(g_log, "warp step=%d sw=%d sh=%d cw=%d ch=%d offx=%.9g offy=%.9g p=", g_step, src.width(), src.height(),
                dst.width(), dst.height(), offx, offy);
        const double p[8] = {H.H[0][0], H.H[0][1], H.H[0][2], H.H[1][0], H.H[1][1], H.H[1][2], H.H[2][0], H.H[2][1]};
        for (int i = 0; i < 8; ++i) fprintf(g_log, "%.17g%s", p[i], i < 7 ? "," : "\n");
        fflush(g_log);
    }
    dump("frame", src);
    o_warp(this, src, dst, H, offx, offy);
}

void ImageProcess::movingImageByOffset(const U8Img &src, U8Img &dst, int ox, int oy) {
    if (g_log) {
        fprintf(g_log, "move step=%d sw=%d sh=%d cw=%d ch=%d ox=%d oy=%d\n", g_step, src.width(), src.height(), dst.width(),
                dst.height(), ox, oy);
        fflush(g_log);
    }
    dump("mosaic", src);
    o_move(this, src, dst, ox, oy);
}

void ImageProcess::updateFeaturesByHomography(FeatMap &feature, Homography &H, float offset_x, float offset_y) {
    if (g_log) {
        fprintf(g_log, "fwd step=%d n=%d offx=%.9g offy=%.9g p=", g_step, (int)feature.size(), offset_x, offset_y);
        const double p[8] = {H.H[0][0], H.H[0][1], H.H[0][2], H.H[1][0], H.H[1][1], H.H[1][2], H.H[2][0], H.H[2][1]};
        for (int i = 0; i < 8; ++i) fprintf(g_log, "%.17g%s", p[i], i < 7 ? "," : "\n");
        fflush(g_log);
    }
    o_updh(this, feature, H, offset_x, offset_y);
}

U8Img ImageProcess::blendTwoImages(const U8Img &a, const U8Img &b) {
    dump("a", a);
    dump("b", b);
    U8Img r = o_blend(this, a, b);
    dump("out", r);
    if (g_log) {
        fprintf(g_log, "blend step=%d w=%d h=%d\n", g_step, r.width(), r.height());
        fflush(g_log);
    }
    ++g_step;
    return r;
}
