// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// The `src/ex6` VARIANT of the reference (the later, submitted version: Deriche blur, pyramid depth from min(w,h), seam
// scan on all three channels with double ratios), compiled from the sources where they lie under
// /root/reference/src/ex6 (nothing is copied into this repo).  oracle/_ref/libref6_hotpath.so exists to pin the
// variant options of the oracle (seam_rule = 1, level_rule = 1, blur_kind = 1) on the WHOLE blend function:
//   ImageProcess::blend   src/ex6/ImageProcess.cpp:638-742
// Built by oracle/Makefile (target `ref`) with hidden visibility and -Bsymbolic, so that it can be loaded next to the root
// variant's library (same class names) without either seeing the other's symbols.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <queue>
#include <set>
#include <string>
#include <thread>
#include <vector>
#include <time.h>

#include "CImg.h"

#define private public
#define display(...) is_empty()  // result.display("result") throws when cimg_display == 0 (CImg.h:7228-7232)
#include "ImageProcess.cpp"
#undef display
#undef private

typedef CImg<unsigned char> U8Img;

extern "C" __attribute__((visibility("default"))) int ref6_blend_u8(const uint8_t *a, const uint8_t *b, int w, int h, uint8_t *out) {
    // blend() never touches `this`: any suitably aligned storage will do (the constructor runs the whole SIFT pipeline)
    static std::aligned_storage<sizeof(ImageProcess), alignof(ImageProcess)>::type buf;
    U8Img A(a, w, h, 1, 3, true), B(b, w, h, 1, 3, true);
    U8Img r = reinterpret_cast<ImageProcess *>(&buf)->blend(A, B);
    if (r.width() != w || r.height() != h || r.spectrum() != 3) return -1;
    std::memcpy(out, r.data(), (size_t)w * h * 3);
    return 0;
}
