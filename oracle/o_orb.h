// ORACLE -- test infrastructure, NOT the product (see o_math.h header).
// o_orb.h: CPU restatement of cv::ORB::detectAndCompute as the reference configures it
// (src/frontend.cpp:35-37, :150-154; OpenCV-3.1 defaults listed in SURVEY.md 8a-1).
#pragma once
#include <cstdint>
#include <vector>

#include "../include/vo_hip.h"

namespace orc {

struct OrbPlan {
    int W = 0, H = 0, nlevels = 0, nfeatures = 0, fast_thr = 20, edge = 31;
    std::vector<int> lw, lh, quota;
    std::vector<float> scale;
    // bilinear resize tables for level l (from level l-1); index 0 unused
    std::vector<std::vector<int>> xofs, yofs;
    std::vector<std::vector<short>> ialpha, ibeta;   // 2 per destination column / row
    int umax[16];
    int gk[7];
};

struct OrbLevelDebug {
    std::vector<uint8_t> gray;      // level image
    std::vector<uint8_t> blurred;   // 7x7 sigma-2 blurred level image
    std::vector<uint8_t> score;     // FAST score map (0 = not a corner)
    std::vector<int> cand_xy;       // NMS survivors inside the border, row-major, packed y*w+x
};

void orb_build_plan(const vo_params& p, OrbPlan& plan);

// Full detect + describe for one frame. `dbg` (optional) receives per-level intermediates.
void orb_detect_describe(const OrbPlan& plan, const uint8_t* bgr, int bgr_stride, const uint16_t* depth,
                         int depth_stride, std::vector<vo_keypoint>& kps, std::vector<uint8_t>& desc,
                         std::vector<OrbLevelDebug>* dbg);

// pieces exposed for unit tests
void bgr_to_gray(const uint8_t* bgr, int stride, int w, int h, uint8_t* gray);
void resize_level(const OrbPlan& plan, int level, const uint8_t* src, uint8_t* dst);
int fast_score(const uint8_t* img, int stride, int x, int y);   // max-threshold corner score, <0 if flat
void gauss_blur7(const OrbPlan& plan, const uint8_t* src, int w, int h, uint8_t* dst);
long long harris_key(const uint8_t* img, int stride, int x, int y);
float fast_atan2_deg(float y, float x);

}  // namespace orc
