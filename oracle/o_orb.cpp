// ORACLE -- test infrastructure, NOT the product (see o_math.h header).
// o_orb.cpp: CPU restatement of the ORB detector/descriptor behind
// FrontEnd::ExtractKeyPointsAndComputeDescriptors (reference src/frontend.cpp:150-154,
// constructed at :35-37 with nfeatures/scaleFactor/nlevels; everything else OpenCV-3.1
// defaults: edgeThreshold 31, firstLevel 0, WTA_K 2, HARRIS score, patchSize 31, FAST 20).
// OpenCV 3.1.0 is un-vendored and absent here; the algorithm follows its published
// structure.  Deliberate, documented differences:
//   * BRIEF pattern: seeded G-II pattern (include/vo_brief_pattern.h), not bit_pattern_31_
//   * steering uses the exact centroid direction (cos,sin from m10,m01), not fastAtan2's angle
//   * Harris ranking uses the exact integer 25(ab-c^2)-(a+b)^2 (k = 0.04 = 1/25)
#include "o_orb.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "../include/vo_brief_pattern.h"

namespace orc {

static inline short sat_short(long v) { return (short)std::min(32767L, std::max(-32768L, v)); }

void orb_build_plan(const vo_params& p, OrbPlan& pl) {
    pl.W = p.width; pl.H = p.height; pl.nlevels = p.n_levels; pl.nfeatures = p.n_features;
    pl.fast_thr = p.fast_threshold; pl.edge = p.edge_threshold;
    int L = pl.nlevels;
    pl.lw.assign(L, 0); pl.lh.assign(L, 0); pl.quota.assign(L, 0); pl.scale.assign(L, 1.f);
    pl.xofs.assign(L, {}); pl.yofs.assign(L, {}); pl.ialpha.assign(L, {}); pl.ibeta.assign(L, {});
    const double sf = (double)p.scale_factor;
    for (int l = 0; l < L; ++l) {
        pl.scale[l] = (float)std::pow(sf, (double)l);
        pl.lw[l] = (int)lrintf((float)pl.W / pl.scale[l]);
        pl.lh[l] = (int)lrintf((float)pl.H / pl.scale[l]);
    }
    // per-level feature quota: geometric series, last level takes the remainder
    float factor = (float)(1.0 / sf);
    float per = pl.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; ++l) { pl.quota[l] = (int)lrintf(per); sum += pl.quota[l]; per *= factor; }
    pl.quota[L - 1] = std::max(pl.nfeatures - sum, 0);
    // bilinear tables (11-bit fixed point, as cv::resize INTER_LINEAR does for 8-bit images)
    for (int l = 1; l < L; ++l) {
        int sw = pl.lw[l - 1], sh = pl.lh[l - 1], dw = pl.lw[l], dh = pl.lh[l];
        double sx_ = (double)sw / dw, sy_ = (double)sh / dh;
        pl.xofs[l].resize(dw); pl.ialpha[l].resize(2 * dw);
        for (int dx = 0; dx < dw; ++dx) {
            float fx = (float)((dx + 0.5) * sx_ - 0.5);
            int sx = (int)std::floor(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            pl.xofs[l][dx] = sx;
            pl.ialpha[l][2 * dx] = sat_short(lrintf((1.f - fx) * 2048.f));
            pl.ialpha[l][2 * dx + 1] = sat_short(lrintf(fx * 2048.f));
        }
        pl.yofs[l].resize(dh); pl.ibeta[l].resize(2 * dh);
        for (int dy = 0; dy < dh; ++dy) {
            float fy = (float)((dy + 0.5) * sy_ - 0.5);
            int sy = (int)std::floor(fy);
            fy -= sy;
            pl.yofs[l][dy] = sy;
            pl.ibeta[l][2 * dy] = sat_short(lrintf((1.f - fy) * 2048.f));
            pl.ibeta[l][2 * dy + 1] = sat_short(lrintf(fy * 2048.f));
        }
    }
    // half-width of the radius-15 disc per row (intensity-centroid patch)
    const int hp = 15;
    int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
    for (int v = 0; v <= vmax; ++v) pl.umax[v] = (int)lrint(std::sqrt((double)hp * hp - v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (pl.umax[v0] == pl.umax[v0 + 1]) ++v0;
        pl.umax[v] = v0;
        ++v0;
    }
    // 7-tap sigma=2 Gaussian in 8-bit fixed point
    double g[7], gs = 0;
    for (int i = 0; i < 7; ++i) { double x = i - 3; g[i] = (double)(float)std::exp(-0.5 / 4.0 * x * x); gs += g[i]; }
    for (int i = 0; i < 7; ++i) pl.gk[i] = (int)lrintf((float)(g[i] / gs) * 256.f);
}

void bgr_to_gray(const uint8_t* bgr, int stride, int w, int h, uint8_t* gray) {
    for (int y = 0; y < h; ++y) {
        const uint8_t* r = bgr + (size_t)y * stride;
        for (int x = 0; x < w; ++x)
            gray[(size_t)y * w + x] = (uint8_t)((r[3 * x] * 1868 + r[3 * x + 1] * 9617 + r[3 * x + 2] * 4899 + 8192) >> 14);
    }
}

void resize_level(const OrbPlan& pl, int l, const uint8_t* src, uint8_t* dst) {
    int sw = pl.lw[l - 1], sh = pl.lh[l - 1], dw = pl.lw[l], dh = pl.lh[l];
    for (int dy = 0; dy < dh; ++dy) {
        int sy = pl.yofs[l][dy];
        int r0 = std::min(std::max(sy, 0), sh - 1), r1 = std::min(std::max(sy + 1, 0), sh - 1);
        int b0 = pl.ibeta[l][2 * dy], b1 = pl.ibeta[l][2 * dy + 1];
        const uint8_t* S0 = src + (size_t)r0 * sw;
        const uint8_t* S1 = src + (size_t)r1 * sw;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = pl.xofs[l][dx], sx1 = std::min(sx + 1, sw - 1);
            int a0 = pl.ialpha[l][2 * dx], a1 = pl.ialpha[l][2 * dx + 1];
            int h0 = S0[sx] * a0 + S0[sx1] * a1;
            int h1 = S1[sx] * a0 + S1[sx1] * a1;
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            dst[(size_t)dy * dw + dx] = (uint8_t)std::min(255, std::max(0, v));
        }
    }
}

static const int RING[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// Largest threshold t for which (x,y) is still a FAST-9/16 corner: max over the 16 arcs of 9
// contiguous ring pixels of the smallest |ring - centre| difference of one sign, minus 1.
int fast_score(const uint8_t* img, int stride, int x, int y) {
    int p = img[(size_t)y * stride + x], d[25];
    for (int i = 0; i < 16; ++i) d[i] = img[(size_t)(y + RING[i][1]) * stride + x + RING[i][0]] - p;
    for (int i = 0; i < 9; ++i) d[16 + i] = d[i];
    int best = -256;
    for (int s = 0; s < 16; ++s) {
        int mn = 255, mx = -255;
        for (int j = 0; j < 9; ++j) { mn = std::min(mn, d[s + j]); mx = std::max(mx, d[s + j]); }
        best = std::max(best, std::max(mn, -mx));
    }
    return best - 1;
}

void gauss_blur7(const OrbPlan& pl, const uint8_t* src, int w, int h, uint8_t* dst) {
    std::vector<int> row((size_t)w * h);
    auto refl = [](int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int k = -3; k <= 3; ++k) s += pl.gk[k + 3] * src[(size_t)y * w + refl(x + k, w)];
            row[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int k = -3; k <= 3; ++k) s += pl.gk[k + 3] * row[(size_t)refl(y + k, h) * w + x];
            dst[(size_t)y * w + x] = (uint8_t)std::min(255, (s + (1 << 15)) >> 16);
        }
}

// Harris measure over a 7x7 block of 3x3-Sobel-like gradients, as an exact integer:
// 25*(a*b - c^2) - (a+b)^2  ==  25 * (det - 0.04 * trace^2).
long long harris_key(const uint8_t* img, int stride, int x, int y) {
    long long a = 0, b = 0, c = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t* q = img + (size_t)(y + dy) * stride + (x + dx);
            int ix = (q[1] - q[-1]) * 2 + (q[-stride + 1] - q[-stride - 1]) + (q[stride + 1] - q[stride - 1]);
            int iy = (q[stride] - q[-stride]) * 2 + (q[stride - 1] - q[-stride - 1]) + (q[stride + 1] - q[-stride + 1]);
            a += ix * ix; b += iy * iy; c += ix * iy;
        }
    return 25 * (a * b - c * c) - (a + b) * (a + b);
}

// 7th-order odd polynomial atan2 in degrees (0.3 deg accuracy class, like cv::fastAtan2).
float fast_atan2_deg(float y, float x) {
    const float k = 57.29577951308232f;
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k, p5 = 0.1555786518463281f * k,
                p7 = -0.04432655554792128f * k;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + 2.220446e-16f); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + 2.220446e-16f); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

void orb_detect_describe(const OrbPlan& pl, const uint8_t* bgr, int bgr_stride, const uint16_t* depth,
                         int depth_stride, std::vector<vo_keypoint>& kps, std::vector<uint8_t>& desc,
                         std::vector<OrbLevelDebug>* dbg) {
    const int L = pl.nlevels;
    kps.clear(); desc.clear();
    if (dbg) dbg->assign(L, {});
    std::vector<std::vector<uint8_t>> pyr(L);
    pyr[0].resize((size_t)pl.W * pl.H);
    bgr_to_gray(bgr, bgr_stride, pl.W, pl.H, pyr[0].data());
    for (int l = 1; l < L; ++l) {
        pyr[l].resize((size_t)pl.lw[l] * pl.lh[l]);
        resize_level(pl, l, pyr[l - 1].data(), pyr[l].data());
    }
    const double harris_scale = std::pow(1.0 / (4.0 * 7.0 * 255.0), 4.0) / 25.0;

    for (int l = 0; l < L; ++l) {
        const int w = pl.lw[l], h = pl.lh[l];
        const uint8_t* img = pyr[l].data();
        // FAST-9/16 score map (corner iff score >= threshold)
        std::vector<uint8_t> score((size_t)w * h, 0);
        const int t = pl.fast_thr;
        for (int y = 3; y < h - 3; ++y)
            for (int x = 3; x < w - 3; ++x) {
                int p = img[(size_t)y * w + x];
                // a 9-arc always contains one pixel of every opposite pair -> cheap reject
                int d0 = std::abs(img[(size_t)(y + 3) * w + x] - p), d8 = std::abs(img[(size_t)(y - 3) * w + x] - p);
                if (d0 <= t && d8 <= t) continue;
                int d4 = std::abs(img[(size_t)y * w + x + 3] - p), d12 = std::abs(img[(size_t)y * w + x - 3] - p);
                if (d4 <= t && d12 <= t) continue;
                int s = fast_score(img, w, x, y);
                if (s >= t) score[(size_t)y * w + x] = (uint8_t)std::min(s, 255);
            }
        // 3x3 non-max suppression (strictly greater than all 8 neighbours), then border filter
        struct Cand { int x, y, s; long long hk; };
        std::vector<Cand> cand;
        const int e = pl.edge;
        for (int y = e; y < h - e; ++y)
            for (int x = e; x < w - e; ++x) {
                int s = score[(size_t)y * w + x];
                if (!s) continue;
                const uint8_t* q = &score[(size_t)y * w + x];
                if (s > q[-1] && s > q[1] && s > q[-w - 1] && s > q[-w] && s > q[-w + 1] && s > q[w - 1] && s > q[w] && s > q[w + 1])
                    cand.push_back({x, y, s, 0});
            }
        if (dbg) {
            (*dbg)[l].gray = pyr[l]; (*dbg)[l].score = score;
            for (auto& c : cand) (*dbg)[l].cand_xy.push_back(c.y * w + c.x);
        }
        // retain best 2*quota by FAST score, keeping all ties at the cut (KeyPointsFilter::retainBest);
        // if the ties would exceed the 4*quota working capacity, they are ranked by pixel index (row, then column)
        // and exactly as many as 2*quota needs stay.
        const int quota = pl.quota[l];
        int thr = 0, need = -1;
        if ((int)cand.size() > 2 * quota) {
            int hist[256] = {0};
            for (auto& c : cand) hist[c.s]++;
            int acc = 0, s = 255;
            for (; s >= 0; --s) { acc += hist[s]; if (acc >= 2 * quota) break; }
            thr = s;
            if (acc > 4 * quota) need = 2 * quota - (acc - hist[s]);
        }
        std::vector<Cand> kept;
        for (auto& c : cand) if (c.s > thr || (c.s == thr && need < 0)) { c.hk = harris_key(img, w, c.x, c.y); kept.push_back(c); }
        if (need >= 0) {
            std::vector<Cand> ties;
            for (auto& c : cand) if (c.s == thr) ties.push_back(c);
            std::sort(ties.begin(), ties.end(), [](const Cand& a, const Cand& b) { return a.y != b.y ? a.y < b.y : a.x < b.x; });
            for (int i = 0; i < need && i < (int)ties.size(); ++i) { ties[i].hk = harris_key(img, w, ties[i].x, ties[i].y); kept.push_back(ties[i]); }
        }
        // retain best `quota` by Harris (descending; ties by ascending pixel index)
        std::sort(kept.begin(), kept.end(), [w](const Cand& a, const Cand& b) {
            if (a.hk != b.hk) return a.hk > b.hk;
            return a.y * w + a.x < b.y * w + b.x;
        });
        if ((int)kept.size() > quota) kept.resize(quota);

        std::vector<uint8_t> blur((size_t)w * h);
        gauss_blur7(pl, img, w, h, blur.data());
        if (dbg) (*dbg)[l].blurred = blur;

        for (auto& c : kept) {
            // intensity centroid over the radius-15 disc
            long long m10 = 0, m01 = 0;
            for (int v = -15; v <= 15; ++v) {
                int um = pl.umax[std::abs(v)];
                const uint8_t* row = img + (size_t)(c.y + v) * w + c.x;
                for (int u = -um; u <= um; ++u) { m10 += u * row[u]; m01 += v * row[u]; }
            }
            vo_keypoint kp;
            kp.x = (float)c.x * pl.scale[l];
            kp.y = (float)c.y * pl.scale[l];
            kp.size = 31.f * pl.scale[l];
            kp.angle = fast_atan2_deg((float)m01, (float)m10);
            kp.response = (float)((double)c.hk * harris_scale);
            kp.octave = l; kp.class_id = -1;
            // Frame::GetDepth (reference src/frame.cpp:43-67)
            int px = (int)lrintf(kp.x), py = (int)lrintf(kp.y);
            auto D = [&](int xx, int yy) -> int {
                if (xx < 0 || yy < 0 || xx >= pl.W || yy >= pl.H) return 0;
                return *(const uint16_t*)((const uint8_t*)depth + (size_t)yy * depth_stride + 2 * (size_t)xx);
            };
            int dr = D(px, py);
            static const int nx[4] = {-1, 0, 1, 0}, ny[4] = {0, -1, 0, 1};
            for (int i = 0; i < 4 && dr == 0; ++i) dr = D(px + nx[i], py + ny[i]);
            kp.depth_raw = dr;
            kps.push_back(kp);
            // steered BRIEF on the blurred level
            double cs = 1.0, sn = 0.0;
            if (m10 != 0 || m01 != 0) {
                double nrm = std::sqrt((double)m10 * (double)m10 + (double)m01 * (double)m01);
                cs = (double)m10 / nrm; sn = (double)m01 / nrm;
            }
            uint8_t d[32] = {0};
            const uint8_t* ctr = blur.data() + (size_t)c.y * w + c.x;
            for (int i = 0; i < 256; ++i) {
                const int8_t* q = VO_BRIEF_PATTERN[i];
                int x1 = (int)lrint(q[0] * cs - q[1] * sn), y1 = (int)lrint(q[0] * sn + q[1] * cs);
                int x2 = (int)lrint(q[2] * cs - q[3] * sn), y2 = (int)lrint(q[2] * sn + q[3] * cs);
                if (ctr[y1 * w + x1] < ctr[y2 * w + x2]) d[i >> 3] |= (uint8_t)(1u << (i & 7));
            }
            desc.insert(desc.end(), d, d + 32);
        }
    }
}

}  // namespace orc
