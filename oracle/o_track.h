// ORACLE -- test infrastructure, NOT the product (see o_math.h header).
// o_track.h: CPU restatement of the per-frame tracking numerics: frustum/view-angle candidate
// filter + exact Hamming 1-NN + distance gate (reference src/frontend.cpp:156-215,
// src/frame.cpp:70-91), P3P-RANSAC (src/frontend.cpp:238-254, standing in for
// cv::solvePnPRansac), pose-only Levenberg-Marquardt with Huber kernel as g2o runs it
// (src/frontend.cpp:257-329, include/myslam/g2o_types.h:47-108) and local BA
// (src/backend.cpp:19-195, g2o_types.h:111-179).
#pragma once
#include <cstdint>
#include <vector>

#include "../include/vo_hip.h"
#include "o_math.h"

namespace orc {

struct Cam { double fx, fy, cx, cy; int W, H; };

struct MapStore {
    std::vector<double> pos, nrm;       // 3 per point
    std::vector<uint8_t> desc, flags;   // 32 per point, 1 per point
    std::vector<int32_t> active;
};

struct Corr { std::vector<float> xyz, uv; int n = 0; };   // float32 as the reference casts (frontend.cpp:228)

void match_active(const Cam& cam, const MapStore& map, const SE3& T, const uint8_t* desc, int n_kp,
                  float ratio, float floor_dist, std::vector<vo_match>& out, int& n_cand, int& min_dist);

// quartic a4 x^4 + .. + a0 = 0, real roots only; deterministic (sqrt + basic ops only)
int solve_quartic(double a4, double a3, double a2, double a1, double a0, double roots[4]);
// Grunert P3P: world points P[3], unit bearings f[3] -> up to 4 (R,t) with X_c = R X_w + t
int p3p_grunert(const V3 P[3], const V3 f[3], M3 R[4], V3 t[4]);
uint64_t rng_draw(uint64_t seed, uint64_t hyp, uint64_t j);
void sample4(uint64_t seed, int hyp, int n, int idx[4]);
int ransac_update_iters(double conf, int n_pts, int n_inl, int max_iters);

struct RansacOut {
    SE3 T; std::vector<int32_t> inliers; std::vector<int32_t> hyp_counts;
    std::vector<double> hyp_pose;   // 12 per hypothesis (valid flag in hyp_counts >= 0)
    int iters_used = 0, best = -1;
};
struct HypShard { int rank = 0, world = 1; void (*exchange)(void*, int32_t*, int) = nullptr; void* user = nullptr; };
void pnp_ransac(const Cam& cam, const Corr& c, int n_hyp, float reproj_px, float conf, uint64_t seed,
                const SE3& prior, RansacOut& out, const HypShard* shard = nullptr);

struct LmOut { SE3 T; std::vector<uint8_t> inlier_mask; int iters = 0; double chi2 = 0; };
void pose_lm(const Cam& cam, const Corr& c, const std::vector<int32_t>& edges, const SE3& T0, double huber_delta,
             double chi2_cut, int it_robust, int it_plain, LmOut& out);

// e-3 (include/vo_hip.h: vo_set_ba_shard): this rank linearises the edges of the points k % world == rank; `exchange` sums n doubles over the ranks in place
struct BaShard { int rank = 0, world = 1; void (*exchange)(void*, double*, int) = nullptr; void* user = nullptr; };
int local_ba(const Cam& cam, const vo_ba_problem& in, vo_ba_result& out, const BaShard* shard = nullptr);

// linear N-view triangulation of one point (reference include/myslam/util.h:16-34): smallest eigenvector of A^T A by cyclic Jacobi
bool triangulate_point(int n_views, const double* T_cw, const double* xy, double xyz[3]);

}  // namespace orc
