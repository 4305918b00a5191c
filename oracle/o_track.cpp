// ORACLE -- test infrastructure, NOT the product (see o_math.h header).
// o_track.cpp: matching, P3P-RANSAC and pose-only LM restated on the CPU.  Reference call
// sites: src/frontend.cpp:156-215 (match), :217-254 (solvePnPRansac), :256-332 (g2o pose BA).
#include "o_track.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace orc {

// ------------------------------------------------------------------------------------------
// candidate filter + exact Hamming nearest neighbour + distance gate
// ------------------------------------------------------------------------------------------
static inline int hamming256(const uint8_t* a, const uint8_t* b) {
    const uint64_t* x = (const uint64_t*)a;
    const uint64_t* y = (const uint64_t*)b;
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

void match_active(const Cam& cam, const MapStore& map, const SE3& T, const uint8_t* desc, int n_kp,
                  float ratio, float floor_dist, std::vector<vo_match>& out, int& n_cand, int& min_dist) {
    out.clear(); n_cand = 0; min_dist = -1;
    const V3 C = T.inverse().t;                       // camera centre (frame.h:57-59)
    const double cos_max = 0.8660254037844387;        // acos(d) > pi/6  <=>  d < cos(pi/6)
    std::vector<vo_match> all;
    for (int32_t mi : map.active) {
        if (map.flags[mi] & VO_MAP_FLAG_OUTLIER) continue;                       // frontend.cpp:177
        V3 pw(map.pos[3 * mi], map.pos[3 * mi + 1], map.pos[3 * mi + 2]);
        V3 pc = T * pw;
        if (!(pc.z > 0)) continue;                                               // frame.cpp:73
        double u = cam.fx * pc.x / pc.z + cam.cx, v = cam.fy * pc.y / pc.z + cam.cy;
        if (u < 0 || u >= cam.W || v < 0 || v >= cam.H) continue;                // frame.cpp:78-80
        V3 dir = normalized(pw - C);
        double d = dir.x * map.nrm[3 * mi] + dir.y * map.nrm[3 * mi + 1] + dir.z * map.nrm[3 * mi + 2];
        if (d < cos_max) continue;                                               // frame.cpp:84-88
        ++n_cand;
        if (n_kp == 0) continue;
        int best = 1 << 30, bi = -1;
        for (int k = 0; k < n_kp; ++k) {
            int h = hamming256(&map.desc[32 * (size_t)mi], desc + 32 * (size_t)k);
            if (h < best) { best = h; bi = k; }                                  // first minimum wins
        }
        all.push_back({mi, bi, best, 0});
    }
    if (all.empty()) return;
    int mn = 1 << 30;
    for (auto& m : all) mn = std::min(mn, m.distance);
    min_dist = mn;
    float max_dis = std::max((float)mn * ratio, floor_dist);                     // frontend.cpp:196
    for (auto& m : all) if ((float)m.distance <= max_dis) out.push_back(m);      // frontend.cpp:206
}

// ------------------------------------------------------------------------------------------
// quartic (Ferrari, resolvent root by bisection + Newton: only + - * / sqrt, deterministic)
// ------------------------------------------------------------------------------------------
int solve_quartic(double a4, double a3, double a2, double a1, double a0, double roots[4]) {
    if (!(std::fabs(a4) > 1e-300)) return 0;
    const double b3 = a3 / a4, b2 = a2 / a4, b1 = a1 / a4, b0 = a0 / a4;
    const double p = b2 - 3.0 * b3 * b3 / 8.0;
    const double q = b1 - b2 * b3 / 2.0 + b3 * b3 * b3 / 8.0;
    const double r = b0 - b1 * b3 / 4.0 + b2 * b3 * b3 / 16.0 - 3.0 * b3 * b3 * b3 * b3 / 256.0;
    double ys[4]; int n = 0;
    const double qtol = 1e-13 * (1.0 + std::fabs(p) * std::sqrt(std::fabs(p)) + std::fabs(r));
    if (std::fabs(q) <= qtol) {                       // biquadratic
        double disc = p * p - 4.0 * r;
        if (disc < 0) return 0;
        double sd = std::sqrt(disc);
        double z[2] = {(-p + sd) / 2.0, (-p - sd) / 2.0};
        for (int i = 0; i < 2; ++i) if (z[i] >= 0) { double y = std::sqrt(z[i]); ys[n++] = y; ys[n++] = -y; }
    } else {
        // g(m) = 8m^3 + 8p m^2 + (2p^2 - 8r) m - q^2, g(0) < 0, one root m > 0
        const double c2 = 8.0 * p, c1 = 2.0 * p * p - 8.0 * r, c0 = -q * q;
        auto g = [&](double m) { return ((8.0 * m + c2) * m + c1) * m + c0; };
        double lo = 0.0, hi = 1.0;
        int guard = 0;
        while (!(g(hi) > 0.0) && guard < 600) { hi *= 2.0; ++guard; }
        if (guard >= 600) return 0;
        for (int it = 0; it < 64; ++it) {
            double mid = 0.5 * (lo + hi);
            if (g(mid) > 0.0) hi = mid; else lo = mid;
        }
        double m = 0.5 * (lo + hi);
        for (int it = 0; it < 4; ++it) {
            double gd = (24.0 * m + 2.0 * c2) * m + c1;
            if (gd == 0.0) break;
            double mn = m - g(m) / gd;
            if (mn > 0.0) m = mn;
        }
        if (!(m > 0.0)) return 0;
        const double s = std::sqrt(2.0 * m), h = q / (2.0 * s), base = p / 2.0 + m;
        // y^2 - s y + (base + h) = 0  and  y^2 + s y + (base - h) = 0
        double d1 = s * s - 4.0 * (base + h);
        if (d1 >= 0) { double sd = std::sqrt(d1); ys[n++] = (s + sd) / 2.0; ys[n++] = (s - sd) / 2.0; }
        double d2 = s * s - 4.0 * (base - h);
        if (d2 >= 0) { double sd = std::sqrt(d2); ys[n++] = (-s + sd) / 2.0; ys[n++] = (-s - sd) / 2.0; }
    }
    for (int i = 0; i < n; ++i) {
        double x = ys[i] - b3 / 4.0;
        for (int it = 0; it < 2; ++it) {               // polish on the normalised quartic
            double f = (((x + b3) * x + b2) * x + b1) * x + b0;
            double fd = ((4.0 * x + 3.0 * b3) * x + 2.0 * b2) * x + b1;
            if (fd != 0.0) x -= f / fd;
        }
        roots[i] = x;
    }
    return n;
}

// ------------------------------------------------------------------------------------------
// P3P (Grunert 1841, coefficients as in Haralick et al. 1994), pose from the 3 recovered points
// ------------------------------------------------------------------------------------------
static inline void frame_of(V3 A, V3 B, V3 Cc, V3 e[3]) {
    e[0] = normalized(B - A);
    e[2] = normalized(cross(e[0], Cc - A));
    e[1] = cross(e[2], e[0]);
}

int p3p_grunert(const V3 P[3], const V3 f[3], M3 R[4], V3 t[4]) {
    const double a2 = dot(P[1] - P[2], P[1] - P[2]), b2 = dot(P[0] - P[2], P[0] - P[2]), c2 = dot(P[0] - P[1], P[0] - P[1]);
    if (!(a2 > 1e-18 && b2 > 1e-18 && c2 > 1e-18)) return 0;
    const double ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
    const double q = (a2 - c2) / b2, ac = (a2 + c2) / b2;
    const double A4 = (q - 1.0) * (q - 1.0) - 4.0 * c2 / b2 * ca * ca;
    const double A3 = 4.0 * (q * (1.0 - q) * cb - (1.0 - ac) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
    const double A2 = 2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * ((b2 - c2) / b2) * ca * ca -
                             4.0 * ac * ca * cb * cg + 2.0 * ((b2 - a2) / b2) * cg * cg);
    const double A1 = 4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - ac) * ca * cg);
    const double A0 = (1.0 + q) * (1.0 + q) - 4.0 * a2 / b2 * cg * cg;
    double vs[4];
    int nr = solve_quartic(A4, A3, A2, A1, A0, vs), ns = 0;
    V3 ep[3];
    frame_of(P[0], P[1], P[2], ep);
    for (int i = 0; i < nr; ++i) {
        const double v = vs[i];
        if (!(v > 0.0)) continue;
        const double den = 2.0 * (cg - v * ca);
        if (std::fabs(den) < 1e-12) continue;
        const double u = ((q - 1.0) * v * v - 2.0 * q * cb * v + 1.0 + q) / den;
        if (!(u > 0.0)) continue;
        const double s1sq = b2 / (1.0 + v * v - 2.0 * v * cb);
        if (!(s1sq > 0.0)) continue;
        const double s1 = std::sqrt(s1sq), s2 = u * s1, s3 = v * s1;
        V3 Q0 = s1 * f[0], Q1 = s2 * f[1], Q2 = s3 * f[2], eq[3];
        frame_of(Q0, Q1, Q2, eq);
        M3 Rm;
        for (int r_ = 0; r_ < 3; ++r_)
            for (int c_ = 0; c_ < 3; ++c_)
                Rm(r_, c_) = eq[0][r_] * ep[0][c_] + eq[1][r_] * ep[1][c_] + eq[2][r_] * ep[2][c_];
        V3 tt = Q0 - Rm * P[0];
        if (!(std::isfinite(tt.x) && std::isfinite(tt.y) && std::isfinite(tt.z))) continue;
        R[ns] = Rm; t[ns] = tt; ++ns;
    }
    return ns;
}

// ------------------------------------------------------------------------------------------
// counter-based sampler
// ------------------------------------------------------------------------------------------
static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint64_t rng_draw(uint64_t seed, uint64_t hyp, uint64_t j) { return mix64(seed ^ mix64(hyp * 0x100000001B3ull + j)); }

void sample4(uint64_t seed, int hyp, int n, int idx[4]) {
    uint64_t j = 0;
    for (int k = 0; k < 4; ++k) {
        for (;;) {
            int c = (int)(rng_draw(seed, (uint64_t)hyp, j++) % (uint64_t)n);
            bool dup = false;
            for (int i = 0; i < k; ++i) dup |= (idx[i] == c);
            if (!dup) { idx[k] = c; break; }
            if (j > 64) {   // pathological: fall back to the first unused index
                for (c = 0; c < n; ++c) { dup = false; for (int i = 0; i < k; ++i) dup |= (idx[i] == c); if (!dup) break; }
                idx[k] = c; break;
            }
        }
    }
}

// Smallest k with (1 - w^4)^k <= 1 - conf, capped at max_iters (w = inlier ratio).  Same role as
// OpenCV's RANSACUpdateNumIters, written as a product loop so CPU and GPU agree bit for bit.
int ransac_update_iters(double conf, int n_pts, int n_inl, int max_iters) {
    double w = (double)n_inl / (double)n_pts;
    double qf = 1.0 - w * w * w * w, target = 1.0 - conf;
    if (!(qf > 0.0)) return 0;
    double acc = 1.0; int k = 0;
    while (acc > target && k < max_iters) { acc *= qf; ++k; }
    return k;
}

// inlier test without divisions: |fx X + (cx-u) Z, fy Y + (cy-v) Z|^2 <= thr^2 Z^2  (same predicate as
// |proj - z|^2 <= thr^2 for Z > 0; written so that CPU and GPU evaluate identical operations)
static inline bool reproj_ok(const Cam& cam, const M3& R, V3 t, const float* X, const float* z, double thr2) {
    V3 pc = R * V3(X[0], X[1], X[2]) + t;
    if (!(pc.z > 0)) return false;
    double du = cam.fx * pc.x + (cam.cx - (double)z[0]) * pc.z, dv = cam.fy * pc.y + (cam.cy - (double)z[1]) * pc.z;
    return du * du + dv * dv <= thr2 * (pc.z * pc.z);
}

void pnp_ransac(const Cam& cam, const Corr& c, int n_hyp, float reproj_px, float conf, uint64_t seed,
                const SE3& prior, RansacOut& out, const HypShard* shard) {
    const bool sharded = shard && shard->world > 1 && shard->exchange;
    out.T = prior; out.inliers.clear(); out.iters_used = 0; out.best = -1;
    out.hyp_counts.assign(n_hyp, 0); out.hyp_pose.assign((size_t)12 * n_hyp, 0.0);
    const int n = c.n;
    if (n < 4) return;                                     // solvePnPRansac fails -> prior pose kept
    const double thr2 = (double)reproj_px * (double)reproj_px;
    for (int h = 0; h < n_hyp; ++h) {
        int id[4];
        sample4(seed, h, n, id);
        V3 P[3], f[3];
        for (int k = 0; k < 3; ++k) {
            P[k] = V3(c.xyz[3 * id[k]], c.xyz[3 * id[k] + 1], c.xyz[3 * id[k] + 2]);
            f[k] = normalized(V3(((double)c.uv[2 * id[k]] - cam.cx) / cam.fx, ((double)c.uv[2 * id[k] + 1] - cam.cy) / cam.fy, 1.0));
        }
        M3 R[4]; V3 t[4];
        int ns = p3p_grunert(P, f, R, t), bi = -1;
        double be = DBL_MAX;
        for (int s = 0; s < ns; ++s) {                     // 4th point picks the branch
            V3 pc = R[s] * V3(c.xyz[3 * id[3]], c.xyz[3 * id[3] + 1], c.xyz[3 * id[3] + 2]) + t[s];
            if (!(pc.z > 0)) continue;
            double du = cam.fx * pc.x / pc.z + cam.cx - (double)c.uv[2 * id[3]], dv = cam.fy * pc.y / pc.z + cam.cy - (double)c.uv[2 * id[3] + 1];
            double e = du * du + dv * dv;
            if (e < be) { be = e; bi = s; }
        }
        if (bi < 0) { out.hyp_counts[h] = (!sharded || h % shard->world == shard->rank) ? -1 : 0; continue; }
        SE3(R[bi], t[bi]).to12(&out.hyp_pose[(size_t)12 * h]);
        if (sharded && h % shard->world != shard->rank) continue;      // scored by its owner; 0 goes into the sum
        int cnt = 0;
        for (int k = 0; k < n; ++k) cnt += reproj_ok(cam, R[bi], t[bi], &c.xyz[3 * k], &c.uv[2 * k], thr2);
        out.hyp_counts[h] = cnt;
    }
    if (sharded) shard->exchange(shard->user, out.hyp_counts.data(), n_hyp);      // element-wise sum over the ranks
    // sequential adaptive-stop scan (cv RANSAC loop semantics: frontend.cpp:240 iters=100, conf .99)
    int best_cnt = 3, niters = n_hyp, h = 0;
    for (; h < niters; ++h) {
        int cnt = out.hyp_counts[h];
        if (cnt > best_cnt) {
            best_cnt = cnt; out.best = h;
            niters = std::min(niters, ransac_update_iters((double)conf, n, cnt, niters));
        }
    }
    out.iters_used = h;
    if (out.best < 0) return;
    out.T = SE3::from12(&out.hyp_pose[(size_t)12 * out.best]);
    for (int k = 0; k < n; ++k)
        if (reproj_ok(cam, out.T.R, out.T.t, &c.xyz[3 * k], &c.uv[2 * k], thr2)) out.inliers.push_back(k);
}

// ------------------------------------------------------------------------------------------
// pose-only LM (g2o OptimizationAlgorithmLevenberg + RobustKernelHuber semantics)
// ------------------------------------------------------------------------------------------
static inline void huber(double e2, double delta, bool robust, double& rho0, double& rho1) {
    if (!robust || e2 <= delta * delta) { rho0 = e2; rho1 = 1.0; return; }
    double se = std::sqrt(e2);
    rho0 = 2.0 * se * delta - delta * delta; rho1 = delta / se;
}

static inline void edge_err(const Cam& cam, const SE3& T, const float* X, const float* z, double e[2], V3& pc) {
    pc = T * V3(X[0], X[1], X[2]);
    e[0] = (double)z[0] - (cam.fx * pc.x / pc.z + cam.cx);            // g2o_types.h:83
    e[1] = (double)z[1] - (cam.fy * pc.y / pc.z + cam.cy);
}

static double lm_chi(const Cam& cam, const Corr& c, const std::vector<int32_t>& act, const SE3& T, bool robust, double delta) {
    double s = 0;
    for (int32_t k : act) {
        double e[2], r0, r1; V3 pc;
        edge_err(cam, T, &c.xyz[3 * k], &c.uv[2 * k], e, pc);
        huber(e[0] * e[0] + e[1] * e[1], delta, robust, r0, r1);
        s += r0;
    }
    return s;
}

static int lm_optimize(const Cam& cam, const Corr& c, const std::vector<int32_t>& act, SE3& T, bool robust,
                       double delta, int max_it) {
    if (act.empty()) return 0;
    double lambda = 0, ni = 2;
    int it = 0;
    for (; it < max_it; ++it) {
        double H[36] = {0}, b[6] = {0}, cur = 0;
        for (int32_t k : act) {
            double e[2], r0, r1; V3 pc;
            edge_err(cam, T, &c.xyz[3 * k], &c.uv[2 * k], e, pc);
            huber(e[0] * e[0] + e[1] * e[1], delta, robust, r0, r1);
            cur += r0;
            const double X = pc.x, Y = pc.y, Zi = 1.0 / (pc.z + 1e-18), Zi2 = Zi * Zi, fx = cam.fx, fy = cam.fy;
            const double J[2][6] = {{-fx * Zi, 0, fx * X * Zi2, fx * X * Y * Zi2, -fx - fx * X * X * Zi2, fx * Y * Zi},
                                    {0, -fy * Zi, fy * Y * Zi2, fy + fy * Y * Y * Zi2, -fy * X * Y * Zi2, -fy * X * Zi}};  // g2o_types.h:97-99
            for (int i = 0; i < 6; ++i) {
                b[i] -= r1 * (J[0][i] * e[0] + J[1][i] * e[1]);
                for (int j = 0; j < 6; ++j) H[i * 6 + j] += r1 * (J[0][i] * J[0][j] + J[1][i] * J[1][j]);
            }
        }
        if (it == 0) {
            double md = 0;
            for (int i = 0; i < 6; ++i) md = std::max(md, std::fabs(H[i * 7]));
            lambda = 1e-5 * md; ni = 2;
        }
        double rho = 0; int qmax = 0; bool converged = false;
        do {
            double A[36], x[6];
            for (int i = 0; i < 36; ++i) A[i] = H[i];
            for (int i = 0; i < 6; ++i) { A[i * 7] += lambda; x[i] = b[i]; }
            bool ok = chol_solve(6, A, x);
            SE3 Tn = T;
            double tmp = DBL_MAX;
            if (ok) { Tn = SE3::exp(x) * T; tmp = lm_chi(cam, c, act, Tn, robust, delta); }      // g2o_types.h:59
            rho = cur - tmp;
            double scale = 1e-3;
            if (ok) for (int i = 0; i < 6; ++i) scale += x[i] * (lambda * x[i] + b[i]);
            rho /= scale;
            if (rho > 0 && std::isfinite(tmp)) {
                double a = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                a = std::min(a, 2.0 / 3.0);
                lambda *= std::max(1.0 / 3.0, a); ni = 2; cur = tmp; T = Tn;
            } else { lambda *= ni; ni *= 2; }
            if (ok) {                                       // a step below 1e-10 (taken or not) ends the optimisation:
                double mx = 0;                              // later iterations / retries cannot move the pose any further
                for (int i = 0; i < 6; ++i) mx = std::max(mx, std::fabs(x[i]));
                converged = mx < 1e-10;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10 && !converged);
        if (qmax == 10 || rho == 0 || converged) { ++it; break; }
    }
    return it;
}

void pose_lm(const Cam& cam, const Corr& c, const std::vector<int32_t>& edges, const SE3& T0, double huber_delta,
             double chi2_cut, int it_robust, int it_plain, LmOut& out) {
    out.T = T0; out.iters = 0;
    out.inlier_mask.assign(edges.size(), 0);
    std::vector<int32_t> act = edges;
    out.iters += lm_optimize(cam, c, act, out.T, true, huber_delta, it_robust);       // frontend.cpp:290-291
    std::vector<int32_t> act2;
    for (int32_t k : edges) {                                                         // frontend.cpp:294-306
        double e[2]; V3 pc;
        edge_err(cam, out.T, &c.xyz[3 * k], &c.uv[2 * k], e, pc);
        if (!(e[0] * e[0] + e[1] * e[1] > chi2_cut)) act2.push_back(k);
    }
    out.iters += lm_optimize(cam, c, act2, out.T, false, huber_delta, it_plain);      // frontend.cpp:309-310
    out.chi2 = 0;
    for (size_t i = 0; i < edges.size(); ++i) {                                       // frontend.cpp:317-329
        double e[2]; V3 pc;
        edge_err(cam, out.T, &c.xyz[3 * edges[i]], &c.uv[2 * edges[i]], e, pc);
        double c2 = e[0] * e[0] + e[1] * e[1];
        out.inlier_mask[i] = !(c2 > chi2_cut);
        if (out.inlier_mask[i]) out.chi2 += c2;
    }
}

}  // namespace orc
