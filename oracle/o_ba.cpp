// ORACLE -- test infrastructure, NOT the product (see o_math.h header).
// o_ba.cpp: local bundle adjustment as Backend::Optimize runs it through g2o
// (reference src/backend.cpp:19-195): free + fixed SE3 pose vertices, marginalised 3-D point
// vertices, BinaryEdgeProjection (include/myslam/g2o_types.h:135-179: error :143-148,
// Jacobians :150-167, point Jacobian = J[:,0:3]*R :166), Huber delta sqrt(7.815), LM with
// Schur complement on the points (BlockSolver_6_3), 10 robust iterations, chi2 cull, 10 plain.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "o_track.h"

namespace orc {
namespace {

struct BA {
    const Cam& cam; const vo_ba_problem& in;
    std::vector<SE3> pose; std::vector<V3> pt;
    std::vector<uint8_t> active;                    // per edge (level 0)
    std::vector<int> pt_start, pt_edges;            // CSR point -> edges
    const BaShard* sh = nullptr;                    // e-3: this rank's share (its points' edges are the active ones); the three exchanges per LM step make the shares one system
    BA(const Cam& c, const vo_ba_problem& p) : cam(c), in(p) {}
    bool mine(int k) const { return !sh || k % sh->world == sh->rank; }
    void xsum(std::vector<double>& v) const { if (sh) sh->exchange(sh->user, v.data(), (int)v.size()); }

    void err(int e, const SE3& T, V3 p, double r[2], V3& pc) const {
        pc = T * p;
        r[0] = (double)in.edge_uv[2 * e] - (cam.fx * pc.x / pc.z + cam.cx);
        r[1] = (double)in.edge_uv[2 * e + 1] - (cam.fy * pc.y / pc.z + cam.cy);
    }
    double chi(bool robust, const std::vector<SE3>& P, const std::vector<V3>& X) const {
        double s = 0, d = in.huber_delta;
        for (int e = 0; e < in.n_edges; ++e) {
            if (!active[e]) continue;
            double r[2]; V3 pc;
            err(e, P[in.edge_pose[e]], X[in.edge_point[e]], r, pc);
            double e2 = r[0] * r[0] + r[1] * r[1];
            if (robust && e2 > d * d) s += 2.0 * std::sqrt(e2) * d - d * d; else s += e2;
        }
        return s;
    }

    int optimize(bool robust, int max_it) {
        const int nf = in.n_free, np = in.n_points, ne = in.n_edges, D = 6 * nf;
        int nact = 0;
        for (int e = 0; e < ne; ++e) nact += active[e];
        if ((!nact && !sh) || nf == 0) return 0;          // (a sharded rank without active edges of its own still takes part in the exchanges)
        double lambda = 0, ni = 2;
        std::vector<double> Hpp((size_t)D * D), bp(D), Hll((size_t)9 * np), bl((size_t)3 * np), W((size_t)18 * ne);
        int it = 0;
        for (; it < max_it; ++it) {
            std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(bp.begin(), bp.end(), 0.0);
            std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
            double cur = 0;
            for (int e = 0; e < ne; ++e) {
                if (!active[e]) continue;
                const int j = in.edge_pose[e], k = in.edge_point[e];
                double r[2]; V3 pc;
                err(e, pose[j], pt[k], r, pc);
                double e2 = r[0] * r[0] + r[1] * r[1], w = 1.0, d = in.huber_delta;
                if (robust && e2 > d * d) { double se = std::sqrt(e2); cur += 2.0 * se * d - d * d; w = d / se; } else cur += e2;
                const double X = pc.x, Y = pc.y, Zi = 1.0 / (pc.z + 1e-18), Zi2 = Zi * Zi, fx = cam.fx, fy = cam.fy;
                const double Jp[2][6] = {{-fx * Zi, 0, fx * X * Zi2, fx * X * Y * Zi2, -fx - fx * X * X * Zi2, fx * Y * Zi},
                                         {0, -fy * Zi, fy * Y * Zi2, fy + fy * Y * Y * Zi2, -fy * X * Y * Zi2, -fy * X * Zi}};
                double Jl[2][3];
                const M3& R = pose[j].R;
                for (int a = 0; a < 2; ++a)
                    for (int c_ = 0; c_ < 3; ++c_) Jl[a][c_] = Jp[a][0] * R(0, c_) + Jp[a][1] * R(1, c_) + Jp[a][2] * R(2, c_);
                for (int a = 0; a < 3; ++a) {
                    bl[3 * k + a] -= w * (Jl[0][a] * r[0] + Jl[1][a] * r[1]);
                    for (int c_ = 0; c_ < 3; ++c_) Hll[9 * (size_t)k + 3 * a + c_] += w * (Jl[0][a] * Jl[0][c_] + Jl[1][a] * Jl[1][c_]);
                }
                if (j < nf) {
                    for (int a = 0; a < 6; ++a) {
                        bp[6 * j + a] -= w * (Jp[0][a] * r[0] + Jp[1][a] * r[1]);
                        for (int c_ = 0; c_ < 6; ++c_) Hpp[(size_t)(6 * j + a) * D + 6 * j + c_] += w * (Jp[0][a] * Jp[0][c_] + Jp[1][a] * Jp[1][c_]);
                        for (int c_ = 0; c_ < 3; ++c_) W[18 * (size_t)e + 3 * a + c_] = w * (Jp[0][a] * Jl[0][c_] + Jp[1][a] * Jl[1][c_]);
                    }
                }
            }
            double md_ll = 0;
            if (it == 0) for (int k = 0; k < np; ++k) for (int a = 0; a < 3; ++a) md_ll = std::max(md_ll, std::fabs(Hll[9 * (size_t)k + 4 * a]));
            if (sh) {                                       // exchange 1: the pose blocks, b_p, the chi2 of the linearisation, every rank's largest point-block entry
                std::vector<double> x((size_t)36 * nf + D + 1 + sh->world, 0.0);
                for (int j = 0; j < nf; ++j) for (int a = 0; a < 6; ++a) for (int c_ = 0; c_ < 6; ++c_) x[36 * (size_t)j + 6 * a + c_] = Hpp[(size_t)(6 * j + a) * D + 6 * j + c_];
                for (int i = 0; i < D; ++i) x[36 * (size_t)nf + i] = bp[i];
                x[36 * (size_t)nf + D] = cur; x[36 * (size_t)nf + D + 1 + sh->rank] = md_ll;
                xsum(x);
                for (int j = 0; j < nf; ++j) for (int a = 0; a < 6; ++a) for (int c_ = 0; c_ < 6; ++c_) Hpp[(size_t)(6 * j + a) * D + 6 * j + c_] = x[36 * (size_t)j + 6 * a + c_];
                for (int i = 0; i < D; ++i) bp[i] = x[36 * (size_t)nf + i];
                cur = x[36 * (size_t)nf + D];
                for (int r = 0; r < sh->world; ++r) md_ll = std::max(md_ll, x[36 * (size_t)nf + D + 1 + r]);
            }
            if (it == 0) {
                double md = md_ll;
                for (int i = 0; i < D; ++i) md = std::max(md, std::fabs(Hpp[(size_t)i * D + i]));
                lambda = 1e-5 * md; ni = 2;
            }
            double rho = 0; int qmax = 0; bool converged = false;
            do {
                // Schur complement on the points
                std::vector<double> S = Hpp, bs = bp, Hinv((size_t)9 * np), dl((size_t)3 * np);
                for (int i = 0; i < D; ++i) S[(size_t)i * D + i] += lambda;
                if (sh && sh->rank > 0) { std::fill(S.begin(), S.end(), 0.0); std::fill(bs.begin(), bs.end(), 0.0); }      // (H_pp, b_p and lambda enter the summed system once, through rank 0)
                bool ok = true;
                for (int k = 0; k < np; ++k) {
                    double a[9];
                    for (int i = 0; i < 9; ++i) a[i] = Hll[9 * (size_t)k + i];
                    a[0] += lambda; a[4] += lambda; a[8] += lambda;
                    double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
                    double* h = &Hinv[9 * (size_t)k];
                    if (!(std::fabs(det) > 0)) { for (int i = 0; i < 9; ++i) h[i] = 0; continue; }
                    double id = 1.0 / det;
                    h[0] = (a[4] * a[8] - a[5] * a[7]) * id; h[1] = (a[2] * a[7] - a[1] * a[8]) * id; h[2] = (a[1] * a[5] - a[2] * a[4]) * id;
                    h[3] = (a[5] * a[6] - a[3] * a[8]) * id; h[4] = (a[0] * a[8] - a[2] * a[6]) * id; h[5] = (a[2] * a[3] - a[0] * a[5]) * id;
                    h[6] = (a[3] * a[7] - a[4] * a[6]) * id; h[7] = (a[1] * a[6] - a[0] * a[7]) * id; h[8] = (a[0] * a[4] - a[1] * a[3]) * id;
                    // edges of this point that touch free poses
                    for (int p1 = pt_start[k]; p1 < pt_start[k + 1]; ++p1) {
                        int e1 = pt_edges[p1], j1 = in.edge_pose[e1];
                        if (!active[e1] || j1 >= nf) continue;
                        double WH[18];                       // W_e1 * Hinv (6x3)
                        for (int a_ = 0; a_ < 6; ++a_)
                            for (int c_ = 0; c_ < 3; ++c_)
                                WH[3 * a_ + c_] = W[18 * (size_t)e1 + 3 * a_] * h[c_] + W[18 * (size_t)e1 + 3 * a_ + 1] * h[3 + c_] + W[18 * (size_t)e1 + 3 * a_ + 2] * h[6 + c_];
                        for (int a_ = 0; a_ < 6; ++a_)
                            bs[6 * j1 + a_] -= WH[3 * a_] * bl[3 * k] + WH[3 * a_ + 1] * bl[3 * k + 1] + WH[3 * a_ + 2] * bl[3 * k + 2];
                        for (int p2 = pt_start[k]; p2 < pt_start[k + 1]; ++p2) {
                            int e2 = pt_edges[p2], j2 = in.edge_pose[e2];
                            if (!active[e2] || j2 >= nf) continue;
                            for (int a_ = 0; a_ < 6; ++a_)
                                for (int c_ = 0; c_ < 6; ++c_)
                                    S[(size_t)(6 * j1 + a_) * D + 6 * j2 + c_] -= WH[3 * a_] * W[18 * (size_t)e2 + 3 * c_] + WH[3 * a_ + 1] * W[18 * (size_t)e2 + 3 * c_ + 1] + WH[3 * a_ + 2] * W[18 * (size_t)e2 + 3 * c_ + 2];
                        }
                    }
                }
                if (sh) {                                   // exchange 2: the reduced system
                    std::vector<double> x(S); x.insert(x.end(), bs.begin(), bs.end());
                    xsum(x);
                    std::copy(x.begin(), x.begin() + (size_t)D * D, S.begin()); std::copy(x.begin() + (size_t)D * D, x.end(), bs.begin());
                }
                std::vector<double> dp = bs;
                ok = chol_solve(D, S.data(), dp.data());
                std::vector<SE3> Pn = pose; std::vector<V3> Xn = pt;
                double tmp = DBL_MAX, scale = 1e-3, tmp_pts_scale = 0;
                if (ok) {
                    for (int k = 0; k < np; ++k) {
                        double rhs[3] = {bl[3 * k], bl[3 * k + 1], bl[3 * k + 2]};
                        for (int p1 = pt_start[k]; p1 < pt_start[k + 1]; ++p1) {
                            int e1 = pt_edges[p1], j1 = in.edge_pose[e1];
                            if (!active[e1] || j1 >= nf) continue;
                            for (int c_ = 0; c_ < 3; ++c_)
                                for (int a_ = 0; a_ < 6; ++a_) rhs[c_] -= W[18 * (size_t)e1 + 3 * a_ + c_] * dp[6 * j1 + a_];
                        }
                        const double* h = &Hinv[9 * (size_t)k];
                        for (int a_ = 0; a_ < 3; ++a_) dl[3 * k + a_] = h[3 * a_] * rhs[0] + h[3 * a_ + 1] * rhs[1] + h[3 * a_ + 2] * rhs[2];
                    }
                    for (int j = 0; j < nf; ++j) Pn[j] = SE3::exp(&dp[6 * j]) * pose[j];          // g2o_types.h:56-60
                    for (int k = 0; k < np; ++k) Xn[k] = pt[k] + V3(dl[3 * k], dl[3 * k + 1], dl[3 * k + 2]);  // :121-125
                    tmp = chi(robust, Pn, Xn);
                    for (int i = 0; i < D; ++i) scale += dp[i] * (lambda * dp[i] + bp[i]);
                    double sc_pts = 0;
                    for (int i = 0; i < 3 * np; ++i) sc_pts += dl[i] * (lambda * dl[i] + bl[i]);
                    if (!sh) scale += sc_pts;
                    else tmp_pts_scale = sc_pts;
                }
                double mx_pts = 0;
                if (ok) for (int i = 0; i < 3 * np; ++i) mx_pts = std::max(mx_pts, std::fabs(dl[i]));
                if (sh) {                                   // exchange 3: the trial state's chi2, the points' part of the gain ratio's denominator, every rank's largest point step
                    std::vector<double> x(2 + (size_t)sh->world, 0.0);
                    x[0] = ok ? tmp : 0.0; x[1] = ok ? tmp_pts_scale : 0.0; x[2 + sh->rank] = mx_pts;
                    xsum(x);
                    if (ok) { tmp = x[0]; scale += x[1]; }
                    for (int r = 0; r < sh->world; ++r) mx_pts = std::max(mx_pts, x[2 + r]);
                }
                rho = (cur - tmp) / scale;
                if (rho > 0 && std::isfinite(tmp)) {
                    double a = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                    a = std::min(a, 2.0 / 3.0);
                    lambda *= std::max(1.0 / 3.0, a); ni = 2; cur = tmp; pose.swap(Pn); pt.swap(Xn);
                } else { lambda *= ni; ni *= 2; }
                if (ok) {
                    double mx = mx_pts;
                    for (int i = 0; i < D; ++i) mx = std::max(mx, std::fabs(dp[i]));
                    converged = mx < 1e-10;
                }
                ++qmax;
            } while (rho < 0 && qmax < 10 && !converged);
            if (qmax == 10 || rho == 0 || converged) { ++it; break; }
        }
        return it;
    }
};

}  // namespace

int local_ba(const Cam& cam, const vo_ba_problem& in, vo_ba_result& out, const BaShard* shard) {
    if (in.n_free < 0 || in.n_free > in.n_poses || in.n_points < 0 || in.n_edges < 0) return VO_E_INVALID;
    BA ba(cam, in);
    ba.pose.resize(in.n_poses); ba.pt.resize(in.n_points);
    for (int j = 0; j < in.n_poses; ++j) ba.pose[j] = SE3::from12(in.poses + 12 * (size_t)j);
    for (int k = 0; k < in.n_points; ++k) ba.pt[k] = V3(in.points[3 * k], in.points[3 * k + 1], in.points[3 * k + 2]);
    for (int e = 0; e < in.n_edges; ++e)
        if (in.edge_pose[e] < 0 || in.edge_pose[e] >= in.n_poses || in.edge_point[e] < 0 || in.edge_point[e] >= in.n_points) return VO_E_INVALID;
    ba.active.assign(in.n_edges, 1);
    if (shard && shard->world > 1) { ba.sh = shard; for (int e = 0; e < in.n_edges; ++e) ba.active[e] = ba.mine(in.edge_point[e]) ? 1 : 0; }
    ba.pt_start.assign(in.n_points + 1, 0);
    for (int e = 0; e < in.n_edges; ++e) ba.pt_start[in.edge_point[e] + 1]++;
    for (int k = 0; k < in.n_points; ++k) ba.pt_start[k + 1] += ba.pt_start[k];
    ba.pt_edges.resize(in.n_edges);
    { std::vector<int> fill(ba.pt_start.begin(), ba.pt_start.end() - 1);
      for (int e = 0; e < in.n_edges; ++e) ba.pt_edges[fill[in.edge_point[e]]++] = e; }

    out.chi2_initial = ba.chi(false, ba.pose, ba.pt);
    out.lm_iters = ba.optimize(true, in.it_robust);                        // backend.cpp:140-141
    for (int e = 0; e < in.n_edges; ++e) {                                 // backend.cpp:144-156
        double r[2]; V3 pc;
        ba.err(e, ba.pose[in.edge_pose[e]], ba.pt[in.edge_point[e]], r, pc);
        out.edge_flags[e] = 0;
        if (!ba.mine(in.edge_point[e])) continue;
        if (r[0] * r[0] + r[1] * r[1] > in.chi2_th) { out.edge_flags[e] |= 1; ba.active[e] = 0; }
    }
    out.lm_iters += ba.optimize(false, in.it_plain);                       // backend.cpp:158-159
    out.chi2_final = 0;
    for (int e = 0; e < in.n_edges; ++e) {                                 // backend.cpp:162-172
        double r[2]; V3 pc;
        ba.err(e, ba.pose[in.edge_pose[e]], ba.pt[in.edge_point[e]], r, pc);
        double c2 = r[0] * r[0] + r[1] * r[1];
        if (ba.active[e]) { if (c2 > in.chi2_th) out.edge_flags[e] |= 2; else out.chi2_final += c2; }
    }
    for (int j = 0; j < in.n_free; ++j) ba.pose[j].to12(out.poses + 12 * (size_t)j);
    for (int k = 0; k < in.n_points; ++k) { out.points[3 * k] = ba.pt[k].x; out.points[3 * k + 1] = ba.pt[k].y; out.points[3 * k + 2] = ba.pt[k].z; }
    if (ba.sh) {                                            // the last exchange: every rank gets the whole result (its own points / flags, zeros elsewhere, summed)
        const size_t nx = (size_t)in.n_points, ne = (size_t)in.n_edges;
        std::vector<double> x(3 * nx + ne + 2, 0.0);
        for (size_t k = 0; k < nx; ++k) if (ba.mine((int)k)) for (int a = 0; a < 3; ++a) x[3 * k + a] = out.points[3 * k + a];
        for (size_t e = 0; e < ne; ++e) if (ba.mine(in.edge_point[e])) x[3 * nx + e] = out.edge_flags[e];
        x[3 * nx + ne] = out.chi2_initial; x[3 * nx + ne + 1] = out.chi2_final;
        ba.xsum(x);
        for (size_t i = 0; i < 3 * nx; ++i) out.points[i] = x[i];
        for (size_t e = 0; e < ne; ++e) out.edge_flags[e] = (uint8_t)std::lrint(x[3 * nx + e]);
        out.chi2_initial = x[3 * nx + ne]; out.chi2_final = x[3 * nx + ne + 1];
    }
    return VO_OK;
}

}  // namespace orc
