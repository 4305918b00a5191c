// ORACLE -- test infrastructure, NOT the product (see o_math.h header).
// o_tri.cpp: linear N-view triangulation, CPU restatement of reference include/myslam/util.h:16-34 (rows x m2 - m0,
// y m2 - m1 per view; the reference takes the last right singular vector of A by Eigen::bdcSvd and succeeds iff
// sigma4 / sigma3 < 1e-2).  Here: eigen-decomposition of the 4x4 A^T A by cyclic Jacobi rotations (singular values =
// square roots of the eigenvalues), every operation in the order the HIP kernel k_triangulate uses.
#include <cmath>
#include <cstring>

#include "o_track.h"

namespace orc {

bool triangulate_point(int nv, const double* T, const double* xy, double xyz[3]) {
    double a[16];
    for (int i = 0; i < 16; ++i) a[i] = 0.0;
    for (int v = 0; v < nv; ++v) {
        const double* P = T + 12 * (size_t)v;              // R row-major (9), t (3): row r of [R|t] = (P[3r], P[3r+1], P[3r+2], P[9+r])
        for (int r = 0; r < 2; ++r) {
            double row[4];
            for (int c = 0; c < 3; ++c) row[c] = xy[2 * (size_t)v + r] * P[6 + c] - P[3 * r + c];
            row[3] = xy[2 * (size_t)v + r] * P[11] - P[9 + r];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) a[4 * i + j] += row[i] * row[j];
        }
    }
    double V[16];
    for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += a[4 * p + q] * a[4 * p + q];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (std::fabs(a[4 * p + q]) < 1e-300) continue;
                const double th = (a[4 * q + q] - a[4 * p + p]) / (2 * a[4 * p + q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; ++k) { const double x = a[4 * k + p], y = a[4 * k + q]; a[4 * k + p] = c * x - s * y; a[4 * k + q] = s * x + c * y; }
                for (int k = 0; k < 4; ++k) { const double x = a[4 * p + k], y = a[4 * q + k]; a[4 * p + k] = c * x - s * y; a[4 * q + k] = s * x + c * y; }
                for (int k = 0; k < 4; ++k) { const double x = V[4 * k + p], y = V[4 * k + q]; V[4 * k + p] = c * x - s * y; V[4 * k + q] = s * x + c * y; }
            }
    }
    int i0 = 0;                                             // smallest and second smallest eigenvalue (first index wins ties)
    for (int i = 1; i < 4; ++i) if (a[5 * i] < a[5 * i0]) i0 = i;
    int i1 = i0 == 0 ? 1 : 0;
    for (int i = 0; i < 4; ++i) if (i != i0 && a[5 * i] < a[5 * i1]) i1 = i;
    const double w = V[12 + i0];
    xyz[0] = V[i0] / w; xyz[1] = V[4 + i0] / w; xyz[2] = V[8 + i0] / w;
    const double s4 = std::sqrt(std::fmax(a[5 * i0], 0.0)), s3 = std::sqrt(std::fmax(a[5 * i1], 0.0));
    return s4 / s3 < 1e-2;
}

}  // namespace orc
