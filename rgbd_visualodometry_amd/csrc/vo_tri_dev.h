// vo_tri_dev.h -- the 4x4 eigen-problem of the linear N-view triangulation (reference include/myslam/util.h:16-34) as a device function:
// cyclic Jacobi in registers on A^T A, smallest eigenvector, sigma4 : sigma3 test.  Shared by k_triangulate (vo_tri.hip) and the
// first-success triangulation of a keyframe commit (vo_kf.hip).  Operation order of oracle/o_tri.cpp.
#pragma once
#include <hip/hip_runtime.h>

// accumulate one view's two rows into a (4x4, row-major): P = T_cw as R row-major (9) + t (3); (mx, my) on the normalised image plane
__device__ __forceinline__ void tri_accumulate(double a[16], const double p[12], double mx, double my) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double m = r == 0 ? mx : my;
        double row[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) row[c] = m * p[6 + c] - p[3 * r + c];
        row[3] = m * p[11] - p[9 + r];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) a[4 * x + y] += row[x] * row[y];
    }
}

// returns sigma4 / sigma3 < 1e-2; xyz = smallest eigenvector of a, dehomogenised (a is destroyed)
__device__ __forceinline__ bool tri_solve(double a[16], double xyz[3]) {
    double V[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) V[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) off += a[4 * p + q] * a[4 * p + q];
        if (off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                if (fabs(a[4 * p + q]) < 1e-300) continue;
                const double th = (a[4 * q + q] - a[4 * p + p]) / (2 * a[4 * p + q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double x = a[4 * k + p], y = a[4 * k + q]; a[4 * k + p] = c * x - s * y; a[4 * k + q] = s * x + c * y; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double x = a[4 * p + k], y = a[4 * q + k]; a[4 * p + k] = c * x - s * y; a[4 * q + k] = s * x + c * y; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double x = V[4 * k + p], y = V[4 * k + q]; V[4 * k + p] = c * x - s * y; V[4 * k + q] = s * x + c * y; }
            }
    }
    const double d[4] = {a[0], a[5], a[10], a[15]};
    int i0 = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (d[k] < d[i0]) i0 = k;
    int i1 = i0 == 0 ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k != i0 && d[k] < d[i1]) i1 = k;
    double e[4] = {0, 0, 0, 0};                           // column i0 of V without dynamic register indexing
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = i0 == 0 ? V[4 * k] : (i0 == 1 ? V[4 * k + 1] : (i0 == 2 ? V[4 * k + 2] : V[4 * k + 3]));
    xyz[0] = e[0] / e[3]; xyz[1] = e[1] / e[3]; xyz[2] = e[2] / e[3];
    const double s4 = sqrt(fmax(d[i0], 0.0)), s3 = sqrt(fmax(d[i1], 0.0));
    return s4 / s3 < 1e-2;
}
