// vo_tri.hip -- batched linear N-view triangulation on gfx950 (reference include/myslam/util.h:16-34, applied to the LM
// inliers of a keyframe by FrontEnd::TriangulateMappointsInTrackingMap, src/frontend.cpp:465-506).
//
//   k_triangulate   one lane per map point: A^T A (4x4) accumulated over the point's views, cyclic Jacobi
//                   eigen-decomposition in registers, smallest eigenvector / sigma4 : sigma3 test.
//
// Views are few (2 .. number of keyframes that see the point) and the 4x4 problem lives in registers, so the kernel
// is bound by the gather of its inputs (108 B per view); the point of the batch is one launch + one round trip for all
// the points of a keyframe instead of a host loop.  Double arithmetic in the operation order of oracle/o_tri.cpp
// (compiled with -ffp-contract=off).
#include <cmath>
#include <cstring>

#include "vo_internal.h"
#include "vo_tri_dev.h"

__global__ __launch_bounds__(256) void k_triangulate(int n, const int32_t* __restrict__ vs, const double* __restrict__ T, const double* __restrict__ xy,
                                                     double* __restrict__ xyz, uint8_t* __restrict__ ok) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int v0 = vs[i], nv = vs[i + 1] - v0;
    if (nv < 2) { ok[i] = 0; return; }
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = 0.0;
    for (int v = 0; v < nv; ++v) {
        const double* P = T + 12 * (size_t)(v0 + v);
        double p[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) p[k] = P[k];
        tri_accumulate(a, p, xy[2 * (size_t)(v0 + v)], xy[2 * (size_t)(v0 + v) + 1]);
    }
    double x[3];
    const bool good = tri_solve(a, x);
    xyz[3 * (size_t)i] = x[0]; xyz[3 * (size_t)i + 1] = x[1]; xyz[3 * (size_t)i + 2] = x[2];
    ok[i] = good ? 1 : 0;
}

extern "C" int vo_triangulate_batch(vo_ctx* c, int n, const int32_t* vs, const double* T, const double* xy, double* xyz, uint8_t* ok) {
    if (!c || n < 0 || (n && (!vs || !T || !xy || !xyz || !ok))) return VO_E_INVALID;
    if (n == 0) return VO_OK;
    const int nviews = vs[n];
    for (int i = 0; i < n; ++i) if (vs[i + 1] < vs[i]) return VO_E_INVALID;
    if (vs[0] != 0 || nviews < 0) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // one pinned staging buffer (inputs then outputs) and one device slab: one H2D, one launch, one D2H
    const size_t N = (size_t)n, NV = (size_t)nviews;
    const size_t o_vs = 0, o_T = (4 * (N + 1) + 255) & ~(size_t)255, o_xy = o_T + ((96 * NV + 255) & ~(size_t)255), o_xyz = o_xy + ((16 * NV + 255) & ~(size_t)255),
                 o_ok = o_xyz + ((24 * N + 255) & ~(size_t)255), total = o_ok + ((N + 255) & ~(size_t)255);
    uint8_t* h = (uint8_t*)vo_stage(c, total);
    if (!h) return VO_E_NOMEM;
    int rc = vo_scratch(c, total);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));                      // staging / scratch may still serve an earlier call
    memcpy(h + o_vs, vs, 4 * (N + 1)); memcpy(h + o_T, T, 96 * NV); memcpy(h + o_xy, xy, 16 * NV);
    uint8_t* d = (uint8_t*)c->d_ba;
    HIP_TRY(hipMemcpyAsync(d, h, o_xyz, hipMemcpyHostToDevice, st));
    { ProfScope ps(c, "k_triangulate");
      hipLaunchKernelGGL(k_triangulate, dim3((n + 255) / 256), dim3(256), 0, st, n, (const int32_t*)(d + o_vs), (const double*)(d + o_T), (const double*)(d + o_xy),
                         (double*)(d + o_xyz), d + o_ok); }
    HIP_TRY(hipMemcpyAsync(h + o_xyz, d + o_xyz, total - o_xyz, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    memcpy(xyz, h + o_xyz, 24 * N); memcpy(ok, h + o_ok, N);
    return VO_OK;
}
