// vo_ba.hip -- local bundle adjustment on gfx950, replacing the g2o optimisation inside
// Backend::Optimize (reference src/backend.cpp:19-195): SE3 pose vertices (free + fixed), 3-D point
// vertices marginalised by a Schur complement (g2o BlockSolver_6_3), BinaryEdgeProjection
// residual/Jacobians (include/myslam/g2o_types.h:143-167), Huber delta sqrt(7.815), Levenberg-
// Marquardt with g2o's lambda/rho policy, 10 robust iterations + chi2 cull + 10 plain ones.
//
//   k_ba_linearize  one lane per edge: r, J_pose (2x6), J_point (2x3) = J_pose[:,0:3] R, Huber weight;
//                   H_pp / b_p / H_ll / b_l accumulated with f64 global atomics, W_e = w J_p^T J_l stored
//   k_ba_schur      one lane per point: (H_ll + lambda I)^-1, S -= W_i Hinv W_j^T, b_s -= W_i Hinv b_l
//   k_ba_chol       dense Cholesky of the reduced 6K x 6K system in one workgroup
//   k_ba_backsub    one lane per point: dl = Hinv (b_l - sum W^T dp), trial point, gain-ratio terms
//   k_ba_pose       one lane per free pose: trial pose exp(dp) * T, gain-ratio terms
//   k_ba_chi        robust chi2 of the trial state
// The LM accept/reject decision is a handful of scalars: it is taken on the host after one small
// D2H copy per trial (BA runs once per keyframe, not per frame).  The reduced system is dense and
// tiny at default.yaml scale (6K <= ~100); config 5 of BASELINE.json (window 20, 1.6e5 points) is
// where S would be rebuilt as an MFMA contraction -- not needed for correctness here.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "vo_internal.h"

struct BaCam { double fx, fy, cx, cy; };

struct BaDev {
    int n_poses, n_free, n_points, n_edges, D;
    double* poses; double* pts; double* poses_n; double* pts_n;
    const int32_t* e_pose; const int32_t* e_pt; const float* e_uv; uint8_t* active; uint8_t* flags;
    const int32_t* pt_start; const int32_t* pt_edges;
    double* Hpp; double* bp; double* Hll; double* bl; double* W;
    double* S; double* bs; double* Hinv; double* dl;
    double* scal;       // [0] chi cur  [1] chi trial  [2] scale  [3] ok  [4] maxdiag (as u64 bits)
};

__device__ __forceinline__ void ba_err(const BaCam& cam, const double* T, const double* p, const float* uv, double r[2], double pc[3]) {
    pc[0] = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[9];
    pc[1] = T[3] * p[0] + T[4] * p[1] + T[5] * p[2] + T[10];
    pc[2] = T[6] * p[0] + T[7] * p[1] + T[8] * p[2] + T[11];
    r[0] = (double)uv[0] - (cam.fx * pc[0] / pc[2] + cam.cx);
    r[1] = (double)uv[1] - (cam.fy * pc[1] / pc[2] + cam.cy);
}

__global__ void k_ba_linearize(BaCam cam, BaDev B, int robust, double delta) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B.n_edges || !B.active[e]) return;
    const int j = B.e_pose[e], k = B.e_pt[e];
    const double* T = B.poses + 12 * (size_t)j;
    double r[2], pc[3];
    ba_err(cam, T, B.pts + 3 * (size_t)k, B.e_uv + 2 * (size_t)e, r, pc);
    const double e2 = r[0] * r[0] + r[1] * r[1];
    double w = 1.0, rho0 = e2;
    if (robust && e2 > delta * delta) { const double se = sqrt(e2); rho0 = 2.0 * se * delta - delta * delta; w = delta / se; }
    atomicAdd(&B.scal[0], rho0);
    const double X = pc[0], Y = pc[1], Zi = 1.0 / (pc[2] + 1e-18), Zi2 = Zi * Zi, fx = cam.fx, fy = cam.fy;
    const double Jp[2][6] = {{-fx * Zi, 0, fx * X * Zi2, fx * X * Y * Zi2, -fx - fx * X * X * Zi2, fx * Y * Zi},
                             {0, -fy * Zi, fy * Y * Zi2, fy + fy * Y * Y * Zi2, -fy * X * Y * Zi2, -fy * X * Zi}};
    double Jl[2][3];
    for (int a = 0; a < 2; ++a) for (int c = 0; c < 3; ++c) Jl[a][c] = Jp[a][0] * T[c] + Jp[a][1] * T[3 + c] + Jp[a][2] * T[6 + c];
    for (int a = 0; a < 3; ++a) {
        atomicAdd(&B.bl[3 * (size_t)k + a], -w * (Jl[0][a] * r[0] + Jl[1][a] * r[1]));
        for (int c = 0; c < 3; ++c) atomicAdd(&B.Hll[9 * (size_t)k + 3 * a + c], w * (Jl[0][a] * Jl[0][c] + Jl[1][a] * Jl[1][c]));
    }
    if (j < B.n_free) {
        for (int a = 0; a < 6; ++a) {
            atomicAdd(&B.bp[6 * j + a], -w * (Jp[0][a] * r[0] + Jp[1][a] * r[1]));
            for (int c = 0; c < 6; ++c) atomicAdd(&B.Hpp[36 * (size_t)j + 6 * a + c], w * (Jp[0][a] * Jp[0][c] + Jp[1][a] * Jp[1][c]));
            for (int c = 0; c < 3; ++c) B.W[18 * (size_t)e + 3 * a + c] = w * (Jp[0][a] * Jl[0][c] + Jp[1][a] * Jl[1][c]);
        }
    }
}

__global__ void k_ba_maxdiag(BaDev B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0;
    if (i < B.D) v = fabs(B.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)]);
    else if (i < B.D + 3 * B.n_points) { const int k = (i - B.D) / 3, a = (i - B.D) % 3; v = fabs(B.Hll[9 * (size_t)k + 4 * a]); }
    else return;
    atomicMax((unsigned long long*)&B.scal[4], (unsigned long long)__double_as_longlong(v));   // v >= 0: bit order == value order
}

__global__ void k_ba_init_S(BaDev B, double lambda) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.D * B.D) return;
    const int r = i / B.D, c = i % B.D;
    double v = 0;
    if (r / 6 == c / 6) v = B.Hpp[36 * (size_t)(r / 6) + 6 * (r % 6) + (c % 6)];
    if (r == c) { v += lambda; B.bs[r] = B.bp[r]; }
    B.S[i] = v;
}

__global__ void k_ba_schur(BaDev B, double lambda) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B.n_points) return;
    double a[9];
    for (int i = 0; i < 9; ++i) a[i] = B.Hll[9 * (size_t)k + i];
    a[0] += lambda; a[4] += lambda; a[8] += lambda;
    const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    double h[9];
    if (!(fabs(det) > 0)) { for (int i = 0; i < 9; ++i) { h[i] = 0; B.Hinv[9 * (size_t)k + i] = 0; } return; }
    const double id = 1.0 / det;
    h[0] = (a[4] * a[8] - a[5] * a[7]) * id; h[1] = (a[2] * a[7] - a[1] * a[8]) * id; h[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    h[3] = (a[5] * a[6] - a[3] * a[8]) * id; h[4] = (a[0] * a[8] - a[2] * a[6]) * id; h[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    h[6] = (a[3] * a[7] - a[4] * a[6]) * id; h[7] = (a[1] * a[6] - a[0] * a[7]) * id; h[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    for (int i = 0; i < 9; ++i) B.Hinv[9 * (size_t)k + i] = h[i];
    const double bl0 = B.bl[3 * (size_t)k], bl1 = B.bl[3 * (size_t)k + 1], bl2 = B.bl[3 * (size_t)k + 2];
    for (int p1 = B.pt_start[k]; p1 < B.pt_start[k + 1]; ++p1) {
        const int e1 = B.pt_edges[p1], j1 = B.e_pose[e1];
        if (!B.active[e1] || j1 >= B.n_free) continue;
        const double* W1 = B.W + 18 * (size_t)e1;
        double WH[18];
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) WH[3 * r + c] = W1[3 * r] * h[c] + W1[3 * r + 1] * h[3 + c] + W1[3 * r + 2] * h[6 + c];
        for (int r = 0; r < 6; ++r) atomicAdd(&B.bs[6 * j1 + r], -(WH[3 * r] * bl0 + WH[3 * r + 1] * bl1 + WH[3 * r + 2] * bl2));
        for (int p2 = B.pt_start[k]; p2 < B.pt_start[k + 1]; ++p2) {
            const int e2 = B.pt_edges[p2], j2 = B.e_pose[e2];
            if (!B.active[e2] || j2 >= B.n_free) continue;
            const double* W2 = B.W + 18 * (size_t)e2;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c)
                    atomicAdd(&B.S[(size_t)(6 * j1 + r) * B.D + 6 * j2 + c], -(WH[3 * r] * W2[3 * c] + WH[3 * r + 1] * W2[3 * c + 1] + WH[3 * r + 2] * W2[3 * c + 2]));
        }
    }
}

// in-place Cholesky + solve, one workgroup; result in bs, ok flag in scal[3]
__global__ __launch_bounds__(256) void k_ba_chol(BaDev B) {
    const int D = B.D, tid = threadIdx.x;
    double* A = B.S; double* b = B.bs;
    __shared__ int s_ok;
    if (tid == 0) s_ok = 1;
    __syncthreads();
    for (int j = 0; j < D; ++j) {
        if (tid == 0) {
            double d = A[(size_t)j * D + j];
            for (int k = 0; k < j; ++k) d -= A[(size_t)j * D + k] * A[(size_t)j * D + k];
            if (!(d > 0.0)) s_ok = 0; else A[(size_t)j * D + j] = sqrt(d);
        }
        __syncthreads();
        if (!s_ok) break;
        const double d = A[(size_t)j * D + j];
        for (int i = j + 1 + tid; i < D; i += 256) {
            double s = A[(size_t)i * D + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * D + k] * A[(size_t)j * D + k];
            A[(size_t)i * D + j] = s / d;
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (s_ok) {
            for (int i = 0; i < D; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * D + k] * b[k]; b[i] = s / A[(size_t)i * D + i]; }
            for (int i = D - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < D; ++k) s -= A[(size_t)k * D + i] * b[k]; b[i] = s / A[(size_t)i * D + i]; }
        }
        B.scal[3] = s_ok ? 1.0 : 0.0;
    }
}

__global__ void k_ba_backsub(BaDev B, double lambda) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B.n_points) return;
    if (B.scal[3] == 0.0) return;
    double rhs[3] = {B.bl[3 * (size_t)k], B.bl[3 * (size_t)k + 1], B.bl[3 * (size_t)k + 2]};
    for (int p1 = B.pt_start[k]; p1 < B.pt_start[k + 1]; ++p1) {
        const int e1 = B.pt_edges[p1], j1 = B.e_pose[e1];
        if (!B.active[e1] || j1 >= B.n_free) continue;
        const double* W1 = B.W + 18 * (size_t)e1;
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 6; ++r) rhs[c] -= W1[3 * r + c] * B.bs[6 * j1 + r];
    }
    const double* h = B.Hinv + 9 * (size_t)k;
    double sc = 0;
    for (int a = 0; a < 3; ++a) {
        const double d = h[3 * a] * rhs[0] + h[3 * a + 1] * rhs[1] + h[3 * a + 2] * rhs[2];
        B.dl[3 * (size_t)k + a] = d;
        B.pts_n[3 * (size_t)k + a] = B.pts[3 * (size_t)k + a] + d;
        sc += d * (lambda * d + B.bl[3 * (size_t)k + a]);
    }
    atomicAdd(&B.scal[2], sc);
}

__global__ void k_ba_pose(BaDev B, double lambda) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B.n_poses) return;
    const double* T = B.poses + 12 * (size_t)j;
    double* Tn = B.poses_n + 12 * (size_t)j;
    if (j >= B.n_free || B.scal[3] == 0.0) { for (int i = 0; i < 12; ++i) Tn[i] = T[i]; return; }
    const double* d = B.bs + 6 * j;
    const double w[3] = {d[3], d[4], d[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double A, Bc, C;
    if (th < 1e-8) { A = 1.0 - th2 / 6.0; Bc = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0; }
    else { A = sin(th) / th; Bc = (1.0 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double Wm[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9], R[9], V[9];
    for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += Wm[3 * i + k] * Wm[3 * k + c]; W2[3 * i + c] = s; }
    for (int i = 0; i < 9; ++i) { const double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + A * Wm[i] + Bc * W2[i]; V[i] = I + Bc * Wm[i] + C * W2[i]; }
    const double tx = V[0] * d[0] + V[1] * d[1] + V[2] * d[2], ty = V[3] * d[0] + V[4] * d[1] + V[5] * d[2], tz = V[6] * d[0] + V[7] * d[1] + V[8] * d[2];
    for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += R[3 * i + k] * T[3 * k + c]; Tn[3 * i + c] = s; }
    Tn[9] = R[0] * T[9] + R[1] * T[10] + R[2] * T[11] + tx;
    Tn[10] = R[3] * T[9] + R[4] * T[10] + R[5] * T[11] + ty;
    Tn[11] = R[6] * T[9] + R[7] * T[10] + R[8] * T[11] + tz;
    double sc = 0;
    for (int a = 0; a < 6; ++a) sc += d[a] * (lambda * d[a] + B.bp[6 * j + a]);
    atomicAdd(&B.scal[2], sc);
}

// which: 0 -> current state into scal[0]-free slot scal[5]; 1 -> trial state into scal[1]
__global__ void k_ba_chi(BaCam cam, BaDev B, int trial, int robust, double delta) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0;
    if (e < B.n_edges && B.active[e]) {
        const double* P = (trial ? B.poses_n : B.poses) + 12 * (size_t)B.e_pose[e];
        const double* X = (trial ? B.pts_n : B.pts) + 3 * (size_t)B.e_pt[e];
        double r[2], pc[3];
        ba_err(cam, P, X, B.e_uv + 2 * (size_t)e, r, pc);
        const double e2 = r[0] * r[0] + r[1] * r[1];
        v = (robust && e2 > delta * delta) ? 2.0 * sqrt(e2) * delta - delta * delta : e2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(&B.scal[trial ? 1 : 5], v);
}

// stage 0: cull after the robust round (bit0, deactivate); stage 1: flag level-0 outliers (bit1)
__global__ void k_ba_cull(BaCam cam, BaDev B, int stage, double th) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B.n_edges) return;
    double r[2], pc[3];
    ba_err(cam, B.poses + 12 * (size_t)B.e_pose[e], B.pts + 3 * (size_t)B.e_pt[e], B.e_uv + 2 * (size_t)e, r, pc);
    const double c2 = r[0] * r[0] + r[1] * r[1];
    if (stage == 0) { if (c2 > th) { B.flags[e] = 1; B.active[e] = 0; } else B.flags[e] = 0; }
    else if (B.active[e]) { if (c2 > th) B.flags[e] |= 2; else atomicAdd(&B.scal[6], c2); }
}

int vo_ba_run(vo_ctx* c, const vo_ba_problem* in, vo_ba_result* out) {
    hipStream_t st = c->stream;
    const int np = in->n_poses, nf = in->n_free, nx = in->n_points, ne = in->n_edges, D = 6 * nf;
    out->lm_iters = 0; out->chi2_initial = 0; out->chi2_final = 0;
    if (ne == 0 || nf == 0 || nx == 0) {
        memcpy(out->poses, in->poses, sizeof(double) * 12 * (size_t)nf);
        memcpy(out->points, in->points, sizeof(double) * 3 * (size_t)nx);
        memset(out->edge_flags, 0, ne);
        return VO_OK;
    }
    // CSR point -> edges
    std::vector<int32_t> pt_start(nx + 1, 0), pt_edges(ne);
    for (int e = 0; e < ne; ++e) pt_start[in->edge_point[e] + 1]++;
    for (int k = 0; k < nx; ++k) pt_start[k + 1] += pt_start[k];
    { std::vector<int32_t> fill(pt_start.begin(), pt_start.end() - 1);
      for (int e = 0; e < ne; ++e) pt_edges[fill[in->edge_point[e]]++] = e; }

    // carve the scratch slab
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_poses = carve(96 * (size_t)np), o_pts = carve(24 * (size_t)nx), o_poses_n = carve(96 * (size_t)np), o_pts_n = carve(24 * (size_t)nx);
    const size_t o_epose = carve(4 * (size_t)ne), o_ept = carve(4 * (size_t)ne), o_euv = carve(8 * (size_t)ne), o_act = carve(ne), o_flags = carve(ne);
    const size_t o_ps = carve(4 * (size_t)(nx + 1)), o_pe = carve(4 * (size_t)ne);
    const size_t o_lin = off;      // zeroed before every linearisation: Hpp bp Hll bl scal
    const size_t o_Hpp = carve(288 * (size_t)nf), o_bp = carve(8 * (size_t)D), o_Hll = carve(72 * (size_t)nx), o_bl = carve(24 * (size_t)nx), o_scal = carve(64);
    const size_t lin_bytes = off - o_lin;
    const size_t o_W = carve(144 * (size_t)ne), o_S = carve(8 * (size_t)D * D), o_bs = carve(8 * (size_t)D), o_Hinv = carve(72 * (size_t)nx), o_dl = carve(24 * (size_t)nx);
    int rc = vo_scratch(c, off);
    if (rc) return rc;
    uint8_t* base = (uint8_t*)c->d_ba;
    BaDev B;
    B.n_poses = np; B.n_free = nf; B.n_points = nx; B.n_edges = ne; B.D = D;
    B.poses = (double*)(base + o_poses); B.pts = (double*)(base + o_pts); B.poses_n = (double*)(base + o_poses_n); B.pts_n = (double*)(base + o_pts_n);
    B.e_pose = (const int32_t*)(base + o_epose); B.e_pt = (const int32_t*)(base + o_ept); B.e_uv = (const float*)(base + o_euv);
    B.active = base + o_act; B.flags = base + o_flags; B.pt_start = (const int32_t*)(base + o_ps); B.pt_edges = (const int32_t*)(base + o_pe);
    B.Hpp = (double*)(base + o_Hpp); B.bp = (double*)(base + o_bp); B.Hll = (double*)(base + o_Hll); B.bl = (double*)(base + o_bl); B.scal = (double*)(base + o_scal);
    B.W = (double*)(base + o_W); B.S = (double*)(base + o_S); B.bs = (double*)(base + o_bs); B.Hinv = (double*)(base + o_Hinv); B.dl = (double*)(base + o_dl);
    BaCam cam{(double)c->p.fx, (double)c->p.fy, (double)c->p.cx, (double)c->p.cy};

    HIP_TRY(hipMemcpyAsync(base + o_poses, in->poses, 96 * (size_t)np, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + o_pts, in->points, 24 * (size_t)nx, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + o_epose, in->edge_pose, 4 * (size_t)ne, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + o_ept, in->edge_point, 4 * (size_t)ne, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + o_euv, in->edge_uv, 8 * (size_t)ne, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + o_ps, pt_start.data(), 4 * (size_t)(nx + 1), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(base + o_pe, pt_edges.data(), 4 * (size_t)ne, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(base + o_act, 1, ne, st));
    HIP_TRY(hipMemsetAsync(base + o_flags, 0, ne, st));
    HIP_TRY(hipStreamSynchronize(st));       // pageable sources

    double* h_scal = (double*)vo_stage(c, 64);
    if (!h_scal) return VO_E_NOMEM;
    const dim3 blk(256), gE((ne + 255) / 256), gP((nx + 255) / 256), gJ((np + 255) / 256);
    auto read_scal = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(h_scal, B.scal, 64, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return VO_OK;
    };
    auto optimize = [&](int robust, int max_it, int& iters) -> int {
        double lambda = 0, ni = 2;
        for (int it = 0; it < max_it; ++it) {
            HIP_TRY(hipMemsetAsync(base + o_lin, 0, lin_bytes, st));
            { ProfScope ps(c, "k_ba_linearize"); hipLaunchKernelGGL(k_ba_linearize, gE, blk, 0, st, cam, B, robust, in->huber_delta); }
            if (it == 0) hipLaunchKernelGGL(k_ba_maxdiag, dim3((D + 3 * nx + 255) / 256), blk, 0, st, B);
            if ((rc = read_scal())) return rc;
            double cur = h_scal[0];
            if (it == 0) { double md; memcpy(&md, &h_scal[4], 8); lambda = 1e-5 * md; ni = 2; }
            double rho = 0; int qmax = 0;
            do {
                HIP_TRY(hipMemsetAsync(B.scal + 1, 0, 24, st));        // trial chi, scale, ok
                hipLaunchKernelGGL(k_ba_init_S, dim3((D * D + 255) / 256), blk, 0, st, B, lambda);
                { ProfScope ps(c, "k_ba_schur"); hipLaunchKernelGGL(k_ba_schur, gP, blk, 0, st, B, lambda); }
                { ProfScope ps(c, "k_ba_chol"); hipLaunchKernelGGL(k_ba_chol, dim3(1), blk, 0, st, B); }
                hipLaunchKernelGGL(k_ba_backsub, gP, blk, 0, st, B, lambda);
                hipLaunchKernelGGL(k_ba_pose, gJ, blk, 0, st, B, lambda);
                hipLaunchKernelGGL(k_ba_chi, gE, blk, 0, st, cam, B, 1, robust, in->huber_delta);
                if ((rc = read_scal())) return rc;
                const bool ok = h_scal[3] != 0.0;
                const double tmp = ok ? h_scal[1] : DBL_MAX;
                const double scale = (ok ? h_scal[2] : 0.0) + 1e-3;
                rho = (cur - tmp) / scale;
                if (rho > 0 && std::isfinite(tmp)) {
                    double a = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                    a = std::min(a, 2.0 / 3.0);
                    lambda *= std::max(1.0 / 3.0, a); ni = 2; cur = tmp;
                    std::swap(B.poses, B.poses_n); std::swap(B.pts, B.pts_n);
                } else { lambda *= ni; ni *= 2; }
                ++qmax;
            } while (rho < 0 && qmax < 10);
            ++iters;
            if (qmax == 10 || rho == 0) break;
        }
        return VO_OK;
    };

    // initial plain chi2 (reporting only)
    HIP_TRY(hipMemsetAsync(B.scal, 0, 64, st));
    hipLaunchKernelGGL(k_ba_chi, gE, blk, 0, st, cam, B, 0, 0, in->huber_delta);
    if ((rc = read_scal())) return rc;
    out->chi2_initial = h_scal[5];

    int iters = 0;
    if ((rc = optimize(1, in->it_robust, iters))) return rc;                       // backend.cpp:140-141
    hipLaunchKernelGGL(k_ba_cull, gE, blk, 0, st, cam, B, 0, in->chi2_th);         // backend.cpp:144-156
    if ((rc = optimize(0, in->it_plain, iters))) return rc;                        // backend.cpp:158-159
    HIP_TRY(hipMemsetAsync(B.scal, 0, 64, st));
    hipLaunchKernelGGL(k_ba_cull, gE, blk, 0, st, cam, B, 1, in->chi2_th);         // backend.cpp:162-172
    if ((rc = read_scal())) return rc;
    out->chi2_final = h_scal[6];
    out->lm_iters = iters;
    HIP_TRY(hipMemcpyAsync(out->poses, B.poses, 96 * (size_t)nf, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->points, B.pts, 24 * (size_t)nx, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->edge_flags, B.flags, ne, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return VO_OK;
}
