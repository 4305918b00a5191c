// vo_ba.hip -- local bundle adjustment on gfx950, replacing the g2o optimisation inside
// Backend::Optimize (reference src/backend.cpp:19-195): SE3 pose vertices (free + fixed), 3-D point
// vertices marginalised by a Schur complement (g2o BlockSolver_6_3), BinaryEdgeProjection
// residual/Jacobians (include/myslam/g2o_types.h:143-167), Huber delta sqrt(7.815), Levenberg-
// Marquardt with g2o's lambda/rho policy, 10 robust iterations + chi2 cull + 10 plain ones.
//
// Systems the LDS-resident Cholesky solves (6K <= 192, i.e. up to 32 free poses) take the three-launch LM step of vo_ba_phase2.h:
//   k_ba_schur2       pair lists -> packed lower triangle of S from per-point records (rank-2 form) + H_pp / b_p sums
//   k_ba_chol16       dense Cholesky + solve of [S b; b^T 0] in one workgroup (packed triangle by LDS-DMA, 16-column DPP panels,
//                     f64 MFMA trailing update); clears S behind its load
//   k_ba_upchi2       back-substitution, trial state, robust chi2, LM decision, and the linearisation at the trial state
//   k_ba_lin2                   first step of a round only (linearisation + the largest diagonal entry for lambda_0)
// Larger systems keep the first-generation step:
//   k_ba_lin          4 lanes per point: r, J_pose (2x6), J_point (2x3) = J_pose[:,0:3] R, Huber weight, H_ll / b_l / W_e
//                     (no atomics); 4 workgroups per free pose: H_pp / b_p
//   k_ba_init_S       S = blockdiag(H_pp) + lambda I, b_s = b_p, (H_ll + lambda I)^-1
//   k_ba_schur_blocks one workgroup per <= BA_SLICE-pair slice of a 6x6 block: S -= W_e1 Hinv W_e2^T, b_s -= W_e Hinv b_l
//   k_ba_chol16g      the same Cholesky with the matrix in global memory (L2), panel in LDS
//   k_ba_update       trial points (back-substitution) and trial poses exp(dp) * T, gain-ratio terms
//   k_ba_chi_control  robust chi2 of the trial state; the last workgroup runs the LM accept / lambda policy
// The LM state (lambda, current chi2, iteration counters, round, which of the two state / linearisation buffers is current) lives in
// a device-resident control block: k_ba_admit starts a problem, k_ba_round moves it from the robust round through the cull to the
// plain round and on to "done" and reports to pinned host memory -- the BA engine (further down) enqueues chunks of steps over all
// problems in flight without waiting for any of that.  The second half
// of this file is the device-resident graph cut and merge.
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

#include "vo_internal.h"
#include "vo_reduce.h"

struct BaCam { double fx, fy, cx, cy; };

struct BaBlock { int j1, j2, start, count; };      // one 6x6 block of the reduced system and its pair list

// Device-resident Levenberg-Marquardt state: the accept/reject decision, the lambda policy and the
// iteration bookkeeping run in a one-lane kernel (k_ba_control) so that several LM steps can be enqueued
// back to back; the host only polls `finished` once per chunk of steps.
struct BaCtl {
    double lambda, ni, cur;
    int it, qmax, max_it, need_lin, first, finished, buf, iters_done, steps, arrived;
    int robust, lbuf;                // Huber kernel on (round 1) / off (round 2): backend.cpp:138-160; lbuf: which record / weight buffer holds the current linearisation (vo_ba_phase2.h)
    // the rounds of one local BA (k_ba_admit / k_ba_round): the device moves from the robust round to the plain one and on to "done" by itself
    int stage;                       // 0: robust round, 1: plain round, 2: done (final cull made, BaStat written)
    int max_it_next;                 // iterations of the plain round
    int lin_ticket;                  // k_ba_lin2: workgroups that have added their sums (the last one takes the largest diagonal entry; back to 0 behind it)
    int iters_total, ticket, gen, chol_seq;      // chol_seq: steps + 1, written by the Cholesky behind its results (k_ba_cholup: the update workgroups of the same launch wait for it)
    double chi0;                     // plain chi2 of the initial state (reporting)
};

// What the host needs to know about a slot, written by k_ba_round straight into pinned host memory (no read-back copy per chunk of steps)
struct BaStat { int gen, stage, finished, it, buf, iters_total, steps, n_culled; double chi0, chi_final; int n_pairs, pad_[3]; };

struct BaDev {
    int n_poses, n_free, n_points, n_edges, D, n_blocks;
    int gp;                                                  // point workgroups of the 4-lanes-per-point kernels (64 points each)
    int edges_by_point;                                      // 1: the edges are grouped by point, pt_edges is the identity
    BaCam cam; double delta, chi2_th;
    int it_robust, it_plain, gen, s_tiles;                   // iterations of the two rounds (backend.cpp:141,159); gen: the engine's admission counter (stale status records are told apart by it); s_tiles: S is kept as 16x16 tiles (vo_ba_chol2.h: ch2_fits(D)) instead of packed rows
    // resident graphs only (nullptr otherwise): observation id per edge; the ids of the culled edges are collected by k_ba_round's final
    // stage (device list for the merge kernel, first cull_host_cap of them also in pinned host memory: no list kernel, no read-back copy)
    const long long* e_obs; long long* cull; int* ncull; int cull_cap; long long* cull_host; int cull_host_cap;
    BaCtl* ctl;
    double* posesA; double* ptsA; double* posesB; double* ptsB;      // double-buffered state, ctl->buf selects the current one
    const int32_t* e_pose; const int32_t* e_pt; const float* e_uv; uint8_t* active; uint8_t* flags;
    const int32_t* pt_start; const int32_t* pt_edges;       // CSR point -> edges
    const int32_t* ps_start; const int32_t* ps_edges;       // CSR free pose -> edges
    const int32_t* ps_pt;                                   // optional: the points of the per-pose lists' edges (same positions as ps_edges): one load level less in the pose workgroups of the Schur launch; nullptr: e_pt[e]
    const BaBlock* blocks; const int2* pairs;               // (e1, e2) pairs sharing a point, grouped by (pose(e1) <= pose(e2))
    const int32_t* pair_pt;                                 // optional: the pairs' points (same positions as `pairs`), so that the Schur slices read the point record one load level earlier; nullptr: e_pt[e1]
    const int* n_slices;                                     // device-built pair lists: number of valid entries of `blocks` (nullptr: n_blocks)
    double* Hpp; double* bp; double* Hll; double* bl; double* W;
    double* S; double* bs; double* Hinv; double* dl;
    double* partU; double* partC; int nU;   // per-workgroup partial sums (no same-address atomics): update {gain term, max step} x nU, trial chi2 x grid of k_ba_chi_control
    double* scal;       // [0] chi cur  [1] chi trial  [2] scale  [3] ok  [4] maxdiag (as u64 bits) [5] chi report [6] chi final
    // e-3 (ba_shard_solve): the problem is one rank's share of a BA sharded by point.  gen1: the launch-per-phase step with S in global memory whatever D is;
    // xbuf: the exchange regions behind S / b_s (which live in it: [S D*D][b_s D][x1: H_pp 36 nf, b_p D, chi, maxima per rank][x3: chi trial, scale, max step per rank])
    int gen1, shard_rank, shard_world, upc_ovf; double* xbuf;
    int upc_ppw;                                             // points per update workgroup of a lone problem's fused launch (vo_ba_phase2.h; 0: 128)      // upc_ovf: the update workgroups' LDS holds the overflow list + cells beside the poses (vo_ba_phase2.h; 0: too many poses)
};
#define BA_FOLD(B) ((B).D <= BA_FOLD_D && !(B).gen1)
__host__ __device__ inline size_t ba_x1_off(int D) { return (size_t)D * D + D; }
__host__ __device__ inline size_t ba_x1_len(int nf, int world) { return 36 * (size_t)nf + 6 * (size_t)nf + 1 + world; }
__host__ __device__ inline size_t ba_x3_off(int D, int nf, int world) { return ba_x1_off(D) + ba_x1_len(nf, world); }
__host__ __device__ inline size_t ba_x3_len(int world) { return 2 + (size_t)world; }

// One launch serves every active problem of a batch (blockIdx.z): the local BAs of several streams step through the same
// kernel sequence, each with its own state, control block and sizes; grids are sized for the largest problem and every
// kernel trims to its own problem's extent.
#define BA_SLOTS 16
// ctls: the engine's control blocks, indexed by slot like Bs -- a kernel reads its control block straight from the argument
// (B.ctl holds the same address, but behind a dependent load of the descriptor: one memory latency more at every kernel start)
struct BaBatch { const BaDev* Bs; BaCtl* ctls; BaStat* stat; int n; int slot[BA_SLOTS]; };
#define BA_PROBLEM(Q) const BaDev& B = Q.Bs[Q.slot[blockIdx.z]]; BaCtl* const ctl_ = Q.ctls + Q.slot[blockIdx.z];
// The same with the descriptor copied at kernel entry: there every field is a scalar load (nothing has been stored yet, so the compiler may treat
// the descriptor as unclobbered); a field read through the reference AFTER the kernel's first store comes back through a vector load into vector
// registers -- per use, and for good where it is a base pointer of a loop.
#define BA_PROBLEM_COPY(Q) const BaDev B_copy_ = Q.Bs[Q.slot[blockIdx.z]]; const BaDev& B = B_copy_; BaCtl* const ctl_ = Q.ctls + Q.slot[blockIdx.z];

#define BA_STATE(B) \
    const int buf_ = __builtin_amdgcn_readfirstlane(ctl_->buf); \
    double* const poses_c = buf_ ? B.posesB : B.posesA; double* const pts_c = buf_ ? B.ptsB : B.ptsA; \
    double* const poses_t = buf_ ? B.posesA : B.posesB; double* const pts_t = buf_ ? B.ptsA : B.ptsB; \
    (void)poses_c; (void)pts_c; (void)poses_t; (void)pts_t;

__device__ __forceinline__ void ba_err(const BaCam& cam, const double* T, const double* p, const float* uv, double r[2], double pc[3]) {
    pc[0] = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[9];
    pc[1] = T[3] * p[0] + T[4] * p[1] + T[5] * p[2] + T[10];
    pc[2] = T[6] * p[0] + T[7] * p[1] + T[8] * p[2] + T[11];
    r[0] = (double)uv[0] - (cam.fx * pc[0] / pc[2] + cam.cx);
    r[1] = (double)uv[1] - (cam.fy * pc[1] / pc[2] + cam.cy);
}

// residual, Huber weight and both Jacobians of one edge (g2o_types.h:143-167)
__device__ __forceinline__ void ba_edge(const BaCam& cam, const double* T, const double* p, const float* uv, int robust, double delta,
                                        double r[2], double& w, double& rho0, double Jp[2][6], double Jl[2][3]) {
#pragma clang fp contract(fast)
    double pc[3];
    ba_err(cam, T, p, uv, r, pc);
    const double e2 = r[0] * r[0] + r[1] * r[1];
    w = 1.0; rho0 = e2;
    if (robust && e2 > delta * delta) { const double se = sqrt(e2); rho0 = 2.0 * se * delta - delta * delta; w = delta / se; }
    const double X = pc[0], Y = pc[1], Zi = 1.0 / (pc[2] + 1e-18), Zi2 = Zi * Zi, fx = cam.fx, fy = cam.fy;
    Jp[0][0] = -fx * Zi; Jp[0][1] = 0; Jp[0][2] = fx * X * Zi2; Jp[0][3] = fx * X * Y * Zi2; Jp[0][4] = -fx - fx * X * X * Zi2; Jp[0][5] = fx * Y * Zi;
    Jp[1][0] = 0; Jp[1][1] = -fy * Zi; Jp[1][2] = fy * Y * Zi2; Jp[1][3] = fy + fy * Y * Y * Zi2; Jp[1][4] = -fy * X * Y * Zi2; Jp[1][5] = -fy * X * Zi;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Jl[a][c] = Jp[a][0] * T[c] + Jp[a][1] * T[3 + c] + Jp[a][2] * T[6 + c];
}

template <int NV>
__device__ __forceinline__ void ba_block_reduce(double* v, double* s_part) {      // 256 threads; result valid in thread 0
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = vo_wave_sum_f64(v[i]);
    __syncthreads();
    if (lane == 0) for (int i = 0; i < NV; ++i) s_part[wave * NV + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < NV; ++i) v[i] = s_part[i] + s_part[NV + i] + s_part[2 * NV + i] + s_part[3 * NV + i];
}

template <int NV>
__device__ __forceinline__ void ba_wave_reduce(double* v) {      // 64 threads; result valid in every lane
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = vo_wave_sum_f64(v[i]);
}

// Linearisation, one launch: blocks [0, gp) own 64 points each, 4 lanes per point (H_ll, b_l, W_e, chi2: no atomics, no zeroing),
// blocks [gp, gp + PSPLIT n_free) reduce slices of a free pose's edges into H_pp / b_p (27 f64 atomics per block;
// H_pp / b_p are zeroed by the step that accepted the state, see k_ba_chi_control).
#ifndef PSPLIT
#define PSPLIT 4
#endif
#define PSPLIT_LONE 16                                       // a lone problem's step / linearisation launches (round 6): a pose's list is one round of sixteen workgroups; launches over several problems keep PSPLIT (their grids are wide already: 8 / 16 streams lost 6-7 % with sixteen)
#define BA_SLICE 256                                         // pairs per Schur workgroup: one per lane (k_ba_schur2 keeps nothing across pairs)
// Four lanes share a point (a quad): each takes every fourth edge, the quad sums H_ll / b_l with two DPP quad
// permutes.  A point seen by all ~30 keyframes of the window no longer makes one lane walk 30 edges in a row.
__device__ __forceinline__ double ba_quad_sum(double x) {
    x += vo_dpp_mov_f64<0xB1, 0xF>(x);      // quad_perm [1,0,3,2]
    x += vo_dpp_mov_f64<0x4E, 0xF>(x);      // quad_perm [2,3,0,1]
    return x;
}
__device__ __forceinline__ void ba_inv3_damped(const double* __restrict__ Hll, double lambda, double* __restrict__ h) {      // (H_ll + lambda I)^-1, zero if singular
    double a[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) a[q] = Hll[q];
    a[0] += lambda; a[4] += lambda; a[8] += lambda;
    const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
    if (!(fabs(det) > 0)) {
#pragma unroll
        for (int q = 0; q < 9; ++q) h[q] = 0;
    } else {
        const double id = 1.0 / det;
        h[0] = (a[4] * a[8] - a[5] * a[7]) * id; h[1] = (a[2] * a[7] - a[1] * a[8]) * id; h[2] = (a[1] * a[5] - a[2] * a[4]) * id;
        h[3] = (a[5] * a[6] - a[3] * a[8]) * id; h[4] = (a[0] * a[8] - a[2] * a[6]) * id; h[5] = (a[2] * a[3] - a[0] * a[5]) * id;
        h[6] = (a[3] * a[7] - a[4] * a[6]) * id; h[7] = (a[1] * a[6] - a[0] * a[7]) * id; h[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    }
}
// Systems that the LDS-resident Cholesky kernel solves (D <= BA_FOLD_D) have no separate "init S" launch: the linearisation
// launch zeroes S / b_s and inverts the damped point blocks, the Schur kernel accumulates into the zeroed S, and the Cholesky
// kernel adds blockdiag(H_pp) + lambda I and b_p while it loads the system.  Larger systems keep k_ba_init_S.
#define BA_FOLD_D 192
// Systems of the LDS-resident Cholesky (D <= BA_FOLD_D) keep S as a PACKED lower triangle, entry (r, c <= r) at r (r + 1) / 2 + c -- the
// layout the factorisation uses in LDS, so that k_ba_chol16 brings it in with straight global -> LDS copies.  Larger systems (k_ba_chol16g)
// keep the full row-major matrix.
__device__ __forceinline__ size_t ba_tri(int r, int c) { return (size_t)r * (size_t)(r + 1) / 2 + (size_t)c; }
// ... or, for the second-generation Cholesky (vo_ba_chol2.h, ch2_fits(D)), as 16x16 tiles of row stride 17 in tile order
#define BA_TILE_RS 17
#define BA_TILE_TS 272
__host__ __device__ inline size_t ba_tile_idx(int r, int c) { const int i = r >> 4, j = c >> 4; return (size_t)(i * (i + 1) / 2 + j) * BA_TILE_TS + (size_t)((r & 15) * BA_TILE_RS + (c & 15)); }
__host__ __device__ inline size_t ba_tile_doubles(int D) { const int T = (D + 15) >> 4; return (size_t)T * (T + 1) / 2 * BA_TILE_TS; }
__device__ __forceinline__ size_t ba_sidx(const BaDev& B, int r, int c) { return B.s_tiles ? ba_tile_idx(r, c) : ba_tri(r, c); }
__device__ __forceinline__ void ba_fold_zero(const BaDev& B, int blk, int nblk) {            // S = 0, b_s = 0 (grid-stride over the point blocks)
    const int n = B.s_tiles ? (int)ba_tile_doubles(B.D) : B.D * (B.D + 1) / 2;
    for (int i = blk * 256 + threadIdx.x; i < n; i += nblk * 256) B.S[i] = 0.0;
    for (int i = blk * 256 + threadIdx.x; i < B.D; i += nblk * 256) B.bs[i] = 0.0;
}
__device__ __forceinline__ void ba_lin_points_body(const BaCam& cam, const BaDev& B, const BaCtl* ctl_, int robust, double delta, int blk,
                                                   const double* poses_c, const double* pts_c, double* s_part) {
    const int k = blk * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    double chi[1] = {0.0};
    double H[6] = {0, 0, 0, 0, 0, 0}, b3[3] = {0, 0, 0};
    if (k < B.n_points) {
        const double* p = pts_c + 3 * (size_t)k;
        const int q1 = B.pt_start[k + 1];
        for (int q = B.pt_start[k] + sub; q < q1; q += 4) {
            const int e = B.pt_edges[q];
            if (!B.active[e]) continue;
            const int j = B.e_pose[e];
            double r[2], w, rho0, Jp[2][6], Jl[2][3];
            ba_edge(cam, poses_c + 12 * (size_t)j, p, B.e_uv + 2 * (size_t)e, robust, delta, r, w, rho0, Jp, Jl);
            chi[0] += rho0;
            b3[0] -= w * (Jl[0][0] * r[0] + Jl[1][0] * r[1]); b3[1] -= w * (Jl[0][1] * r[0] + Jl[1][1] * r[1]); b3[2] -= w * (Jl[0][2] * r[0] + Jl[1][2] * r[1]);
            H[0] += w * (Jl[0][0] * Jl[0][0] + Jl[1][0] * Jl[1][0]); H[1] += w * (Jl[0][0] * Jl[0][1] + Jl[1][0] * Jl[1][1]); H[2] += w * (Jl[0][0] * Jl[0][2] + Jl[1][0] * Jl[1][2]);
            H[3] += w * (Jl[0][1] * Jl[0][1] + Jl[1][1] * Jl[1][1]); H[4] += w * (Jl[0][1] * Jl[0][2] + Jl[1][1] * Jl[1][2]); H[5] += w * (Jl[0][2] * Jl[0][2] + Jl[1][2] * Jl[1][2]);
            if (j < B.n_free) {
                double* We = B.W + 18 * (size_t)e;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) We[3 * a + c] = w * (Jp[0][a] * Jl[0][c] + Jp[1][a] * Jl[1][c]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) H[i] = ba_quad_sum(H[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b3[i] = ba_quad_sum(b3[i]);
    if (k < B.n_points && sub == 0) {
        double* Ho = B.Hll + 9 * (size_t)k;
        Ho[0] = H[0]; Ho[1] = H[1]; Ho[2] = H[2]; Ho[3] = H[1]; Ho[4] = H[3]; Ho[5] = H[4]; Ho[6] = H[2]; Ho[7] = H[4]; Ho[8] = H[5];
        B.bl[3 * (size_t)k] = b3[0]; B.bl[3 * (size_t)k + 1] = b3[1]; B.bl[3 * (size_t)k + 2] = b3[2];
        // folded init: lambda of this step is known unless this is the first step of a round (then k_ba_init_S follows k_ba_maxdiag)
        if (BA_FOLD(B) && !ctl_->first) {
            const double Hs[9] = {H[0], H[1], H[2], H[1], H[3], H[4], H[2], H[4], H[5]};
            ba_inv3_damped(Hs, ctl_->lambda, B.Hinv + 9 * (size_t)k);
        }
    }
    if (BA_FOLD(B)) ba_fold_zero(B, blk, B.gp);
    ba_block_reduce<1>(chi, s_part);
    if (threadIdx.x == 0 && chi[0] != 0.0) atomicAdd(&B.scal[0], chi[0]);
}

__device__ __forceinline__ void ba_lin_poses_body(const BaCam& cam, const BaDev& B, int robust, double delta, int blk,
                                                  const double* poses_c, const double* pts_c, double* s_part) {
    const int j = blk / PSPLIT, part = blk % PSPLIT;
    const double* T = poses_c + 12 * (size_t)j;
    double v[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) v[i] = 0;
    for (int q = B.ps_start[j] + part * 256 + threadIdx.x; q < B.ps_start[j + 1]; q += 256 * PSPLIT) {
        const int e = B.ps_edges[q];
        if (!B.active[e]) continue;
        double r[2], w, rho0, Jp[2][6], Jl[2][3];
        ba_edge(cam, T, pts_c + 3 * (size_t)B.e_pt[e], B.e_uv + 2 * (size_t)e, robust, delta, r, w, rho0, Jp, Jl);
        int c = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            v[21 + a] -= w * (Jp[0][a] * r[0] + Jp[1][a] * r[1]);
#pragma unroll
            for (int b = a; b < 6; ++b) v[c++] += w * (Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b]);
        }
    }
    // 27 sums over the 256 threads: transposed 32-value wave reduction, then the 4 wave partials through LDS
    {
        double v32[32], r8[8];
#pragma unroll
        for (int i = 0; i < 32; ++i) v32[i] = i < 27 ? v[i] : 0.0;
        vo_wave_reduce32(v32, r8);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if ((lane & 15) == 0) {
            const int slot = VO_R32_SLOT(lane >> 4);
#pragma unroll
            for (int k8 = 0; k8 < 7; ++k8) s_part[wave * 32 + 4 * k8 + slot] = r8[k8];
        }
        __syncthreads();
    }
    if (threadIdx.x < 27) {
        const int i = threadIdx.x;
        const double t = s_part[i] + s_part[32 + i] + s_part[64 + i] + s_part[96 + i];
        if (i < 21) {
            int a = 0, rem = i;
            while (rem >= 6 - a) { rem -= 6 - a; ++a; }        // upper-triangle index -> (a, b)
            const int b = a + rem;
            atomicAdd(&B.Hpp[36 * (size_t)j + 6 * a + b], t);
            if (a != b) atomicAdd(&B.Hpp[36 * (size_t)j + 6 * b + a], t);
        } else atomicAdd(&B.bp[6 * j + (i - 21)], t);
    }
}

__global__ __launch_bounds__(256) void k_ba_lin(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished) return;
    if (!ctl_->need_lin) {                                 // a rejected step is solved again at the same linearisation with a larger lambda
        if (BA_FOLD(B) && (int)blockIdx.x < B.gp) {
            const int k = blockIdx.x * 64 + (threadIdx.x >> 2);
            if (k < B.n_points && (threadIdx.x & 3) == 0) ba_inv3_damped(B.Hll + 9 * (size_t)k, ctl_->lambda, B.Hinv + 9 * (size_t)k);
            ba_fold_zero(B, blockIdx.x, B.gp);
        }
        return;
    }
    BA_STATE(B)
    __shared__ double s_part[4 * 32];
    const int gp = B.gp, robust = ctl_->robust;
    if ((int)blockIdx.x < gp) ba_lin_points_body(B.cam, B, ctl_, robust, B.delta, blockIdx.x, poses_c, pts_c, s_part);
    else if ((int)blockIdx.x - gp < B.n_free * PSPLIT) ba_lin_poses_body(B.cam, B, robust, B.delta, blockIdx.x - gp, poses_c, pts_c, s_part);
}

__global__ void k_ba_maxdiag(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished || !ctl_->need_lin || !ctl_->first) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0;
    if (i < B.D) v = fabs(B.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)]);
    else if (i < B.D + 3 * B.n_points) { const int k = (i - B.D) / 3, a = (i - B.D) % 3; v = fabs(B.Hll[9 * (size_t)k + 4 * a]); }
    else return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)&B.scal[4], (unsigned long long)__double_as_longlong(v));   // v >= 0: bit order == value order
}

// D > BA_FOLD_D: S = blockdiag(H_pp) + lambda I, b_s = b_p, and (H_ll + lambda I)^-1 per point, every step.
// D <= BA_FOLD_D: only the first step of a round, and only the inverses (lambda comes from k_ba_maxdiag just before).
__global__ void k_ba_init_S(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished) return;
    const bool fold = BA_FOLD(B), first = ctl_->need_lin && ctl_->first;
    if (fold && !first) return;
    // first step of a round: lambda = 1e-5 * max diag(H) (g2o computeLambdaInit); the control block is updated later in
    // this step by the Cholesky kernel, so every lane derives the same value here
    const double lambda = first ? 1e-5 * B.scal[4] : ctl_->lambda;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (!fold && i < B.D * B.D) {
        const int r = i / B.D, c = i % B.D;
        double v = 0;
        const bool root = B.shard_rank == 0;                // (a sharded BA: H_pp and b_p are the exchanged sums; they and lambda enter the summed S once, through rank 0)
        if (root && r / 6 == c / 6) v = B.Hpp[36 * (size_t)(r / 6) + 6 * (r % 6) + (c % 6)];
        if (r == c) { if (root) v += lambda; B.bs[r] = root ? B.bp[r] : 0.0; }
        B.S[i] = v;
    }
    if (i < B.n_points) ba_inv3_damped(B.Hll + 9 * (size_t)i, lambda, B.Hinv + 9 * (size_t)i);
}

// Schur complement, one workgroup per slice (<= BA_SLICE pairs) of a 6x6 block (j1 <= j2) of the reduced system:
//   S[j1][j2] -= sum over points seen by both poses of W_e1 (H_ll+lambda)^-1 W_e2^T
// Diagonal blocks also produce b_s[j] = b_p[j] - sum W_e (H_ll+lambda)^-1 b_l.  Half of all pairs are diagonal (every free edge pairs
// with itself): there e1 == e2, so W is loaded once; off-diagonal slices carry 36 sums instead of 42 and never touch b_l.
template <bool DIAG>
__device__ __forceinline__ void ba_schur_slice(const BaDev& B, const BaBlock blk, double* s_part, double* s_tot) {
    constexpr int NV = DIAG ? 42 : 36;
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0;
    for (int q = blk.start + threadIdx.x; q < blk.start + blk.count; q += 256) {
        const int2 pr = B.pairs[q];
        if (!B.active[pr.x] || (!DIAG && !B.active[pr.y])) continue;
        const int k = B.e_pt[pr.x];
        const double* h = B.Hinv + 9 * (size_t)k;
        const double* W2 = B.W + 18 * (size_t)pr.y;
        double w2[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) w2[i] = W2[i];
        const double* W1 = B.W + 18 * (size_t)pr.x;
        double b0 = 0, b1 = 0, b2 = 0;
        if (DIAG) { b0 = B.bl[3 * (size_t)k]; b1 = B.bl[3 * (size_t)k + 1]; b2 = B.bl[3 * (size_t)k + 2]; }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const double a0 = DIAG ? w2[3 * r] : W1[3 * r], a1 = DIAG ? w2[3 * r + 1] : W1[3 * r + 1], a2 = DIAG ? w2[3 * r + 2] : W1[3 * r + 2];
            const double y0 = a0 * h[0] + a1 * h[3] + a2 * h[6];
            const double y1 = a0 * h[1] + a1 * h[4] + a2 * h[7];
            const double y2 = a0 * h[2] + a1 * h[5] + a2 * h[8];
#pragma unroll
            for (int c = 0; c < 6; ++c) v[6 * r + c] += y0 * w2[3 * c] + y1 * w2[3 * c + 1] + y2 * w2[3 * c + 2];
            if (DIAG) v[36 + r] += y0 * b0 + y1 * b1 + y2 * b2;
        }
    }
    {   // NV sums: a transposed 32-value wave reduction for v[0..31], a second (padded) one for the rest; wave partials through LDS
        double r0[8], r1[8];
        {
            double lo[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) lo[i] = v[i];
            vo_wave_reduce32(lo, r0);
        }
        {
            double hi[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) hi[i] = i < NV - 32 ? v[32 + i] : 0.0;
            vo_wave_reduce32(hi, r1);
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if ((lane & 15) == 0) {
            const int slot = VO_R32_SLOT(lane >> 4);
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) s_part[wave * 42 + 4 * k8 + slot] = r0[k8];
#pragma unroll
            for (int k8 = 0; k8 < 3; ++k8) if (4 * k8 + slot < NV - 32) s_part[wave * 42 + 32 + 4 * k8 + slot] = r1[k8];
        }
    }
    __syncthreads();
    if (threadIdx.x < NV) s_tot[threadIdx.x] = s_part[threadIdx.x] + s_part[42 + threadIdx.x] + s_part[84 + threadIdx.x] + s_part[126 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 36) {
        const int r = threadIdx.x / 6, c = threadIdx.x % 6;
        const double val = s_tot[threadIdx.x];
        atomicAdd(&B.S[(size_t)(6 * blk.j1 + r) * B.D + 6 * blk.j2 + c], -val);        // a block may be split over several workgroups
        if (!DIAG) atomicAdd(&B.S[(size_t)(6 * blk.j2 + c) * B.D + 6 * blk.j1 + r], -val);
    } else if (DIAG && threadIdx.x < 42) {
        atomicAdd(&B.bs[6 * blk.j1 + threadIdx.x - 36], -s_tot[threadIdx.x]);
    }
}
__global__ __launch_bounds__(256) void k_ba_schur_blocks(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished) return;
    __shared__ double s_part[4 * 42];
    __shared__ double s_tot[42];
    if ((int)blockIdx.x >= B.n_blocks || (B.n_slices && (int)blockIdx.x >= *B.n_slices)) return;
    const BaBlock blk = B.blocks[blockIdx.x];
    if (blk.j1 == blk.j2) ba_schur_slice<true>(B, blk, s_part, s_tot);
    else ba_schur_slice<false>(B, blk, s_part, s_tot);
}

// ---- pair lists of the reduced system, built on the device ---------------------------------------------------------
// Block (j1 <= j2) needs the pairs (e1, e2) of edges of poses j1 / j2 that see the same point.  With the edges sorted
// by point, the per-pose edge lists (ps_edges) are sorted by point too, so a block's list is the intersection of two
// sorted lists: the workgroup keeps the point ids of pose j2 in LDS and binary-searches the points of pose j1.  Three
// launches (count, scan + slice table, ordered fill) replace a host enumeration of ~4 pairs per edge and the upload
// of the lists; the order inside a block (ascending point) is the host builder's, so the sums are the same.
#define PAIR_LDS_CAP 8192
struct BaPairPlan { const int32_t* ps_start; const int32_t* ps_edges; const int32_t* ps_pt; int nf; int* cnt; int* off; int* n_slices; int* n_pairs; BaBlock* blocks; int2* pairs; int32_t* pair_pt; int lds_cap; };      // lds_cap: entries of the launch's dynamic LDS (a longer list is searched in global memory)

__device__ __forceinline__ void ba_block_of(int b, int nf, int& j1, int& j2) {     // b-th (j1 <= j2) pair in row-major order
    j1 = 0;
    while (b >= nf - j1) { b -= nf - j1; ++j1; }
    j2 = j1 + b;
}
__device__ __forceinline__ int ba_find_sorted(const int* s_pts, int n, int key) {  // index of key in the sorted list, -1 if absent
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pts[mid] < key) lo = mid + 1; else hi = mid; }
    return (lo < n && s_pts[lo] == key) ? lo : -1;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_ba_pairs(BaPairPlan Q) {
    extern __shared__ int s_pts[];
    __shared__ int s_w[16];
    int j1, j2;
    ba_block_of(blockIdx.x, Q.nf, j1, j2);
    const int a0 = Q.ps_start[j1], n1 = Q.ps_start[j1 + 1] - a0, b0 = Q.ps_start[j2], n2 = Q.ps_start[j2 + 1] - b0;
    if (!FILL && j1 == j2) { if (threadIdx.x == 0) Q.cnt[blockIdx.x] = n1; return; }
    if (FILL && Q.cnt[blockIdx.x] == 0) return;
    const int base = FILL ? Q.off[blockIdx.x] : 0;
    if (FILL && j1 == j2) {
        for (int q = threadIdx.x; q < n1; q += 256) { const int e = Q.ps_edges[a0 + q]; Q.pairs[base + q] = make_int2(e, e); }
        return;
    }
    // pose j2's sorted point list: in LDS when it fits this launch's allocation, searched in global memory otherwise (config 5: 11 k edges per pose)
    const int* pts = s_pts;
    if (n2 <= Q.lds_cap) { for (int i = threadIdx.x; i < n2; i += 256) s_pts[i] = Q.ps_pt[b0 + i]; }
    else pts = Q.ps_pt + b0;
    __syncthreads();
    int run = 0;
    for (int c0 = 0; c0 < n1; c0 += 256) {
        const int q = c0 + threadIdx.x;
        const int hit = q < n1 ? ba_find_sorted(pts, n2, Q.ps_pt[a0 + q]) : -1;
        // order-preserving position of the hits inside this chunk of 256
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned long long m = __ballot(hit >= 0);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int before = __popcll(m & ((1ull << lane) - 1ull)), tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) before += s_w[w]; tot += s_w[w]; }
        if (FILL && hit >= 0) Q.pairs[base + run + before] = make_int2(Q.ps_edges[a0 + q], Q.ps_edges[b0 + hit]);
        run += tot;
        __syncthreads();
    }
    if (!FILL && threadIdx.x == 0) Q.cnt[blockIdx.x] = run;
}

// one workgroup: offsets of the blocks' lists, the table of <= BA_SLICE-pair slices, totals.  A thread takes a run of consecutive blocks
// (nb = nf (nf + 1) / 2 <= 12880 for the 160 free poses a cut may have: 13 per thread), the runs' sums are scanned over the workgroup.
__global__ __launch_bounds__(1024) void k_ba_pairs_scan(BaPairPlan Q, int nb) {
    __shared__ int s_wa[16], s_wb[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nb + 1023) / 1024, i0 = min(nb, tid * per), i1 = min(nb, i0 + per);
    int a = 0, b = 0;
    for (int i = i0; i < i1; ++i) { const int c = Q.cnt[i]; a += c; b += (c + BA_SLICE - 1) / BA_SLICE; }
    int ia = a, ib = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64); if (lane >= o) { ia += ta; ib += tb; } }
    if (lane == 63) { s_wa[wave] = ia; s_wb[wave] = ib; }
    __syncthreads();
    int oa = ia - a, ob = ib - b, ta = 0, tb = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) { oa += s_wa[w]; ob += s_wb[w]; } ta += s_wa[w]; tb += s_wb[w]; }
    if (tid == 0) { *Q.n_pairs = ta; *Q.n_slices = tb; }
    for (int i = i0; i < i1; ++i) {
        const int c = Q.cnt[i];
        Q.off[i] = oa;
        int j1, j2;
        ba_block_of(i, Q.nf, j1, j2);
        for (int k = 0, sl = ob; k < c; k += BA_SLICE, ++sl) Q.blocks[sl] = BaBlock{j1, j2, oa + k, min(BA_SLICE, c - k)};
        oa += c; ob += (c + BA_SLICE - 1) / BA_SLICE;
    }
}

// The pair plan in ONE launch (the resident cut; the three launches above remain for vo_local_ba's own slab).  A block's list cannot be longer than the
// shorter of its two per-pose lists, so the lists are laid out at offsets of those bounds -- known from ps_start alone, no counting pass -- and
// the pair array has gaps (a slice names its range; nothing walks the array end to end).  Every workgroup stores its block's count, takes a ticket,
// and the last one builds the slice table from the counts (agent-scope loads: the counts come from workgroups on other XCDs).
// Q.off[0] is the ticket (zero at launch: k_cut_report clears it), Q.off[1 ..] unused.  Same pairs, same order inside a block as k_ba_pairs<true>.
__global__ __launch_bounds__(256) void k_ba_pairs_one(BaPairPlan Q, int nb) {
    extern __shared__ int s_pts[];
    __shared__ int s_w[16];
    __shared__ int s_row[VO_BA_RESIDENT_MAX_FREE];       // row r: sum over j2 >= r of min(len r, len j2)
    __shared__ int s_len[VO_BA_RESIDENT_MAX_FREE];
    __shared__ int s_last;
    const int nf = Q.nf;
    int j1, j2;
    ba_block_of(blockIdx.x, nf, j1, j2);
    for (int j = threadIdx.x; j < nf; j += 256) s_len[j] = Q.ps_start[j + 1] - Q.ps_start[j];
    __syncthreads();
    for (int r = threadIdx.x; r < nf; r += 256) { const int lr = s_len[r]; int a = 0; for (int c = r; c < nf; ++c) a += min(lr, s_len[c]); s_row[r] = a; }
    __syncthreads();
    int base = 0;
    for (int r = 0; r < j1; ++r) base += s_row[r];
    for (int c = j1; c < j2; ++c) base += min(s_len[j1], s_len[c]);
    const int a0 = Q.ps_start[j1], n1 = s_len[j1], b0 = Q.ps_start[j2], n2 = s_len[j2];
    int run = 0;
    if (j1 == j2) {
        for (int q = threadIdx.x; q < n1; q += 256) { const int e = Q.ps_edges[a0 + q]; Q.pairs[base + q] = make_int2(e, e); Q.pair_pt[base + q] = Q.ps_pt[a0 + q]; }
        run = n1;
    } else {
        // pose j2's sorted point list: in LDS when it fits this launch's allocation, searched in global memory otherwise (config 5: 11 k edges per pose)
        const int* pts = s_pts;
        if (n2 <= Q.lds_cap) { for (int i = threadIdx.x; i < n2; i += 256) s_pts[i] = Q.ps_pt[b0 + i]; }
        else pts = Q.ps_pt + b0;
        __syncthreads();
        for (int c0 = 0; c0 < n1; c0 += 256) {
            const int q = c0 + threadIdx.x;
            const int ptq = q < n1 ? Q.ps_pt[a0 + q] : -1;
            const int hit = q < n1 ? ba_find_sorted(pts, n2, ptq) : -1;
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;      // order-preserving position of the hits inside this chunk of 256
            const unsigned long long m = __ballot(hit >= 0);
            if (lane == 0) s_w[wave] = __popcll(m);
            __syncthreads();
            int before = __popcll(m & ((1ull << lane) - 1ull)), tot = 0;
            for (int w = 0; w < 4; ++w) { if (w < wave) before += s_w[w]; tot += s_w[w]; }
            if (hit >= 0) { Q.pairs[base + run + before] = make_int2(Q.ps_edges[a0 + q], Q.ps_edges[b0 + hit]); Q.pair_pt[base + run + before] = ptq; }
            run += tot;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&Q.cnt[blockIdx.x], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the count has been performed before the ticket is taken
        s_last = __hip_atomic_fetch_add(&Q.off[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;
    }
    __syncthreads();
    if (!s_last) return;
    // the last workgroup: the table of <= BA_SLICE-pair slices over all blocks, in block order.  A thread takes a run of consecutive blocks.
    __shared__ int s_wa[4], s_wb[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nb + 255) / 256, i0 = min(nb, tid * per), i1 = min(nb, i0 + per);
    int a = 0, b = 0, ub0 = 0;
    {   // offset of block i0 (bound-based), then this run's pair and slice counts
        int r1, r2;
        if (i0 < nb) { ba_block_of(i0, nf, r1, r2); for (int r = 0; r < r1; ++r) ub0 += s_row[r]; for (int c = r1; c < r2; ++c) ub0 += min(s_len[r1], s_len[c]); }
    }
    for (int i = i0; i < i1; ++i) { const int c = __hip_atomic_load(&Q.cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); a += c; b += (c + BA_SLICE - 1) / BA_SLICE; }
    int ia = a, ib = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64); if (lane >= o) { ia += ta; ib += tb; } }
    if (lane == 63) { s_wa[wave] = ia; s_wb[wave] = ib; }
    __syncthreads();
    int ob = ib - b, ta = 0, tb = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) ob += s_wb[w]; ta += s_wa[w]; tb += s_wb[w]; }
    if (tid == 0) { *Q.n_pairs = ta; *Q.n_slices = tb; Q.off[0] = 0; }
    int r1 = 0, r2 = 0, oa = ub0;
    if (i0 < nb) ba_block_of(i0, nf, r1, r2);
    for (int i = i0; i < i1; ++i) {
        const int c = __hip_atomic_load(&Q.cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0, sl = ob; k < c; k += BA_SLICE, ++sl) Q.blocks[sl] = BaBlock{r1, r2, oa + k, min(BA_SLICE, c - k)};
        ob += (c + BA_SLICE - 1) / BA_SLICE;
        oa += min(s_len[r1], s_len[r2]);                  // the next block's offset: this block's bound
        if (++r2 == nf) { ++r1; r2 = r1; }
    }
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---- k_ba_chol16: the D <= 192 Cholesky + solve, 16-column panels ---------------------------------------------
// The right-hand side rides along as row D of the augmented matrix [S b; b^T .], so the forward substitution falls out
// of the factorisation (the last row of the augmented factor is y = L^-1 b).  Blocked so that the dependent chain is short:
//  * panel = 16 columns = one DPP row.  The diagonal block is factored by ONE wave entirely in registers: lane r
//    (0..15) keeps row r; the pivot and the multipliers of column j reach the other lanes through the DP-ALU DPP
//    row broadcast (v_mov_b64_dpp / v_fmac_f64_dpp row_newbcast:j), i.e. ONE instruction per rank-1 term, no
//    LDS and no SGPR hop; the pivot uses v_rsq_f64 + 2 Newton steps (1/sqrt(d) directly, sqrt(d) = d * rsqrt(d))
//    instead of an IEEE sqrt followed by an IEEE divide;
//  * panel solve: one lane per row below the block, the factored block arrives as software-pipelined LDS broadcasts;
//  * trailing update: 16x16 tiles, K = 16 -> 4 x v_mfma_f64_16x16x4_f64 per tile, spread over the waves;
//  * backward substitution: wave 0 solves the 16x16 triangle in registers (DPP again), the other waves push x_p
//    into the rows above -- one barrier per panel.
#define CH_NB 16
#define TRI32(r, c) ((int)(r) * ((int)(r) + 1) / 2 + (int)(c))   // packed lower triangle, 32-bit index math (< 2^15 entries)
#define CH_THREADS 512
__device__ __forceinline__ double ba_readlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// 1/sqrt(d) = y (1 + q):  y = v_rsq_f64 (2^-24), q = e (1/2 + 3/8 e), e = 1 - d y^2 -- one cubic step, 2^-52.7
// (tools/chol_bench.hip probes both)
__device__ __forceinline__ void ba_rsqrt_parts(double d, double& y, double& q) {
#pragma clang fp contract(fast)
    y = __builtin_amdgcn_rsq(d);
    const double e = 1.0 - (d * y) * y;
    q = e * (0.5 + 0.375 * e);
}
// DP-ALU DPP helpers.  A VGPR written by a VALU op needs 2 wait states before a DPP op reads it; the compiler cannot
// see inside the asm, so the producers below carry the s_nop themselves.
template <int K> __device__ __forceinline__ double ch_bcast(double v) {           // value of lane K of each 16-lane row
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v), "n"(K));
    return r;
}
__device__ __forceinline__ double ch_mul_for_dpp(double a, double b) {             // a * b, safe to feed a DPP read next
    double r;
    asm("v_mul_f64 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double ch_fma_for_dpp(double a, double b, double c) {      // a * b + c, safe to feed a DPP read next
    double r;
    asm("v_fma_f64 %0, %1, %2, %3\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int K> __device__ __forceinline__ double ch_mul_bcast(double bsrc, double m) {       // bsrc[lane K] * m
    double acc = 0.0;
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(m), "n"(K));
    return acc;
}
template <int K> __device__ __forceinline__ void ch_fnma_bcast(double& acc, double bsrc, double m) {   // acc -= bsrc[lane K] * m
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(m), "n"(K));
}
// the same, pinned in program order against the other volatile statements: the rank-1 updates of the block factorisation must stay
// right-looking (left to itself the scheduler sinks them next to the pivot that consumes them -- a chain of J dependent
// accumulations in front of pivot J instead of one)
template <int K> __device__ __forceinline__ void ch_fnma_bcast_v(double& acc, double bsrc, double m) {
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(m), "n"(K));
}
template <int K> __device__ __forceinline__ double ch_bcast_v(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v), "n"(K));
    return r;
}
template <int J, int K> struct ChRank1 {
    static __device__ __forceinline__ void run(double (&a)[CH_NB], double l) { ch_fnma_bcast_v<K>(a[K], l, l); ChRank1<J, K + 1>::run(a, l); }
};
template <int J> struct ChRank1<J, CH_NB> { static __device__ __forceinline__ void run(double (&)[CH_NB], double) {} };
template <int J> struct ChCol {                                 // column J of the in-register block factorisation
    // No divergent branch may sit between the DPP ops (an EXEC write needs 5 wait states before the next DPP and the
    // broadcast source lanes must be active), and this wave is issue-bound (~28 VALU instructions per column at ~8 clocks each, the
    // dependent chain of ~110 clocks hides underneath): nothing per column that is not arithmetic.  Lanes above the diagonal (r < J)
    // carry the symmetric counterpart in a[J] and are NOT masked: what they compute lands in entries of their row right of the diagonal,
    // which nobody reads.  A non-positive pivot is not tested here either: rsq turns it into NaN / inf, which reaches every later pivot
    // inverse of the block (ch_pivots_ok looks at them once).
    static __device__ __forceinline__ void run(double (&a)[CH_NB], double* s_pinv) {
        const double d = ch_bcast_v<J>(a[J]);
        double y, q;
        ba_rsqrt_parts(d, y, q);
        const double l0 = a[J] * y;                             // off the dependent chain (parallel to e, q)
        const double l = ch_fma_for_dpp(l0, q, l0);             // a * rsqrt(d); lane J: d * rsqrt(d) = sqrt(d)
        s_pinv[J] = fma(y, q, y);
        a[J] = l;
        ChRank1<J, J + 1>::run(a, l);
        ChCol<J + 1>::run(a, s_pinv);
    }
};
template <> struct ChCol<CH_NB> { static __device__ __forceinline__ void run(double (&)[CH_NB], double*) {} };
// every pivot of the block was positive: its inverse square root is finite and positive in every lane (NaN compares false)
__device__ __forceinline__ bool ch_pivots_ok(double myinv) { return __ballot(!(myinv > 0.0 && myinv < 1e300)) == 0ull; }
__device__ __forceinline__ void ch_exec_settle(double& v) { asm("s_nop 4" : "+v"(v)); }   // EXEC write -> DPP: 5 wait states
template <int K> struct ChBack {                                // step K of the in-register back substitution
    static __device__ __forceinline__ void run(const double (&col)[CH_NB], double& y, double inv, double& xf, int r) {
        const double xv = ch_mul_for_dpp(y, inv);
        xf = r == K ? xv : xf;
        ch_fnma_bcast<K>(y, xv, col[K]);                        // col[K] = 0 for lanes >= K
        ChBack<K - 1>::run(col, y, inv, xf, r);
    }
};
template <> struct ChBack<-1> { static __device__ __forceinline__ void run(const double (&)[CH_NB], double&, double, double&, int) {} };
// Back substitution on pre-scaled residuals z_j = y_j / L[j][j]: x_k = z_k when its turn comes and
// z_j -= (L[k][j] / L[j][j]) x_k, so one dependent v_fmac_f64_dpp per step (no multiply, no select on the chain).
template <int K> struct ChBackZ {
    static __device__ __forceinline__ void run(const double (&colS)[CH_NB], double& z) {
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(z) : "v"(colS[K]), "n"(K));   // colS[K] = 0 for lanes >= K
        ChBackZ<K - 1>::run(colS, z);
    }
};
template <> struct ChBackZ<0> { static __device__ __forceinline__ void run(const double (&)[CH_NB], double&) {} };   // step 0 would only touch lanes < 0
template <int K> struct ChPush2 {                               // two accumulators: acc_(K&1) -= x[lane K] * Lx[K]
    static __device__ __forceinline__ void run(const double (&Lx)[CH_NB], double& a0, double& a1, double x) {
        ch_fnma_bcast<K>((K & 1) ? a1 : a0, x, Lx[K]);
        ChPush2<K + 1>::run(Lx, a0, a1, x);
    }
};
template <> struct ChPush2<CH_NB> { static __device__ __forceinline__ void run(const double (&)[CH_NB], double&, double&, double) {} };
template <int K> struct ChFwd {                                 // step K of the in-register forward substitution (rhs row only)
    static __device__ __forceinline__ void run(const double (&Lr)[CH_NB], double& bc, double inv, double& yf, int r) {
        const double yv = ch_mul_for_dpp(bc, inv);
        yf = r == K ? yv : yf;
        ch_fnma_bcast<K>(bc, yv, Lr[K]);                        // Lr[K] = 0 for lanes <= K
        ChFwd<K + 1>::run(Lr, bc, inv, yf, r);
    }
};
template <> struct ChFwd<CH_NB> { static __device__ __forceinline__ void run(const double (&)[CH_NB], double&, double, double&, int) {} };

template <int K> struct ChPush {                               // y -= sum_k x[lane k] * Lx[k]: x_p pushed into the panel above, in registers
    static __device__ __forceinline__ void run(const double (&Lx)[CH_NB], double& y, double x) { ch_fnma_bcast<K>(y, x, Lx[K]); ChPush<K + 1>::run(Lx, y, x); }
};
template <> struct ChPush<CH_NB> { static __device__ __forceinline__ void run(const double (&)[CH_NB], double&, double) {} };

template <int K, int C> struct ChSolveRow {                     // x[C] -= L[C][K] * x[K], L[C][K] = lane C's Lk[K]
    static __device__ __forceinline__ void run(double (&x)[CH_NB], const double (&Lk)[CH_NB]) { ch_fnma_bcast<C>(x[C], Lk[K], x[K]); ChSolveRow<K, C + 1>::run(x, Lk); }
};
template <int K> struct ChSolveRow<K, CH_NB> { static __device__ __forceinline__ void run(double (&)[CH_NB], const double (&)[CH_NB]) {} };
template <int K> struct ChSolve {                               // panel solve of one matrix row per lane, the block row k in lanes 0..15 of each DPP row
    static __device__ __forceinline__ void run(double (&x)[CH_NB], const double (&Lk)[CH_NB]) {
        x[K] = ch_mul_bcast<K>(Lk[K], x[K]);                    // diagonal slot of the block copy holds 1 / L[K][K]
        ChSolveRow<K, K + 1>::run(x, Lk);
        ChSolve<K + 1>::run(x, Lk);
    }
};
template <> struct ChSolve<CH_NB> { static __device__ __forceinline__ void run(double (&)[CH_NB], const double (&)[CH_NB]) {} };

// One wave factors the 16x16 block at (j0, j0) of the packed triangle s_L in registers and leaves: the block's L in
// s_L, its transposed copy (1/L[k][k] on the diagonal) in s_dg, the pivot inverses in s_inv.  Returns false on a
// non-positive pivot.  `a` arrives loaded (row r16 of the block in every 16-lane DPP row, identity padding).
__device__ __forceinline__ bool ch_factor_block(double (&a)[CH_NB], double* s_L, double* s_dg, double* s_inv, double* s_pinv, int j0, int nb, int lane) {
    const int r16 = lane & 15;
    ch_exec_settle(a[0]);
    ChCol<0>::run(a, s_pinv);
    const double myinv = s_pinv[r16];
    const bool ok = ch_pivots_ok(myinv);
    if (lane < CH_NB) {                                         // transposed copy: column k of the block is contiguous
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) s_dg[c * CH_NB + lane] = c < lane ? a[c] : (c == lane ? myinv : 0.0);
    }
    if (lane < nb) {
        double* wrow = s_L + TRI32(j0 + lane, j0);
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) if (c <= lane) wrow[c] = a[c];
        s_inv[j0 + lane] = myinv;
    }
    return ok;
}

// 16-byte write-through (sc1) store: the line leaves this XCD's L2, so a later remote atomic / a later load after an acquire sees
// memory, not a stale copy (persistent LM kernel: the solver workgroup clears S behind its own load)
__device__ __forceinline__ void ba_st16_sc1(double* p, double a, double b) {
    typedef double f64x2_ __attribute__((ext_vector_type(2)));
    const f64x2_ v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
// PB = false: the body of k_ba_chol16 (launch-per-phase path).  PB = true: the solver workgroup of the persistent LM kernel
// (k_ba_persist): lambda comes from the caller, the control block is not touched, S / b_s are cleared behind the load (write-through:
// they are the targets of the next step's Schur atomics) and the solution goes to x_out with write-through stores.
template <bool PB>
__device__ __forceinline__ bool ba_chol16_body(const BaDev& B, BaCtl* ctl_, double lambda_in, double* s_mem, double* x_out, bool clear_after_load = false) {
    int D_ = B.D;
    // inside the persistent kernel's step loop everything derived from D and the LDS base is loop invariant; hoisted out of the loop it
    // stays live across the whole kernel and spills: the values are laundered so that they are recomputed per call
    int tid_ = threadIdx.x;
    const double* A_ = B.S;
    if (PB) {
        // the LDS base is laundered as an address_space(3) pointer: through a generic pointer the compiler would lose the address
        // space and fall back to flat_load / flat_store for every LDS access of the factorisation
        typedef __attribute__((address_space(3))) double lds_f64;
        typedef __attribute__((address_space(1))) const double glb_f64;
        lds_f64* sm3 = (lds_f64*)s_mem;
        glb_f64* a1 = (glb_f64*)A_;
        asm volatile("" : "+s"(D_), "+s"(sm3), "+s"(a1));
        asm volatile("" : "+v"(tid_));
        s_mem = (double*)sm3; A_ = (const double*)a1;
    }
    const int D = D_, DA = D + 1, tid = tid_, lane = tid & 63, wave = tid >> 6, r16 = lane & 15;
    const double* const A = A_;
    double* const s_dg = s_mem;                                 // [16][16] factored diagonal block, transposed, 1/L[k][k] on the diagonal
    double* const s_L = s_mem + CH_NB * CH_NB;                  // packed lower triangle of [S b; b^T 0]
    double* const s_b = s_L + TRI32(DA, 0);                       // y, then x
    double* const s_inv = s_b + D;                              // 1 / L[j][j]
    __shared__ int s_ok;
    __shared__ double s_pinv[CH_NB];                            // 1 / pivot of the block being factored
#ifdef CH_STAMPS
    long long t_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_ = clock64();
#define CH_STAMP(i) { const long long n_ = clock64(); t_[i] += n_ - tl_; tl_ = n_; }
#else
#define CH_STAMP(i)
#endif
    if (tid == 0) s_ok = 1;
    // the Schur kernel accumulated -sum W H^-1 W^T into a zeroed S: blockdiag(H_pp) + lambda I and b_p join while the system is
    // loaded (no separate init launch).  Every lane derives this step's lambda the way k_ba_init_S does; the control block is
    // taken over only after the barrier below, when nobody reads it any more.
    double lambda = lambda_in;
    if (!PB) lambda = (ctl_->need_lin && ctl_->first) ? 1e-5 * B.scal[4] : ctl_->lambda;
    const double* const Hpp = B.Hpp;
    // ---- load.  S arrives as the packed lower triangle the factorisation works on (ba_tri), so it is copied global -> LDS by the DMA
    // path (global_load_lds_dwordx4: 1 KiB per wave instruction, no registers, every piece in flight at once); blockdiag(H_pp) + lambda I
    // and the right-hand side row b_s + b_p join in LDS.  (Row-by-row register loads of a full matrix: two dependent batches per wave, 10 us
    // at D = 144.)
    const int ntri = D * (D + 1) / 2;
    {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef __attribute__((address_space(1))) const void glb_void;
        const int npiece = (ntri + 127) >> 7;                   // 128 doubles = 1 KiB per piece; the slab behind S is readable up to the next multiple (host: carve)
        for (int pc = wave; pc < npiece; pc += CH_THREADS / 64)
            if (pc * 128 + 2 * lane < ntri)                     // the last piece is cut at the end of the triangle (at most one double beyond it, inside both buffers)
                __builtin_amdgcn_global_load_lds((glb_void*)(A + (size_t)pc * 128 + 2 * lane), (lds_void*)(s_L + (size_t)pc * 128), 16, 0, 0);
    }
    const double rhs_v = tid < D ? B.bs[tid] + B.bp[tid] : 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 36 * (D / 6); i += CH_THREADS) {     // the 6x6 diagonal blocks of H_pp (lower halves) and lambda on the diagonal
        const int j = i / 36, a = (i % 36) / 6, b = i % 6;
        if (b <= a) s_L[TRI32(6 * j + a, 6 * j + b)] += Hpp[i] + (a == b ? lambda : 0.0);
    }
    if (tid < D) s_L[TRI32(D, tid)] = rhs_v;
    if (tid == 0) s_L[TRI32(D, D)] = 0.0;
    if (PB) {                                                   // clear S / b_s behind the copy for the next step's atomics (write-through: see ba_st16_sc1)
        double* g = const_cast<double*>(A);
        for (int i = 2 * tid; i < ntri; i += 2 * CH_THREADS) ba_st16_sc1(g + i, 0.0, 0.0);      // the slab is padded: a pair past the end is harmless
        if (tid < D) __hip_atomic_store(B.bs + tid, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (clear_after_load) {                              // launch path, second generation: plain stores (the kernel boundary publishes them)
        double2* g = reinterpret_cast<double2*>(const_cast<double*>(A));
        for (int i = tid; 2 * i < ntri; i += CH_THREADS) g[i] = make_double2(0.0, 0.0);
        if (tid < D) B.bs[tid] = 0.0;
    }
    __syncthreads();
    if (wave == 0) {                                            // first diagonal block, in registers (the other waves wait at the barrier)
        const int nb = min(CH_NB, D);
        double a[CH_NB];
        const bool mine = r16 < nb;
        const double* row = s_L + TRI32(mine ? r16 : 0, 0);
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) a[c] = row[min(c, mine ? r16 : 0)];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) a[c] = (mine && c <= r16) ? a[c] : (c == r16 ? 1.0 : 0.0);
        if (!ch_factor_block(a, s_L, s_dg, s_inv, s_pinv, 0, nb, lane) && lane == 0) s_ok = 0;
    }
    __syncthreads();
    if (!PB && tid == 0) {
        BaCtl* c = ctl_;                               // take over the fresh linearisation, clear the trial sums
        if (c->need_lin) {
            c->cur = B.scal[0];
            if (c->first) { c->lambda = lambda; c->ni = 2; c->first = 0; }
            c->need_lin = 0;
        }
        B.scal[1] = 0; B.scal[2] = 0; B.scal[7] = 0;
    }
    CH_STAMP(0)
    // one 16x16 tile of the trailing update S22 -= L21 L21^T on the f64 matrix cores (K = 16: 4 MFMAs).  Lane l holds
    // A[l&15][l>>4], B[l>>4][l&15]; D: col = l&15, row = (l>>4) + 4 reg.  Rows past the end are clamped (their
    // products land in entries that are never written back).
    auto tile = [&](int tr, int tc, int base, int j0, int m) {
        const int ra = min(16 * tr + r16, m - 1), rb = min(16 * tc + r16, m - 1), kq = lane >> 4;
        const double* pa = s_L + TRI32(base + ra, j0) + kq;
        const double* pb = s_L + TRI32(base + rb, j0) + kq;
        const double a0 = pa[0], a1 = pa[4], a2 = pa[8], a3 = pa[12], b0 = pb[0], b1 = pb[4], b2 = pb[8], b3 = pb[12];
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 0);
        const int col = 16 * tc + r16;
        double* pc[4]; double cv[4]; bool st[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rw = 16 * tr + (lane >> 4) + 4 * q;
            st[q] = rw < m && col <= rw;
            pc[q] = s_L + TRI32(base + min(rw, m - 1), base + min(col, min(rw, m - 1)));   // clamped: the load is unconditional
            cv[q] = *pc[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) if (st[q]) *pc[q] = cv[q] - acc[q];
    };
    auto tile2 = [&](int trA, int tcA, int trB, int tcB, int base, int j0, int m) {      // two independent tiles, interleaved
        const int kq = lane >> 4;
        const int raA = min(16 * trA + r16, m - 1), rbA = min(16 * tcA + r16, m - 1), raB = min(16 * trB + r16, m - 1), rbB = min(16 * tcB + r16, m - 1);
        const double* paA = s_L + TRI32(base + raA, j0) + kq; const double* pbA = s_L + TRI32(base + rbA, j0) + kq;
        const double* paB = s_L + TRI32(base + raB, j0) + kq; const double* pbB = s_L + TRI32(base + rbB, j0) + kq;
        const double a0 = paA[0], a1 = paA[4], a2 = paA[8], a3 = paA[12], b0 = pbA[0], b1 = pbA[4], b2 = pbA[8], b3 = pbA[12];
        const double c0 = paB[0], c1 = paB[4], c2 = paB[8], c3 = paB[12], d0 = pbB[0], d1 = pbB[4], d2 = pbB[8], d3 = pbB[12];
        double* pcA[4]; double* pcB[4]; double cvA[4], cvB[4]; bool stA[4], stB[4];
        const int colA = 16 * tcA + r16, colB = 16 * tcB + r16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rwA = 16 * trA + kq + 4 * q, rwB = 16 * trB + kq + 4 * q;
            stA[q] = rwA < m && colA <= rwA; stB[q] = rwB < m && colB <= rwB;
            pcA[q] = s_L + TRI32(base + min(rwA, m - 1), base + min(colA, min(rwA, m - 1)));
            pcB[q] = s_L + TRI32(base + min(rwB, m - 1), base + min(colB, min(rwB, m - 1)));
            cvA[q] = *pcA[q]; cvB[q] = *pcB[q];
        }
        f64x4 accA = {0.0, 0.0, 0.0, 0.0}, accB = {0.0, 0.0, 0.0, 0.0};
        accA = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f64_16x16x4f64(c0, d0, accB, 0, 0, 0);
        accA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f64_16x16x4f64(c1, d1, accB, 0, 0, 0);
        accA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f64_16x16x4f64(c2, d2, accB, 0, 0, 0);
        accA = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f64_16x16x4f64(c3, d3, accB, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { if (stA[q]) *pcA[q] = cvA[q] - accA[q]; if (stB[q]) *pcB[q] = cvB[q] - accB[q]; }
    };
    for (int j0 = 0; j0 < D; j0 += CH_NB) {
        const int nb = min(CH_NB, D - j0);
        if (!s_ok) break;
        const int base = j0 + nb, m = DA - base;                // rows below the panel, incl. the rhs row
        if (nb == CH_NB) {                                      // ---- panel solve, one lane per row
            if (wave * 64 < m) {
                // right-looking within the row: x[c] -= L[c][k] x[k] as soon as x[k] is final.  Lane r16 of every DPP
                // row keeps column r16 of the transposed block copy (16 LDS reads per lane in total), L[c][k] reaches
                // the FMA as the row broadcast of lane c: one v_fmac_f64_dpp per term, no LDS broadcast traffic.
                // Every lane of the wave stays active (DPP sources); rows past the end are clamped and not stored.
                double* prow = s_L + TRI32(base + min(tid, m - 1), j0);
                double x[CH_NB], Lk[CH_NB];
#pragma unroll
                for (int k = 0; k < CH_NB; ++k) Lk[k] = s_dg[k * CH_NB + r16];
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) x[c] = prow[c];
                ch_exec_settle(x[0]);
                ChSolve<0>::run(x, Lk);
                if (tid < m) {
#pragma unroll
                    for (int c = 0; c < CH_NB; ++c) prow[c] = x[c];
                }
            }
        } else if (wave == 0) {
            // partial panel = the last one: only the rhs row is left below it.  Lane c keeps b_c and row c of the
            // block; y_k travels by DPP broadcast.
            double* prow = s_L + TRI32(D, j0);
            double Lr[CH_NB];
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) Lr[k] = s_dg[k * CH_NB + r16];
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) Lr[k] = k < r16 ? Lr[k] : 0.0;
            double bc = prow[min(r16, nb - 1)], yf = 0.0;
            bc = r16 < nb ? bc : 0.0;
            const double inv = s_dg[r16 * CH_NB + r16];
            ch_exec_settle(bc);
            ChFwd<0>::run(Lr, bc, inv, yf, r16);
            if (lane < nb) prow[lane] = yf;
        }
        __syncthreads();
        CH_STAMP(1)
        if (base >= D) break;                                   // only the rhs row was left: nothing to update
        // ---- trailing update with look-ahead: wave 0 updates the tile holding the next diagonal block and factors it
        // while the other waves update the rest of the trailing matrix.
        const int T = (m + 15) >> 4, ntile = T * (T + 1) / 2;
        if (wave == 0) {
            // the tile holding the next diagonal block: updated on the matrix cores like the others, but handed to the factorisation's
            // row-per-lane layout through the (now idle) 16x16 scratch s_dg instead of the packed triangle -- the block's pre-factor values
            // are never needed there.  Entry (row, col) sits at row * 16 + (col ^ row): conflict-free both for the MFMA layout's
            // writes (a row per 16 lanes) and for the row reads (a column per instruction).
            const int nb2 = min(CH_NB, D - base);
            double a[CH_NB];
            {
                const int kq = lane >> 4, ra = min(r16, m - 1);
                const double* pa = s_L + TRI32(base + ra, j0) + kq;
                const double a0 = pa[0], a1 = pa[4], a2 = pa[8], a3 = pa[12];
                double cv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {                  // old values, by symmetry from the packed lower triangle (clamped past the end)
                    const int rw = kq + 4 * q, hi = min(max(rw, r16), m - 1), lo = min(min(rw, r16), hi);
                    cv[q] = s_L[TRI32(base + hi, base + lo)];
                }
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, a3, acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rw = kq + 4 * q;
                    const double val = cv[q] - acc[q];
                    s_dg[rw * CH_NB + (r16 ^ rw)] = val;
                    if (rw >= nb2 && rw < m && r16 <= rw) s_L[TRI32(base + rw, base + r16)] = val;      // a partial last block: the rhs row below it stays in the triangle
                }
            }
            CH_STAMP(5)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool mine = r16 < nb2;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = s_dg[r16 * CH_NB + (c ^ r16)];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = (mine && c <= r16) ? a[c] : (c == r16 ? 1.0 : 0.0);
            CH_STAMP(6)
            if (!ch_factor_block(a, s_L, s_dg, s_inv, s_pinv, base, nb2, lane) && lane == 0) s_ok = 0;
            CH_STAMP(7)
        } else {
            // tiles 1.. in (tr, tc <= tr) order over 7 waves, two tiles in flight per wave: a tile is a dependent chain (operand loads ->
            // 4 MFMAs on one accumulator -> read-modify-write of the target), ~1400 clocks of mostly latency; with D >= 130 the first panels'
            // 45 / 36 tiles are more than wave 0's chain hides
            constexpr int NW = CH_THREADS / 64 - 1;
            int tr = 0, tc = wave;
            while (tc > tr) { tc -= tr + 1; ++tr; }
            for (int t = wave; t < ntile; t += 2 * NW) {
                int tr2 = tr, tc2 = tc + NW;
                while (tc2 > tr2) { tc2 -= tr2 + 1; ++tr2; }
                if (t + NW < ntile) tile2(tr, tc, tr2, tc2, base, j0, m);
                else tile(tr, tc, base, j0, m);
                tr = tr2; tc = tc2 + NW;
                while (tc > tr) { tc -= tr + 1; ++tr; }
            }
        }
        __syncthreads();
        CH_STAMP(2)
    }
    CH_STAMP(3)
    if (s_ok) {
        for (int i = tid; i < D; i += CH_THREADS) s_b[i] = s_L[TRI32(D, i)];    // y = L^-1 b (last row of the augmented factor)
        __syncthreads();
        // ---- L^T x = y, panel by panel from the bottom.  Wave 0 owns the dependent chain: it solves the 16x16 triangle
        // in registers, pushes x_p into the panel above by DPP (no LDS round trip) and goes on; the other waves push x_p
        // into the rows further up.  One barrier per panel.
        const int np = (D + CH_NB - 1) / CH_NB;
        double xf = 0.0;                                        // wave 0, lane k (of every DPP row): x[j0 + k]
        double Lx[CH_NB], colS[CH_NB], inv = 0.0;               // wave 0: operands of the NEXT panel up, loaded one panel ahead
        auto load_panel = [&](int jt, int jb, int nbb) {        // target panel jt (full), pushed from panel jb of width nbb
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) Lx[k] = s_L[TRI32(jb + min(k, nbb - 1), jt) + r16];
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) colS[k] = s_L[TRI32(jt + k, jt) + min(r16, k)];      // uniform row base + lane offset
            inv = s_inv[jt + r16];
        };
        auto mask_panel = [&](int nbb) {
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) { Lx[k] = k < nbb ? Lx[k] : 0.0; colS[k] = k > r16 ? colS[k] * inv : 0.0; }
        };
        if (wave == 0) {
            const int j0 = CH_NB * (np - 1), nb = D - j0;
            double col[CH_NB];                                  // lane j: L[j0+k][j0+j] / L[j][j], k > j
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) col[k] = s_L[TRI32(j0 + min(max(k, r16), nb - 1), j0 + min(r16, nb - 1))];
            double y = s_b[j0 + min(r16, nb - 1)];
            double iv = s_inv[j0 + min(r16, nb - 1)];
            y = r16 < nb ? y : 0.0; iv = r16 < nb ? iv : 0.0;
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) col[k] = (r16 < nb && k < nb && k > r16) ? col[k] * iv : 0.0;
            if (np > 1) load_panel(j0 - CH_NB, j0, nb);          // in flight during the chain
            double z = y * iv;
            ch_exec_settle(z);
            ChBackZ<CH_NB - 1>::run(col, z);
            xf = z;
            if (lane < nb) s_b[j0 + lane] = xf;
            if (np > 1) mask_panel(nb);
        }
        __syncthreads();
        for (int p = np - 1; p >= 1; --p) {
            const int j0 = CH_NB * p, nb = min(CH_NB, D - j0), jn = j0 - CH_NB;
            if (wave == 0) {
                double y = s_b[jn + r16];
                double cLx[CH_NB], cS[CH_NB];
                const double cinv = inv;
#pragma unroll
                for (int k = 0; k < CH_NB; ++k) { cLx[k] = Lx[k]; cS[k] = colS[k]; }
                if (p > 1) load_panel(jn - CH_NB, jn, CH_NB);   // next panel's operands: in flight during push + chain
                CH_STAMP(4)
                const double xp = ch_mul_for_dpp(xf, 1.0);
                double a0 = 0.0, a1 = 0.0;
                ch_exec_settle(a0);
                ChPush2<0>::run(cLx, a0, a1, xp);
                double z = (y + (a0 + a1)) * cinv;
                ChBackZ<CH_NB - 1>::run(cS, z);
                xf = z;
                if (lane < CH_NB) s_b[jn + lane] = xf;
                if (p > 1) mask_panel(CH_NB);
                CH_STAMP(8)
            } else {
                const int r = tid - 64;
                if (r < jn) {
#pragma clang fp contract(fast)
                    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                    if (nb == CH_NB) {
#pragma unroll
                        for (int k = 0; k < CH_NB; k += 4) {
                            s0 += s_L[TRI32(j0 + k, r)] * s_b[j0 + k];
                            s1 += s_L[TRI32(j0 + k + 1, r)] * s_b[j0 + k + 1];
                            s2 += s_L[TRI32(j0 + k + 2, r)] * s_b[j0 + k + 2];
                            s3 += s_L[TRI32(j0 + k + 3, r)] * s_b[j0 + k + 3];
                        }
                    } else {
                        for (int k = 0; k + 1 < nb; k += 2) {   // nb is even
                            s0 += s_L[TRI32(j0 + k, r)] * s_b[j0 + k];
                            s1 += s_L[TRI32(j0 + k + 1, r)] * s_b[j0 + k + 1];
                        }
                    }
                    s_b[r] -= (s0 + s1) + (s2 + s3);
                }
            }
            __syncthreads();
            CH_STAMP(9)
        }
        if (PB) { for (int i = tid; i < D; i += CH_THREADS) __hip_atomic_store(x_out + i, s_b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else { for (int i = tid; i < D; i += CH_THREADS) x_out[i] = s_b[i]; }
    }
    if (!PB && tid == 0) B.scal[3] = s_ok ? 1.0 : 0.0;
#ifdef CH_STAMPS
    CH_STAMP(10)
    if (tid == 0) for (int i = 0; i < 12; ++i) B.dl[i] = (double)t_[i];
#endif
    return s_ok != 0;
}

__device__ __forceinline__ void ba_pose_body(const BaDev& B, double lambda, int blk, const double* poses_c, double* poses_t);
// trial_poses = 1 (second-generation phases, vo_ba_phase2.h): the kernel also writes the trial poses exp(dp) T and the pose part of the
// gain ratio, which is what k_ba_update's pose workgroups do in the first generation
// phase2 = 1 (vo_ba_phase2.h): S and b_s are cleared behind the load (the next step's Schur kernel accumulates into them: no separate zeroing
// launch) and the solution goes to B.dl instead of overwriting b_s
__global__ __launch_bounds__(CH_THREADS) void k_ba_chol16(BaBatch Q, int trial_poses, int phase2) {
    BA_PROBLEM(Q)
    if (ctl_->finished || B.D > 192 || B.s_tiles) return;      // larger systems: k_ba_chol16g; tile-major ones: k_ba_chol16v2
    extern __shared__ double s_mem[];
    (void)ba_chol16_body<false>(B, ctl_, 0.0, s_mem, phase2 ? B.dl : B.bs, phase2 != 0);
    if (trial_poses) {
        __syncthreads();                                       // dp, the ok flag and the control block are complete
        BA_STATE(B)
        const double lambda = ctl_->lambda;
        for (int blk = 0; blk * CH_THREADS < B.n_poses; ++blk) ba_pose_body(B, lambda, blk, poses_c, poses_t);
    }
}

#include "vo_ba_chol2.h"
// second generation of the same solve (vo_ba_chol2.h): the waves have roles and hand work over through words in LDS
__global__ __launch_bounds__(CH2_T) void k_ba_chol16v2(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished || !B.s_tiles) return;
    extern __shared__ double s_mem[];
    ba_chol16v2_body<false>(B, ctl_, s_mem, B.dl, true);
}

// ---- k_ba_chol16g: the same 16-column scheme for D > 192, where the packed triangle no longer fits in LDS -----------
// The matrix stays in global memory (S is rebuilt every step, so its lower triangle is overwritten by L in place; at
// these sizes it sits in L2), LDS holds the solved panel (DA x 16, the MFMA operands of the trailing update), the
// augmented rhs row, y / x and the pivots.  (A volatile pointer would make every access system-coherent, i.e. miss
// every cache: 5x slower.  Workgroup-scope coherence needs nothing beyond the barriers: one CU, one L1.)
#define CHG_TB 8
__global__ __launch_bounds__(CH_THREADS) void k_ba_chol16g(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished || (B.D <= 192 && !B.gen1)) return;
    extern __shared__ double s_mem[];
    const int D = B.D, DA = D + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15;
    double* const S = B.S;              // plain accesses: the waves of one workgroup share the CU's L1, barriers order them
    double* const s_dg = s_mem;                                 // [16][16] factored diagonal block, transposed, 1/L[k][k] on the diagonal
    double* const s_P = s_mem + CH_NB * CH_NB;                  // [DA][16] solved panel rows (row index relative to `base`)
    double* const s_aug = s_P + (size_t)DA * CH_NB;             // [DA] rhs row of the augmented matrix
    double* const s_b = s_aug + DA;                             // y, then x
    double* const s_inv = s_b + D;                              // 1 / L[j][j]
    __shared__ int s_ok;
    __shared__ double s_pinv[CH_NB];
    if (tid == 0) {
        s_ok = 1;
        BaCtl* c = ctl_;                               // take over the fresh linearisation, clear the trial sums
        if (c->need_lin) {
            c->cur = B.scal[0];
            if (c->first) { c->lambda = 1e-5 * B.scal[4]; c->ni = 2; c->first = 0; }
            c->need_lin = 0;
        }
        B.scal[1] = 0; B.scal[2] = 0; B.scal[7] = 0;
    }
    for (int i = tid; i < D; i += CH_THREADS) s_aug[i] = B.bs[i];
    if (tid == 0) s_aug[D] = 0.0;
    __syncthreads();
    for (int j0 = 0; j0 < D; j0 += CH_NB) {
        const int nb = min(CH_NB, D - j0);
        const int base = j0 + nb, m = DA - base;                // rows below the panel, incl. the rhs row (local row m - 1)
        // rows of the panel solve are fetched before the barrier: they do not depend on the block factor
        double x[CH_NB];
        const bool solver = nb == CH_NB && wave * 64 < m;
        if (solver) {
            const int rl = min(tid, m - 1), r = base + rl;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) x[c] = r < D ? S[(size_t)r * D + j0 + c] : s_aug[j0 + c];
        }
        if (wave == 0) {                                        // ---- diagonal block, in registers
            double a[CH_NB];
            const bool mine = r16 < nb;
            const double* row = S + (size_t)(j0 + (mine ? r16 : 0)) * D + j0;
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = row[min(c, nb - 1)];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = (mine && c <= r16) ? a[c] : (c == r16 ? 1.0 : 0.0);
            ch_exec_settle(a[0]);
            ChCol<0>::run(a, s_pinv);
            const double myinv = s_pinv[r16];
            const bool ok = ch_pivots_ok(myinv);
            if (lane < CH_NB) {
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) s_dg[c * CH_NB + lane] = c < lane ? a[c] : (c == lane ? myinv : 0.0);
            }
            if (lane < nb) {
                double* wrow = S + (size_t)(j0 + lane) * D + j0;
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) if (c <= lane) wrow[c] = a[c];
                s_inv[j0 + lane] = myinv;
            }
            if (!ok && lane == 0) s_ok = 0;
        }
        __syncthreads();
        if (!s_ok) break;
        if (nb == CH_NB) {                                      // ---- panel solve, one lane per row, CH_THREADS rows per pass
            // (round 6: the rows beyond the first CH_THREADS were never solved -- systems above D = 527, i.e. 88 free keyframes, were factored
            // wrongly; the parity tests stopped at D = 360 against the restatement and compared HIP with HIP above)
            for (int row0 = 0; row0 < m; row0 += CH_THREADS) {
                if (wave * 64 + row0 >= m) break;               // (wave-uniform)
                if (row0) {
                    const int rl = min(row0 + tid, m - 1), r = base + rl;
#pragma unroll
                    for (int c = 0; c < CH_NB; ++c) x[c] = r < D ? S[(size_t)r * D + j0 + c] : s_aug[j0 + c];
                }
                double Lk[CH_NB];
#pragma unroll
                for (int k = 0; k < CH_NB; ++k) Lk[k] = s_dg[k * CH_NB + r16];
                ch_exec_settle(x[0]);
                ChSolve<0>::run(x, Lk);
                if (row0 + tid < m) {
                    const int r = base + row0 + tid;
#pragma unroll
                    for (int c = 0; c < CH_NB; ++c) {
                        s_P[(row0 + tid) * CH_NB + c] = x[c];
                        if (r < D) S[(size_t)r * D + j0 + c] = x[c]; else s_aug[j0 + c] = x[c];
                    }
                }
            }
        } else if (wave == 0) {                                 // partial last panel: only the rhs row is below it
            double Lr[CH_NB];
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) Lr[k] = s_dg[k * CH_NB + r16];
#pragma unroll
            for (int k = 0; k < CH_NB; ++k) Lr[k] = k < r16 ? Lr[k] : 0.0;
            double bc = s_aug[j0 + min(r16, nb - 1)], yf = 0.0;
            bc = r16 < nb ? bc : 0.0;
            const double inv = s_dg[r16 * CH_NB + r16];
            ch_exec_settle(bc);
            ChFwd<0>::run(Lr, bc, inv, yf, r16);
            if (lane < nb) s_aug[j0 + lane] = yf;
        }
        __syncthreads();
        if (base >= D) break;
        // ---- trailing update S22 -= L21 L21^T: 16x16 tiles, operands from the LDS panel, C read-modify-written in L2
        // (rhs row: in s_aug).  Lane l holds A[l&15][l>>4], B[l>>4][l&15]; D: col = l&15, row = (l>>4) + 4 reg.
        // A wave takes its tiles in groups of CHG_TB: all C loads of the group are issued first (independent L2 round
        // trips in flight together), then the MFMAs, then the stores.
        const int T = (m + 15) >> 4, ntile = T * (T + 1) / 2;
        int tr = 0, tc = wave;
        while (tc > tr) { tc -= tr + 1; ++tr; }
        for (int t0 = wave; t0 < ntile; t0 += CHG_TB * (CH_THREADS / 64)) {
            double cv[CHG_TB][4];
            int trs[CHG_TB], tcs[CHG_TB];
#pragma unroll
            for (int g = 0; g < CHG_TB; ++g) {
                trs[g] = tr; tcs[g] = tc;
                const bool live = t0 + g * (CH_THREADS / 64) < ntile;
                const int col = 16 * tc + r16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rwq = 16 * tr + (lane >> 4) + 4 * q;
                    const int rc = min(rwq, m - 1), cc = min(col, rc);
                    // both sources are read (clamped addresses) and the VALUE is selected: a select between an LDS and a
                    // global pointer does not survive instruction selection
                    const int rg = live ? min(rc, m - 2) : 0, cg = live ? min(cc, rg) : 0;
                    const double vg = S[(size_t)(base + rg) * D + base + cg], va = s_aug[base + cc];
                    cv[g][q] = rc == m - 1 ? va : vg;
                }
                tc += CH_THREADS / 64;
                while (tc > tr) { tc -= tr + 1; ++tr; }
            }
#pragma unroll
            for (int g = 0; g < CHG_TB; ++g) {
                if (t0 + g * (CH_THREADS / 64) >= ntile) break;
                const int trg = trs[g], tcg = tcs[g];
                const int ra = min(16 * trg + r16, m - 1), rb = min(16 * tcg + r16, m - 1), kq = lane >> 4;
                const int col = 16 * tcg + r16;
                const double* pa = s_P + ra * CH_NB + kq;
                const double* pb = s_P + rb * CH_NB + kq;
                const double a0 = pa[0], a1 = pa[4], a2 = pa[8], a3 = pa[12], b0 = pb[0], b1 = pb[4], b2 = pb[8], b3 = pb[12];
                f64x4 acc = {0.0, 0.0, 0.0, 0.0};
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rwq = 16 * trg + (lane >> 4) + 4 * q;
                    if (!(rwq < m && col <= rwq)) continue;
                    const double v = cv[g][q] - acc[q];
                    if (rwq == m - 1) s_aug[base + col] = v; else S[(size_t)(base + rwq) * D + base + col] = v;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (s_ok) {
        for (int i = tid; i < D; i += CH_THREADS) s_b[i] = s_aug[i];          // y = L^-1 b
        __syncthreads();
        const int np = (D + CH_NB - 1) / CH_NB;
        for (int p = np - 1; p >= 0; --p) {                     // L^T x = y, panel by panel from the bottom
            const int j0 = CH_NB * p, nb = min(CH_NB, D - j0);
            if (wave == 0) {
                double col[CH_NB];                              // lane j: L[j0+k][j0+j], k > j
#pragma unroll
                for (int k = 0; k < CH_NB; ++k) col[k] = S[(size_t)(j0 + min(max(k, r16), nb - 1)) * D + j0 + min(r16, nb - 1)];
#pragma unroll
                for (int k = 0; k < CH_NB; ++k) col[k] = (r16 < nb && k < nb && k > r16) ? col[k] : 0.0;
                double y = s_b[j0 + min(r16, nb - 1)], inv = s_inv[j0 + min(r16, nb - 1)];
                y = r16 < nb ? y : 0.0; inv = r16 < nb ? inv : 0.0;
                double xf = 0.0;
                ch_exec_settle(y);
                ChBack<CH_NB - 1>::run(col, y, inv, xf, r16);
                if (lane < nb) s_b[j0 + lane] = xf;
            }
            __syncthreads();
            for (int r = tid; r < j0; r += CH_THREADS) {        // rows above: L[j0+k][r] is contiguous in r
                double s0 = 0, s1 = 0;
                for (int k = 0; k + 1 < nb; k += 2) {           // nb is even
                    s0 += S[(size_t)(j0 + k) * D + r] * s_b[j0 + k];
                    s1 += S[(size_t)(j0 + k + 1) * D + r] * s_b[j0 + k + 1];
                }
                s_b[r] -= s0 + s1;
            }
            __syncthreads();
        }
        for (int i = tid; i < D; i += CH_THREADS) B.bs[i] = s_b[i];
    }
    if (tid == 0) B.scal[3] = s_ok ? 1.0 : 0.0;
}

__device__ __forceinline__ void ba_backsub_body(const BaDev& B, double lambda, int blk, const double* pts_c, double* pts_t) {
    const int k = blk * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;       // 4 lanes per point, as in the linearisation
    double sc = 0, mx = 0;
    const bool live = k < B.n_points && B.scal[3] != 0.0;
    double rhs[3] = {0, 0, 0};
    if (live) {
        if (sub == 0) { rhs[0] = B.bl[3 * (size_t)k]; rhs[1] = B.bl[3 * (size_t)k + 1]; rhs[2] = B.bl[3 * (size_t)k + 2]; }
        const int q1 = B.pt_start[k + 1];
        for (int p1 = B.pt_start[k] + sub; p1 < q1; p1 += 4) {
            const int e1 = B.pt_edges[p1], j1 = B.e_pose[e1];
            if (!B.active[e1] || j1 >= B.n_free) continue;
            const double* W1 = B.W + 18 * (size_t)e1;
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 6; ++r) rhs[c] -= W1[3 * r + c] * B.bs[6 * j1 + r];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) rhs[c] = ba_quad_sum(rhs[c]);
    if (live && sub == 0) {
        const double* h = B.Hinv + 9 * (size_t)k;
        for (int a = 0; a < 3; ++a) {
            const double d = h[3 * a] * rhs[0] + h[3 * a + 1] * rhs[1] + h[3 * a + 2] * rhs[2];
            B.dl[3 * (size_t)k + a] = d;
            pts_t[3 * (size_t)k + a] = pts_c[3 * (size_t)k + a] + d;
            sc += d * (lambda * d + B.bl[3 * (size_t)k + a]);
            mx = fmax(mx, fabs(d));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sc += __shfl_xor(sc, o, 64); mx = fmax(mx, __shfl_xor(mx, o, 64)); }
    // one partial per workgroup, summed by the last workgroup of k_ba_chi_control: hundreds of f64 atomics on ONE
    // address serialise in L2 and used to outlast the kernel body
    __shared__ double s_u[8];
    if ((threadIdx.x & 63) == 0) { s_u[threadIdx.x >> 6] = sc; s_u[4 + (threadIdx.x >> 6)] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        B.partU[2 * blk] = (s_u[0] + s_u[1]) + (s_u[2] + s_u[3]);
        B.partU[2 * blk + 1] = fmax(fmax(s_u[4], s_u[5]), fmax(s_u[6], s_u[7]));
    }
}

__device__ __forceinline__ void ba_pose_body(const BaDev& B, double lambda, int blk, const double* poses_c, double* poses_t) {
    const int j = blk * blockDim.x + threadIdx.x;
    if (j >= B.n_poses) return;
    const double* T = poses_c + 12 * (size_t)j;
    double* Tn = poses_t + 12 * (size_t)j;
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
    double sc = 0, mx = 0;
    for (int a = 0; a < 6; ++a) { sc += d[a] * (lambda * d[a] + B.bp[6 * j + a]); mx = fmax(mx, fabs(d[a])); }
    atomicAdd(&B.scal[2], sc);
    atomicMax((unsigned long long*)&B.scal[7], (unsigned long long)__double_as_longlong(mx));
}

// trial state: new points (blocks [0, gp)) and new poses (blocks [gp, ...)), gain-ratio terms, max |step|
__global__ void k_ba_update(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished) return;
    const double lambda = ctl_->lambda;
    BA_STATE(B)
    const int gp = B.gp;
    if ((int)blockIdx.x < gp) ba_backsub_body(B, lambda, blockIdx.x, pts_c, pts_t);
    else if (((int)blockIdx.x - gp) * (int)blockDim.x < B.n_poses) ba_pose_body(B, lambda, blockIdx.x - gp, poses_c, poses_t);
}

// chi2 of the current (trial = 0 -> scal[5]) or trial (-> scal[1]) state
__global__ void k_ba_chi(BaBatch Q, int trial, int robust, int guard) {
    BA_PROBLEM(Q)
    if (guard && ctl_->finished) return;
    const BaCam cam = B.cam; const double delta = B.delta;
    BA_STATE(B)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0;
    if (e < B.n_edges && B.active[e]) {
        const double* P = (trial ? poses_t : poses_c) + 12 * (size_t)B.e_pose[e];
        const double* X = (trial ? pts_t : pts_c) + 3 * (size_t)B.e_pt[e];
        double r[2], pc[3];
        ba_err(cam, P, X, B.e_uv + 2 * (size_t)e, r, pc);
        const double e2 = r[0] * r[0] + r[1] * r[1];
        v = (robust && e2 > delta * delta) ? 2.0 * sqrt(e2) * delta - delta * delta : e2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(&B.scal[trial ? 1 : 5], v);
}

// g2o's gain-ratio test and lambda policy (OptimizationAlgorithmLevenberg::solve) on the control block: s1 = robust chi2 of the trial state, s2 = the
// gain ratio's denominator without its 1e-3, m7 = the largest step entry.  Returns whether the step is accepted (the caller clears H_pp / b_p then).
__device__ __forceinline__ int ba_lm_decide(const BaDev& B, BaCtl* c, double s1, double s2, double m7, bool ok) {
    const double tmp = ok ? s1 : DBL_MAX;
    const double scale = (ok ? s2 : 0.0) + 1e-3;
    const double rho = (c->cur - tmp) / scale;
    bool converged = false;
    int accept = 0;
    if (rho > 0 && isfinite(tmp)) {
        double a = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
        a = fmin(a, 2.0 / 3.0);
        c->lambda *= fmax(1.0 / 3.0, a); c->ni = 2; c->cur = tmp;
        c->buf ^= 1; c->need_lin = 1; accept = 1;       // trial state becomes the current state
        B.scal[0] = 0; B.scal[4] = 0;
    } else { c->lambda *= c->ni; c->ni *= 2; }
    if (ok) converged = m7 < 1e-10;
    c->qmax += 1; c->steps += 1;
    if (!(rho < 0 && c->qmax < 10 && !converged)) {     // this LM iteration is over
        c->iters_done += 1;
        if (c->qmax == 10 || rho == 0 || converged || c->it + 1 >= c->max_it) c->finished = 1;
        c->it += 1; c->qmax = 0;
    }
    return accept;
}

// Robust chi2 of the trial state; the LAST workgroup to finish runs g2o's gain-ratio test and lambda policy
// (OptimizationAlgorithmLevenberg::solve) on the control block and, when the step is accepted, clears
// H_pp / b_p for the next linearisation.  Partial sums reach L2 through f64 atomics; the arrival counter is
// taken after a device-scope fence (threadfence reduction).
__global__ __launch_bounds__(256) void k_ba_chi_control(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished) return;
    const int nblk_e = (B.n_edges + 1023) / 1024;               // this problem's share of the grid: 4 edges per lane (a quarter of the workgroups = a quarter of the arrival atomics)
    if ((int)blockIdx.x >= nblk_e) return;
    const BaCam cam = B.cam; const double delta = B.delta; const int robust = ctl_->robust;
    BA_STATE(B)
    __shared__ int s_last;
    double v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (e < B.n_edges && B.active[e]) {
            double r[2], pc[3];
            ba_err(cam, poses_t + 12 * (size_t)B.e_pose[e], pts_t + 3 * (size_t)B.e_pt[e], B.e_uv + 2 * (size_t)e, r, pc);
            const double e2 = r[0] * r[0] + r[1] * r[1];
            v += (robust && e2 > delta * delta) ? 2.0 * sqrt(e2) * delta - delta * delta : e2;
        }
    }
    v = vo_wave_sum_f64(v);
    __shared__ double s_w[12];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        B.partC[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);       // one partial per workgroup, no same-address atomics
        __threadfence();
        s_last = atomicAdd(&ctl_->arrived, 1) == nblk_e - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __shared__ int s_accept;
    {   // the last workgroup sums the partials of this kernel (trial chi2) and of k_ba_update (gain term, max step)
        __threadfence();
        const volatile double* pc = B.partC; const volatile double* pu = B.partU;
        double a = 0, b = 0, m = 0;
        for (int i = threadIdx.x; i < nblk_e; i += 256) a += pc[i];
        for (int i = threadIdx.x; i < B.nU; i += 256) { b += pu[2 * i]; m = fmax(m, pu[2 * i + 1]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); m = fmax(m, __shfl_xor(m, o, 64)); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; s_w[w] = a; s_w[4 + w] = b; s_w[8 + w] = m; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        BaCtl* c = ctl_;
        c->arrived = 0;
        const double s1 = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        // scal[2], [3], [7] were written by earlier kernels of this step (k_ba_update's pose part, the Cholesky kernel): plain loads,
        // issued together -- three dependent atomic round trips here were a third of this kernel's time
        const volatile double* sc = B.scal;
        const double sc2 = sc[2], s3 = sc[3], sc7 = sc[7];
        const double s2 = sc2 + ((s_w[4] + s_w[5]) + (s_w[6] + s_w[7]));
        const double m7 = fmax(sc7, fmax(fmax(s_w[8], s_w[9]), fmax(s_w[10], s_w[11])));
        if (B.shard_world > 1) {
            // a sharded BA (e-3): these are ONE rank's sums.  They go to the exchange region -- the pose part of the gain ratio's denominator, which every rank
            // computes identically, through rank 0 only; the largest step in this rank's slot -- and k_ba_shard_decide takes the decision behind the exchange
            double* x3 = B.xbuf + ba_x3_off(B.D, B.n_free, B.shard_world);
            x3[0] = s1; x3[1] = s2 - (B.shard_rank == 0 ? 0.0 : sc2);
            for (int r = 0; r < B.shard_world; ++r) x3[2 + r] = r == B.shard_rank ? m7 : 0.0;
            s_accept = 0;
        } else s_accept = ba_lm_decide(B, c, s1, s2, m7, s3 != 0.0);
    }
    __syncthreads();
    if (s_accept) {
        for (int i = threadIdx.x; i < 36 * B.n_free; i += 256) B.Hpp[i] = 0;
        for (int i = threadIdx.x; i < B.D; i += 256) B.bp[i] = 0;
    }
}

// ---- e-3: the local BA sharded over ranks by point (include/vo_hip.h, vo_set_ba_shard) -- the kernels at its three exchanges -------------------------
// after k_ba_lin: this rank's H_pp, b_p, robust chi2 and (first step of a round) largest diagonal entry -> region x1; a step that does not linearise
// (a rejected one: the sums in place are already the exchanged ones) sends zeros and keeps what it has
__global__ __launch_bounds__(256) void k_ba_shard_pack1(BaBatch Q) {
    BA_PROBLEM(Q)
    double* x1 = B.xbuf + ba_x1_off(B.D);
    const int n = (int)ba_x1_len(B.n_free, B.shard_world), nh = 36 * B.n_free, live = !ctl_->finished && ctl_->need_lin;
    for (int i = threadIdx.x; i < n; i += 256) {
        double v = 0.0;
        if (live) {
            if (i < nh) v = B.Hpp[i];
            else if (i < nh + B.D) v = B.bp[i - nh];
            else if (i == nh + B.D) v = B.scal[0];
            else if (i - (nh + B.D + 1) == B.shard_rank) v = ctl_->first ? B.scal[4] : 0.0;      // (k_ba_maxdiag ran on this rank's sums: its H_pp part is a lower bound of the full one's, see unpack1)
        }
        x1[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_ba_shard_unpack1(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished || !ctl_->need_lin) return;
    const double* x1 = B.xbuf + ba_x1_off(B.D);
    const int nh = 36 * B.n_free;
    for (int i = threadIdx.x; i < nh; i += 256) B.Hpp[i] = x1[i];
    for (int i = threadIdx.x; i < B.D; i += 256) B.bp[i] = x1[nh + i];
    __shared__ double s_m[4];
    double m = 0.0;
    if (ctl_->first) {                                       // lambda_0 = 1e-5 max diag(H): the summed H_pp's diagonal and every rank's point blocks
        for (int i = threadIdx.x; i < B.D; i += 256) m = fmax(m, fabs(x1[36 * (i / 6) + 7 * (i % 6)]));
        for (int r = threadIdx.x; r < B.shard_world; r += 256) m = fmax(m, x1[nh + B.D + 1 + r]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        B.scal[0] = x1[nh + B.D];
        if (ctl_->first) B.scal[4] = fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3]));
    }
}
// behind the third exchange: the decision k_ba_chi_control's last workgroup takes in an un-sharded solve, from the summed values -- on every rank the same
__global__ __launch_bounds__(256) void k_ba_shard_decide(BaBatch Q) {
    BA_PROBLEM(Q)
    if (ctl_->finished) return;
    __shared__ int s_accept;
    if (threadIdx.x == 0) {
        const double* x3 = B.xbuf + ba_x3_off(B.D, B.n_free, B.shard_world);
        double m7 = 0.0;
        for (int r = 0; r < B.shard_world; ++r) m7 = fmax(m7, x3[2 + r]);
        s_accept = ba_lm_decide(B, ctl_, x3[0], x3[1], m7, B.scal[3] != 0.0);
    }
    __syncthreads();
    if (s_accept) {
        for (int i = threadIdx.x; i < 36 * B.n_free; i += 256) B.Hpp[i] = 0;
        for (int i = threadIdx.x; i < B.D; i += 256) B.bp[i] = 0;
    }
}
// a rank's edges: those of its points (point k belongs to rank k % world); the others are inactive from the start
__global__ void k_ba_shard_mask(BaBatch Q) {
    BA_PROBLEM(Q)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < B.n_edges) B.active[e] = (B.e_pt[e] % B.shard_world) == B.shard_rank ? 1 : 0;
}
// the last exchange's buffer: positions of this rank's points, flags of its edges (others zero: the sum is the whole result), its shares of the two chi2 reports
__global__ void k_ba_shard_pack_final(BaBatch Q, double* __restrict__ xf) {
    BA_PROBLEM(Q)
    BA_STATE(B)
    const int i = blockIdx.x * blockDim.x + threadIdx.x, nx = B.n_points, ne = B.n_edges;
    if (i < 3 * nx) xf[i] = ((i / 3) % B.shard_world) == B.shard_rank ? pts_c[i] : 0.0;
    else if (i < 3 * nx + ne) { const int e = i - 3 * nx; xf[i] = (B.e_pt[e] % B.shard_world) == B.shard_rank ? (double)B.flags[e] : 0.0; }
    else if (i == 3 * nx + ne) xf[i] = B.scal[5];
    else if (i == 3 * nx + ne + 1) xf[i] = B.scal[6];
}

// stage 0: cull after the robust round (bit0, deactivate); stage 1: flag level-0 outliers (bit1)
// returns the edge's share of the final chi2 (stage 1, edges that stay): the caller sums a wavefront's shares before it touches scal[6] -- one atomic per
// EDGE on that one address was what the BA's last launch waited for (60-90 k same-address atomics: 25 us)
__device__ __forceinline__ double ba_cull_edge(const BaDev& B, const double* poses_c, const double* pts_c, int e, int stage) {
    double r[2], pc[3];
    ba_err(B.cam, poses_c + 12 * (size_t)B.e_pose[e], pts_c + 3 * (size_t)B.e_pt[e], B.e_uv + 2 * (size_t)e, r, pc);
    const double c2 = r[0] * r[0] + r[1] * r[1], th = B.chi2_th;
    if (stage == 0) { if (c2 > th) { B.flags[e] = 1; B.active[e] = 0; } else B.flags[e] = 0; }
    else if (B.active[e]) { if (c2 > th) B.flags[e] |= 2; else return c2; }
    return 0.0;
}
// A problem enters its slot: control block of the robust round, zeroed accumulators (one workgroup; the plain chi2 of the
// initial state follows in k_ba_chi).  The descriptor (BaDev) is the only thing the host uploads.
// the descriptor travels as a kernel argument (a copy would be a blit kernel with its own launch gap)
// (one lane, a plain struct assignment: the argument is then read with scalar loads through the constant cache -- per-lane vector loads
// from the kernel-argument segment took 14 us for these 500 bytes)
__global__ void k_ba_put_desc(BaDev v, BaDev* __restrict__ dst) {
    if (threadIdx.x == 0) *dst = v;
}
// (one launch: the descriptor arrives as a kernel argument, thread 0 stores it where the step kernels read it, everybody works from the argument)
__global__ __launch_bounds__(256) void k_ba_admit(BaDev B, BaDev* __restrict__ dst, BaCtl* ctl_) {
    if (threadIdx.x == 0) *dst = B;
    if (threadIdx.x == 0) {
        BaCtl c;
        memset(&c, 0, sizeof(c));
        c.max_it = B.it_robust; c.max_it_next = B.it_plain; c.need_lin = 1; c.first = 1; c.ni = 2; c.robust = 1; c.finished = B.it_robust <= 0 ? 1 : 0; c.gen = B.gen;
        *ctl_ = c;
    }
    if (threadIdx.x < 8) B.scal[threadIdx.x] = 0;
    if (threadIdx.x == 8 && B.ncull) *B.ncull = 0;
    for (int i = threadIdx.x; i < 36 * B.n_free; i += 256) B.Hpp[i] = 0;
    for (int i = threadIdx.x; i < B.D; i += 256) B.bp[i] = 0;
}

// End of a chunk of LM steps: a slot whose round has ended is culled (backend.cpp:144-156 after the robust round, :162-172 after
// the plain one) and moved on -- to the plain round (control block reset, accumulators zeroed: the next launches linearise it
// afresh) or to "done" -- by the last of its workgroups; every slot's state goes to the host's status record.  The host
// neither waits for a round to end nor enqueues anything in between: it reads the records when the chunk's event has passed.
__global__ __launch_bounds__(256) void k_ba_round(BaBatch Q) {
    BA_PROBLEM(Q)
    BaStat* const st = Q.stat + Q.slot[blockIdx.z];
    const int stage = ctl_->stage;
    if (stage >= 2) return;
    if (!ctl_->finished) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->stage = stage; st->finished = 0; st->it = ctl_->it; st->steps = ctl_->steps; st->gen = ctl_->gen; }
        return;
    }
    const int nblk = (B.n_edges + 255) / 256;
    if ((int)blockIdx.x >= nblk) return;
    BA_STATE(B)
    const int e = blockIdx.x * 256 + threadIdx.x;
    double chi_keep = 0.0;
    if (e < B.n_edges) {
        chi_keep = ba_cull_edge(B, poses_c, pts_c, e, stage);
        if (stage == 1 && B.e_obs && (B.flags[e] & 3)) {    // culled by either test: its observation id joins the list (agent-scope store: read by another workgroup / the merge kernel)
            const int pos = __hip_atomic_fetch_add(B.ncull, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pos < B.cull_cap) __hip_atomic_store(B.cull + pos, B.e_obs[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __shared__ int s_last;
    __shared__ double s_keep[4];
    if (stage == 1) {                                        // (workgroup-uniform)
        // the kept edges' chi2: one partial per workgroup (B.partC), added up by the last one -- as ~1200 same-address atomics (one per wave) the sum was ~10 of
        // the final round's 23 us, and its value depended on the order the atomics arrived in
        chi_keep = vo_wave_sum_f64(chi_keep);
        if ((threadIdx.x & 63) == 0) s_keep[threadIdx.x >> 6] = chi_keep;
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(B.partC + blockIdx.x, (s_keep[0] + s_keep[1]) + (s_keep[2] + s_keep[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores (the partial, the culled list's entries) have been performed before the barrier (a workgroup-scope barrier alone need not wait for them)
    __syncthreads();
    // (no fences around the ticket: the last workgroup reads nothing the others wrote except the partials and the culled list, which leave as agent-scope
    // stores performed before the barrier above lets thread 0 take the ticket; a device-scope fence here would write this XCD's L2 back)
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&ctl_->ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
    __syncthreads();
    if (!s_last) return;
    if (stage == 1) {
        double t = 0.0;
        for (int i = threadIdx.x; i < nblk; i += 256) t += __hip_atomic_load(B.partC + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = vo_wave_sum_f64(t);
        __syncthreads();                                    // (s_keep's first use has been read)
        if ((threadIdx.x & 63) == 0) s_keep[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&B.scal[6], (s_keep[0] + s_keep[1]) + (s_keep[2] + s_keep[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stage == 0) {
        if (threadIdx.x < 8 && threadIdx.x != 5) B.scal[threadIdx.x] = 0;       // [5]: the initial chi2 (k_ba_chi at admission)
        for (int i = threadIdx.x; i < 36 * B.n_free; i += 256) B.Hpp[i] = 0;
        for (int i = threadIdx.x; i < B.D; i += 256) B.bp[i] = 0;
    }
    if (stage == 1 && B.ncull && B.cull_host) {
        const int n = min(min(__hip_atomic_load(B.ncull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), B.cull_cap), B.cull_host_cap);
        for (int i = threadIdx.x; i < n; i += 256) B.cull_host[i] = __hip_atomic_load(B.cull + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        BaCtl* c = ctl_;
        c->ticket = 0;
        c->iters_total += c->iters_done;
        if (stage == 0) {
            c->chi0 = ((const volatile double*)B.scal)[5];
            c->lambda = 0; c->ni = 2; c->cur = 0; c->it = 0; c->qmax = 0; c->max_it = c->max_it_next; c->need_lin = 1; c->first = 1; c->iters_done = 0; c->steps = 0;
            c->arrived = 0; c->robust = 0; c->lbuf = 0; c->finished = c->max_it_next <= 0 ? 1 : 0;
            c->stage = 1;
            st->stage = 1; st->finished = c->finished; st->it = 0; st->steps = 0; st->gen = c->gen;
        } else {
            c->stage = 2;
            st->n_culled = B.ncull ? __hip_atomic_load(B.ncull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            st->chi0 = c->chi0; st->chi_final = __hip_atomic_load(&B.scal[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            st->n_pairs = B.n_slices ? B.n_slices[1] : 0;      // (the device-built pair plan's count: byte / flop accounting of the Schur launches)
            st->buf = c->buf; st->iters_total = c->iters_total; st->finished = 1; st->it = c->it; st->steps = c->steps; st->gen = c->gen;
            __threadfence_system();
            st->stage = 2;
        }
    }
}

#include "vo_ba_phase2.h"

// The Cholesky and the update in ONE launch (tile-major systems): workgroups 0 .. n-1 are the solvers of the launch's n problems,
// the rest are k_ba_upchi2's (gpmax per problem, problem-major), which request everything that does not depend on the solution
// and then wait for their solver's word (ctl->chol_seq).  Workgroups are dispatched in order and every wait is for a workgroup
// with a lower number, so the wait cannot starve what it waits for; it is bounded all the same.  A step is then TWO launches.
__global__ __launch_bounds__(CH2_T) void k_ba_cholup(BaBatch Q, int n, int gpmax, int rep) {
    static_assert(CH2_T == UPC_T, "one block size for both roles");
    const int b = blockIdx.x, z = b < n ? b : (b - n) / gpmax;
    const BaDev B_copy_ = Q.Bs[Q.slot[z]]; const BaDev& B = B_copy_; BaCtl* const ctl_ = Q.ctls + Q.slot[z];
    if (ctl_->finished) return;
    if (b < n) {
        extern __shared__ double s_mem[];
        ba_chol16v2_body<true>(B, ctl_, s_mem, B.dl, true);
    } else {
        ba_upchi2_body<true>(B, ctl_, rep, (b - n) % gpmax, z);
    }
}

// the same for a lone problem, its descriptor in the kernel's arguments (see k_ba_schur2_one)
__global__ __launch_bounds__(CH2_T) void k_ba_cholup_one(BaDev B, BaCtl* ctl_, int gpmax, int rep) {
    if (blockIdx.x == 0) {
        extern __shared__ double s_mem[];
        ba_chol16v2_body<true>(B, ctl_, s_mem, B.dl, true, true);      // (ctl->finished comes with the solver's head load)
    } else {
        if (ctl_->finished) return;
        ba_upchi2_body<true>(B, ctl_, rep, ((int)blockIdx.x - 1) % gpmax, 0);
    }
}

// per-device function attributes (vo_ctx_create calls this with the context's device current)
int vo_ba_set_attrs() {
    HIP_TRY(hipFuncSetAttribute((const void*)k_ba_chol16, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)k_ba_chol16g, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)k_ba_chol16v2, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)k_ba_cholup, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)k_ba_cholup_one, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)k_ba_upchi2, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return VO_OK;
}

// ---- the BA engine: continuous batching of local BAs ------------------------------------------------------------------
// One engine per GPU and process.  A caller (vo_local_ba on any context of that GPU: the back-end workers of several
// streams) prepares its problem on its own stream and hands it to the engine.  Whichever caller finds the engine without
// a driver drives it (enqueues the steps, polls the control blocks) until its own problem is done, then passes the wheel
// to one of the callers still waiting -- no dedicated thread, no hand-off latency for a lone stream.  The engine keeps up
// to BA_SLOTS problems "in flight": every LM step is ONE sequence of launches over all active slots (blockIdx.z = slot), a
// problem that arrives while others are being solved joins at the next chunk of steps, one that finishes its robust
// round is culled and restarted for the plain round, one that finishes leaves -- none waits for the others.  So eight
// streams' BAs cost about the latency of one (their kernels are small: a 6K x 6K Cholesky is one workgroup), where eight
// HIP streams with ~120 tiny dependent launches each mostly serialise in the command processor.
#include <condition_variable>
#include <deque>
// which layout S has (and which Cholesky kernel a problem takes): tiles + the second generation where ch2_fits says so (D <= 174, 177 .. 191), packed rows + the first otherwise
// points per update workgroup of the fused launch: as few as keeps the launch (solver + point workgroups + pose workgroup) within the chip's 256 compute units,
// whole wavefronts (16 points), at least 64 (the partial-sum arrays are sized for 64-point workgroups)
static int ba_upc_ppw(int nx) { const int p = ((nx + 249) / 250 + 15) / 16 * 16; return std::min(UPC_T / 4, std::max(64, p)); }
static int ba_use_tiles(int D) { return ch2_fits(D) ? 1 : 0; }      // (D <= 174 whole in LDS; 177 .. 191 with the last row block in global memory: vo_ba_chol2.h)
// Host waits on this latency chain poll instead of sleeping: a blocking wait costs the wake-up of a sleeping thread (10-40 us) per hand-off, and a
// local BA has six of them.  vo_spin_event: hipEventSynchronize by polling (bounded: falls back to the blocking call after ~2 ms);
// vo_spin_word: a word in pinned host memory that a kernel stores behind its results (system-scope fence in the kernel).
static inline void vo_cpu_relax() { __builtin_ia32_pause(); }
static hipError_t vo_spin_event(hipEvent_t ev) {
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0;; ++i) {
            const hipError_t e = hipEventQuery(ev);
            if (e != hipErrorNotReady) return e;
            for (int k = 0; k < 8; ++k) vo_cpu_relax();
            if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    return hipEventSynchronize(ev);
}
static bool vo_spin_word(const volatile int* w, int want, int timeout_ms) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0;; ++i) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == want) return true;
        for (int k = 0; k < 4; ++k) vo_cpu_relax();
        if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) return false;
    }
}
static double tnow() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct BaJob {
    vo_ctx* c = nullptr; const vo_ba_problem* in = nullptr; vo_ba_result* out = nullptr;
    BaDev B;
    int grid_lin = 0, grid_initS = 0, grid_upd = 0, grid_e = 0, grid_c = 0, grid_maxdiag = 0; size_t lds = 0;
    int cur_buf = 0, iters = 0, steps = 0;
    int est_stage = 0, est_left = 0;                        // the host's estimate of where the device is (chunk sizes only: the device moves on by itself)
    double chi0 = 0, chi_final = 0; int n_pairs = 0;
    int n_culled = -1; std::vector<long long> culled;       // resident graphs: from k_ba_round's pinned list (-1: not collected)
    int rc = VO_OK; bool done = false;
    hipEvent_t wait_ev = nullptr;                           // the problem's arrays are complete once this event (recorded on the owner's stream) has passed
    hipEvent_t wait_pairs = nullptr;                        // optional: the pair lists are complete once this one has passed (waited for in front of the first Schur launch only)
};
// one enqueued chunk of LM steps: the slots it covers (with their admission numbers) and two events -- before its last step and behind it
struct BaChunk { hipEvent_t ev_near = nullptr, ev_end = nullptr; int n = 0, steps = 0; int sl[BA_SLOTS]; int gen[BA_SLOTS]; };
#define BA_MAX_ENGINES 4
#define BA_CULL_HOST 4096
struct BaEngine {
    int device = 0, refs = 0;
    hipStream_t st = nullptr;
    BaDev* d_Bs = nullptr; BaDev* h_Bs = nullptr;           // [BA_SLOTS] problem descriptors: device / pinned mirror
    BaCtl* d_ctl = nullptr;                                 // [BA_SLOTS] control blocks (device only: k_ba_admit / k_ba_round write them)
    BaStat* h_stat = nullptr;                               // [BA_SLOTS] pinned status records, written by k_ba_round
    long long* h_cull = nullptr;                            // [BA_SLOTS][BA_CULL_HOST] pinned: culled observation ids of a resident graph
    BaJob* slot[BA_SLOTS] = {};
    int slot_gen[BA_SLOTS] = {}; int gen_ctr = 0;
    BaChunk ring[2]; int r_head = 0, r_n = 0;               // chunks in flight (oldest first)
    std::mutex mu; std::condition_variable cv;
    std::deque<BaJob*> pending;
    bool driving = false;                                   // a caller is inside ba_engine_pump
    long long n_steps = 0, n_slot_steps = 0, n_jobs = 0;
    int pending_hint = 0;                                   // queue length seen by the last admission (chunk size policy)
    BaEngine* sib[BA_MAX_ENGINES] = {}; int n_sib = 1, rr = 0;   // engine 0 of a device: its engines (sib[0] = itself) and the rotation
};
static void ba_engine_free(BaEngine* E);
static std::mutex g_eng_mu;
static std::vector<BaEngine*> g_engines;

static BaBatch ba_batch_of(BaEngine* E, const int* slots, int n) {
    BaBatch Q; Q.Bs = E->d_Bs; Q.ctls = E->d_ctl; Q.stat = E->h_stat; Q.n = n;
    for (int i = 0; i < BA_SLOTS; ++i) Q.slot[i] = i < n ? slots[i] : 0;
    return Q;
}

// admit queued problems into free slots (stream order: behind everything enqueued so far).  The descriptor is the only upload:
// k_ba_admit writes the control block of the robust round (backend.cpp:140-141) and clears the accumulators; the plain chi2 of the
// initial state (scal[5], reporting only) comes out of the first linearisation (k_ba_lin2; k_ba_chi for the larger systems)
static int ba_engine_admit(BaEngine* E) {
    hipStream_t st = E->st;
    std::unique_lock<std::mutex> lk(E->mu);
    for (int s = 0; s < BA_SLOTS && !E->pending.empty(); ++s) {
        if (E->slot[s]) continue;
        BaJob* j = E->pending.front(); E->pending.pop_front();
        E->slot[s] = j; j->B.ctl = E->d_ctl + s; j->cur_buf = 0; j->iters = 0; j->steps = 0;
        j->B.it_robust = j->in->it_robust; j->B.it_plain = j->in->it_plain; j->B.gen = E->slot_gen[s] = ++E->gen_ctr;
        j->B.cull_host = j->B.e_obs ? E->h_cull + (size_t)s * BA_CULL_HOST : nullptr; j->B.cull_host_cap = BA_CULL_HOST; j->n_culled = -1;
        j->est_stage = 0; j->est_left = std::max(1, j->in->it_robust);
        lk.unlock();
        int rc = VO_OK;
        if (j->wait_ev && hipStreamWaitEvent(st, j->wait_ev, 0) != hipSuccess) rc = VO_E_DEVICE;
        E->h_Bs[s] = j->B;                                  // the slot's mirror is free: its previous problem is gone
        static_assert(sizeof(BaDev) % 4 == 0 && sizeof(BaDev) <= 2048, "BaDev travels as a kernel argument");
        if (rc == VO_OK) {
            const BaBatch Q = ba_batch_of(E, &s, 1);
            hipLaunchKernelGGL(k_ba_admit, dim3(1, 1, 1), dim3(256), 0, st, j->B, E->d_Bs + s, E->d_ctl + s);
            if (j->B.D > BA_FOLD_D) hipLaunchKernelGGL(k_ba_chi, dim3(j->grid_e, 1, 1), dim3(256), 0, st, Q, 0, 0, 0);      // (D <= 192: k_ba_lin2 sums it in passing)
        }
        lk.lock();
        if (rc != VO_OK) { j->rc = rc; j->done = true; E->slot[s] = nullptr; E->cv.notify_all(); }
        else ++E->n_jobs;
    }
    E->pending_hint = (int)E->pending.size();
    return VO_OK;
}

static bool ba_engine_wants_steps(const BaEngine* E) {
    for (int s = 0; s < BA_SLOTS; ++s) if (E->slot[s] && E->slot[s]->est_stage < 2) return true;
    return false;
}

// One chunk of LM steps over every active slot, enqueued without waiting for anything: [first linearisation of a round, for the
// slots that are at one] K x [Schur, Cholesky, update + chi2 + control] [k_ba_round: round transitions, status records].  Systems the
// LDS-resident Cholesky solves (D <= 192) take the second-generation phases (vo_ba_phase2.h: three launches per step), larger
// ones the first generation (its Schur kernel writes the full matrix k_ba_chol16g reads).
static int ba_engine_enqueue(BaEngine* E) {
    hipStream_t st = E->st;
    int act[BA_SLOTS], na = 0;
    for (int s = 0; s < BA_SLOTS; ++s) if (E->slot[s]) act[na++] = s;
    if (na == 0 || E->r_n >= 2) return VO_OK;
    // chunk length: to the end of the nearest round (a slot that ends its round idles through the rest of the chunk), short while
    // other problems may want to join
    int chunk = (na == 1 && E->pending_hint == 0) ? 16 : 4;
    int sA[BA_SLOTS], nA = 0, sB[BA_SLOTS], nB = 0, nA_tiles = 0;
    int gA_lin = 0, gA_blk = 0, gA_up = 0, gA_md = 0, gA_pose = 0, gA_np = 0, gA_pts = 0, gB_lin = 0, gB_init = 0, gB_blk = 0, gB_upd = 0, gB_c = 0, gB_md = 0, g_e = 0;
    size_t ldsA = 0, ldsA_up = 0, ldsA_up_plain = 0, ldsB = 0;
    for (int i = 0; i < na; ++i) {
        BaJob* j = E->slot[act[i]];
        if (j->est_stage < 2) chunk = std::min(chunk, std::max(1, j->est_left));
        g_e = std::max(g_e, j->grid_e);
        if (j->B.D <= BA_FOLD_D) {
            sA[nA++] = act[i]; nA_tiles += j->B.s_tiles;
            gA_lin = std::max(gA_lin, j->grid_lin); gA_blk = std::max(gA_blk, j->B.n_blocks); gA_pose = std::max(gA_pose, j->B.n_free); gA_pts = std::max(gA_pts, (j->B.n_points + 63) / 64); gA_up = std::max(gA_up, j->B.n_points);
            gA_np = std::max(gA_np, std::min(j->B.n_poses, 512));      // k_ba_lin2 stages up to 512 poses (48 KB) in LDS
            ldsA = std::max(ldsA, j->lds); ldsA_up = std::max(ldsA_up, sizeof(double) * (24 * (size_t)j->B.n_poses + (size_t)j->B.D + (j->B.upc_ovf ? UPC_LDS_EXTRA : 0)));      // (the overflow region: used by the fused launch only)
            ldsA_up_plain = std::max(ldsA_up_plain, sizeof(double) * (24 * (size_t)j->B.n_poses + (size_t)j->B.D));
            gA_md = std::max(gA_md, (j->B.D + j->B.n_points + 255) / 256);
        } else {
            sB[nB++] = act[i];
            gB_lin = std::max(gB_lin, j->grid_lin); gB_init = std::max(gB_init, j->grid_initS); gB_blk = std::max(gB_blk, j->B.n_blocks);
            gB_upd = std::max(gB_upd, j->grid_upd); gB_c = std::max(gB_c, j->grid_c); gB_md = std::max(gB_md, j->grid_maxdiag);
            ldsB = std::max(ldsB, j->lds);
        }
    }
    const int ps_A = nA == 1 ? PSPLIT_LONE : PSPLIT;         // workgroups per free pose's list (see PSPLIT_LONE)
    gA_pose *= ps_A; gA_lin = gA_pts + gA_pose;
    const int up_rep = nA >= 2 ? 2 : 1;                     // points per workgroup of k_ba_upchi2: 128 x up_rep (vo_ba_phase2.h; 8 problems per launch: 45.7 / 40.7 / 41.0 / 88 us for 1 / 2 / 4 / 8)
    // problems per fused launch at most (VO_BA_FUSE_MAX; 0: never -- the three-launch step; read per chunk, so that bench.py's per-kernel
    // timing pass can take the solver's launch apart).  Default 1: with several problems the waiting update workgroups hold compute
    // units other streams want (8 / 16 streams: 3540 / 4530 frames/s fused, 3740 / 4760 not)
    const char* const fm_env = getenv("VO_BA_FUSE_MAX");
    const int fuse_max = fm_env ? atoi(fm_env) : 1;
    const bool fuse_up = nA >= 1 && nA <= fuse_max && nA_tiles == nA && std::max(ldsA, ldsA_up) <= 150 * 1024;
    BaChunk& C = E->ring[(E->r_head + E->r_n) % 2];
    if (!C.ev_end) { HIP_TRY(hipEventCreateWithFlags(&C.ev_near, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&C.ev_end, hipEventDisableTiming)); }
    C.n = na; C.steps = chunk;
    for (int i = 0; i < na; ++i) { C.sl[i] = act[i]; C.gen[i] = E->slot_gen[act[i]]; }
    vo_ctx* prof = E->slot[act[0]]->c;
    const BaBatch QA = ba_batch_of(E, sA, nA), QB = ba_batch_of(E, sB, nB), QE = ba_batch_of(E, act, na);
    const dim3 blk(256);
    if (chunk < 2) HIP_TRY(hipEventRecord(C.ev_near, st));
    for (int sidx = 0; sidx < chunk; ++sidx) {
        if (chunk >= 2 && sidx == chunk - 1) HIP_TRY(hipEventRecord(C.ev_near, st));
        if (nA) {
            // the first step of a round linearises in a launch of its own (lambda_0 needs the largest diagonal entry first; both kernels
            // leave at once for a slot that is not at the start of a round); afterwards the linearisation at the accepted state is a
            // by-product of k_ba_upchi2 and a step is THREE launches
            if (sidx == 0) {
                { ProfScope ps(prof, "k_ba_lin2", st); hipLaunchKernelGGL(k_ba_lin2, dim3(gA_lin, 1, nA), blk, 96 * (size_t)gA_np, st, QA, gA_np, ps_A); }      // (+ the largest diagonal entry: its last workgroup)
                for (int i = 0; i < na; ++i) {              // a problem's pair plan may still be running on its owner's stream: the linearisation above did not need it
                    BaJob* j = E->slot[act[i]];
                    if (j->wait_pairs) { HIP_TRY(hipStreamWaitEvent(st, j->wait_pairs, 0)); j->wait_pairs = nullptr; }
                }
            }
            const bool direct = nA == 1;                               // a lone problem: descriptor by value
            if (direct) { ProfScope ps(prof, "k_ba_schur2_one", st); hipLaunchKernelGGL(k_ba_schur2_one, dim3(gA_blk + gA_pose), blk, 0, st, E->h_Bs[sA[0]], E->d_ctl + sA[0]); }
            else { ProfScope ps(prof, "k_ba_schur2", st); hipLaunchKernelGGL(k_ba_schur2, dim3(gA_blk + gA_pose, 1, nA), blk, 0, st, QA); }
            // both generations in one step: every problem leaves the kernel that is not its own at once (s_tiles says which one is)
            // tile-major problems only: solvers and updates in one launch (vo_ba_phase2.h, FUSED)
            if (fuse_up) {
                ProfScope ps(prof, direct ? "k_ba_cholup_one" : "k_ba_cholup", st);
                const int ppw = (direct && E->h_Bs[sA[0]].upc_ppw > 0) ? E->h_Bs[sA[0]].upc_ppw : UPC_T / 4;
                const int gpmax = (gA_up + up_rep * ppw - 1) / (up_rep * ppw) + 1;      // (+ 1: the pose workgroup behind a problem's point workgroups)
                if (direct) hipLaunchKernelGGL(k_ba_cholup_one, dim3(1 + gpmax), dim3(CH2_T), std::max(ldsA, ldsA_up), st, E->h_Bs[sA[0]], E->d_ctl + sA[0], gpmax, up_rep);
                else hipLaunchKernelGGL(k_ba_cholup, dim3(nA * (1 + gpmax)), dim3(CH2_T), std::max(ldsA, ldsA_up), st, QA, nA, gpmax, up_rep);
            } else {
            if (nA_tiles) { ProfScope ps(prof, "k_ba_chol16v2", st); hipLaunchKernelGGL(k_ba_chol16v2, dim3(1, 1, nA), dim3(CH2_T), ldsA, st, QA); }
            if (nA_tiles < nA) { ProfScope ps(prof, "k_ba_chol16", st); hipLaunchKernelGGL(k_ba_chol16, dim3(1, 1, nA), dim3(CH_THREADS), ldsA, st, QA, 0, 1); }
            { ProfScope ps(prof, "k_ba_upchi2", st); hipLaunchKernelGGL(k_ba_upchi2, dim3((gA_up + up_rep * (UPC_T / 4) - 1) / (up_rep * (UPC_T / 4)) + 1, 1, nA), dim3(UPC_T), ldsA_up_plain, st, QA, up_rep); }
            }
        }
        if (nB) {
            if (sidx == 0) for (int i = 0; i < nB; ++i) { BaJob* j = E->slot[sB[i]]; if (j->wait_pairs) { HIP_TRY(hipStreamWaitEvent(st, j->wait_pairs, 0)); j->wait_pairs = nullptr; } }
            { ProfScope ps(prof, "k_ba_lin", st); hipLaunchKernelGGL(k_ba_lin, dim3(gB_lin, 1, nB), blk, 0, st, QB); }
            if (sidx == 0) hipLaunchKernelGGL(k_ba_maxdiag, dim3(gB_md, 1, nB), blk, 0, st, QB);
            { ProfScope ps(prof, "k_ba_init_S", st); hipLaunchKernelGGL(k_ba_init_S, dim3(gB_init, 1, nB), blk, 0, st, QB); }
            if (gB_blk) { ProfScope ps(prof, "k_ba_schur_blocks", st); hipLaunchKernelGGL(k_ba_schur_blocks, dim3(gB_blk, 1, nB), blk, 0, st, QB); }
            { ProfScope ps(prof, "k_ba_chol16g", st); hipLaunchKernelGGL(k_ba_chol16g, dim3(1, 1, nB), dim3(CH_THREADS), ldsB, st, QB); }
            { ProfScope ps(prof, "k_ba_update", st); hipLaunchKernelGGL(k_ba_update, dim3(gB_upd, 1, nB), blk, 0, st, QB); }
            { ProfScope ps(prof, "k_ba_chi_control", st); hipLaunchKernelGGL(k_ba_chi_control, dim3(gB_c, 1, nB), blk, 0, st, QB); }
        }
    }
    { ProfScope ps(prof, "k_ba_round", st); hipLaunchKernelGGL(k_ba_round, dim3(g_e, 1, na), blk, 0, st, QE); }
    HIP_TRY(hipEventRecord(C.ev_end, st));
    HIP_TRY(hipGetLastError());
    ++E->r_n;
    E->n_steps += chunk; E->n_slot_steps += (long long)chunk * na;
    for (int i = 0; i < na; ++i) {                          // where the device will be behind this chunk if every step is accepted
        BaJob* j = E->slot[act[i]];
        j->steps += chunk;
        if (j->est_stage >= 2) continue;
        j->est_left -= chunk;
        if (j->est_left <= 0) {
            j->est_stage += 1; j->est_left = std::max(1, j->in->it_plain);
            if (j->est_stage == 1 && j->in->it_plain <= 0) j->est_left = 1;      // the plain round ends at once; its transition comes with the next chunk
        }
    }
    return VO_OK;
}

// the oldest chunk has passed: completions, and -- when nothing newer is in flight -- the estimates are set from the device's state
static int ba_engine_retire(BaEngine* E) {
    BaChunk& C = E->ring[E->r_head];
    {   // The chunk is over when its event has passed -- or, sooner, when every problem in it has reported "done": k_ba_round stores a slot's final
        // status record in pinned memory behind a system-scope fence, and for such a slot the rest of the chunk is empty launches.  (The event becomes
        // visible to the host tens of microseconds after the kernel has ended; the record within a few.)
        bool fin = false;
        if (C.n > 0) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0;; ++i) {
                fin = true;
                for (int k = 0; k < C.n && fin; ++k) {
                    const volatile BaStat* t = E->h_stat + C.sl[k];
                    fin = __atomic_load_n(&t->stage, __ATOMIC_ACQUIRE) == 2 && t->gen == C.gen[k];
                }
                if (fin) break;
                if ((i & 7) == 7) { const hipError_t e = hipEventQuery(C.ev_end); if (e == hipSuccess) break; if (e != hipErrorNotReady) HIP_TRY(e); }
                for (int k = 0; k < 8; ++k) vo_cpu_relax();
                if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) break;
            }
        }
        if (!fin) HIP_TRY(vo_spin_event(C.ev_end));
    }
    E->r_head = (E->r_head + 1) % 2; --E->r_n;
    int fin[BA_SLOTS], nfin = 0, stuck = 0;
    for (int i = 0; i < C.n; ++i) {
        const int s = C.sl[i];
        BaJob* j = E->slot[s];
        if (!j || E->slot_gen[s] != C.gen[i]) continue;     // the slot has changed hands since
        const volatile BaStat* t = E->h_stat + s;
        if (t->gen != C.gen[i]) continue;
        if (t->stage == 2) {
            j->iters = t->iters_total; j->cur_buf = t->buf; j->chi0 = t->chi0; j->chi_final = t->chi_final; j->n_pairs = t->n_pairs;
            if (j->B.cull_host) { j->n_culled = t->n_culled; const int take = std::max(0, std::min(j->n_culled, BA_CULL_HOST)); j->culled.assign(E->h_cull + (size_t)s * BA_CULL_HOST, E->h_cull + (size_t)s * BA_CULL_HOST + take); }
            fin[nfin++] = s;
            continue;
        }
        if (E->r_n == 0) {
            const int max_it = t->stage == 0 ? j->in->it_robust : j->in->it_plain;
            j->est_stage = t->stage; j->est_left = t->finished ? 1 : std::max(1, max_it - t->it);
        }
        if (j->steps > (j->in->it_robust + j->in->it_plain) * 10 + 64) { j->rc = VO_E_DEVICE; fin[nfin++] = s; stuck = 1; }      // the control block never reported the end
    }
    if (nfin) {
        if (stuck) (void)hipStreamSynchronize(E->st);      // nothing enqueued may touch a slab its owner is about to reuse
        std::unique_lock<std::mutex> lk(E->mu);
        for (int i = 0; i < nfin; ++i) { E->slot[fin[i]]->done = true; E->slot[fin[i]] = nullptr; }
        E->cv.notify_all();
    }
    return VO_OK;
}

// engine thread (whichever caller drives): one turn = keep a chunk enqueued ahead of the one in flight, retire the oldest
static int ba_engine_pump(BaEngine* E) {
    int rc = ba_engine_admit(E);
    if (rc) return rc;
    if (E->r_n == 0) {
        int na = 0;
        for (int s = 0; s < BA_SLOTS; ++s) if (E->slot[s]) { ++na; if (E->slot[s]->est_stage >= 2) { E->slot[s]->est_stage = 1; E->slot[s]->est_left = 1; } }   // not done after all: keep stepping
        if (na == 0) return VO_OK;
        return ba_engine_enqueue(E);
    }
    if (E->r_n == 1 && ba_engine_wants_steps(E)) {
        // With several problems in flight the next chunk is put together as late as possible -- when the one in flight starts its last
        // step -- so that a problem that arrives meanwhile joins after at most one chunk; a lone problem is enqueued ahead at once.
        int na = 0;
        for (int s = 0; s < BA_SLOTS; ++s) if (E->slot[s]) ++na;
        if (na > 1 || E->pending_hint > 0) {
            HIP_TRY(vo_spin_event(E->ring[E->r_head].ev_near));
            if ((rc = ba_engine_admit(E))) return rc;
        }
        if ((rc = ba_engine_enqueue(E))) return rc;
    }
    return ba_engine_retire(E);
}

static int ba_engine_solve(BaEngine* E, BaJob* j) {
    std::unique_lock<std::mutex> lk(E->mu);
    E->pending.push_back(j);
    while (!j->done) {
        if (E->driving) { E->cv.wait(lk); continue; }
        E->driving = true;                                  // this caller drives the engine until its own problem is done
        while (!j->done) {
            lk.unlock();
            const int rc = ba_engine_pump(E);
            lk.lock();
            if (rc != VO_OK) {                              // a HIP error: fail everything in flight -- once the engine's stream has drained (the owners may free their slabs at once)
                (void)hipStreamSynchronize(E->st);
                E->r_n = 0; E->r_head = 0;
                for (int s = 0; s < BA_SLOTS; ++s) if (E->slot[s]) { E->slot[s]->rc = rc; E->slot[s]->done = true; E->slot[s] = nullptr; }
                while (!E->pending.empty()) { E->pending.front()->rc = rc; E->pending.front()->done = true; E->pending.pop_front(); }
            }
        }
        E->driving = false;
        E->cv.notify_all();                                 // a caller whose problem is still in flight takes over
    }
    return j->rc;
}

// engines are shared by the contexts of a device: the first context creates the engine, the last one ends it
static BaEngine* ba_engine_new(int device) {
    BaEngine* E = new BaEngine();
    E->device = device; E->refs = 1;
    bool ok = vo_stream_create(&E->st, 1) == hipSuccess;      // BA is the latency-critical chain beside tracking
    ok = ok && hipMalloc((void**)&E->d_Bs, sizeof(BaDev) * BA_SLOTS) == hipSuccess && hipMalloc((void**)&E->d_ctl, sizeof(BaCtl) * BA_SLOTS) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&E->h_Bs, sizeof(BaDev) * BA_SLOTS, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&E->h_stat, sizeof(BaStat) * BA_SLOTS, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&E->h_cull, sizeof(long long) * BA_SLOTS * BA_CULL_HOST, hipHostMallocDefault) == hipSuccess;
    if (ok) {
        memset(E->h_stat, 0, sizeof(BaStat) * BA_SLOTS);
        std::vector<BaCtl> z(BA_SLOTS);
        memset(z.data(), 0, sizeof(BaCtl) * BA_SLOTS);
        for (int s = 0; s < BA_SLOTS; ++s) { z[s].finished = 1; z[s].stage = 2; }
        ok = hipMemcpy(E->d_ctl, z.data(), sizeof(BaCtl) * BA_SLOTS, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { fprintf(stderr, "[vo_hip] BA engine: allocation failed on device %d\n", device); ba_engine_free(E); return nullptr; }
    return E;
}

// The engines of a device: the first context creates engine 0, the last one ends them all.  A device may run several engines
// (VO_BA_ENGINES, default 1).  Two were the default while a step was five launches with host round trips between chunks: the second
// engine's kernels filled the first one's gaps.  With the three-launch step and the device-side round transitions the gaps are gone and
// two engines' step kernels only slow each other down (isolated, 16 problems: one engine 2400 BA/s, two 1260; 32 streams: 5556 against
// 4721 frames/s).  A context is bound to an engine by its first solve (ba_engine_of), so contexts that never run a local BA -- the
// trackers of an overlapped back-end -- do not take part in the rotation.
BaEngine* vo_ba_engine_acquire(int device) {
    std::unique_lock<std::mutex> lk(g_eng_mu);
    for (BaEngine* E : g_engines) if (E->device == device) { ++E->refs; return E; }
    BaEngine* E = ba_engine_new(device);
    if (!E) return nullptr;
    const char* env = getenv("VO_BA_ENGINES");
    E->n_sib = std::max(1, std::min(BA_MAX_ENGINES, env ? atoi(env) : 1));
    E->sib[0] = E;
    g_engines.push_back(E);
    return E;
}

void vo_ba_engine_drain(vo_ctx* c) {                        // (see vo_internal.h)
    BaEngine* E = c ? c->ba_engine_sel : nullptr;
    if (E && E->st) (void)hipStreamSynchronize(E->st);
}

static BaEngine* ba_engine_of(vo_ctx* c) {
    if (c->ba_engine_sel) return c->ba_engine_sel;
    BaEngine* base = c->ba_engine;
    if (!base) return nullptr;
    std::unique_lock<std::mutex> lk(g_eng_mu);
    const int k = base->rr++ % base->n_sib;
    if (!base->sib[k] && !(base->sib[k] = ba_engine_new(base->device))) return nullptr;
    return c->ba_engine_sel = base->sib[k];
}

static void ba_engine_free(BaEngine* E) {
    if (vo_trace_level() && E->n_jobs) fprintf(stderr, "[vo_trace] BA engine: %lld problems, %lld step launches, %.2f problems per step launch\n", E->n_jobs, E->n_steps, E->n_steps ? (double)E->n_slot_steps / E->n_steps : 0.0);
    if (E->st) { (void)hipStreamSynchronize(E->st); (void)hipStreamDestroy(E->st); }
    if (E->d_Bs) (void)hipFree(E->d_Bs);
    if (E->d_ctl) (void)hipFree(E->d_ctl);
    if (E->h_Bs) (void)hipHostFree(E->h_Bs);
    if (E->h_stat) (void)hipHostFree(E->h_stat);
    if (E->h_cull) (void)hipHostFree(E->h_cull);
    for (BaChunk& C : E->ring) { if (C.ev_near) (void)hipEventDestroy(C.ev_near); if (C.ev_end) (void)hipEventDestroy(C.ev_end); }
    delete E;
}

void vo_ba_engine_release(BaEngine* E) {
    if (!E) return;
    {
        std::unique_lock<std::mutex> lk(g_eng_mu);
        if (--E->refs > 0) return;
        for (size_t i = 0; i < g_engines.size(); ++i) if (g_engines[i] == E) { g_engines.erase(g_engines.begin() + i); break; }
    }
    (void)hipSetDevice(E->device);
    for (int k = E->n_sib - 1; k >= 0; --k) if (E->sib[k]) ba_engine_free(E->sib[k]);      // sib[0] is E itself
}

// ---- e-3: one rank's share of a local BA sharded over ranks by point (include/vo_hip.h: vo_set_ba_shard*) --------------------------
// The problem arrays are the whole problem's (every rank uploads the same); k_ba_shard_mask leaves only the edges of this rank's points active, so every
// kernel of the launch-per-phase step works on this rank's share, and three in-place SUM exchanges per LM step make the shares one system (see the ABI
// header).  The engine is not involved: the exchanges must appear in the same order on every rank, so the steps are enqueued here, on the context's
// stream, a chunk at a time; the device-side control block decides as in the un-sharded solve (identically on every rank: same sums, same arithmetic).
static int ba_shard_solve(vo_ctx* c, BaJob* j, const vo_ba_problem* in, vo_ba_result* out) {
    hipStream_t st = c->stream;
    const int W = c->ba_shard_world, R = c->ba_shard_rank;
    BaDev B = j->B;
    const int D = B.D, nf = B.n_free, nx = B.n_points, ne = B.n_edges;
    const size_t n_x = ba_x3_off(D, nf, W) + ba_x3_len(W), n_fin = 3 * (size_t)nx + (size_t)ne + 2, n_dbl = std::max(n_x, n_fin) + 64;
    const size_t hdr = 4096, need = hdr + 8 * n_dbl;
    static_assert(sizeof(BaDev) <= 3072 && sizeof(BaCtl) <= 512 && sizeof(BaStat) <= 512, "header layout of the shard buffers");
    if (need > c->ba_shard_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        if (c->d_ba_shard) (void)hipFree(c->d_ba_shard);
        if (c->h_ba_shard) (void)hipHostFree(c->h_ba_shard);
        c->d_ba_shard = nullptr; c->h_ba_shard = nullptr; c->ba_shard_bytes = 0;
        const size_t want = need + need / 2;
        if (hipMalloc(&c->d_ba_shard, want) != hipSuccess) { c->d_ba_shard = nullptr; return VO_E_NOMEM; }
        if (hipHostMalloc(&c->h_ba_shard, want, hipHostMallocDefault) != hipSuccess) { (void)hipFree(c->d_ba_shard); c->d_ba_shard = nullptr; c->h_ba_shard = nullptr; return VO_E_NOMEM; }
        c->ba_shard_bytes = want;
    }
    uint8_t* db = (uint8_t*)c->d_ba_shard; uint8_t* hb = (uint8_t*)c->h_ba_shard;
    BaDev* d_B = (BaDev*)db; BaCtl* d_ctl = (BaCtl*)(db + 3072); BaStat* h_stat = (BaStat*)hb; double* h_x = (double*)(hb + hdr);
    B.gen1 = 1; B.shard_rank = R; B.shard_world = W; B.xbuf = (double*)(db + hdr); B.S = B.xbuf; B.bs = B.xbuf + (size_t)D * D; B.s_tiles = 0;
    B.ctl = d_ctl; B.it_robust = in->it_robust; B.it_plain = in->it_plain; B.gen = 1;
    memset(h_stat, 0, sizeof(BaStat));
    BaBatch Q; Q.Bs = d_B; Q.ctls = d_ctl; Q.stat = h_stat; Q.n = 1;
    for (int i = 0; i < BA_SLOTS; ++i) Q.slot[i] = 0;
    auto xchg = [&](double* dev, size_t n) -> int {        // in-place SUM over the ranks of n doubles in device memory
        if (c->ba_shard_stream_fn) return c->ba_shard_stream_fn(c->ba_shard_user, dev, n, (void*)st) == 0 ? VO_OK : VO_E_DEVICE;
        HIP_TRY(hipMemcpyAsync(h_x, dev, 8 * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        c->ba_shard_fn(c->ba_shard_user, h_x, (int)n);
        HIP_TRY(hipMemcpyAsync(dev, h_x, 8 * n, hipMemcpyHostToDevice, st));
        return VO_OK;
    };
    const dim3 blk(256);
    hipLaunchKernelGGL(k_ba_admit, dim3(1), blk, 0, st, B, d_B, d_ctl);
    hipLaunchKernelGGL(k_ba_shard_mask, dim3(j->grid_e), blk, 0, st, Q);
    hipLaunchKernelGGL(k_ba_chi, dim3(j->grid_e), blk, 0, st, Q, 0, 0, 0);      // this rank's share of the initial state's plain chi2 (reporting)
    const size_t lds = sizeof(double) * (CH_NB * CH_NB + (size_t)(CH_NB + 1) * (D + 1) + 2 * (size_t)D);
    double* const x1 = B.xbuf + ba_x1_off(D); double* const x3 = B.xbuf + ba_x3_off(D, nf, W);
    int rc = VO_OK;
    for (int chunk = 0;; ++chunk) {
        if (chunk > 64) return VO_E_DEVICE;                  // (20 LM iterations of at most 10 trials each; a control block that never reports the end)
        for (int s = 0; s < 5; ++s) {
            { ProfScope ps(c, "k_ba_lin", st); hipLaunchKernelGGL(k_ba_lin, dim3(j->grid_lin), blk, 0, st, Q); }
            hipLaunchKernelGGL(k_ba_maxdiag, dim3(j->grid_maxdiag), blk, 0, st, Q);
            hipLaunchKernelGGL(k_ba_shard_pack1, dim3(1), blk, 0, st, Q);
            if ((rc = xchg(x1, ba_x1_len(nf, W)))) return rc;
            hipLaunchKernelGGL(k_ba_shard_unpack1, dim3(1), blk, 0, st, Q);
            { ProfScope ps(c, "k_ba_init_S", st); hipLaunchKernelGGL(k_ba_init_S, dim3(j->grid_initS), blk, 0, st, Q); }
            if (B.n_blocks) { ProfScope ps(c, "k_ba_schur_blocks", st); hipLaunchKernelGGL(k_ba_schur_blocks, dim3(B.n_blocks), blk, 0, st, Q); }
            if ((rc = xchg(B.xbuf, (size_t)D * D + D))) return rc;      // the reduced system: S and b_s, one all-reduce (115 KB at D = 120)
            { ProfScope ps(c, "k_ba_chol16g", st); hipLaunchKernelGGL(k_ba_chol16g, dim3(1), dim3(CH_THREADS), lds, st, Q); }
            { ProfScope ps(c, "k_ba_update", st); hipLaunchKernelGGL(k_ba_update, dim3(j->grid_upd), blk, 0, st, Q); }
            { ProfScope ps(c, "k_ba_chi_control", st); hipLaunchKernelGGL(k_ba_chi_control, dim3(j->grid_c), blk, 0, st, Q); }
            if ((rc = xchg(x3, ba_x3_len(W)))) return rc;
            hipLaunchKernelGGL(k_ba_shard_decide, dim3(1), blk, 0, st, Q);
        }
        hipLaunchKernelGGL(k_ba_round, dim3(j->grid_e), blk, 0, st, Q);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        if (__atomic_load_n(&h_stat->stage, __ATOMIC_ACQUIRE) == 2) break;
    }
    // the whole result on every rank: this rank's points and edge flags, zeros elsewhere, summed
    hipLaunchKernelGGL(k_ba_shard_pack_final, dim3((int)((n_fin + 255) / 256)), blk, 0, st, Q, B.xbuf);
    if ((rc = xchg(B.xbuf, n_fin))) return rc;
    HIP_TRY(hipMemcpyAsync(h_x, B.xbuf, 8 * n_fin, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->poses, h_stat->buf ? B.posesB : B.posesA, 96 * (size_t)nf, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(out->points, h_x, 24 * (size_t)nx);
    for (int e = 0; e < ne; ++e) out->edge_flags[e] = (uint8_t)lrint(h_x[3 * (size_t)nx + e]);
    out->chi2_initial = h_x[3 * (size_t)nx + ne]; out->chi2_final = h_x[3 * (size_t)nx + ne + 1];
    out->lm_iters = h_stat->iters_total;
    return VO_OK;
}

int vo_ba_run(vo_ctx* c, const vo_ba_problem* in, vo_ba_result* out) {
    hipStream_t st = c->stream;
    BaEngine* E = ba_engine_of(c);
    if (!E) return VO_E_STATE;
    const int np = in->n_poses, nf = in->n_free, nx = in->n_points, ne = in->n_edges, D = 6 * nf;
    if ((CH_NB * CH_NB + (size_t)(CH_NB + 1) * (D + 1) + 2 * (size_t)D) * sizeof(double) > 158 * 1024) return VO_E_UNSUPPORTED;     // D > ~1060 (176 free poses)
    out->lm_iters = 0; out->chi2_initial = 0; out->chi2_final = 0;
    if (ne == 0 || nf == 0 || nx == 0) {
        memcpy(out->poses, in->poses, sizeof(double) * 12 * (size_t)nf);
        memcpy(out->points, in->points, sizeof(double) * 3 * (size_t)nx);
        memset(out->edge_flags, 0, ne);
        return VO_OK;
    }
    const bool trace = vo_trace_level() != 0;
    HIP_TRY(hipStreamSynchronize(st));                      // the pinned staging buffer may still feed an earlier vo_map_upsert
    const double tt0 = tnow();
    // CSR point -> edges and free pose -> edges: ONE counting pass here (the scratch vectors live in the context: no
    // allocation, no zero-fill of edge-sized arrays per problem); the lists themselves are written straight into the pinned
    // upload mirror further down
    std::vector<int32_t>& pt_start = c->ba_pt_start; std::vector<int32_t>& ps_start = c->ba_ps_start;
    pt_start.assign((size_t)nx + 1, 0); ps_start.assign((size_t)nf + 1, 0);
    bool sorted_by_point = true;
    for (int e = 0; e < ne; ++e) {
        const int k = in->edge_point[e], j = in->edge_pose[e];
        pt_start[k + 1]++;
        if (e && k < in->edge_point[e - 1]) sorted_by_point = false;
        if (j < nf) ps_start[j + 1]++;
    }
    for (int k = 0; k < nx; ++k) pt_start[k + 1] += pt_start[k];
    for (int j = 0; j < nf; ++j) ps_start[j + 1] += ps_start[j];
    const size_t n_ps = (size_t)ps_start[nf];
    std::vector<int32_t> pt_edges, ps_edges;                // only the host pair builder below needs them as vectors
    const double tp1 = tnow();
    // Device-built pair lists (k_ba_pairs) need the edges sorted by point (the per-pose lists are then sorted by point
    // and a block's list is a sorted intersection) and the longest per-pose list in LDS; otherwise the host builds them.
    bool dev_pairs = nf <= 64 && sorted_by_point;                 // (beyond: the host builds the pair lists -- the resident cut builds them on the device up to its 160)
    int max_len = 0;
    for (int j = 0; j < nf; ++j) max_len = std::max(max_len, ps_start[j + 1] - ps_start[j]);
    if (max_len > PAIR_LDS_CAP) dev_pairs = false;
    std::vector<BaBlock> blocks;
    int npairs = 0;
    int2* pairs = nullptr;
    int nb_all = 0, slices_ub = 0;
    int* h_counts = nullptr;
    if (dev_pairs) {
        nb_all = nf * (nf + 1) / 2;
        for (int j1 = 0; j1 < nf; ++j1) for (int j2 = j1; j2 < nf; ++j2) {
            const int m = std::min(ps_start[j1 + 1] - ps_start[j1], ps_start[j2 + 1] - ps_start[j2]);
            npairs += m; slices_ub += (m + BA_SLICE - 1) / BA_SLICE;         // upper bounds: the device writes the real counts
        }
    } else {
        pt_edges.resize(ne); ps_edges.resize(n_ps);
        { std::vector<int32_t> fill(pt_start.begin(), pt_start.end() - 1);
          for (int e = 0; e < ne; ++e) pt_edges[fill[in->edge_point[e]]++] = e; }
        { std::vector<int32_t> fill(ps_start.begin(), ps_start.end() - 1);
          for (int e = 0; e < ne; ++e) if (in->edge_pose[e] < nf) ps_edges[fill[in->edge_pose[e]]++] = e; }
        // Pair lists of the 6x6 blocks (j1 <= j2) of the reduced system, grouped by block with the points in ascending
        // order inside a block.  Host threads split the point range: count per (thread, block), prefix over blocks and
        // threads, then every thread writes its pairs straight into pinned memory -- same lists for any thread count.
        // Two edges of one point to the same pose never pair up.
        const int NT = ne > 20000 ? 4 : 1;
        std::vector<std::vector<int32_t>> cnt_t(NT, std::vector<int32_t>((size_t)nf * nf, 0));
        auto k_lo = [&](int t) { return (int)((long long)nx * t / NT); };
        auto enumerate = [&](int t, int2* out, std::vector<int32_t>* fill) {
            std::vector<int32_t>& cnt = cnt_t[t];
            int32_t fe[64], fj[64]; std::vector<int32_t> fev, fjv;
            for (int k = k_lo(t); k < k_lo(t + 1); ++k) {
                const int deg = pt_start[k + 1] - pt_start[k];
                int32_t* pe = fe; int32_t* pj = fj;
                if (deg > 64) { fev.resize(deg); fjv.resize(deg); pe = fev.data(); pj = fjv.data(); }
                int m = 0;
                for (int a = pt_start[k]; a < pt_start[k + 1]; ++a) { const int e = pt_edges[a], j = in->edge_pose[e]; if (j < nf) { pe[m] = e; pj[m] = j; ++m; } }
                for (int a = 0; a < m; ++a) {
                    const int ja = pj[a], ea = pe[a];
                    if (out) out[(*fill)[(size_t)ja * nf + ja]++] = make_int2(ea, ea); else cnt[(size_t)ja * nf + ja]++;
                    for (int b2 = a + 1; b2 < m; ++b2) {
                        const int jb = pj[b2];
                        if (ja == jb) continue;
                        const size_t bid = ja < jb ? (size_t)ja * nf + jb : (size_t)jb * nf + ja;
                        if (out) out[(*fill)[bid]++] = ja < jb ? make_int2(ea, pe[b2]) : make_int2(pe[b2], ea); else cnt[bid]++;
                    }
                }
            }
        };
        auto run_threads = [&](const std::function<void(int)>& fn) {
            std::vector<std::thread> th;
            for (int t = 1; t < NT; ++t) th.emplace_back(fn, t);
            fn(0);
            for (auto& x : th) x.join();
        };
        run_threads([&](int t) { enumerate(t, nullptr, nullptr); });
        std::vector<std::vector<int32_t>> off_t(NT, std::vector<int32_t>((size_t)nf * nf, 0));
        for (int j1 = 0; j1 < nf; ++j1) for (int j2 = j1; j2 < nf; ++j2) {
            const size_t bid = (size_t)j1 * nf + j2;
            int cnt = 0;
            for (int t = 0; t < NT; ++t) { off_t[t][bid] = npairs + cnt; cnt += cnt_t[t][bid]; }
            if (!cnt) continue;
            for (int o = 0; o < cnt; o += BA_SLICE) blocks.push_back(BaBlock{j1, j2, npairs + o, std::min(BA_SLICE, cnt - o)});
            npairs += cnt;
        }
        // pairs are written straight into pinned memory (after the 512-byte mailbox used for scal / ctl read-backs)
        uint8_t* h_pin = (uint8_t*)vo_stage(c, 512 + sizeof(int2) * (size_t)std::max(npairs, 1));
        if (!h_pin) return VO_E_NOMEM;
        pairs = (int2*)(h_pin + 512);
        run_threads([&](int t) { enumerate(t, pairs, &off_t[t]); });
    }
    const int nblk = dev_pairs ? slices_ub : (int)blocks.size();
    const double tp2 = tnow();

    // carve the scratch slab
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    // the host inputs come first and contiguous: they are packed into one pinned mirror and travel in ONE H2D copy (ten
    // pageable copies per problem used to queue behind each other -- and behind the other streams' -- on the copy engines)
    const size_t o_poses = carve(96 * (size_t)np), o_pts = carve(24 * (size_t)nx);
    const size_t o_epose = carve(4 * (size_t)ne), o_ept = carve(4 * (size_t)ne), o_euv = carve(8 * (size_t)ne);
    const size_t o_ps = carve(4 * (size_t)(nx + 1)), o_pe = carve(4 * (size_t)ne);
    const size_t o_qs = carve(4 * (size_t)(nf + 1)), o_qe = carve(4 * std::max<size_t>(n_ps, 1));
    const size_t o_pspt = carve(4 * n_ps + 4);
    const size_t up_end = off;
    const size_t o_poses_n = carve(96 * (size_t)np), o_pts_n = carve(24 * (size_t)nx), o_act = carve(ne), o_flags = carve(ne);
    const size_t o_blk = carve(sizeof(BaBlock) * (size_t)std::max(nblk, 1)), o_pairs = carve(sizeof(int2) * (size_t)std::max(npairs, 1));
    const size_t o_pcnt = carve(4 * (size_t)std::max(nb_all, 1)), o_poff = carve(4 * (size_t)std::max(nb_all, 1)), o_pn = carve(16);
    const size_t o_Hpp = carve(288 * (size_t)nf), o_bp = carve(8 * (size_t)D), o_Hll = carve(72 * (size_t)nx), o_bl = carve(24 * (size_t)nx), o_scal = carve(64);
    const size_t o_partU = carve(24 * ((size_t)(nx + 63) / 64 + 1)), o_partC = carve(8 * ((size_t)(ne + 255) / 256 + 1));
    const size_t o_W = carve(std::max<size_t>(144 * (size_t)ne, 16 * (size_t)ne + 192 * (size_t)nx + 2048)), o_S = carve(std::max<size_t>(8 * (size_t)D * D, 8 * ba_tile_doubles(D)) + 1024), o_bs = carve(8 * (size_t)D), o_Hinv = carve(72 * (size_t)nx), o_dl = carve(std::max<size_t>(24 * (size_t)nx, 8 * (size_t)(D + 8)));      // (dl: the solution and, behind it, lambda / ok / cur / ni for the update workgroups of a fused launch)
    int rc = vo_scratch(c, off);
    if (rc) return rc;
    uint8_t* base = (uint8_t*)c->d_ba;
    BaDev B;
    B.n_poses = np; B.n_free = nf; B.n_points = nx; B.n_edges = ne; B.D = D; B.n_blocks = nblk; B.s_tiles = ba_use_tiles(D);
    B.n_slices = dev_pairs ? (const int*)(base + o_pn) : nullptr;
    B.ps_start = (const int32_t*)(base + o_qs); B.ps_edges = (const int32_t*)(base + o_qe); B.blocks = (const BaBlock*)(base + o_blk); B.pairs = (const int2*)(base + o_pairs); B.pair_pt = nullptr; B.ps_pt = (const int32_t*)(base + o_pspt);
    B.posesA = (double*)(base + o_poses); B.ptsA = (double*)(base + o_pts); B.posesB = (double*)(base + o_poses_n); B.ptsB = (double*)(base + o_pts_n);
    B.ctl = nullptr;                                        // the engine assigns the control block of the problem's slot
    B.partU = (double*)(base + o_partU); B.partC = (double*)(base + o_partC); B.nU = (nx + 63) / 64;
    B.e_pose = (const int32_t*)(base + o_epose); B.e_pt = (const int32_t*)(base + o_ept); B.e_uv = (const float*)(base + o_euv);
    B.active = base + o_act; B.flags = base + o_flags; B.pt_start = (const int32_t*)(base + o_ps); B.pt_edges = (const int32_t*)(base + o_pe);
    B.Hpp = (double*)(base + o_Hpp); B.bp = (double*)(base + o_bp); B.Hll = (double*)(base + o_Hll); B.bl = (double*)(base + o_bl); B.scal = (double*)(base + o_scal);
    B.W = (double*)(base + o_W); B.S = (double*)(base + o_S); B.bs = (double*)(base + o_bs); B.Hinv = (double*)(base + o_Hinv); B.dl = (double*)(base + o_dl);
    B.cam = BaCam{(double)c->p.fx, (double)c->p.fy, (double)c->p.cx, (double)c->p.cy};
    B.delta = in->huber_delta; B.chi2_th = in->chi2_th; B.gp = (nx + 63) / 64; B.edges_by_point = sorted_by_point ? 1 : 0;
    B.e_obs = nullptr; B.cull = nullptr; B.ncull = nullptr; B.cull_cap = 0; B.cull_host = nullptr; B.cull_host_cap = 0;
    B.gen1 = 0; B.shard_rank = 0; B.shard_world = 1; B.upc_ovf = sizeof(double) * (24 * (size_t)np + (size_t)D + UPC_LDS_EXTRA) <= 150 * 1024 ? 1 : 0; B.upc_ppw = ba_upc_ppw(nx); B.xbuf = nullptr;

    {
        if (up_end > c->h_ba_up_bytes) {                    // pinned mirror of the upload region, grown geometrically
            if (c->h_ba_up) (void)hipHostFree(c->h_ba_up);
            c->h_ba_up = nullptr; c->h_ba_up_bytes = 0;
            const size_t want = up_end + up_end / 2 + (1 << 20);
            if (hipHostMalloc(&c->h_ba_up, want, hipHostMallocDefault) != hipSuccess) { c->h_ba_up = nullptr; return VO_E_NOMEM; }
            c->h_ba_up_bytes = want;
        }
        uint8_t* m = (uint8_t*)c->h_ba_up;                  // free: the stream was drained at the top of this function
        memcpy(m + o_poses, in->poses, 96 * (size_t)np); memcpy(m + o_pts, in->points, 24 * (size_t)nx);
        memcpy(m + o_epose, in->edge_pose, 4 * (size_t)ne); memcpy(m + o_ept, in->edge_point, 4 * (size_t)ne); memcpy(m + o_euv, in->edge_uv, 8 * (size_t)ne);
        memcpy(m + o_ps, pt_start.data(), 4 * (size_t)(nx + 1)); memcpy(m + o_qs, ps_start.data(), 4 * (size_t)(nf + 1));
        {   // the edge lists, written in place: point -> edges (identity when the caller's edges are grouped by point), free pose -> edges (+ their points)
            int32_t* m_pe = (int32_t*)(m + o_pe); int32_t* m_qe = (int32_t*)(m + o_qe); int32_t* m_qp = (int32_t*)(m + o_pspt);
            std::vector<int32_t>& cur = c->ba_cursor;
            cur.assign(ps_start.begin(), ps_start.end());
            if (sorted_by_point) {
                for (int e = 0; e < ne; ++e) {
                    m_pe[e] = e;
                    const int j = in->edge_pose[e];
                    if (j < nf) { const int pos = cur[j]++; m_qe[pos] = e; m_qp[pos] = in->edge_point[e]; }
                }
            } else {
                memcpy(m_pe, pt_edges.data(), 4 * (size_t)ne);
                for (int e = 0; e < ne; ++e) { const int j = in->edge_pose[e]; if (j < nf) { const int pos = cur[j]++; m_qe[pos] = e; m_qp[pos] = in->edge_point[e]; } }
            }
        }
        HIP_TRY(hipMemcpyAsync(base, m, up_end, hipMemcpyHostToDevice, st));
    }
    if (dev_pairs) {
        BaPairPlan Q;
        Q.ps_start = B.ps_start; Q.ps_edges = B.ps_edges; Q.ps_pt = (const int32_t*)(base + o_pspt); Q.nf = nf;
        Q.cnt = (int*)(base + o_pcnt); Q.off = (int*)(base + o_poff); Q.n_slices = (int*)(base + o_pn); Q.n_pairs = (int*)(base + o_pn) + 1;
        Q.blocks = (BaBlock*)(base + o_blk); Q.pairs = (int2*)(base + o_pairs); Q.lds_cap = std::max(max_len, 1);
        if (nb_all) {
            hipLaunchKernelGGL(k_ba_pairs<false>, dim3(nb_all), dim3(256), 4 * (size_t)std::max(max_len, 1), st, Q);
            hipLaunchKernelGGL(k_ba_pairs_scan, dim3(1), dim3(1024), 0, st, Q, nb_all);
            hipLaunchKernelGGL(k_ba_pairs<true>, dim3(nb_all), dim3(256), 4 * (size_t)std::max(max_len, 1), st, Q);
        }
        // the real slice / pair counts come back with the synchronisation below (it waits for the uploads anyway), so the
        // Schur kernel is launched on the exact grid instead of the upper bound
        if (!(h_counts = (int*)vo_stage(c, 512))) return VO_E_NOMEM;
        h_counts[0] = 0; h_counts[1] = 0;
        HIP_TRY(hipMemcpyAsync(h_counts, base + o_pn, 8, hipMemcpyDeviceToHost, st));
    } else {
        if (nblk) HIP_TRY(hipMemcpyAsync(base + o_blk, blocks.data(), sizeof(BaBlock) * (size_t)nblk, hipMemcpyHostToDevice, st));
        if (npairs) HIP_TRY(hipMemcpyAsync(base + o_pairs, pairs, sizeof(int2) * (size_t)npairs, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemsetAsync(base + o_act, 1, ne, st));
    HIP_TRY(hipMemsetAsync(base + o_flags, 0, ne, st));
    const double tp3 = tnow();
    HIP_TRY(hipStreamSynchronize(st));       // pageable sources
    int nblk_launch = nblk;
    if (dev_pairs && h_counts) { nblk_launch = std::min(nblk, h_counts[0]); npairs = h_counts[1]; }

    const double tt1 = tnow();
    // ---- solve: the engine steps this problem together with whatever other local BAs are in flight on this GPU ----------
    BaJob job;
    job.c = c; job.in = in; job.out = out; job.B = B; job.B.n_blocks = nblk_launch;
    job.grid_lin = (nx + 63) / 64 + nf * PSPLIT; job.grid_initS = (std::max(D * D, nx) + 255) / 256; job.grid_upd = (nx + 63) / 64 + (np + 255) / 256;
    job.grid_e = (ne + 255) / 256; job.grid_c = (ne + 1023) / 1024; job.grid_maxdiag = (D + 3 * nx + 255) / 256;
    job.lds = job.B.s_tiles ? ch2_lds_bytes(D) : D <= 192 ? sizeof(double) * (CH_NB * CH_NB + (size_t)(D + 1) * (D + 2) / 2 + 2 * (size_t)D)
                       : sizeof(double) * (CH_NB * CH_NB + (size_t)(CH_NB + 1) * (D + 1) + 2 * (size_t)D);
    if (c->ba_shard_world > 1) return ba_shard_solve(c, &job, in, out);      // e-3: this rank's share, in lockstep with the other ranks (no engine: the exchanges order the launches)
    rc = ba_engine_solve(E, &job);
    if (rc) return rc;
    out->lm_iters = job.iters;
    const double tt2 = tnow();
    const int cur_buf = job.cur_buf;
    HIP_TRY(hipMemcpyAsync(out->poses, cur_buf ? B.posesB : B.posesA, 96 * (size_t)nf, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->points, cur_buf ? B.ptsB : B.ptsA, 24 * (size_t)nx, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->edge_flags, B.flags, ne, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    out->chi2_final = job.chi_final;
    out->chi2_initial = job.chi0;
    if (trace) { static double a0 = 0, a1 = 0, a2 = 0; static int n = 0; a0 += tt1 - tt0; a1 += tt2 - tt1; a2 += tnow() - tt2; if (++n % 10 == 0) fprintf(stderr, "[vo_trace] vo_ba_run avg ms: prep+upload %.2f optimise %.2f download %.2f (D=%d edges=%d pairs=%d) | last prep: csr %.2f pairs %.2f enqueue %.2f sync %.2f\n", a0 / n, a1 / n, a2 / n, D, ne, npairs, tp1 - tt0, tp2 - tp1, tp3 - tp2, tt1 - tp3); }
    return VO_OK;
}

// =====================================================================================================================
// Resident graph cut: the local BA's problem arrays are built ON THE DEVICE from the observation table
// and the keyframe poses (reference src/backend.cpp:36-135, which walks hash maps of shared pointers), so a keyframe costs
// the host neither the graph cut (0.45 ms of pointer chasing) nor the CSR build and 3 MB upload (0.4 ms).
//
//   k_cut_init        keyframe -> pose index for the free keyframes (list passed by value)
//   k_cut_points      observation-parallel: flag the map points some FREE keyframe observes (outliers excluded)
//   k_scan_*          exclusive scan of int32 arrays (flags -> dense point index, per-point edge counts -> pt_start)
//   k_cut_count       observation-parallel: edges per point (one atomic per observation, ~3 per address); flag the fixed keyframes
//   k_cut_fixed_scan  one workgroup: numbers the fixed keyframes behind the free ones
//   k_cut_fill        observation-parallel: observation ids into their point's segment (arrival order); map slot -> point list
//   k_cut_emit        edge-parallel: an observation's place in its point's segment is its rank by keyframe number (what the host's
//                     graph cut emits: observation lists are in keyframe order), then edge_pose / edge_point / edge_uv / pt_edges;
//                     the same launch gathers poses from the keyframe table and positions from the map
//   k_ps_hist / k_ps_offsets / k_ps_fill   per free pose: its edges in ascending edge order (stable counting sort by pose)
// 16 launches, one memset, two small read-backs (sizes; per-pose list lengths). The pair plan (k_ba_pairs*) is queued behind
// them and an event hands the finished arrays to the engine's stream.
// Point order = ascending map slot, fixed poses = ascending keyframe number: deterministic, and the same as oracle/o_capi.cpp.
// =====================================================================================================================
// obs_lo / map_lo: where the tables are entered.  Both tables only grow (observations are appended, map slots are handed out in ascending
// order), so everything a cut can touch lies behind two bounds the host keeps per keyframe (vo_obs_append, vo_ctx::kf_reach): the first
// observation of the oldest point a free keyframe observes -- every observation of every point in the graph sits behind it -- and that
// point's slot.  The slot-indexed scratch arrays (pt_flag, pidx) are indexed by slot - map_lo.  A cut costs what its window holds, not
// what the run has accumulated.
struct CutTabs { const int32_t* obs_kf; const int32_t* obs_mp; const float* obs_uv; const uint8_t* alive; long long n_obs;
                 const uint8_t* map_flags; const double* map_pos; const double* kf_pose; int n_kf, map_hi; long long obs_lo; int map_lo; };

__global__ void k_cut_points(CutTabs T, const int* __restrict__ kf_idx, int* __restrict__ pt_flag) {
    const long long o = T.obs_lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= T.n_obs || !T.alive[o]) return;
    const int k = T.obs_kf[o], m = T.obs_mp[o];
    if (k < T.n_kf && m >= T.map_lo && m < T.map_hi && kf_idx[k] >= 0 && !(T.map_flags[m] & VO_MAP_FLAG_OUTLIER)) pt_flag[m - T.map_lo] = 1;      // idempotent
}
// three-pass exclusive scan of n int32 (n <= 16 Mi): a workgroup covers SCAN_TILE = 1024 lanes x 16 consecutive elements; block sums,
// scan of the (<= 1024) sums, per-block scan + offset; total -> *total_out
#define SCAN_PER 16
#define SCAN_TILE (1024 * SCAN_PER)
// Exclusive scan of an int32 array in two launches.  A workgroup owns a tile of 16 Ki elements = 4 sub-tiles of 4 Ki; a thread takes one
// int4 per sub-tile (consecutive lanes, consecutive 16 bytes: the arrays are 256-byte aligned and padded).  k_scan_blocksum leaves the tile
// totals; k_scan_final adds up the totals in front of its tile itself (<= 1024 of them) instead of waiting for a third launch.
__device__ __forceinline__ int4 scan_load4(const int* __restrict__ in, long long i, int n) {
    int4 v = make_int4(0, 0, 0, 0);
    if (i + 3 < n) v = *reinterpret_cast<const int4*>(in + i);
    else { if (i < n) v.x = in[i]; if (i + 1 < n) v.y = in[i + 1]; if (i + 2 < n) v.z = in[i + 2]; }
    return v;
}
__global__ __launch_bounds__(1024) void k_scan_blocksum(const int* __restrict__ in, int n, int* __restrict__ bsum) {
    __shared__ int s_w[16];
    int v = 0;
#pragma unroll
    for (int q = 0; q < SCAN_PER / 4; ++q) {
        const int4 e = scan_load4(in, (long long)blockIdx.x * SCAN_TILE + ((long long)q * 1024 + threadIdx.x) * 4, n);
        v += (e.x + e.y) + (e.z + e.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_w[w]; bsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void k_scan_final(const int* __restrict__ in, int n, const int* __restrict__ bsum, int nb, int* __restrict__ out, int* __restrict__ total_out) {
    __shared__ int s_w[16], s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // totals of the tiles in front of this one
        int p = 0;
        for (int i = threadIdx.x; i < (int)blockIdx.x; i += 1024) p += bsum[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if (lane == 0) s_w[wave] = p;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_w[w]; s_base = t; if ((int)blockIdx.x == nb - 1) *total_out = t + bsum[blockIdx.x]; }
        __syncthreads();
    }
    int run = s_base;
#pragma unroll
    for (int q = 0; q < SCAN_PER / 4; ++q) {
        const long long i = (long long)blockIdx.x * SCAN_TILE + ((long long)q * 1024 + threadIdx.x) * 4;
        const int4 e = scan_load4(in, i, n);
        const int v = (e.x + e.y) + (e.z + e.w);
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        __syncthreads();                                    // s_w of the previous round has been read
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int off = run + inc - v, tot = 0;
        for (int w = 0; w < 16; ++w) { const int sw = s_w[w]; if (w < wave) off += sw; tot += sw; }
        const int4 o4 = make_int4(off, off + e.x, off + e.x + e.y, off + e.x + e.y + e.z);
        if (i + 3 < n) *reinterpret_cast<int4*>(out + i) = o4;
        else { if (i < n) out[i] = o4.x; if (i + 1 < n) out[i + 1] = o4.y; if (i + 2 < n) out[i + 2] = o4.z; }
        run += tot;
    }
}
__global__ void k_cut_count(CutTabs T, const int* __restrict__ kf_idx, const int* __restrict__ pt_flag, const int* __restrict__ pidx,
                            int* __restrict__ cnt, int* __restrict__ fixed_flag) {
    const long long o = T.obs_lo + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= T.n_obs || !T.alive[o]) return;
    const int k = T.obs_kf[o], m = T.obs_mp[o];
    if (k >= T.n_kf || m < T.map_lo || m >= T.map_hi || !pt_flag[m - T.map_lo]) return;
    atomicAdd(&cnt[pidx[m - T.map_lo]], 1);
    if (kf_idx[k] < 0) fixed_flag[k] = 1;
}
struct CutFree { int n; int kf[VO_BA_RESIDENT_MAX_FREE]; };       // by value: 644 bytes of kernel arguments
// keyframe -> pose index: the free keyframes take 0 .. n-1 in the caller's order, everything else -1 until k_cut_fixed_scan
// (the same launch zeroes the cut's flag / count / cursor arrays: a memset is a blit kernel with a launch gap of its own)
__global__ void k_cut_init(int n_kf, CutFree F, int* __restrict__ kf_idx, int* __restrict__ pose_kf, int4* __restrict__ zero, int n_zero16) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = k; i < n_zero16; i += gridDim.x * blockDim.x) zero[i] = make_int4(0, 0, 0, 0);
    if (k < F.n) pose_kf[k] = F.kf[k];
    if (k >= n_kf) return;
    int idx = -1;
    for (int i = 0; i < F.n; ++i) if (F.kf[i] == k) idx = i;
    kf_idx[k] = idx;
}
// a kernel's results for the host, in pinned memory: values first, a system-scope fence, then the word the host polls (vo_spin_word)
__global__ void k_cut_report(const int* __restrict__ src, int n, int* __restrict__ host, int* __restrict__ word, int seq, int* __restrict__ clear) {
    if ((int)threadIdx.x < n) host[threadIdx.x] = src[threadIdx.x];
    if (threadIdx.x == 0 && clear) *clear = 0;              // (k_ba_pairs_one's ticket)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
// one workgroup: fixed poses follow the free ones in ascending keyframe number (running scan over the keyframe table)
__global__ __launch_bounds__(1024) void k_cut_fixed_scan(int n_kf, int n_free, const int* __restrict__ fixed_flag, int* __restrict__ kf_idx, int* __restrict__ pose_kf,
                                                         int* __restrict__ n_fixed_out, const int* __restrict__ tot, int* __restrict__ host, int* __restrict__ word, int seq) {
    __shared__ int s_w[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int run = 0;
    for (int k0 = 0; k0 < n_kf; k0 += 1024) {
        const int k = k0 + threadIdx.x;
        const bool hit = k < n_kf && fixed_flag[k];
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int before = __popcll(m & ((1ull << lane) - 1ull)), tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) before += s_w[w]; tot += s_w[w]; }
        if (hit) { kf_idx[k] = n_free + run + before; pose_kf[n_free + run + before] = k; }
        run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *n_fixed_out = run;
        // the sizes go to the host at once (pinned memory, then the word it polls: k_cut_report's protocol without a launch of its own)
        host[0] = tot[0]; host[1] = tot[1]; host[2] = run;
        __threadfence_system();
        __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// observation -> its slot in its point's edge range (arrival order; k_cut_emit sorts), and map slot -> dense point list
__global__ void k_cut_fill(CutTabs T, const int* __restrict__ pt_flag, const int* __restrict__ pidx, const int* __restrict__ pt_start,
                           int* __restrict__ fill, long long* __restrict__ e_obs, int* __restrict__ point_slots) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, o = T.obs_lo + i;
    if (i < T.map_hi - T.map_lo && pt_flag[i]) point_slots[pidx[i]] = T.map_lo + (int)i;
    if (o >= T.n_obs || !T.alive[o]) return;
    const int k = T.obs_kf[o], m = T.obs_mp[o];
    if (k >= T.n_kf || m < T.map_lo || m >= T.map_hi || !pt_flag[m - T.map_lo]) return;
    const int p = pidx[m - T.map_lo];
    e_obs[pt_start[p] + atomicAdd(&fill[p], 1)] = o;
}
// edge-parallel: slot i of its point's segment holds some observation (arrival order); its place among the point's edges is its
// rank by keyframe number (a keyframe observes a point once: no ties).  The same launch gathers poses and positions.
__global__ void k_cut_emit(CutTabs T, int nf, const int* __restrict__ tot /* nx, ne, n_fixed: k_scan_*, k_cut_fixed_scan */, const int* __restrict__ pidx, const int* __restrict__ pt_start, const int* __restrict__ kf_idx,
                           const int* __restrict__ pose_kf, const int* __restrict__ point_slots, const long long* __restrict__ e_arr, long long* __restrict__ e_obs,
                           int32_t* __restrict__ e_pose, int32_t* __restrict__ e_pt, float* __restrict__ e_uv, int32_t* __restrict__ pt_edges, uint8_t* __restrict__ active,
                           uint8_t* __restrict__ flags, double* __restrict__ posesA, double* __restrict__ posesB, double* __restrict__ ptsA) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nx = tot[0], ne = tot[1], np = nf + tot[2];      // (the launch covers the host's upper bounds: the sizes themselves never leave the device before the chain's end)
    if (i < np * 12) { const double v = T.kf_pose[12 * (size_t)pose_kf[i / 12] + i % 12]; posesA[i] = v; posesB[i] = v; }
    if (i < nx * 3) ptsA[i] = T.map_pos[3 * (size_t)point_slots[i / 3] + i % 3];
    if (i >= ne) return;
    const long long o = e_arr[i];
    const int ko = T.obs_kf[o], p = pidx[T.obs_mp[o] - T.map_lo];
    const int a = pt_start[p], b = pt_start[p + 1];
    int rank = 0;
    for (int j = a; j < b; ++j) rank += T.obs_kf[e_arr[j]] < ko;
    const int d = a + rank;
    e_obs[d] = o; e_pose[d] = kf_idx[ko]; e_pt[d] = p; e_uv[2 * d] = T.obs_uv[2 * o]; e_uv[2 * d + 1] = T.obs_uv[2 * o + 1]; pt_edges[d] = d;
    active[d] = 1; flags[d] = 0;
}
// per-pose edge lists (free poses only) in ascending edge index = ascending point: a stable counting sort by pose over chunks
// of 1024 edges.  k_ps_hist: per-chunk counts; k_ps_offsets: one workgroup per pose scans its counts over the chunks;
// k_ps_fill: recomputes the in-chunk ranks (ballot per pose) and writes the lists.
#define PS_CHUNK 1024
__device__ __forceinline__ int ps_wave_rank(int bin, int nf, int lane, int* __restrict__ wave_cnt /* [nf] of this wave, or nullptr */) {
    int rank = 0;
    for (int j = 0; j < nf; ++j) {
        const unsigned long long m = __ballot(bin == j);
        if (bin == j) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (wave_cnt && lane == 0) wave_cnt[j] = __popcll(m);
    }
    return rank;
}
__global__ __launch_bounds__(PS_CHUNK) void k_ps_hist(const int* __restrict__ ne_p, int nf, const int32_t* __restrict__ e_pose, int* __restrict__ hist /* [chunks][nf] */) {
    __shared__ int s_cnt[PS_CHUNK / 64][VO_BA_RESIDENT_MAX_FREE];
    const int ne = *ne_p;                                   // (a chunk behind the last edge counts nothing: its row of hist is zero)
    const int e = blockIdx.x * PS_CHUNK + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = e < ne ? e_pose[e] : nf;
    (void)ps_wave_rank(q < nf ? q : -1, nf, lane, s_cnt[wave]);
    __syncthreads();
    if ((int)threadIdx.x < nf) { int t = 0; for (int w = 0; w < PS_CHUNK / 64; ++w) t += s_cnt[w][threadIdx.x]; hist[blockIdx.x * nf + threadIdx.x] = t; }
}
__global__ __launch_bounds__(256) void k_ps_offsets(int chunks, int nf, const int* __restrict__ hist, int* __restrict__ offs /* [chunks][nf] */, int* __restrict__ total /* [nf] */) {
    __shared__ int s_w[4];
    const int j = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int run = 0;
    for (int g0 = 0; g0 < chunks; g0 += 256) {
        const int g = g0 + threadIdx.x;
        const int v = g < chunks ? hist[g * nf + j] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int before = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) before += s_w[w]; tot += s_w[w]; }
        if (g < chunks) offs[g * nf + j] = run + before + inc - v;
        run += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[j] = run;
}
__global__ __launch_bounds__(PS_CHUNK) void k_ps_fill(const int* __restrict__ ne_p, int nf, const int32_t* __restrict__ e_pose, const int32_t* __restrict__ e_pt, const int* __restrict__ offs,
                                                      const int* __restrict__ total, int* __restrict__ ps_start, int32_t* __restrict__ ps_edges, int32_t* __restrict__ ps_pt) {
    __shared__ int s_cnt[PS_CHUNK / 64][VO_BA_RESIDENT_MAX_FREE];
    __shared__ int s_base[VO_BA_RESIDENT_MAX_FREE + 1];
    const int ne = *ne_p;
    const int e = blockIdx.x * PS_CHUNK + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {                                        // start of each pose's list: exclusive scan of the totals, 64 at a time
        int run = 0;
        for (int j0 = 0; j0 < nf; j0 += 64) {
            const int v = j0 + lane < nf ? total[j0 + lane] : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            if (j0 + lane < nf) s_base[j0 + lane] = run + inc - v;
            run += __shfl(inc, 63, 64);
        }
        if (lane == 0) s_base[nf] = run;
    }
    const int q = e < ne ? e_pose[e] : nf;
    const int bin = q < nf ? q : -1;
    const int rank = ps_wave_rank(bin, nf, lane, s_cnt[wave]);
    __syncthreads();
    if (blockIdx.x == 0 && (int)threadIdx.x <= nf) ps_start[threadIdx.x] = s_base[threadIdx.x];
    if (bin < 0) return;
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_cnt[w][bin];
    const int pos = s_base[bin] + offs[blockIdx.x * nf + bin] + before + rank;
    ps_edges[pos] = e; ps_pt[pos] = e_pt[e];
}
__global__ void k_culled_list(int ne, const uint8_t* __restrict__ flags, const long long* __restrict__ e_obs, int* __restrict__ n_out, long long* __restrict__ out, int cap) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < ne && (flags[e] & 3)) { const int pos = atomicAdd(n_out, 1); if (pos < cap) out[pos] = e_obs[e]; }
}

struct BaResident {
    bool ready = false;
    long long win_obs = -1, win_slots = -1;                 // what the last cut visited (vo_ba_resident_window)
    int np = 0, nf = 0, nx = 0, ne = 0, n_fixed = 0, nblk_launch = 0, npairs = 0;
    BaDev B;
    int32_t* d_point_slots = nullptr; int* d_pose_kf = nullptr; long long* d_e_obs = nullptr; int* d_ncull = nullptr; long long* d_cull = nullptr; int cull_cap = 0;
    hipEvent_t ev = nullptr, ev_arrays = nullptr;           // ev_arrays: recorded in FRONT of the pair-plan kernels (the first linearisation needs no pairs); ev: behind them (the first Schur launch waits for it)
    // the last solve, until the next cut: where its result lies (vo_local_ba_resident_merge / _fetch)
    // (a back-end thread writes the first line in _cut / _solve; the caller's thread runs _merge while that thread idles and _fetch
    // beside its next _cut: _fetch reads only what _merge copied into the st_* fields)
    bool solved = false; int solve_seq = 0, cur_buf = 0, n_culled = 0; double chi0 = 0, chi_final = 0; int lm_iters = 0;
    void* d_stage = nullptr; size_t stage_bytes = 0;        // merged result, out of the slab the next cut reuses: [poses nf x 12][points nx x 3][slots nx]
    bool has_stage = false; int merged_seq = 0, st_nf = 0, st_nx = 0, st_ne = 0, st_fixed = 0, st_culled = 0, st_iters = 0; double st_chi0 = 0, st_chi1 = 0;
    bool ledger_pending = false; int ledger_total = 0;      // vo_local_ba_resident_merge_ledger answered VO_E_OVERFLOW: the culled observations are still marked, the next call pages the pairs out
    hipStream_t fetch_stream = nullptr; hipEvent_t ev_merge = nullptr;      // _fetch copies on a stream of its own (behind ev_merge): the tables' stream may be running the tracker's next launch chain, the context's own stream the next cut
};
void vo_ba_resident_free(vo_ctx* c) {
    if (c->resident) {
        if (c->resident->ev) (void)hipEventDestroy(c->resident->ev);
        if (c->resident->ev_arrays) (void)hipEventDestroy(c->resident->ev_arrays);
        if (c->resident->ev_merge) (void)hipEventDestroy(c->resident->ev_merge);
        if (c->resident->d_stage) (void)hipFree(c->resident->d_stage);
        if (c->resident->fetch_stream) (void)hipStreamDestroy(c->resident->fetch_stream);
    }
    delete c->resident; c->resident = nullptr;
}

// merge of a solved graph into the tracker's tables (vo_local_ba_resident_merge): what vo_map_upsert (positions only), vo_kf_set_pose and
// vo_obs_kill do from host arrays, straight from the solve's buffers; the same values also go to a staging buffer for the host's copy
// ledger = 1 (vo_local_ba_resident_merge_ledger: the caller keeps no host map objects): every point of the graph is flagged optimised
// (src/backend.cpp:190), the free poses also go to pinned host memory, and the culled observations are left to k_merge_ledger
__global__ void k_ba_merge(int nf, int nx, const double* __restrict__ poses, const double* __restrict__ pts, const int* __restrict__ pose_kf, const int32_t* __restrict__ point_slots,
                           const int* __restrict__ n_cull, const long long* __restrict__ cull, int cull_cap, double* __restrict__ map_pos, uint8_t* __restrict__ map_flags,
                           double* __restrict__ kf_pose, uint8_t* __restrict__ obs_alive, double* __restrict__ st_poses, double* __restrict__ st_pts, int32_t* __restrict__ st_slots,
                           int ledger, double* __restrict__ host_poses) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nx) {
        const int slot = point_slots[i];
        const double x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
        st_pts[3 * (size_t)i] = x; st_pts[3 * (size_t)i + 1] = y; st_pts[3 * (size_t)i + 2] = z; st_slots[i] = slot;
        const uint8_t fl = map_flags[slot];
        if (ledger) map_flags[slot] = fl | VO_MAP_FLAG_OPTIMIZED;
        if (!(fl & VO_MAP_FLAG_OUTLIER)) { map_pos[3 * (size_t)slot] = x; map_pos[3 * (size_t)slot + 1] = y; map_pos[3 * (size_t)slot + 2] = z; }
    }
    if (i < 12 * nf) { const double v = poses[i]; st_poses[i] = v; kf_pose[12 * (size_t)pose_kf[i / 12] + i % 12] = v; if (host_poses) host_poses[i] = v; }
    if (!ledger && i < min(*n_cull, cull_cap)) obs_alive[cull[i]] = 0;
}
// The covisibility ledger's side of a merge (Frame::RemoveObservedMappoint, reference src/frame.cpp:122-152; Mappoint::RemoveObservedByKeyframe,
// src/mappoint.cpp:40-45), one workgroup: culling observation (K, P) costs K and every keyframe that still sees P one shared point -- the pairs go to
// pinned host memory, the caller's ledger applies them -- and a point nobody sees any more becomes an outlier.  The reference removes the
// observations one after the other; with c < c' both culled from one point, the pair (K, K') is therefore reported once, from c.  Phases: mark the
// culled observations (alive = 2), walk each one's point chain, clear them.
// mode 0: the three phases in one launch -- unless the walk finds more than pair_cap pairs: then the marks STAY (nothing is lost: the host pages the pairs
// out with mode 1 launches over ranges [i0, i1) of the culled list, the marks being what makes every walk see the same survivors, and ends with mode 2,
// the clearing phase alone).  The outlier flags a walk sets are the same ones whichever launch sets them.
__global__ __launch_bounds__(256) void k_merge_ledger(const int* __restrict__ n_cull, const long long* __restrict__ cull, int cull_cap, const int32_t* __restrict__ obs_kf,
                                                      const int32_t* __restrict__ obs_mp, uint8_t* __restrict__ obs_alive, const int2* __restrict__ obs_link, const int32_t* __restrict__ pt_last,
                                                      uint8_t* __restrict__ map_flags, int* __restrict__ pair_a, int* __restrict__ pair_b, int pair_cap, int* __restrict__ n_pairs_total,
                                                      int mode, int i0, int i1) {
    __shared__ int s_n;
    const int n = min(*n_cull, cull_cap);
    if (threadIdx.x == 0) s_n = 0;
    if (mode == 0) for (int i = threadIdx.x; i < n; i += 256) { const long long c = cull[i]; if (obs_alive[c]) obs_alive[c] = 2; }
    __syncthreads();
    if (mode != 2)
    for (int i = (mode == 1 ? i0 : 0) + threadIdx.x; i < (mode == 1 ? min(i1, n) : n); i += 256) {
        const int c = (int)cull[i];
        if (obs_alive[c] != 2) continue;
        const int K = obs_kf[c], P = obs_mp[c];
        int survivors = 0;
        for (int q = pt_last[P]; q >= 0;) {
            const int2 l = obs_link[q];
            const int a = obs_alive[q];
            if (q != c && a) {
                if (a == 1) ++survivors;
                if (a == 1 || q > c) { const int pos = atomicAdd(&s_n, 1); if (pos < pair_cap) { pair_a[pos] = K; pair_b[pos] = l.y; } }
            }
            q = l.x;
        }
        if (survivors == 0) map_flags[P] |= VO_MAP_FLAG_OUTLIER;
    }
    __syncthreads();
    if (mode == 2 || (mode == 0 && s_n <= pair_cap)) for (int i = threadIdx.x; i < n; i += 256) { const long long c = cull[i]; if (obs_alive[c] == 2) obs_alive[c] = 0; }
    if (threadIdx.x == 0 && mode != 2) *n_pairs_total = s_n;
}

// The same scan in ONE launch for up to 512 tiles (8 Mi elements): a workgroup publishes its tile's total -- (call number << 32 | total) in one
// 64-bit agent-scope store, so a stale entry of an earlier scan through the same buffer can never be taken for this call's -- BEFORE it waits for
// anything, then adds up the totals of the tiles in front of it (lane i polls tile i: a workgroup waits only for workgroups with a lower index,
// which the dispatcher started first and which never wait before they publish), scans its tile from the registers it loaded for the total and
// writes the result.  The spins are bounded; a tile that never shows up makes the last tile report -1 as the total.
// (Another one-launch variant -- one workgroup, a contiguous run per lane -- was measured and lost 3 % of a single stream's frames/s: the lanes'
// 256-byte runs do not coalesce.)  `agg` must not hold (call number << 32) of a FUTURE call: the buffers are zeroed when they are allocated.
__global__ __launch_bounds__(1024) void k_scan_one(const int* __restrict__ in, int n, unsigned long long* __restrict__ agg, int nb, int* __restrict__ out, int* __restrict__ total_out, unsigned seq) {
    __shared__ int s_w[16], s_base, s_bad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    int4 e[SCAN_PER / 4]; int v[SCAN_PER / 4];
    int t = 0;
#pragma unroll
    for (int q = 0; q < SCAN_PER / 4; ++q) {
        e[q] = scan_load4(in, (long long)b * SCAN_TILE + ((long long)q * 1024 + threadIdx.x) * 4, n);
        v[q] = (e[q].x + e[q].y) + (e[q].z + e[q].w); t += v[q];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) s_w[wave] = t;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    int mine = 0;
    if (threadIdx.x == 0) {
        for (int w = 0; w < 16; ++w) mine += s_w[w];
        __hip_atomic_store(&agg[b], ((unsigned long long)seq << 32) | (unsigned)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                        // (s_w is reused below)
    int p = 0;
    for (int i = threadIdx.x; i < b; i += 1024) {
        unsigned long long a = 0;
        int spins = 0;
        for (; spins < (1 << 22); ++spins) { a = __hip_atomic_load(&agg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((unsigned)(a >> 32) == seq) break; __builtin_amdgcn_s_sleep(1); }
        if ((unsigned)(a >> 32) != seq) s_bad = 1; else p += (int)(unsigned)a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
    if (lane == 0) s_w[wave] = p;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += s_w[w]; s_base = s; if (b == nb - 1) *total_out = s_bad ? -1 : s + mine; }
    __syncthreads();
    int run = s_base;
#pragma unroll
    for (int q = 0; q < SCAN_PER / 4; ++q) {
        const long long i = (long long)b * SCAN_TILE + ((long long)q * 1024 + threadIdx.x) * 4;
        int inc = v[q];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
        __syncthreads();                                    // s_w of the previous round has been read
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int off = run + inc - v[q], tot = 0;
        for (int w = 0; w < 16; ++w) { const int sw = s_w[w]; if (w < wave) off += sw; tot += sw; }
        const int4 o4 = make_int4(off, off + e[q].x, off + e[q].x + e[q].y, off + e[q].x + e[q].y + e[q].z);
        if (i + 3 < n) *reinterpret_cast<int4*>(out + i) = o4;
        else { if (i < n) out[i] = o4.x; if (i + 1 < n) out[i + 1] = o4.y; if (i + 2 < n) out[i + 2] = o4.z; }
        run += tot;
    }
}
static std::atomic<unsigned> g_scan_seq{0};
extern "C" long long vo_scan_call_number(long long set_to) { if (set_to >= 0) g_scan_seq = (unsigned)set_to; return (long long)g_scan_seq.load(); }      // (test tap: include/vo_hip.h)
int vo_scan_i32(hipStream_t st, const int* in, int n, int* bsum, int* out, int* total) {        // n <= 16 Mi; bsum: 4096 bytes, 256-byte aligned, zeroed when allocated
    const int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb > 1024) return VO_E_UNSUPPORTED;
    if (nb <= 512) {
        unsigned seq = ++g_scan_seq;
        if (seq == 0) seq = ++g_scan_seq;                       // (0 is what a fresh buffer holds)
        hipLaunchKernelGGL(k_scan_one, dim3(std::max(nb, 1)), dim3(1024), 0, st, in, n, (unsigned long long*)bsum, std::max(nb, 1), out, total, seq);
        return VO_OK;
    }
    hipLaunchKernelGGL(k_scan_blocksum, dim3(std::max(nb, 1)), dim3(1024), 0, st, in, n, bsum);
    hipLaunchKernelGGL(k_scan_final, dim3(std::max(nb, 1)), dim3(1024), 0, st, in, n, (const int*)bsum, std::max(nb, 1), out, total);
    // (windows beyond 8 Mi elements only.)  The raw tile sums must not stay: a later one-launch scan reads a word of this buffer as
    // (call number << 32 | tile total) and would take a stale sum whose upper half happens to equal its call number for a published total
    HIP_TRY(hipMemsetAsync(bsum, 0, 4096, st));
    return VO_OK;
}

// cut the graph of `free_kf` out of `t`'s tables into c's BA slab; returns when the arrays are complete (t may change afterwards)
static int ba_resident_cut(vo_ctx* c, vo_ctx* t, const int32_t* free_kf, int nf, double huber_delta, double chi2_th) {
    hipStream_t st = c->stream;
    if (!t->d_obs_kf || t->n_kf <= 0) return VO_E_STATE;
    if (!c->resident) c->resident = new BaResident();
    BaResident& R = *c->resident;
    R.ready = false; R.solved = false;
    if (nf == 0) { R.np = 0; R.nf = 0; R.nx = 0; R.ne = 0; R.n_fixed = 0; R.ready = true; return VO_OK; }      // nothing to optimise (and no launch with an empty grid)
    const int nkf = t->n_kf, D = 6 * nf;
    const long long no = t->n_obs;
    // the window of this cut (see CutTabs): entered behind the oldest point any free keyframe observes
    long long obs_lo = no; int map_lo = std::max(t->map_hi, 1) - 1;
    for (int i = 0; i < nf; ++i)
        if (free_kf[i] >= 0 && free_kf[i] < (int)t->kf_reach.size() && t->kf_reach[free_kf[i]].obs_lo >= 0) {
            obs_lo = std::min(obs_lo, (long long)t->kf_reach[free_kf[i]].obs_lo); map_lo = std::min(map_lo, (int)t->kf_reach[free_kf[i]].slot_lo);
        }
    obs_lo &= ~63ll; map_lo = std::max(0, map_lo) & ~63;      // (aligned: the scans read int4s)
    const int mh = std::max(t->map_hi, 1) - map_lo;          // slots the cut can see
    if (mh >= 16 * 1024 * 1024 || nkf > 1024 * 1024) return VO_E_UNSUPPORTED;         // scan_i32's range
    if ((CH_NB * CH_NB + (size_t)(CH_NB + 1) * (D + 1) + 2 * (size_t)D) * sizeof(double) > 158 * 1024 || nf > VO_BA_RESIDENT_MAX_FREE) return VO_E_UNSUPPORTED;
    CutFree F; F.n = nf;
    for (int i = 0; i < nf; ++i) {
        if (free_kf[i] < 0 || free_kf[i] >= nkf) return VO_E_INVALID;
        for (int k = 0; k < i; ++k) if (free_kf[k] == free_kf[i]) return VO_E_INVALID;       // a keyframe is free once
        F.kf[i] = free_kf[i];
    }
    // ---- cut scratch (lives until c's next cut: the solve reads pt_start, pose_kf and point_slots from it)
    //      zeroed per cut: [fixed_flag nkf][pt_flag mh][cnt mh + 1][fill mh]     written by kernels: [kf_idx][pose_kf][pidx][pt_start mh + 1][point_slots][totals]
    size_t co = 0;
    auto cc = [&](size_t bytes) { size_t o = co; co += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_ffl = cc(4 * (size_t)nkf), o_pfl = cc(4 * (size_t)mh), o_cnt = cc(4 * (size_t)(mh + 1)), o_fil = cc(4 * (size_t)mh), zero_end = co;
    const size_t o_kfi = cc(4 * (size_t)nkf), o_pkf = cc(4 * (size_t)(nkf + VO_BA_RESIDENT_MAX_FREE)), o_pid = cc(4 * (size_t)mh), o_pst = cc(4 * (size_t)(mh + 1)), o_psl = cc(4 * (size_t)mh),
                 o_tot = cc(64);
    if (co > c->d_cut_bytes) {
        if (c->d_cut) { (void)hipStreamSynchronize(st); vo_ba_engine_drain(c); (void)hipFree(c->d_cut); }
        c->d_cut = nullptr; c->d_cut_bytes = 0;
        if (hipMalloc(&c->d_cut, co + co / 2) != hipSuccess) return VO_E_NOMEM;
        c->d_cut_bytes = co + co / 2;
        HIP_TRY(hipMemsetAsync(c->d_cut, 0, c->d_cut_bytes, st));      // (k_scan_one's published totals: see there)
    }
    if (!c->d_cut_sync) {                                   // (see vo_internal.h)
        if (hipMalloc(&c->d_cut_sync, 4096) != hipSuccess) { c->d_cut_sync = nullptr; return VO_E_NOMEM; }
        HIP_TRY(hipMemsetAsync(c->d_cut_sync, 0, 4096, st));
    }
    uint8_t* cb = (uint8_t*)c->d_cut;
    int* kf_idx = (int*)(cb + o_kfi); int* fixed_flag = (int*)(cb + o_ffl); int* pose_kf = (int*)(cb + o_pkf); int* pt_flag = (int*)(cb + o_pfl); int* pidx = (int*)(cb + o_pid);
    int* cnt = (int*)(cb + o_cnt); int* fill = (int*)(cb + o_fil); int* pt_start = (int*)(cb + o_pst); int* point_slots = (int*)(cb + o_psl); int* bsum = (int*)c->d_cut_sync;
    int* tot = (int*)(cb + o_tot);
    int* h = (int*)vo_stage(c, 4096);
    if (!h) return VO_E_NOMEM;
    HIP_TRY(hipStreamSynchronize(st));
    CutTabs T{t->d_obs_kf, t->d_obs_mp, t->d_obs_uv, t->d_obs_alive, no, t->d_map_flags, t->d_map_pos, t->d_kf_pose, nkf, map_lo + mh, obs_lo, map_lo};
    R.win_obs = no - obs_lo; R.win_slots = mh;
    const int n_zero16 = (int)(zero_end / 16);              // (every carve is a multiple of 256 bytes)
    hipLaunchKernelGGL(k_cut_init, dim3(std::max((std::max(nkf, nf) + 255) / 256, std::min(64, (n_zero16 + 255) / 256))), dim3(256), 0, st, nkf, F, kf_idx, pose_kf, (int4*)cb, n_zero16);
    const int seq = ++c->cut_seq;
    h[132] = 0; h[133] = 0;
    const int gO = (int)((no - obs_lo + 255) / 256);
    if (gO) hipLaunchKernelGGL(k_cut_points, dim3(gO), dim3(256), 0, st, T, kf_idx, pt_flag);
    int rc = vo_scan_i32(st, pt_flag, mh, bsum, pidx, tot);                        // dense point index, nx
    if (rc) return rc;
    if (gO) hipLaunchKernelGGL(k_cut_count, dim3(gO), dim3(256), 0, st, T, kf_idx, pt_flag, pidx, cnt, fixed_flag);
    // pt_start[0 .. nx] over the dense indices (cnt is zero from nx on; one spare entry so that pt_start[nx] exists when every slot of
    // the map is in the graph), ne
    if ((rc = vo_scan_i32(st, cnt, mh + 1, bsum, pt_start, tot + 1))) return rc;
    hipLaunchKernelGGL(k_cut_fixed_scan, dim3(1), dim3(1024), 0, st, nkf, nf, fixed_flag, kf_idx, pose_kf, tot + 2, (const int*)tot, h + 128, h + 132, seq);      // (sizes to the host without a copy or a blocking wait)
    HIP_TRY(hipGetLastError());
    // The slab is carved for UPPER BOUNDS of the three sizes -- every slot of the window a point, every observation of the window an edge, every
    // keyframe a pose (the graph is typically half of its window) -- so that the host does not wait for the sizes in the middle of the chain:
    // the kernels below take nx, ne and n_fixed from `tot` on the device, their grids cover the bounds (workgroups behind the real sizes leave
    // at once), and the host reads everything -- sizes and list lengths -- behind the last kernel.  When the bounds ask for more than the
    // context's slab budget (vo_ba_resident_set_slab_budget; a revisit whose window spans most of the tables, with ~100 free keyframes: GBs of
    // pair scratch for a graph a fraction of that size), the cut waits for the sizes instead and carves exactly.
    bool sizes_first = false;
    int nx = 0, ne = 0, n_fixed = 0, np = nf;
    int nx_c = mh, np_c = std::max(nkf, nf);
    long long ne_cl = std::max<long long>(no - obs_lo, 1);
    const int nb_all = nf * (nf + 1) / 2;
    int ne_c = 0, chunks = 0;
    size_t off = 0, pairs_ub = 0, slices_cap = 0;
    size_t o_poses = 0, o_pts = 0, o_epose = 0, o_ept = 0, o_euv = 0, o_pe = 0, o_qs = 0, o_qe = 0, o_pspt = 0, o_poses_n = 0, o_pts_n = 0, o_act = 0, o_flags = 0, o_eobs = 0, o_earr = 0, o_ncull = 0, o_cull = 0, o_hist = 0, o_offs = 0, o_ptot = 0, o_pcnt = 0, o_poff = 0, o_pn = 0, o_Hpp = 0, o_bp = 0, o_Hll = 0, o_bl = 0, o_scal = 0, o_partU = 0, o_partC = 0, o_W = 0, o_S = 0, o_bs2 = 0, o_Hinv = 0, o_dl = 0, o_blk = 0, o_pairs = 0, o_ppt = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    auto layout = [&]() {
        off = 0;
        ne_c = (int)std::min<long long>(ne_cl, 1ll << 30);
        // ---- BA slab (same layout as vo_ba_run's, filled by kernels instead of an upload)
        o_poses = carve(96 * (size_t)np_c), o_pts = carve(24 * (size_t)nx_c);
        o_epose = carve(4 * (size_t)ne_c), o_ept = carve(4 * (size_t)ne_c), o_euv = carve(8 * (size_t)ne_c), o_pe = carve(4 * (size_t)ne_c);
        o_qs = carve(4 * (size_t)(nf + 1)), o_qe = carve(4 * (size_t)ne_c), o_pspt = carve(4 * (size_t)ne_c + 4);
        o_poses_n = carve(96 * (size_t)np_c), o_pts_n = carve(24 * (size_t)nx_c), o_act = carve(ne_c), o_flags = carve(ne_c);
        chunks = (ne_c + PS_CHUNK - 1) / PS_CHUNK;
        o_eobs = carve(8 * (size_t)ne_c), o_earr = carve(8 * (size_t)ne_c), o_ncull = carve(64), o_cull = carve(8 * (size_t)ne_c);
        o_hist = carve(4 * (size_t)chunks * nf), o_offs = carve(4 * (size_t)chunks * nf), o_ptot = carve(4 * VO_BA_RESIDENT_MAX_FREE);
        o_pcnt = carve(4 * (size_t)std::max(nb_all, 1)), o_poff = carve(4 * (size_t)std::max(nb_all, 1)), o_pn = carve(16);
        o_Hpp = carve(288 * (size_t)nf), o_bp = carve(8 * (size_t)D), o_Hll = carve(72 * (size_t)nx_c), o_bl = carve(24 * (size_t)nx_c), o_scal = carve(64);
        o_partU = carve(24 * ((size_t)(nx_c + 63) / 64 + 1)), o_partC = carve(8 * ((size_t)(ne_c + 255) / 256 + 1));
        o_W = carve(std::max<size_t>(144 * (size_t)ne_c, 16 * (size_t)ne_c + 192 * (size_t)nx_c + 2048)), o_S = carve(std::max<size_t>(8 * (size_t)D * D, 8 * ba_tile_doubles(D)) + 1024), o_bs2 = carve(8 * (size_t)D), o_Hinv = carve(72 * (size_t)nx_c), o_dl = carve(std::max<size_t>(24 * (size_t)nx_c, 8 * (size_t)(D + 8)));      // (dl: the solution and, behind it, lambda / ok / cur / ni for the update workgroups of a fused launch)
        // pair lists: a point seen by m free poses gives m (m + 1) / 2 <= m (nf + 1) / 2 pairs, so ne (nf + 1) / 2 bounds them before the
        // per-pose lists exist; slices: one per BA_SLICE pairs plus a partial one per block
        pairs_ub = (size_t)ne_c * (size_t)(nf + 1) / 2 + 1; slices_cap = pairs_ub / BA_SLICE + (size_t)nb_all + 1;
        o_blk = carve(sizeof(BaBlock) * slices_cap), o_pairs = carve(sizeof(int2) * pairs_ub), o_ppt = carve(sizeof(int32_t) * pairs_ub);
    };
    if (ne_cl > (1ll << 30)) return VO_E_UNSUPPORTED;
    layout();
    if (off > (size_t)c->cut_slab_budget) {
        sizes_first = true;
        if (!vo_spin_word(h + 132, seq, 2000)) HIP_TRY(hipStreamSynchronize(st));
        nx = h[128]; ne = h[129]; n_fixed = h[130]; np = nf + n_fixed;
        if (nx < 0 || ne < 0) return VO_E_DEVICE;               // (a scan gave up waiting for one of its tiles)
        R.np = np; R.nf = nf; R.nx = nx; R.ne = ne; R.n_fixed = n_fixed;
        if (nx == 0 || ne == 0 || nf == 0) { R.ready = true; return VO_OK; }         // nothing to optimise
        nx_c = nx; ne_cl = ne; np_c = np;
        layout();
    }
    // (may reallocate: nothing of this problem lives in the slab yet.  The bounds follow the window, which keeps growing for the first ~200 frames of a
    // stream: a slab that has to grow takes twice what is asked for, or the short runs meet a hipFree + hipMalloc in every other cut)
    if ((rc = vo_scratch(c, (!sizes_first && off > c->d_ba_bytes) ? 2 * off : off))) return rc;
    uint8_t* base = (uint8_t*)c->d_ba;
    int32_t* e_pose = (int32_t*)(base + o_epose); int32_t* e_pt = (int32_t*)(base + o_ept); float* e_uv = (float*)(base + o_euv);
    long long* e_obs = (long long*)(base + o_eobs);
    long long* e_arr = (long long*)(base + o_earr);
    hipLaunchKernelGGL(k_cut_fill, dim3((int)((std::max<long long>(no - obs_lo, mh) + 255) / 256)), dim3(256), 0, st, T, pt_flag, pidx, pt_start, fill, e_arr, point_slots);
    hipLaunchKernelGGL(k_cut_emit, dim3((std::max(ne_c, std::max(np_c * 12, nx_c * 3)) + 255) / 256), dim3(256), 0, st, T, nf, (const int*)tot, pidx, pt_start, kf_idx, pose_kf, point_slots,
                       (const long long*)e_arr, e_obs, e_pose, e_pt, e_uv, (int32_t*)(base + o_pe), base + o_act, base + o_flags, (double*)(base + o_poses),
                       (double*)(base + o_poses_n), (double*)(base + o_pts));
    hipLaunchKernelGGL(k_ps_hist, dim3(chunks), dim3(PS_CHUNK), 0, st, (const int*)(tot + 1), nf, e_pose, (int*)(base + o_hist));
    hipLaunchKernelGGL(k_ps_offsets, dim3(nf), dim3(256), 0, st, chunks, nf, (const int*)(base + o_hist), (int*)(base + o_offs), (int*)(base + o_ptot));
    hipLaunchKernelGGL(k_ps_fill, dim3(chunks), dim3(PS_CHUNK), 0, st, (const int*)(tot + 1), nf, e_pose, e_pt, (const int*)(base + o_offs), (const int*)(base + o_ptot), (int*)(base + o_qs),
                       (int32_t*)(base + o_qe), (int32_t*)(base + o_pspt));
    hipLaunchKernelGGL(k_cut_report, dim3(1), dim3(256), 0, st, (const int*)(base + o_qs), nf + 1, h + 256, h + 133, seq, (int*)(base + o_poff));      // the list lengths; behind it every input of `t` has been gathered
    // the pair plan goes out BEFORE the host looks at the list lengths (with the largest LDS a list may need): the stream works through it
    // while the host wakes up, checks the lengths and fills in the descriptor (~25 us per cut)
    BaPairPlan Q;
    Q.ps_start = (const int32_t*)(base + o_qs); Q.ps_edges = (const int32_t*)(base + o_qe); Q.ps_pt = (const int32_t*)(base + o_pspt); Q.nf = nf;
    Q.cnt = (int*)(base + o_pcnt); Q.off = (int*)(base + o_poff); Q.n_slices = (int*)(base + o_pn); Q.n_pairs = (int*)(base + o_pn) + 1;
    Q.blocks = (BaBlock*)(base + o_blk); Q.pairs = (int2*)(base + o_pairs); Q.pair_pt = (int32_t*)(base + o_ppt);
    if (!R.ev_arrays) HIP_TRY(hipEventCreateWithFlags(&R.ev_arrays, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(R.ev_arrays, st));               // (the first linearisation needs no pairs: it may start here)
    Q.lds_cap = PAIR_LDS_CAP;
    hipLaunchKernelGGL(k_ba_pairs_one, dim3(nb_all), dim3(256), 4 * (size_t)PAIR_LDS_CAP, st, Q, nb_all);      // (a list beyond the cap is searched in global memory)
    if (!R.ev) HIP_TRY(hipEventCreateWithFlags(&R.ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(R.ev, st));
    if (!vo_spin_word(h + 133, seq, 2000)) HIP_TRY(hipEventSynchronize(R.ev_arrays));      // from here on `t` may change: every input has been gathered (and the list lengths are in h)
    if (!sizes_first) {                                     // (k_cut_fixed_scan's words were performed before k_cut_report started)
        nx = h[128]; ne = h[129]; n_fixed = h[130]; np = nf + n_fixed;
        if (nx < 0 || ne < 0 || nx > nx_c || ne > ne_c || np > np_c) return VO_E_DEVICE;
        R.np = np; R.nf = nf; R.nx = nx; R.ne = ne; R.n_fixed = n_fixed;
        if (nx == 0 || ne == 0 || nf == 0) { R.ready = true; return VO_OK; }         // nothing to optimise (the kernels above found nothing to do)
    }
    const int* ps_start = h + 256;
    int max_len = 0, npairs = 0, slices_ub = 0;
    for (int j = 0; j < nf; ++j) max_len = std::max(max_len, ps_start[j + 1] - ps_start[j]);
    // (a list longer than PAIR_LDS_CAP -- config 5's 11 k edges per pose -- stays in global memory for the binary searches: slower per pair, same lists)
    for (int j1 = 0; j1 < nf; ++j1) for (int j2 = j1; j2 < nf; ++j2) {
        const int m = std::min(ps_start[j1 + 1] - ps_start[j1], ps_start[j2 + 1] - ps_start[j2]);
        npairs += m; slices_ub += (m + BA_SLICE - 1) / BA_SLICE;
    }
    if ((size_t)npairs > pairs_ub || (size_t)slices_ub > slices_cap) return VO_E_OVERFLOW;
    BaDev B;
    B.n_poses = np; B.n_free = nf; B.n_points = nx; B.n_edges = ne; B.D = D; B.s_tiles = ba_use_tiles(D);
    B.n_blocks = slices_ub;                                 // launch bound; the Schur kernel stops at *n_slices, which the plan kernels below write
    B.n_slices = (const int*)(base + o_pn);
    B.ps_start = (const int32_t*)(base + o_qs); B.ps_edges = (const int32_t*)(base + o_qe); B.blocks = (const BaBlock*)(base + o_blk); B.pairs = (const int2*)(base + o_pairs); B.pair_pt = (const int32_t*)(base + o_ppt); B.ps_pt = (const int32_t*)(base + o_pspt);
    B.posesA = (double*)(base + o_poses); B.ptsA = (double*)(base + o_pts); B.posesB = (double*)(base + o_poses_n); B.ptsB = (double*)(base + o_pts_n);
    B.ctl = nullptr;
    B.partU = (double*)(base + o_partU); B.partC = (double*)(base + o_partC); B.nU = (nx + 63) / 64;
    B.e_pose = e_pose; B.e_pt = e_pt; B.e_uv = e_uv;
    B.active = base + o_act; B.flags = base + o_flags; B.pt_start = (const int32_t*)pt_start; B.pt_edges = (const int32_t*)(base + o_pe);
    B.Hpp = (double*)(base + o_Hpp); B.bp = (double*)(base + o_bp); B.Hll = (double*)(base + o_Hll); B.bl = (double*)(base + o_bl); B.scal = (double*)(base + o_scal);
    B.W = (double*)(base + o_W); B.S = (double*)(base + o_S); B.bs = (double*)(base + o_bs2); B.Hinv = (double*)(base + o_Hinv); B.dl = (double*)(base + o_dl);
    B.cam = BaCam{(double)c->p.fx, (double)c->p.fy, (double)c->p.cx, (double)c->p.cy};
    B.delta = huber_delta; B.chi2_th = chi2_th; B.gp = (nx + 63) / 64; B.edges_by_point = 1;
    B.e_obs = e_obs; B.cull = (long long*)(base + o_cull); B.ncull = (int*)(base + o_ncull); B.cull_cap = ne; B.cull_host = nullptr; B.cull_host_cap = 0;      // (the engine points cull_host at its slot's pinned list)
    B.gen1 = 0; B.shard_rank = 0; B.shard_world = 1; B.upc_ovf = sizeof(double) * (24 * (size_t)np + (size_t)D + UPC_LDS_EXTRA) <= 150 * 1024 ? 1 : 0; B.upc_ppw = ba_upc_ppw(nx); B.xbuf = nullptr;
    HIP_TRY(hipGetLastError());
    R.nblk_launch = slices_ub; R.npairs = npairs;
    R.B = B;
    R.d_point_slots = point_slots; R.d_pose_kf = pose_kf; R.d_e_obs = e_obs; R.d_ncull = (int*)(base + o_ncull); R.d_cull = (long long*)(base + o_cull); R.cull_cap = ne;
    R.ready = true;
    return VO_OK;
}

extern "C" int vo_local_ba_resident_cut(vo_ctx* c, vo_ctx* t, const int32_t* free_kf, int n_free, double huber_delta, double chi2_th,
                                        int32_t* n_points, int32_t* n_fixed, int32_t* n_edges) {
    if (c && c->ba_shard_world > 1) return VO_E_UNSUPPORTED;      // (e-3 shards vo_local_ba's explicit problems; the device graph cut is one rank's)
    if (!c || !t || n_free < 0 || (n_free && !free_kf) || c->device != t->device) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    const bool trace = vo_trace_level() != 0;
    const double t0 = trace ? tnow() : 0.0;
    const int rc = ba_resident_cut(c, t, free_kf, n_free, huber_delta, chi2_th);
    if (trace) { static double a = 0; static int n = 0; a += tnow() - t0; if (++n % 10 == 0 && c->resident) fprintf(stderr, "[vo_trace] resident cut avg ms: %.3f (this one: window %lld observations / %lld slots -> %d points, %d edges, %d free + %d fixed poses)\n", a / n, c->resident->win_obs, c->resident->win_slots, c->resident->nx, c->resident->ne, c->resident->nf, c->resident->n_fixed); }
    if (rc == VO_OK && c->resident) { if (n_points) *n_points = c->resident->nx; if (n_fixed) *n_fixed = c->resident->n_fixed; if (n_edges) *n_edges = c->resident->ne; }
    return rc;
}

extern "C" int vo_local_ba_resident_solve(vo_ctx* c, int it_robust, int it_plain, vo_ba_resident_result* out) {
    if (!c || !out || !c->resident || !c->resident->ready || !out->culled_obs) return VO_E_INVALID;
    const bool deferred = !out->poses && !out->points && !out->point_slots;       // the result stays on the device: vo_local_ba_resident_merge / _fetch
    if (!deferred && (!out->poses || !out->points || !out->point_slots)) return VO_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    BaResident& R = *c->resident;
    hipStream_t st = c->stream;
    const int np = R.np, nf = R.nf, nx = R.nx, ne = R.ne, D = 6 * nf;
    out->n_points = nx; out->n_fixed = R.n_fixed; out->n_edges = ne; out->n_culled = 0; out->chi2_initial = out->chi2_final = 0; out->lm_iters = 0;
    R.ready = false;
    if (nx == 0 || ne == 0 || nf == 0) return VO_OK;
    if (!deferred && nx > out->cap_points) return VO_E_OVERFLOW;
    vo_ba_problem pr;
    memset(&pr, 0, sizeof(pr));
    pr.n_poses = np; pr.n_free = nf; pr.n_points = nx; pr.n_edges = ne; pr.huber_delta = R.B.delta; pr.chi2_th = R.B.chi2_th; pr.it_robust = it_robust; pr.it_plain = it_plain;
    BaJob job;
    job.c = c; job.in = &pr; job.out = nullptr; job.B = R.B; job.wait_ev = R.ev_arrays; job.wait_pairs = R.ev;
    job.grid_lin = (nx + 63) / 64 + nf * PSPLIT; job.grid_initS = (std::max(D * D, nx) + 255) / 256; job.grid_upd = (nx + 63) / 64 + (np + 255) / 256;
    job.grid_e = (ne + 255) / 256; job.grid_c = (ne + 1023) / 1024; job.grid_maxdiag = (D + 3 * nx + 255) / 256;
    job.lds = job.B.s_tiles ? ch2_lds_bytes(D) : D <= 192 ? sizeof(double) * (CH_NB * CH_NB + (size_t)(D + 1) * (D + 2) / 2 + 2 * (size_t)D)
                       : sizeof(double) * (CH_NB * CH_NB + (size_t)(CH_NB + 1) * (D + 1) + 2 * (size_t)D);
    const bool trace = vo_trace_level() != 0;
    const double t0 = trace ? tnow() : 0.0;
    BaEngine* E = ba_engine_of(c);
    if (!E) return VO_E_STATE;
    int rc = ba_engine_solve(E, &job);
    if (rc) return rc;
    const double t1 = trace ? tnow() : 0.0;
    const BaDev& B = R.B;
    int* h = (int*)vo_stage(c, 4096);
    if (!h) return VO_E_NOMEM;
    if (job.n_culled >= 0 && job.n_culled <= BA_CULL_HOST) {
        // the engine's final k_ba_round left the culled observations' ids in the device list (for the merge kernel) and in pinned host memory
        out->n_culled = job.n_culled;
        const int take = std::min(out->n_culled, out->cap_culled);
        if (take > 0) { memcpy(out->culled_obs, job.culled.data(), 8 * (size_t)take); std::sort(out->culled_obs, out->culled_obs + take); }
        if (!deferred) {
            HIP_TRY(hipMemcpyAsync(out->poses, job.cur_buf ? B.posesB : B.posesA, 96 * (size_t)nf, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out->points, job.cur_buf ? B.ptsB : B.ptsA, 24 * (size_t)nx, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out->point_slots, R.d_point_slots, 4 * (size_t)nx, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    } else {
        // (more culled observations than the pinned list holds: list kernel + read-back)
        HIP_TRY(hipMemsetAsync(R.d_ncull, 0, 4, st));
        hipLaunchKernelGGL(k_culled_list, dim3((ne + 255) / 256), dim3(256), 0, st, ne, (const uint8_t*)B.flags, (const long long*)R.d_e_obs, R.d_ncull, R.d_cull, R.cull_cap);
        HIP_TRY(hipMemcpyAsync(h, R.d_ncull, 4, hipMemcpyDeviceToHost, st));
        if (!deferred) {
            HIP_TRY(hipMemcpyAsync(out->poses, job.cur_buf ? B.posesB : B.posesA, 96 * (size_t)nf, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out->points, job.cur_buf ? B.ptsB : B.ptsA, 24 * (size_t)nx, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out->point_slots, R.d_point_slots, 4 * (size_t)nx, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        out->n_culled = h[0];
        const int take = std::min(out->n_culled, out->cap_culled);
        if (take > 0) {
            HIP_TRY(hipMemcpy(out->culled_obs, R.d_cull, 8 * (size_t)take, hipMemcpyDeviceToHost));
            std::sort(out->culled_obs, out->culled_obs + take);                     // arrival order of an atomic append -> ascending observation id
        }
    }
    out->chi2_initial = job.chi0; out->chi2_final = job.chi_final; out->lm_iters = job.iters; out->n_pairs = job.n_pairs;
    HIP_TRY(hipGetLastError());
    R.solved = true; ++R.solve_seq; R.cur_buf = job.cur_buf; R.n_culled = out->n_culled; R.chi0 = job.chi0; R.chi_final = job.chi_final; R.lm_iters = job.iters;
    if (trace) { static double a = 0, b = 0, st = 0; static int n = 0; a += t1 - t0; b += tnow() - t1; st += job.steps; if (++n % 10 == 0) fprintf(stderr, "[vo_trace] resident solve avg ms: optimise %.3f (%.1f step launches) result %.3f (D=%d edges=%d)\n", a / n, st / n, b / n, D, ne);
                 static const bool each = vo_trace_level() >= 2; if (each) fprintf(stderr, "[vo_trace] BA %d: D=%d points=%d edges=%d optimise %.3f ms\n", n, D, nx, ne, t1 - t0); }
    return out->n_culled > out->cap_culled ? VO_E_OVERFLOW : VO_OK;
}

// The result of the last solve goes into t's tables on the device (t's stream: the tracker's own work is ordered around it; c's
// stream waits for it, so the next cut of `c` reads merged tables and may reuse the slab) and into c's staging buffer.
static int ba_resident_merge(vo_ctx* c, vo_ctx* t, bool ledger, int32_t* pair_a, int32_t* pair_b, int cap_pairs, int32_t* n_pairs, double* poses, int cap_poses);
extern "C" int vo_local_ba_resident_merge(vo_ctx* c, vo_ctx* t) { return ba_resident_merge(c, t, false, nullptr, nullptr, 0, nullptr, nullptr, 0); }
extern "C" int vo_local_ba_resident_merge_ledger(vo_ctx* c, vo_ctx* t, int32_t* pair_a, int32_t* pair_b, int cap_pairs, int32_t* n_pairs, double* poses, int cap_poses) {
    if (!n_pairs || cap_pairs < 0 || (cap_pairs && (!pair_a || !pair_b)) || cap_poses < 0 || (cap_poses && !poses)) return VO_E_INVALID;
    *n_pairs = 0;
    return ba_resident_merge(c, t, true, pair_a, pair_b, cap_pairs, n_pairs, poses, cap_poses);
}
static int ba_resident_merge(vo_ctx* c, vo_ctx* t, bool ledger, int32_t* pair_a, int32_t* pair_b, int cap_pairs, int32_t* n_pairs, double* poses, int cap_poses) {
    if (!c || !t || c->device != t->device || !c->resident || !c->resident->solved) return VO_E_STATE;
    const bool resume = ledger && c->resident->ledger_pending;      // the call that follows a VO_E_OVERFLOW
    if (c->resident->merged_seq == c->resident->solve_seq && !resume) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    BaResident& R = *c->resident;
    int* h_pa = nullptr; int* h_pb = nullptr; int* h_np = nullptr; double* h_poses = nullptr; int h_cap = 0;
    if (ledger) {
        if (!t->d_obs_link || !t->d_pt_last) return VO_E_STATE;
        int rc = vo_kf_host_pairs(t, &h_pa, &h_pb, &h_cap, &h_np, &h_poses);
        if (rc) return rc;
        if (!resume) *h_np = 0;
    }
    const int commit_cap = ledger ? std::min(h_cap, cap_pairs) : 0;      // mode 0 of k_merge_ledger clears its marks only if this many pairs suffice
    const int nf = R.nf, nx = R.nx;
    R.st_nf = nf; R.st_nx = nx; R.st_ne = R.ne; R.st_fixed = R.n_fixed; R.st_culled = R.n_culled; R.st_iters = R.lm_iters; R.st_chi0 = R.chi0; R.st_chi1 = R.chi_final;
    R.merged_seq = R.solve_seq; R.has_stage = true;
    if (nx == 0 || R.ne == 0 || nf == 0) return VO_OK;
    if (!t->d_obs_alive || !t->d_kf_pose || R.n_culled > R.cull_cap) return VO_E_STATE;
    const size_t o_pts = (96 * (size_t)nf + 255) & ~(size_t)255, o_sl = o_pts + ((24 * (size_t)nx + 255) & ~(size_t)255), total = o_sl + 4 * (size_t)nx;
    if (!resume && total > R.stage_bytes) {
        if (R.d_stage) { (void)hipStreamSynchronize(t->stream); if (R.fetch_stream) (void)hipStreamSynchronize(R.fetch_stream); (void)hipFree(R.d_stage); }
        R.d_stage = nullptr; R.stage_bytes = 0;
        if (hipMalloc(&R.d_stage, total + total / 2) != hipSuccess) return VO_E_NOMEM;
        R.stage_bytes = total + total / 2;
    }
    uint8_t* sb = (uint8_t*)R.d_stage;
    const BaDev& B = R.B;
    const int n = std::max(std::max(nx, 12 * nf), R.n_culled);
    auto ledger_launch = [&](int mode, int cap, int i0, int i1) {
        hipLaunchKernelGGL(k_merge_ledger, dim3(1), dim3(256), 0, t->stream, (const int*)R.d_ncull, (const long long*)R.d_cull, R.cull_cap, (const int32_t*)t->d_obs_kf, (const int32_t*)t->d_obs_mp,
                           t->d_obs_alive, (const int2*)t->d_obs_link, (const int32_t*)t->d_pt_last, t->d_map_flags, h_pa, h_pb, cap, h_np, mode, i0, i1);
    };
    if (ledger && !resume) ledger_launch(0, commit_cap, 0, 0);      // first: a point that loses its last observation here keeps its position (src/backend.cpp:191: outliers are skipped)
    if (!resume) hipLaunchKernelGGL(k_ba_merge, dim3((n + 255) / 256), dim3(256), 0, t->stream, nf, nx, (const double*)(R.cur_buf ? B.posesB : B.posesA), (const double*)(R.cur_buf ? B.ptsB : B.ptsA),
                       (const int*)R.d_pose_kf, (const int32_t*)R.d_point_slots, (const int*)R.d_ncull, (const long long*)R.d_cull, R.cull_cap, t->d_map_pos, t->d_map_flags,
                       t->d_kf_pose, t->d_obs_alive, (double*)sb, (double*)(sb + o_pts), (int32_t*)(sb + o_sl), ledger ? 1 : 0, ledger ? h_poses : (double*)nullptr);
    if (!ledger) {                                          // (with the ledger the call waits for the tables' stream below: the next cut and a fetch are behind it anyway)
        if (!R.ev_merge) HIP_TRY(hipEventCreateWithFlags(&R.ev_merge, hipEventDisableTiming));
        if (!R.fetch_stream) HIP_TRY(vo_stream_create(&R.fetch_stream, -1));      // lowest class: three small copies behind an event wait must not sit in a queue a chain uses
        HIP_TRY(hipEventRecord(R.ev_merge, t->stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, R.ev_merge, 0));
        HIP_TRY(hipStreamWaitEvent(R.fetch_stream, R.ev_merge, 0));
    }
    HIP_TRY(hipGetLastError());
    if (ledger) {                                           // the caller's ledger needs the pairs before it picks the next free keyframes
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (!R.fetch_stream) HIP_TRY(vo_stream_create(&R.fetch_stream, -1));      // (a later _fetch of this result finds its stream; idle, it needs no event)
        const int total_pairs = resume ? R.ledger_total : *h_np;
        if (poses) memcpy(poses, h_poses, 96 * (size_t)std::min(nf, cap_poses));
        if (!resume && total_pairs <= commit_cap) {           // the usual case: one launch, the marks are gone
            if (total_pairs > 0) { memcpy(pair_a, h_pa, 4 * (size_t)total_pairs); memcpy(pair_b, h_pb, 4 * (size_t)total_pairs); }
            *n_pairs = total_pairs;
            return VO_OK;
        }
        // More pairs than the pinned block (or the caller's arrays) hold: the culled observations are still marked.  Too many for the caller: say how
        // many and wait for the repeated call; otherwise page them out -- walks over ranges of the culled list sized from the average pairs per entry,
        // halved when a range overflows the block -- and clear the marks at the end (this overflow used to end the run).
        R.ledger_pending = true; R.ledger_total = total_pairs;
        if (total_pairs > cap_pairs) { *n_pairs = total_pairs; return VO_E_OVERFLOW; }
        const int n_c = std::min(R.n_culled, R.cull_cap);
        int got = 0, range = (int)std::max<long long>(1, (long long)n_c * h_cap / (2 * (long long)std::max(total_pairs, 1)));
        for (int i0 = 0; i0 < n_c;) {
            const int i1 = std::min(n_c, i0 + range);
            ledger_launch(1, h_cap, i0, i1);
            HIP_TRY(hipStreamSynchronize(t->stream));
            const int cnt = *h_np;
            if (cnt > h_cap) { if (range == 1) return VO_E_UNSUPPORTED; range = std::max(1, range / 2); continue; }      // (one observation with more co-observers than the block holds pairs: 16 Ki keyframes on one point)
            if (got + cnt > cap_pairs) return VO_E_DEVICE;
            if (cnt > 0) { memcpy(pair_a + got, h_pa, 4 * (size_t)cnt); memcpy(pair_b + got, h_pb, 4 * (size_t)cnt); }
            got += cnt; i0 = i1;
            if (cnt < h_cap / 4) range *= 2;
        }
        ledger_launch(2, 0, 0, 0);
        HIP_TRY(hipGetLastError());
        R.ledger_pending = false;
        *n_pairs = got;
        if (got != total_pairs) return VO_E_DEVICE;
    }
    return VO_OK;
}

// the host's copy of a merged result (poses, point slots, positions; the culled observations came back with the solve)
extern "C" int vo_local_ba_resident_fetch(vo_ctx* c, vo_ba_resident_result* out) {
    if (!c || !out || !c->resident || !c->resident->has_stage || !out->poses || !out->points || !out->point_slots) return VO_E_STATE;
    HIP_TRY(hipSetDevice(c->device));
    BaResident& R = *c->resident;
    const int nf = R.st_nf, nx = R.st_nx;
    out->n_points = nx; out->n_fixed = R.st_fixed; out->n_edges = R.st_ne; out->n_culled = R.st_culled; out->chi2_initial = R.st_chi0; out->chi2_final = R.st_chi1; out->lm_iters = R.st_iters;
    if (nx == 0 || R.st_ne == 0 || nf == 0) return VO_OK;
    if (nx > out->cap_points) return VO_E_OVERFLOW;
    const size_t o_pts = (96 * (size_t)nf + 255) & ~(size_t)255, o_sl = o_pts + ((24 * (size_t)nx + 255) & ~(size_t)255);
    const uint8_t* sb = (const uint8_t*)R.d_stage;
    HIP_TRY(hipMemcpyAsync(out->poses, sb, 96 * (size_t)nf, hipMemcpyDeviceToHost, R.fetch_stream));
    HIP_TRY(hipMemcpyAsync(out->points, sb + o_pts, 24 * (size_t)nx, hipMemcpyDeviceToHost, R.fetch_stream));
    HIP_TRY(hipMemcpyAsync(out->point_slots, sb + o_sl, 4 * (size_t)nx, hipMemcpyDeviceToHost, R.fetch_stream));
    HIP_TRY(hipStreamSynchronize(R.fetch_stream));
    return VO_OK;
}

extern "C" int vo_local_ba_resident(vo_ctx* c, vo_ctx* t, const int32_t* free_kf, int n_free, double huber_delta, double chi2_th, int it_robust, int it_plain,
                                    vo_ba_resident_result* out) {
    int rc = vo_local_ba_resident_cut(c, t, free_kf, n_free, huber_delta, chi2_th, nullptr, nullptr, nullptr);
    if (rc) return rc;
    return vo_local_ba_resident_solve(c, it_robust, it_plain, out);
}

extern "C" int vo_ba_resident_set_slab_budget(vo_ctx* c, int64_t bytes) {
    if (!c || bytes <= 0) return VO_E_INVALID;
    c->cut_slab_budget = bytes;
    return VO_OK;
}

extern "C" int vo_ba_resident_window(vo_ctx* c, int64_t* observations_visited, int64_t* map_slots_visited) {
    if (!c || !observations_visited || !map_slots_visited) return VO_E_INVALID;
    if (!c->resident || c->resident->win_obs < 0) return VO_E_STATE;
    *observations_visited = c->resident->win_obs; *map_slots_visited = c->resident->win_slots;
    return VO_OK;
}

extern "C" int vo_ba_resident_graph(vo_ctx* c, vo_ctx* t, const int32_t* free_kf, int n_free, int32_t* n_poses, int32_t* pose_kf, int cap_poses, int32_t* n_points,
                                    int32_t* point_slots, int cap_points, int32_t* n_edges, int32_t* edge_pose, int32_t* edge_point, float* edge_uv, int64_t* edge_obs, int cap_edges) {
    if (!c || !t || !n_poses || !n_points || !n_edges) return VO_E_INVALID;
    int rc = vo_local_ba_resident_cut(c, t, free_kf, n_free, 1.0, 1.0, nullptr, nullptr, nullptr);
    if (rc) return rc;
    BaResident& R = *c->resident;
    HIP_TRY(hipStreamSynchronize(c->stream));               // the cut leaves its last kernels in flight
    *n_poses = R.np; *n_points = R.nx; *n_edges = R.ne;
    R.ready = false;
    if (R.nx == 0 || R.ne == 0) return VO_OK;
    const BaDev& B = R.B;
    if (pose_kf) HIP_TRY(hipMemcpy(pose_kf, R.d_pose_kf, 4 * (size_t)std::min(R.np, cap_poses), hipMemcpyDeviceToHost));
    if (point_slots) HIP_TRY(hipMemcpy(point_slots, R.d_point_slots, 4 * (size_t)std::min(R.nx, cap_points), hipMemcpyDeviceToHost));
    const size_t k = (size_t)std::min(R.ne, cap_edges);
    if (edge_pose) HIP_TRY(hipMemcpy(edge_pose, B.e_pose, 4 * k, hipMemcpyDeviceToHost));
    if (edge_point) HIP_TRY(hipMemcpy(edge_point, B.e_pt, 4 * k, hipMemcpyDeviceToHost));
    if (edge_uv) HIP_TRY(hipMemcpy(edge_uv, B.e_uv, 8 * k, hipMemcpyDeviceToHost));
    if (edge_obs) HIP_TRY(hipMemcpy(edge_obs, R.d_e_obs, 8 * k, hipMemcpyDeviceToHost));
    return VO_OK;
}
