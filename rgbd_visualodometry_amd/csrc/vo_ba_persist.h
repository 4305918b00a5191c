// vo_ba_persist.h -- the persistent Levenberg-Marquardt kernel of the local BA (included by vo_ba.hip).
//
// Reference: Backend::Optimize, src/backend.cpp:138-172 -- optimizer.optimize(10) with Huber kernels, chi2 cull,
// optimizer.optimize(10) without, final outlier flags.  The launch-per-phase path above spends ~5 launches per LM step;
// with ~20 dependent steps per BA the kernel boundaries and the lock-step over problems were the larger part of the BA's
// latency.  Here ONE launch runs the whole BA of one problem: both rounds, every LM step, both culls.
//
// Roles.  Workgroup 0 is the SOLVER (dense Cholesky of the reduced system, trial poses), workgroups 1..G-1 are WORKERS that
// own a contiguous range of points (balanced by edges).  One LM step:
//   workers  linearise their points (H_ll, b_l -> HB, write-through) and their share of the free poses' edge lists
//            (27 sums per (pose, part) -> hpart)                                   [only when the state changed]
//            B1: worker barrier (points, HB, hpart of every worker are now visible)
//            Schur slices (pair lists as in k_ba_schur_blocks; W_e is RECOMPUTED from the state instead of being read:
//            122 B per pair instead of ~370) -> f64 atomics into S (lower triangle only) and b_s;   arrive(FAN)
//   solver   wait(FAN): H_pp / b_p from hpart, Cholesky (clears S / b_s behind its load), dp, trial poses, record; publish(SOLVED)
//   workers  wait(SOLVED): back-substitution + trial point + trial chi2 of their own points in one pass; partials; arrive(CHI)
//   all      wait(CHI): every workgroup sums the partials in the same order and takes g2o's accept / lambda decision
//            itself (no broadcast hop); the LM state is replicated, never communicated.
// Hand-offs follow the recipe of cdna_hip_programming.md Guideline 16 (R1): payload stored write-through (sc1), every storing wave
// drains vmcnt, workgroup barrier, ONE lane adds to an agent-scope counter; the consumer polls that counter relaxed, then
// ONE agent-scope acquire, barrier, loads.  Every word another workgroup wrote during the launch is read with an
// agent-scope (sc1) load.  Every spin is bounded; a timeout raises the abort word and every workgroup leaves.
// All workgroups of a launch must be co-resident: the host sizes G against a per-device budget (vo_ba.hip, ba_persist_*).
#pragma once

#define PB_NT 512
#define PB_PS 8                   // a free pose's edge list is linearised in <= PB_PS parts
#define PB_HP 28                  // doubles per (pose, part) record of hpart: 21 upper-triangle sums of H_pp, 6 of b_p, pad
enum { PB_C_B1 = 0, PB_C_FAN, PB_C_SOLVED, PB_C_CHI, PB_C_FIN, PB_C_ABORT, PB_C_N };
#define PB_LINE 32                // unsigned per counter: each on its own 128-byte line
enum { PB_P_CHILIN = 0, PB_P_MAXD, PB_P_CHIT, PB_P_GAIN, PB_P_MAXS, PB_P_CHI0, PB_P_CHIF, PB_P_N = 8 };

struct PbArgs {
    BaDev B;
    unsigned* sync;               // PB_C_N counters, zeroed before the launch
    double* part;                 // [G][PB_P_N] per-worker partial sums
    double* hpart;                // [n_free][PB_PS][PB_HP]
    double* rec;                  // [8] the solver's record of a step: ok, pose gain term, pose max |step|
    double* dp;                   // [D] pose increments of the step
    double* mail;                 // [8] results: chi2 initial, chi2 final, LM iterations, current buffer, steps, status
    int G, it_robust, it_plain, ps, dbg;
    unsigned long long spin_ticks;            // bound of every wait, in 100 MHz ticks
};

// in-kernel time stamps (100 MHz): `tk` accumulators and `tl` (last stamp) are locals of the function that uses the macro
#define PB_TICK(i) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); tk[i] += n_ - tl; tl = n_; }

struct PbLM { double lambda, ni, cur; int it, qmax, need_lin, first, finished, buf, iters_done, steps; };

// A worker's static tables (LDS, filled once per BA): what the dependent index loads of every step would fetch again and again.
#define PB_ECAP 4096              // edges a worker may own
#define PB_SLOTS 8                // Schur slices per worker kept in the table (more: read through the global lists)
struct PbTab {
    int q_lo, n_own;
    int* e_ps;                    // [n_own] pose of own edge q_lo + i
    float2* e_uv;                 // [n_own] its pixel
    uint8_t* e_act;               // [n_own] active flag (the owner culls its own edges)
    int* sl_e1; int* sl_e2; int* sl_k;      // [PB_SLOTS][PB_NT] pair of thread t in the worker's s-th slice: edges and point (-1: none)
    int* sl_j;                    // [PB_SLOTS][2] the slice's block (j1 <= j2), j1 = -1: no slice
};
__host__ __device__ inline size_t pb_tab_bytes() { return (size_t)PB_ECAP * 13 + (size_t)PB_SLOTS * PB_NT * 12 + 64 + 64; }

__device__ __forceinline__ double pb_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pb_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int pb_ldb(const uint8_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pb_stb(uint8_t* p, uint8_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte agent-scope (sc1: aux bit 4) loads / stores through a buffer descriptor: the per-point records are read by 64 lanes from 64
// different cache lines, where the address unit pays per instruction and line -- half the instructions of 8-byte accesses
typedef unsigned pb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pb_rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ void pb_ld16(__amdgpu_buffer_rsrc_t r, unsigned off, double& a, double& b) {
    const pb_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
    a = __hiloint2double((int)v[1], (int)v[0]); b = __hiloint2double((int)v[3], (int)v[2]);
}
__device__ __forceinline__ void pb_st16(__amdgpu_buffer_rsrc_t r, unsigned off, double a, double b) {
    const pb_u32x4 v = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}
// Per-point record of a linearisation (96 bytes, 32-byte aligned): H_ll (6 unique), b_l (3), the point (3).  It lives behind the
// edge weights in the slab that the launch-per-phase path uses for W_e (144 bytes per edge).
#define PB_REC 12
__device__ __forceinline__ double* pb_rec_base(const BaDev& B) { return B.W + (((size_t)B.n_edges + 31) & ~(size_t)31); }

// every wave's write-through stores and atomics have left, then one lane signals
__device__ __forceinline__ void pb_arrive(unsigned* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane polls (relaxed), ONE acquire, barrier.  Returns false (uniformly) when the launch is being aborted.
__device__ __forceinline__ bool pb_wait(unsigned* ctr, unsigned target, unsigned* abort_w, unsigned long long limit, int* s_okw) {
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned spins = 0;
        unsigned long long t0 = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0) {
                if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (!t0) t0 = now;
                else if (now - t0 > limit) { __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_okw = ok;
    }
    __syncthreads();
    const int r = *s_okw;
    __syncthreads();
    return r != 0;
}

// sums of NS values over the 512 threads, returned in every thread; the order of the additions is fixed
template <int NS>
__device__ __forceinline__ void pb_wg_sum(double (&v)[NS], double* s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) v[i] = vo_wave_sum_f64(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) s_red[wave * NS + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double t = 0;
#pragma unroll
        for (int wv = 0; wv < PB_NT / 64; ++wv) t += s_red[wv * NS + i];
        v[i] = t;
    }
    __syncthreads();
}
__device__ __forceinline__ double pb_wg_max(double m, double* s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int wv = 0; wv < PB_NT / 64; ++wv) t = fmax(t, s_red[wv]);
    __syncthreads();
    return t;
}

// g2o's gain-ratio test and lambda policy (OptimizationAlgorithmLevenberg::solve), as in k_ba_chi_control; every workgroup
// runs it on the same inputs, so the replicated LM states stay identical
__device__ __forceinline__ void pb_decide(PbLM& c, int max_it, int ok, double s1, double s2, double m7) {
    const double tmp = ok ? s1 : DBL_MAX;
    const double scale = (ok ? s2 : 0.0) + 1e-3;
    const double rho = (c.cur - tmp) / scale;
    bool converged = false;
    if (rho > 0 && isfinite(tmp)) {
        double a = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
        a = fmin(a, 2.0 / 3.0);
        c.lambda *= fmax(1.0 / 3.0, a); c.ni = 2; c.cur = tmp;
        c.buf ^= 1; c.need_lin = 1;
    } else { c.lambda *= c.ni; c.ni *= 2; }
    if (ok) converged = m7 < 1e-10;
    c.qmax += 1; c.steps += 1;
    if (!(rho < 0 && c.qmax < 10 && !converged)) {
        c.iters_done += 1;
        if (c.qmax == 10 || rho == 0 || converged || c.it + 1 >= max_it) c.finished = 1;
        c.it += 1; c.qmax = 0;
    }
}

// chi2 of the fresh linearisation (sum of the workers' partials) and, on the first step of a round, lambda = 1e-5 max diag(H)
// (g2o computeLambdaInit): the same code in every workgroup
__device__ __forceinline__ void pb_take_linearisation(const PbArgs& A, PbLM& lm, double* s_red) {
    const BaDev& B = A.B;
    const int NW = A.G - 1, tid = threadIdx.x;
    double v[1] = {0.0};
    for (int i = tid; i < NW; i += PB_NT) v[0] += pb_ld(A.part + (size_t)(i + 1) * PB_P_N + PB_P_CHILIN);
    pb_wg_sum<1>(v, s_red);
    lm.cur = v[0];
    if (lm.first) {
        double m = 0;
        for (int i = tid; i < NW; i += PB_NT) m = fmax(m, pb_ld(A.part + (size_t)(i + 1) * PB_P_N + PB_P_MAXD));
        for (int i = tid; i < 6 * B.n_free; i += PB_NT) {
            const int j = i / 6, a = i % 6;
            const int di = a * 6 - a * (a - 1) / 2;                         // upper-triangle index of (a, a): 0 6 11 15 18 20
            double tp[PB_PS];
#pragma unroll
            for (int p = 0; p < PB_PS; ++p) tp[p] = pb_ld(A.hpart + ((size_t)j * PB_PS + min(p, A.ps - 1)) * PB_HP + di);
            double t = 0;
#pragma unroll
            for (int p = 0; p < PB_PS; ++p) t += p < A.ps ? tp[p] : 0.0;
            m = fmax(m, fabs(t));
        }
        m = pb_wg_max(m, s_red);
        lm.lambda = 1e-5 * m; lm.ni = 2; lm.first = 0;
    }
    lm.need_lin = 0;
}

// ---- worker phases ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pb_load_pose(const double* s_p, int j, double (&T)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = s_p[12 * j + i];
}

// Jacobians of one edge without the residual (one division): J_pose (2x6), J_point = J_pose[:, 0:3] R (g2o_types.h:143-167)
__device__ __forceinline__ void pb_jac(const BaCam& cam, const double (&T)[12], const double (&p)[3], double (&Jp)[2][6], double (&Jl)[2][3]) {
#pragma clang fp contract(fast)
    const double X = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[9], Y = T[3] * p[0] + T[4] * p[1] + T[5] * p[2] + T[10], Z = T[6] * p[0] + T[7] * p[1] + T[8] * p[2] + T[11];
    const double Zi = 1.0 / (Z + 1e-18), Zi2 = Zi * Zi, fx = cam.fx, fy = cam.fy;
    Jp[0][0] = -fx * Zi; Jp[0][1] = 0; Jp[0][2] = fx * X * Zi2; Jp[0][3] = fx * X * Y * Zi2; Jp[0][4] = -fx - fx * X * X * Zi2; Jp[0][5] = fx * Y * Zi;
    Jp[1][0] = 0; Jp[1][1] = -fy * Zi; Jp[1][2] = fy * Y * Zi2; Jp[1][3] = fy + fy * Y * Y * Zi2; Jp[1][4] = -fy * X * Y * Zi2; Jp[1][5] = -fy * X * Zi;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Jl[a][c] = Jp[a][0] * T[c] + Jp[a][1] * T[3 + c] + Jp[a][2] * T[6 + c];
}

// sum over the LPP (4 or 8) consecutive lanes that share a point
template <int LPP> __device__ __forceinline__ double pb_grp_sum(double x) {
    x += vo_dpp_mov_f64<0xB1, 0xF>(x);      // quad_perm [1,0,3,2]
    x += vo_dpp_mov_f64<0x4E, 0xF>(x);      // quad_perm [2,3,0,1]
    if (LPP == 8) x += vo_dpp_mov_f64<0x141, 0xF>(x);     // row_half_mirror: the other quad of the 8 lanes
    return x;
}
// linearisation of the worker's points, LPP lanes per point: H_ll (6 unique) and b_l (3) -> HB[9 k ..], the Huber weight of every
// edge -> wgt[e] (the Schur slices and the back-substitution need no residual then), all write-through; chi2 and max |diag| partials
template <int LPP>
__device__ __forceinline__ void pb_lin_points(const BaDev& B, const PbTab& W, int p_lo, int p_hi, int robust, const double* s_pc, const double* pts_c, double& chi_out, double& md_out) {
    double chi = 0, md = 0;
    const __amdgpu_buffer_rsrc_t rec = pb_rsrc(pb_rec_base(B), (unsigned)B.n_points * (PB_REC * 8));
    for (int k0 = p_lo; k0 < p_hi; k0 += PB_NT / LPP) {
        const int k = k0 + (int)threadIdx.x / LPP, sub = threadIdx.x % LPP;
        double H[6] = {0, 0, 0, 0, 0, 0}, b3[3] = {0, 0, 0}, pk[3] = {0, 0, 0};
        if (k < p_hi) {
            const double p[3] = {pb_ld(pts_c + 3 * (size_t)k), pb_ld(pts_c + 3 * (size_t)k + 1), pb_ld(pts_c + 3 * (size_t)k + 2)};
            pk[0] = p[0]; pk[1] = p[1]; pk[2] = p[2];
            const int q1 = B.pt_start[k + 1];
            for (int q = B.pt_start[k] + sub; q < q1; q += LPP) {
                const int i = q - W.q_lo;
                if (!W.e_act[i]) continue;                     // a culled edge keeps the weight 0 its owner stored
                double T[12], r[2], w, rho0, Jp[2][6], Jl[2][3];
                pb_load_pose(s_pc, W.e_ps[i], T);
                const float2 uvf = W.e_uv[i];
                const float uv[2] = {uvf.x, uvf.y};
                ba_edge(B.cam, T, p, uv, robust, B.delta, r, w, rho0, Jp, Jl);
                pb_st(B.W + q, w);
                chi += rho0;
                b3[0] -= w * (Jl[0][0] * r[0] + Jl[1][0] * r[1]); b3[1] -= w * (Jl[0][1] * r[0] + Jl[1][1] * r[1]); b3[2] -= w * (Jl[0][2] * r[0] + Jl[1][2] * r[1]);
                H[0] += w * (Jl[0][0] * Jl[0][0] + Jl[1][0] * Jl[1][0]); H[1] += w * (Jl[0][0] * Jl[0][1] + Jl[1][0] * Jl[1][1]); H[2] += w * (Jl[0][0] * Jl[0][2] + Jl[1][0] * Jl[1][2]);
                H[3] += w * (Jl[0][1] * Jl[0][1] + Jl[1][1] * Jl[1][1]); H[4] += w * (Jl[0][1] * Jl[0][2] + Jl[1][1] * Jl[1][2]); H[5] += w * (Jl[0][2] * Jl[0][2] + Jl[1][2] * Jl[1][2]);
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) H[i] = pb_grp_sum<LPP>(H[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) b3[i] = pb_grp_sum<LPP>(b3[i]);
        if (k < p_hi && sub == 0) {
            const unsigned o = (unsigned)k * (PB_REC * 8);
            pb_st16(rec, o, H[0], H[1]); pb_st16(rec, o + 16, H[2], H[3]); pb_st16(rec, o + 32, H[4], H[5]);
            pb_st16(rec, o + 48, b3[0], b3[1]); pb_st16(rec, o + 64, b3[2], pk[0]); pb_st16(rec, o + 80, pk[1], pk[2]);
            md = fmax(md, fmax(fabs(H[0]), fmax(fabs(H[3]), fabs(H[5]))));
        }
    }
    chi_out = chi; md_out = md;
}

// one part of a free pose's edge list: the 21 + 6 sums of H_pp / b_p -> hpart (no atomics, no zeroing)
__device__ __forceinline__ void pb_lin_pose_part(const PbArgs& A, int vb, int robust, const double* s_pc, const double* pts_c, double* s_red) {
    const BaDev& B = A.B;
    const int j = vb / A.ps, part = vb % A.ps;
    double T[12];
    pb_load_pose(s_pc, j, T);
    double v[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) v[i] = 0;
    const int q1 = B.ps_start[j + 1];
    for (int q = B.ps_start[j] + part * PB_NT + (int)threadIdx.x; q < q1; q += PB_NT * A.ps) {
        const int e = B.ps_edges[q];
        if (!pb_ldb(B.active + e)) continue;
        const size_t k = (size_t)B.e_pt[e];
        const double p[3] = {pb_ld(pts_c + 3 * k), pb_ld(pts_c + 3 * k + 1), pb_ld(pts_c + 3 * k + 2)};
        double r[2], w, rho0, Jp[2][6], Jl[2][3];
        ba_edge(B.cam, T, p, B.e_uv + 2 * (size_t)e, robust, B.delta, r, w, rho0, Jp, Jl);
        int c = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            v[21 + a] -= w * (Jp[0][a] * r[0] + Jp[1][a] * r[1]);
#pragma unroll
            for (int b = a; b < 6; ++b) v[c++] += w * (Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b]);
        }
    }
    double v32[32], r8[8];
#pragma unroll
    for (int i = 0; i < 32; ++i) v32[i] = i < 27 ? v[i] : 0.0;
    vo_wave_reduce32(v32, r8);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if ((lane & 15) == 0) {
        const int slot = VO_R32_SLOT(lane >> 4);
#pragma unroll
        for (int k8 = 0; k8 < 7; ++k8) s_red[wave * 32 + 4 * k8 + slot] = r8[k8];
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        double t = 0;
#pragma unroll
        for (int wv = 0; wv < PB_NT / 64; ++wv) t += s_red[wv * 32 + threadIdx.x];
        pb_st(A.hpart + ((size_t)j * PB_PS + part) * PB_HP + threadIdx.x, t);
    }
    __syncthreads();
}

// Schur slices (<= 512 pairs each: one pair per thread), as ba_schur_slice.  W_e = w J_pose^T J_point has rank 2, so a pair's
// contribution W_e1 Hinv W_e2^T = J_pose1^T M J_pose2 with the 2x2 matrix M = w1 w2 J_point1 Hinv J_point2^T: it is RECOMPUTED from
// the state (point, H_ll, the two weights: ~130 B per pair) instead of being read as two stored 6x3 blocks (~370 B), at 72 FMAs per
// block instead of 162.  Only the lower triangle of S is accumulated (the Cholesky reads nothing else).  A workgroup takes its
// slices two at a time: the loads of both pairs are in flight together and one barrier pair serves both reductions.
struct PbPair { int on, diag; double p[3], h[9], bl[3], w1, w2; };
// (e1, e2, k) of this thread's pair: from the worker's table (slot >= 0) or through the global lists; the dynamic part -- point, H_ll / b_l,
// the two weights (0 for a culled edge) -- is ONE round of independent loads
__device__ __forceinline__ void pb_pair_load(const BaDev& B, const PbTab& W, __amdgpu_buffer_rsrc_t rec, int slot, bool have, const BaBlock& blk, double lambda, PbPair& P) {
    P.on = 0; P.diag = blk.j1 == blk.j2;
    int e1 = -1, e2 = -1, kk = -1;
    if (slot >= 0) { e1 = W.sl_e1[slot * PB_NT + threadIdx.x]; e2 = W.sl_e2[slot * PB_NT + threadIdx.x]; kk = W.sl_k[slot * PB_NT + threadIdx.x]; }
    else if (have && (int)threadIdx.x < blk.count) { const int2 pr = B.pairs[blk.start + threadIdx.x]; e1 = pr.x; e2 = pr.y; kk = B.e_pt[pr.x]; }
    if (kk < 0) return;
    P.w2 = pb_ld(B.W + e2);
    P.w1 = P.diag ? P.w2 : pb_ld(B.W + e1);
    const unsigned o = (unsigned)kk * (PB_REC * 8);
    double hb[6];
    pb_ld16(rec, o, hb[0], hb[1]); pb_ld16(rec, o + 16, hb[2], hb[3]); pb_ld16(rec, o + 32, hb[4], hb[5]);
    pb_ld16(rec, o + 48, P.bl[0], P.bl[1]); pb_ld16(rec, o + 64, P.bl[2], P.p[0]); pb_ld16(rec, o + 80, P.p[1], P.p[2]);
    if (P.w1 == 0.0 || P.w2 == 0.0) return;                    // a culled edge
    P.on = 1;
    const double Hs[9] = {hb[0], hb[1], hb[2], hb[1], hb[3], hb[4], hb[2], hb[4], hb[5]};
    ba_inv3_damped(Hs, lambda, P.h);
}
__device__ __forceinline__ void pb_pair_sums(const BaDev& B, const BaBlock& blk, const double* s_pc, const PbPair& P, double (&v)[42]) {
#pragma unroll
    for (int i = 0; i < 42; ++i) v[i] = 0;
    if (!P.on) return;
    double Jp2[2][6], Jl2[2][3], G2[2][3], T[12];
    pb_load_pose(s_pc, blk.j2, T);
    pb_jac(B.cam, T, P.p, Jp2, Jl2);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) G2[a][c] = Jl2[a][0] * P.h[c] + Jl2[a][1] * P.h[3 + c] + Jl2[a][2] * P.h[6 + c];      // J_point2 Hinv
    double M[2][2];
    if (P.diag) {                                              // e1 == e2: M = w^2 J_l Hinv J_l^T, and b_s gets W_e Hinv b_l = w J_pose^T (J_l Hinv b_l)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) M[a][b] = P.w2 * P.w2 * (G2[a][0] * Jl2[b][0] + G2[a][1] * Jl2[b][1] + G2[a][2] * Jl2[b][2]);
        const double g0 = P.w2 * (G2[0][0] * P.bl[0] + G2[0][1] * P.bl[1] + G2[0][2] * P.bl[2]), g1 = P.w2 * (G2[1][0] * P.bl[0] + G2[1][1] * P.bl[1] + G2[1][2] * P.bl[2]);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double n0 = M[0][0] * Jp2[0][c] + M[0][1] * Jp2[1][c], n1 = M[1][0] * Jp2[0][c] + M[1][1] * Jp2[1][c];
#pragma unroll
            for (int r6 = 0; r6 < 6; ++r6) v[6 * r6 + c] = Jp2[0][r6] * n0 + Jp2[1][r6] * n1;
        }
#pragma unroll
        for (int r6 = 0; r6 < 6; ++r6) v[36 + r6] = Jp2[0][r6] * g0 + Jp2[1][r6] * g1;
    } else {                                                   // M[a][b] = w1 w2 J_l1[a] . (J_l2 Hinv)[b]   (Hinv is symmetric)
        double Jp1[2][6], Jl1[2][3];
        pb_load_pose(s_pc, blk.j1, T);
        pb_jac(B.cam, T, P.p, Jp1, Jl1);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) M[a][b] = P.w1 * P.w2 * (Jl1[a][0] * G2[b][0] + Jl1[a][1] * G2[b][1] + Jl1[a][2] * G2[b][2]);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double n0 = M[0][0] * Jp2[0][c] + M[0][1] * Jp2[1][c], n1 = M[1][0] * Jp2[0][c] + M[1][1] * Jp2[1][c];
#pragma unroll
            for (int r6 = 0; r6 < 6; ++r6) v[6 * r6 + c] = Jp1[0][r6] * n0 + Jp1[1][r6] * n1;
        }
    }
}
// the sums of one slice over the wavefront -> s_red_w[0..41] (entry 6 r + c of the block, 36 + r of b_s).  A diagonal block is
// symmetric: its 21 lower-triangle sums and the 6 of b_s fit ONE 32-value reduction; an off-diagonal block takes one plus 4 single sums.
__device__ __forceinline__ void pb_slice_wave_reduce(bool diag, double (&v)[42], double* s_red_w) {
    const int lane = threadIdx.x & 63;
    double r0[8];
    double lo[32];
    if (diag) {
        int n = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) lo[n++] = v[6 * r + c];
#pragma unroll
        for (int r = 0; r < 6; ++r) lo[21 + r] = v[36 + r];
#pragma unroll
        for (int i = 27; i < 32; ++i) lo[i] = 0.0;
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) lo[i] = v[i];
    }
    vo_wave_reduce32(lo, r0);
    double x4[4] = {0, 0, 0, 0};
    if (!diag) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x4[i] = vo_wave_sum_f64(v[32 + i]);
    }
    if ((lane & 15) == 0) {
        const int slot = VO_R32_SLOT(lane >> 4);
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) s_red_w[4 * k8 + slot] = r0[k8];
    }
    if (lane < 4 && !diag) s_red_w[32 + lane] = lane == 0 ? x4[0] : lane == 1 ? x4[1] : lane == 2 ? x4[2] : x4[3];
}
__device__ __forceinline__ void pb_slice_commit(const BaDev& B, const BaBlock& blk, const double* s_red_u, int t) {     // t in [0, 36): entry of the slice's sums
    double x = 0;
#pragma unroll
    for (int wv = 0; wv < PB_NT / 64; ++wv) x += s_red_u[wv * 96 + t];
    if (blk.j1 == blk.j2) {                                    // packed: 21 lower-triangle entries (row-major), then 6 of b_s
        if (t < 21) {
            int r = 0, rem = t;
            while (rem > r) { rem -= r + 1; ++r; }
            atomicAdd(&B.S[ba_tri(6 * blk.j1 + r, 6 * blk.j1 + rem)], -x);
        } else if (t < 27) atomicAdd(&B.bs[6 * blk.j1 + t - 21], -x);
    } else {
        const int r = t / 6, c = t % 6;
        atomicAdd(&B.S[ba_tri(6 * blk.j2 + c, 6 * blk.j1 + r)], -x);              // j1 < j2: the block below the diagonal
    }
}
__device__ __forceinline__ void pb_schur(const BaDev& B, const PbTab& W, int dbg, int wi, int NW, double lambda, const double* s_pc, const double* pts_c, double* s_red, unsigned long long* tk) {
    unsigned long long tl = __builtin_amdgcn_s_memrealtime();
    const int n_sl = B.n_slices ? min(*B.n_slices, B.n_blocks) : B.n_blocks;
    const int wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rec = pb_rsrc(pb_rec_base(B), (unsigned)B.n_points * (PB_REC * 8));
    int slot = 0;
    for (int sl0 = wi; sl0 < n_sl; sl0 += 2 * NW, slot += 2) {
        const int sl1 = sl0 + NW;
        const bool have1 = sl1 < n_sl, tab0 = slot < PB_SLOTS && !(dbg & 2), tab1 = slot + 1 < PB_SLOTS && !(dbg & 2);
        BaBlock blk0, blk1;
        if (tab0) { blk0.j1 = W.sl_j[2 * slot]; blk0.j2 = W.sl_j[2 * slot + 1]; blk0.start = 0; blk0.count = 0; } else blk0 = B.blocks[sl0];
        if (tab1 && have1) { blk1.j1 = W.sl_j[2 * slot + 2]; blk1.j2 = W.sl_j[2 * slot + 3]; blk1.start = 0; blk1.count = 0; } else blk1 = B.blocks[have1 ? sl1 : sl0];
        PbPair P0, P1;
        pb_pair_load(B, W, rec, tab0 ? slot : -1, true, blk0, lambda, P0);
        pb_pair_load(B, W, rec, (tab1 && have1) ? slot + 1 : -1, have1, blk1, lambda, P1);
        double v[42];
        if (dbg & 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();                                       // s_red free (previous round's commits have read it)
        PB_TICK(8)
        pb_pair_sums(B, blk0, s_pc, P0, v);
        if (dbg & 4) { __syncthreads(); PB_TICK(9) }
        pb_slice_wave_reduce(blk0.j1 == blk0.j2, v, s_red + wave * 96);
        if (have1) {
            pb_pair_sums(B, blk1, s_pc, P1, v);
            pb_slice_wave_reduce(blk1.j1 == blk1.j2, v, s_red + wave * 96 + 48);
        }
        __syncthreads();
        PB_TICK(10)
        if (threadIdx.x < 36) pb_slice_commit(B, blk0, s_red, threadIdx.x);
        else if (have1 && threadIdx.x >= 64 && threadIdx.x < 64 + 36) pb_slice_commit(B, blk1, s_red + 48, threadIdx.x - 64);
    }
    __syncthreads();
    PB_TICK(11)
}

// back-substitution, trial point and trial chi2 of the worker's points (k_ba_update's point part + k_ba_chi_control's edge pass)
template <int LPP>
__device__ __forceinline__ void pb_update_chi(const BaDev& B, const PbTab& W, int p_lo, int p_hi, int robust, double lambda, const double* s_pc, const double* s_pt, const double* s_dp,
                                              const double* pts_c, double* pts_t, double& chi_out, double& sc_out, double& mx_out) {
    double chi = 0, sc = 0, mx = 0;
    const __amdgpu_buffer_rsrc_t rec = pb_rsrc(pb_rec_base(B), (unsigned)B.n_points * (PB_REC * 8));
    for (int k0 = p_lo; k0 < p_hi; k0 += PB_NT / LPP) {
        const int k = k0 + (int)threadIdx.x / LPP, sub = threadIdx.x % LPP;
        const bool live = k < p_hi;
        double p[3] = {0, 0, 0}, hb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rhs[3] = {0, 0, 0};
        int q0 = 0, q1 = 0;
        if (live) {
            const unsigned o = (unsigned)k * (PB_REC * 8);
            pb_ld16(rec, o, hb[0], hb[1]); pb_ld16(rec, o + 16, hb[2], hb[3]); pb_ld16(rec, o + 32, hb[4], hb[5]);
            pb_ld16(rec, o + 48, hb[6], hb[7]); pb_ld16(rec, o + 64, hb[8], p[0]); pb_ld16(rec, o + 80, p[1], p[2]);
            if (sub == 0) { rhs[0] = hb[6]; rhs[1] = hb[7]; rhs[2] = hb[8]; }
            q0 = B.pt_start[k]; q1 = B.pt_start[k + 1];
            for (int q = q0 + sub; q < q1; q += LPP) {
                const int i = q - W.q_lo, j = W.e_ps[i];
                if (!W.e_act[i] || j >= B.n_free) continue;
                const double w = pb_ld(B.W + q);
                double T[12], Jp[2][6], Jl[2][3];
                pb_load_pose(s_pc, j, T);
                pb_jac(B.cam, T, p, Jp, Jl);
                const double* d6 = s_dp + 6 * j;
                double t0 = 0, t1 = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a) { t0 += Jp[0][a] * d6[a]; t1 += Jp[1][a] * d6[a]; }
#pragma unroll
                for (int c = 0; c < 3; ++c) rhs[c] -= w * (Jl[0][c] * t0 + Jl[1][c] * t1);         // W_e^T dp_j
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rhs[c] = pb_grp_sum<LPP>(rhs[c]);
        if (live) {
            const double Hs[9] = {hb[0], hb[1], hb[2], hb[1], hb[3], hb[4], hb[2], hb[4], hb[5]};
            double h[9], pn[3];
            ba_inv3_damped(Hs, lambda, h);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double d = h[3 * a] * rhs[0] + h[3 * a + 1] * rhs[1] + h[3 * a + 2] * rhs[2];
                pn[a] = p[a] + d;
                if (sub == 0) { pb_st(pts_t + 3 * (size_t)k + a, pn[a]); sc += d * (lambda * d + hb[6 + a]); mx = fmax(mx, fabs(d)); }
            }
            for (int q = q0 + sub; q < q1; q += LPP) {
                const int i = q - W.q_lo;
                if (!W.e_act[i]) continue;
                double T[12], r[2], pc[3];
                pb_load_pose(s_pt, W.e_ps[i], T);
                const float2 uvf = W.e_uv[i];
                const float uv[2] = {uvf.x, uvf.y};
                ba_err(B.cam, T, pn, uv, r, pc);
                const double e2 = r[0] * r[0] + r[1] * r[1];
                chi += (robust && e2 > B.delta * B.delta) ? 2.0 * sqrt(e2) * B.delta - B.delta * B.delta : e2;
            }
        }
    }
    chi_out = chi; sc_out = sc; mx_out = mx;
}

// plain chi2 / cull passes over the worker's own edges (positions q of the point-major edge order)
//   mode 0: initial chi2 (every edge)   mode 1: cull after the robust round (k_ba_cull stage 0)   mode 2: final flags + chi2 (stage 1)
__device__ __forceinline__ double pb_edge_pass(const BaDev& B, const PbTab& W, int mode, const double* s_pc, const double* pts_c) {
    double acc = 0;
    for (int i = threadIdx.x; i < W.n_own; i += PB_NT) {
        const int e = W.q_lo + i;
        const size_t k = (size_t)B.e_pt[e];
        const double p[3] = {pb_ld(pts_c + 3 * k), pb_ld(pts_c + 3 * k + 1), pb_ld(pts_c + 3 * k + 2)};
        double T[12], r[2], pc[3];
        pb_load_pose(s_pc, W.e_ps[i], T);
        const float2 uvf = W.e_uv[i];
        const float uv[2] = {uvf.x, uvf.y};
        ba_err(B.cam, T, p, uv, r, pc);
        const double c2 = r[0] * r[0] + r[1] * r[1];
        if (mode == 0) acc += c2;
        else if (mode == 1) {
            if (c2 > B.chi2_th) { B.flags[e] = 1; W.e_act[i] = 0; pb_stb(B.active + e, 0); pb_st(B.W + e, 0.0); } else B.flags[e] = 0;
        } else if (W.e_act[i]) { if (c2 > B.chi2_th) B.flags[e] |= 2; else acc += c2; }
    }
    return acc;
}

__device__ __forceinline__ int pb_lower_bound(const int32_t* a, int n, int key) {      // first i in [0, n] with a[i] >= key (a ascending, n + 1 entries)
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// ---- solver phases ------------------------------------------------------------------------------------------------------------
// H_pp (full 6x6 blocks) and b_p from the parts' records, fixed order; solver-private arrays (plain stores)
__device__ __forceinline__ void pb_gather_hpp(const PbArgs& A) {
    const BaDev& B = A.B;
    for (int i = threadIdx.x; i < 27 * B.n_free; i += PB_NT) {
        const int j = i / 27, idx = i % 27;
        double tp[PB_PS];
#pragma unroll
        for (int p = 0; p < PB_PS; ++p) tp[p] = pb_ld(A.hpart + ((size_t)j * PB_PS + min(p, A.ps - 1)) * PB_HP + idx);     // all parts in flight together
        double t = 0;
#pragma unroll
        for (int p = 0; p < PB_PS; ++p) t += p < A.ps ? tp[p] : 0.0;
        if (idx < 21) {
            int a = 0, rem = idx;
            while (rem >= 6 - a) { rem -= 6 - a; ++a; }
            const int b = a + rem;
            B.Hpp[36 * (size_t)j + 6 * a + b] = t;
            B.Hpp[36 * (size_t)j + 6 * b + a] = t;
        } else B.bp[6 * j + (idx - 21)] = t;
    }
    __syncthreads();
}

// trial poses exp(dp) * T (free) / copies (fixed), as ba_pose_body; gain-ratio term and max |step| of the pose part
__device__ __forceinline__ void pb_trial_poses(const PbArgs& A, int ok, double lambda, const double* poses_c, double* poses_t, double& sc_out, double& mx_out) {
    const BaDev& B = A.B;
    double sc = 0, mx = 0;
    for (int j = threadIdx.x; j < B.n_poses; j += PB_NT) {
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = pb_ld(poses_c + 12 * (size_t)j + i);
        double* Tn = poses_t + 12 * (size_t)j;
        if (j >= B.n_free || !ok) {
#pragma unroll
            for (int i = 0; i < 12; ++i) pb_st(Tn + i, T[i]);
            continue;
        }
        double d[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) d[a] = pb_ld(A.dp + 6 * j + a);
        const double w[3] = {d[3], d[4], d[5]};
        const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
        double Ac, Bc, C;
        if (th < 1e-8) { Ac = 1.0 - th2 / 6.0; Bc = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0; }
        else { Ac = sin(th) / th; Bc = (1.0 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
        const double Wm[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        double W2[9], R[9], V[9];
        for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += Wm[3 * i + k] * Wm[3 * k + c]; W2[3 * i + c] = s; }
        for (int i = 0; i < 9; ++i) { const double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + Ac * Wm[i] + Bc * W2[i]; V[i] = I + Bc * Wm[i] + C * W2[i]; }
        const double tx = V[0] * d[0] + V[1] * d[1] + V[2] * d[2], ty = V[3] * d[0] + V[4] * d[1] + V[5] * d[2], tz = V[6] * d[0] + V[7] * d[1] + V[8] * d[2];
        for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += R[3 * i + k] * T[3 * k + c]; pb_st(Tn + 3 * i + c, s); }
        pb_st(Tn + 9, R[0] * T[9] + R[1] * T[10] + R[2] * T[11] + tx);
        pb_st(Tn + 10, R[3] * T[9] + R[4] * T[10] + R[5] * T[11] + ty);
        pb_st(Tn + 11, R[6] * T[9] + R[7] * T[10] + R[8] * T[11] + tz);
        for (int a = 0; a < 6; ++a) { sc += d[a] * (lambda * d[a] + B.bp[6 * j + a]); mx = fmax(mx, fabs(d[a])); }
    }
    sc_out = sc; mx_out = mx;
}

struct PbCtrs { unsigned* b1; unsigned* fan; unsigned* solved; unsigned* chi; unsigned* fin; unsigned* abort_w; };
__device__ __forceinline__ PbCtrs pb_ctrs(const PbArgs& A) {
    PbCtrs c;
    c.b1 = A.sync + PB_C_B1 * PB_LINE; c.fan = A.sync + PB_C_FAN * PB_LINE; c.solved = A.sync + PB_C_SOLVED * PB_LINE;
    c.chi = A.sync + PB_C_CHI * PB_LINE; c.fin = A.sync + PB_C_FIN * PB_LINE; c.abort_w = A.sync + PB_C_ABORT * PB_LINE;
    return c;
}
// the partials of a step in the same order in every workgroup, then the same decision
__device__ __forceinline__ void pb_step_decision(const PbArgs& A, PbLM& lm, int max_it, int ok, double pose_sc, double pose_mx, double* red) {
    const int NW = A.G - 1;
    double v[2] = {0.0, 0.0}, m = 0;
    for (int i = threadIdx.x; i < NW; i += PB_NT) {
        v[0] += pb_ld(A.part + (size_t)(i + 1) * PB_P_N + PB_P_CHIT); v[1] += pb_ld(A.part + (size_t)(i + 1) * PB_P_N + PB_P_GAIN);
        m = fmax(m, pb_ld(A.part + (size_t)(i + 1) * PB_P_N + PB_P_MAXS));
    }
    pb_wg_sum<2>(v, red);
    m = pb_wg_max(m, red);
    pb_decide(lm, max_it, ok, v[0], pose_sc + v[1], fmax(pose_mx, m));
}
__device__ __forceinline__ void pb_round_init(PbLM& lm, int max_it) {
    lm.it = 0; lm.qmax = 0; lm.need_lin = 1; lm.first = 1; lm.finished = max_it <= 0; lm.ni = 2; lm.iters_done = 0; lm.steps = 0;
}

__device__ __forceinline__ void pb_solver_main(const PbArgs& A, double* s_mem, int* s_okw) {
    const BaDev& B = A.B;
    const int tid = threadIdx.x, NW = A.G - 1;
    const PbCtrs C = pb_ctrs(A);
    unsigned n_step = 0;
    PbLM lm;
    lm.buf = 0; lm.lambda = 0; lm.ni = 2; lm.cur = 0;
    int iters_total = 0, steps_total = 0;
    bool alive = true;
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memrealtime();      // where the solver's time goes (100 MHz ticks): mail[8..13]
    for (int round = 0; round < 2 && alive; ++round) {
        const int max_it = round == 0 ? A.it_robust : A.it_plain;
        pb_round_init(lm, max_it);
        while (!lm.finished) {
            double* const poses_c = lm.buf ? B.posesB : B.posesA; double* const poses_t = lm.buf ? B.posesA : B.posesB;
            ++n_step;
            if (!pb_wait(C.fan, (unsigned)NW * n_step, C.abort_w, A.spin_ticks, s_okw)) { alive = false; break; }
            PB_TICK(0)
            if (lm.need_lin) {
                pb_gather_hpp(A);
                pb_take_linearisation(A, lm, s_mem);
            }
            PB_TICK(1)
            int ok = 0;
#ifndef PB_NO_CHOL
            ok = ba_chol16_body<true>(B, nullptr, lm.lambda, s_mem, A.dp) ? 1 : 0;
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            PB_TICK(2)
            double pose_sc, pose_mx;
            pb_trial_poses(A, ok, lm.lambda, poses_c, poses_t, pose_sc, pose_mx);
            double v[1] = {pose_sc};
            pb_wg_sum<1>(v, s_mem);
            pose_sc = v[0];
            pose_mx = pb_wg_max(pose_mx, s_mem);
            if (tid == 0) { pb_st(A.rec + 0, ok ? 1.0 : 0.0); pb_st(A.rec + 1, pose_sc); pb_st(A.rec + 2, pose_mx); }
            pb_arrive(C.solved);
            PB_TICK(3)
            if (!pb_wait(C.chi, (unsigned)NW * n_step, C.abort_w, A.spin_ticks, s_okw)) { alive = false; break; }
            PB_TICK(4)
            pb_step_decision(A, lm, max_it, ok, pose_sc, pose_mx, s_mem);
            PB_TICK(5)
        }
        if (!alive) break;
        iters_total += lm.iters_done; steps_total += lm.steps;
    }
    const bool got = pb_wait(C.fin, (unsigned)NW, C.abort_w, A.spin_ticks, s_okw);
    double v[2] = {0.0, 0.0};
    if (got) for (int i = tid; i < NW; i += PB_NT) { v[0] += pb_ld(A.part + (size_t)(i + 1) * PB_P_N + PB_P_CHI0); v[1] += pb_ld(A.part + (size_t)(i + 1) * PB_P_N + PB_P_CHIF); }
    pb_wg_sum<2>(v, s_mem);
    if (tid == 0) {
        A.mail[0] = v[0]; A.mail[1] = v[1]; A.mail[2] = (double)iters_total; A.mail[3] = (double)lm.buf; A.mail[4] = (double)steps_total;
        const unsigned ab = __hip_atomic_load(C.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        A.mail[5] = (got && alive && !ab) ? 0.0 : (ab == 2u ? 2.0 : 1.0);
        for (int i = 0; i < 6; ++i) A.mail[8 + i] = (double)tk[i];
    }
}

__device__ __forceinline__ void pb_worker_main(const PbArgs& A, double* s_mem, int* s_okw, int* s_rng) {
    const BaDev& B = A.B;
    const int tid = threadIdx.x, G = A.G, NW = G - 1, w = blockIdx.x, wi = w - 1, D = B.D, np = B.n_poses;
    const PbCtrs C = pb_ctrs(A);
    unsigned n_b1 = 0, n_step = 0;                       // episodes of the worker barrier / LM steps so far: the same sequence in every workgroup
    double* const s_red = s_mem;                          // [8][96] reduction scratch
    double* s_pc = s_mem + 8 * 96;                        // current poses [np][12]
    double* s_pt = s_pc + 12 * (size_t)np;                // trial poses
    double* const s_dp = s_pt + 12 * (size_t)np;          // pose increments [D]
    PbTab W;
    {
        uint8_t* t = reinterpret_cast<uint8_t*>(s_dp + ((D + 1) & ~1));
        W.e_uv = reinterpret_cast<float2*>(t); t += (size_t)PB_ECAP * 8;
        W.e_ps = reinterpret_cast<int*>(t); t += (size_t)PB_ECAP * 4;
        W.sl_e1 = reinterpret_cast<int*>(t); t += (size_t)PB_SLOTS * PB_NT * 4;
        W.sl_e2 = reinterpret_cast<int*>(t); t += (size_t)PB_SLOTS * PB_NT * 4;
        W.sl_k = reinterpret_cast<int*>(t); t += (size_t)PB_SLOTS * PB_NT * 4;
        W.sl_j = reinterpret_cast<int*>(t); t += 64;
        W.e_act = t;
    }
    // the worker's points: a contiguous range holding about n_edges / NW edges
    if (tid < 2) {
        const long long key = (long long)B.n_edges * (wi + tid) / NW;
        s_rng[tid] = (wi + tid == NW) ? B.n_points : pb_lower_bound(B.pt_start, B.n_points, (int)key);
    }
    for (int i = tid; i < 12 * np; i += PB_NT) s_pc[i] = pb_ld(B.posesA + i);
    __syncthreads();
    const int p_lo = s_rng[0], p_hi = s_rng[1];
    const int q_lo = B.pt_start[p_lo], q_hi = B.pt_start[p_hi];
    const bool lpp8 = p_hi - p_lo <= PB_NT / 8 && !(A.dbg & 1);               // few points per worker: 8 lanes share a point's edges
    W.q_lo = q_lo; W.n_own = q_hi - q_lo;
    if (W.n_own > PB_ECAP) {                                  // the host sized G so that this cannot happen; nothing has been written yet
        if (tid == 0) __hip_atomic_store(C.abort_w, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        W.n_own = 0;
    }
    for (int i = tid; i < W.n_own; i += PB_NT) { W.e_ps[i] = B.e_pose[q_lo + i]; W.e_uv[i] = reinterpret_cast<const float2*>(B.e_uv)[q_lo + i]; W.e_act[i] = 1; }
    {
        const int n_sl = B.n_slices ? min(*B.n_slices, B.n_blocks) : B.n_blocks;
        for (int sidx = 0; sidx < PB_SLOTS; ++sidx) {
            const int sl = wi + sidx * NW;
            int e1 = -1, e2 = -1, kk = -1;
            if (sl < n_sl) {
                const BaBlock blk = B.blocks[sl];
                if (tid == 0) { W.sl_j[2 * sidx] = blk.j1; W.sl_j[2 * sidx + 1] = blk.j2; }
                if (tid < blk.count) { const int2 pr = B.pairs[blk.start + tid]; e1 = pr.x; e2 = pr.y; kk = B.e_pt[pr.x]; }
            }
            W.sl_e1[sidx * PB_NT + tid] = e1; W.sl_e2[sidx * PB_NT + tid] = e2; W.sl_k[sidx * PB_NT + tid] = kk;
        }
    }
    __syncthreads();
    double part_chi0;
    {
        double v[1] = {pb_edge_pass(B, W, 0, s_pc, B.ptsA)};
        pb_wg_sum<1>(v, s_red);
        part_chi0 = v[0];
    }
    PbLM lm;
    lm.buf = 0; lm.lambda = 0; lm.ni = 2; lm.cur = 0;
    bool alive = true;
    unsigned long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memrealtime();     // worker 1's time: mail[16..27]
    for (int round = 0; round < 2 && alive; ++round) {
        const int robust = round == 0, max_it = robust ? A.it_robust : A.it_plain;
        pb_round_init(lm, max_it);
        while (!lm.finished) {
            double* const pts_c = lm.buf ? B.ptsB : B.ptsA; double* const pts_t = lm.buf ? B.ptsA : B.ptsB;
            double* const poses_t = lm.buf ? B.posesA : B.posesB;
            ++n_step;
            if (lm.need_lin) {
                double chi, md;
                if (lpp8) pb_lin_points<8>(B, W, p_lo, p_hi, robust, s_pc, pts_c, chi, md);
                else pb_lin_points<4>(B, W, p_lo, p_hi, robust, s_pc, pts_c, chi, md);
                for (int vb = wi; vb < B.n_free * A.ps; vb += NW) pb_lin_pose_part(A, vb, robust, s_pc, pts_c, s_red);
                double v[1] = {chi};
                pb_wg_sum<1>(v, s_red);
                md = pb_wg_max(md, s_red);
                if (tid == 0) { pb_st(A.part + (size_t)w * PB_P_N + PB_P_CHILIN, v[0]); pb_st(A.part + (size_t)w * PB_P_N + PB_P_MAXD, md); }
                PB_TICK(0)
                pb_arrive(C.b1);
                ++n_b1;
                if (!pb_wait(C.b1, (unsigned)NW * n_b1, C.abort_w, A.spin_ticks, s_okw)) { alive = false; break; }
                PB_TICK(1)
                pb_take_linearisation(A, lm, s_red);
                PB_TICK(2)
            }
            pb_schur(B, W, A.dbg, wi, NW, lm.lambda, s_pc, pts_c, s_red, tk);
            tl = __builtin_amdgcn_s_memrealtime();
            PB_TICK(3)
            pb_arrive(C.fan);
            if (!pb_wait(C.solved, n_step, C.abort_w, A.spin_ticks, s_okw)) { alive = false; break; }
            PB_TICK(4)
            const int ok = pb_ld(A.rec + 0) != 0.0;
            const double pose_sc = pb_ld(A.rec + 1), pose_mx = pb_ld(A.rec + 2);
            double chi = 0, sc = 0, mx = 0;
            if (ok) {
                for (int i = tid; i < 12 * np; i += PB_NT) s_pt[i] = pb_ld(poses_t + i);
                for (int i = tid; i < D; i += PB_NT) s_dp[i] = pb_ld(A.dp + i);
                __syncthreads();
                if (lpp8) pb_update_chi<8>(B, W, p_lo, p_hi, robust, lm.lambda, s_pc, s_pt, s_dp, pts_c, pts_t, chi, sc, mx);
                else pb_update_chi<4>(B, W, p_lo, p_hi, robust, lm.lambda, s_pc, s_pt, s_dp, pts_c, pts_t, chi, sc, mx);
            }
            {
                double v[2] = {chi, sc};
                pb_wg_sum<2>(v, s_red);
                mx = pb_wg_max(mx, s_red);
                if (tid == 0) { pb_st(A.part + (size_t)w * PB_P_N + PB_P_CHIT, v[0]); pb_st(A.part + (size_t)w * PB_P_N + PB_P_GAIN, v[1]); pb_st(A.part + (size_t)w * PB_P_N + PB_P_MAXS, mx); }
            }
            PB_TICK(5)
            pb_arrive(C.chi);
            if (!pb_wait(C.chi, (unsigned)NW * n_step, C.abort_w, A.spin_ticks, s_okw)) { alive = false; break; }
            PB_TICK(6)
            const int buf0 = lm.buf;
            pb_step_decision(A, lm, max_it, ok, pose_sc, pose_mx, s_red);
            PB_TICK(7)
            if (lm.buf != buf0) { double* t = s_pc; s_pc = s_pt; s_pt = t; }         // the trial poses are the current ones now
        }
        if (!alive) break;
        if (round == 0) { (void)pb_edge_pass(B, W, 1, s_pc, lm.buf ? B.ptsB : B.ptsA); __syncthreads(); }      // backend.cpp:144-156; visible to the others behind the next B1
    }
    double v[1] = {alive ? pb_edge_pass(B, W, 2, s_pc, lm.buf ? B.ptsB : B.ptsA) : 0.0};      // backend.cpp:162-172
    pb_wg_sum<1>(v, s_red);
    if (tid == 0) { pb_st(A.part + (size_t)w * PB_P_N + PB_P_CHI0, part_chi0); pb_st(A.part + (size_t)w * PB_P_N + PB_P_CHIF, v[0]); }
    if (tid == 0 && w == 1) for (int i = 0; i < 12; ++i) A.mail[16 + i] = (double)tk[i];
    pb_arrive(C.fin);
}

__global__ __launch_bounds__(PB_NT) void k_ba_persist(PbArgs A) {
    extern __shared__ double s_mem[];
    __shared__ int s_okw;
    __shared__ int s_rng[4];
    if (blockIdx.x == 0) pb_solver_main(A, s_mem, &s_okw);
    else pb_worker_main(A, s_mem, &s_okw, s_rng);
}
