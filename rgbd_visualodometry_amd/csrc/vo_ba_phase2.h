// vo_ba_phase2.h -- second generation of the launch-per-phase LM step for systems the LDS-resident Cholesky solves (D <= 192),
// included by vo_ba.hip.  THREE launches per step in the steady state instead of five, and much less traffic:
//   k_ba_schur2   pair lists as before, but a pair's 6x6 contribution is rebuilt from a 96-byte per-point record (H_ll, b_l, the point) and
//                 the two edges' Huber weights: W_e1 Hinv W_e2^T = J_pose1^T M J_pose2 with the 2x2 matrix M = w1 w2 J_point1 Hinv J_point2^T
//                 (W_e has rank 2): ~130 bytes and 72 FMAs per pair instead of ~370 bytes and 162 (the 144-byte W_e blocks are never
//                 stored: 15 MB per linearisation); lower triangle of S only, packed (ba_tri).  The same launch carries the 4 workgroups per
//                 free pose that sum H_pp / b_p when the state has changed (the Cholesky needs them, the Schur slices do not).
//   k_ba_chol16   as before; it clears S / b_s behind its load (no zeroing launch) and leaves the solution in B.dl
//   k_ba_upchi2   back-substitution, trial point and the robust chi2 of the point's edges in one pass -- and, in that same pass over the
//                 edges, the LINEARISATION AT THE TRIAL STATE (records and weights into the other buffer, `lbuf ^ 1`): when the last
//                 workgroup's LM decision accepts the step (the usual case) the next step starts at the Schur launch; when it rejects,
//                 the current buffers are still the linearisation at the unchanged state.  Every workgroup rebuilds the ~50 trial poses
//                 in LDS (Taylor exp map); the last workgroup runs g2o's accept / lambda policy.
//   k_ba_lin2 / k_ba_maxdiag2   only on the first step of a round: lambda_0 = 1e-5 max diag(H) needs a linearisation before anything else
// Stores and loads are plain: between launches the data stays in the XCDs' L2s.  (A kernel boundary of this chain costs ~5.5 us while the
// tracker's kernels run beside it -- 0.1 us alone, profiles/r03_ba_gaps.txt -- which is what taking launches out of the step buys.)
#pragma once

// agent-scope relaxed accesses: they go past this XCD's L2 (the eight L2s are not coherent with each other), which is what the
// fence-free "last workgroup" tickets of k_ba_upchi2 / k_ba_round read their peers' partial sums with
__device__ __forceinline__ double pb_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pb_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Per-point record of a linearisation (96 bytes, 32-byte aligned): H_ll (6 unique), b_l (3), the point (3).  It lives behind the
// edge weights in the slab that the five-launch path uses for W_e (144 bytes per edge).
#define PB_REC 12
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

// The slab the first generation uses for W_e (>= 16 n_edges + 192 n_points + 2 KiB bytes) holds two weight arrays and two record arrays;
// ctl->lbuf says which pair is the current linearisation.
__device__ __forceinline__ double* p2_w(const BaDev& B, int which) { return B.W + (size_t)which * (((size_t)B.n_edges + 31) & ~(size_t)31); }
__device__ __forceinline__ double* p2_rec(const BaDev& B, int which) {
    return B.W + 2 * (((size_t)B.n_edges + 31) & ~(size_t)31) + (size_t)which * (((size_t)PB_REC * B.n_points + 31) & ~(size_t)31);
}
__device__ __forceinline__ void p2_rec_store(double* rec, int k, const double (&H)[6], const double (&b3)[3], const double (&p)[3]) {
    double2* o = reinterpret_cast<double2*>(rec + (size_t)PB_REC * k);
    o[0] = make_double2(H[0], H[1]); o[1] = make_double2(H[2], H[3]); o[2] = make_double2(H[4], H[5]);
    o[3] = make_double2(b3[0], b3[1]); o[4] = make_double2(b3[2], p[0]); o[5] = make_double2(p[1], p[2]);
}
__device__ __forceinline__ void p2_rec_load(const double* rec, int k, double (&H)[6], double (&b3)[3], double (&p)[3]) {
    const double2* o = reinterpret_cast<const double2*>(rec + (size_t)PB_REC * k);
    const double2 a = o[0], b = o[1], c = o[2], d = o[3], e = o[4], f = o[5];
    H[0] = a.x; H[1] = a.y; H[2] = b.x; H[3] = b.y; H[4] = c.x; H[5] = c.y; b3[0] = d.x; b3[1] = d.y; b3[2] = e.x; p[0] = e.y; p[1] = f.x; p[2] = f.y;
}

// A workgroup-uniform f64 in a scalar register pair (the two poses of a block: 24 values that would otherwise sit in 48 vector registers)
__device__ __forceinline__ double p2_uniform(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// One slice (<= BA_SLICE pairs) of a 6x6 block.  A thread holds ONE pair at a time and nothing across pairs: the pair's 36 (+ 6) products leave in three
// 16-value wavefront reductions (two columns of the block each, + two entries of b_s on a diagonal block) whose results the row leaders add up in
// LDS.  Register need is the two pose Jacobians + one 16-value reduction (round 3 first kept 42 f64 accumulators per thread: 226 VGPRs, two
// waves per SIMD, and a launch over several problems ran its workgroups in as many rounds as it had problems).
// ba_lin_poses_body without per-lane accumulators (27 f64 = 54 registers across the edge loop made this the register peak of k_ba_schur2): every
// round of 256 edges leaves its 21 + 6 products in two 16-value wavefront reductions whose row leaders add them up in LDS (s_part: 4 waves x 32).
// (ps_lo / ps_hi: ps_start[j], ps_start[j + 1] where the caller has them already -- the Schur launch's head load -- or -1: read here)
template <int PS>                                           // PS workgroups per list: PSPLIT (launches over several problems) or PSPLIT_LONE (a lone problem's)
__device__ __forceinline__ void p2_lin_poses_body(const BaCam& cam, const BaDev& B, int robust, double delta, int blk, const double* poses_c, const double* pts_c, double* s_part,
                                                  int ps_lo = -1, int ps_hi = -1) {
    const int j = blk / PS, part = blk % PS;
    if (ps_lo < 0) { ps_lo = __builtin_amdgcn_readfirstlane(B.ps_start[j]); ps_hi = __builtin_amdgcn_readfirstlane(B.ps_start[j + 1]); }
    const int q_lo = ps_lo + part * 256, q_hi = ps_hi;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = p2_uniform(poses_c[12 * (size_t)j + i]);
    if (threadIdx.x < 128) s_part[threadIdx.x] = 0.0;
    __syncthreads();
    // Four rounds' loads at a time (a list of ~3600 edges over four workgroups of 256 lanes is four rounds): the lists' edges and points in one batch, then the
    // activity bytes, observations and positions in one batch -- two dependent trips for the whole list instead of two to three per round (this workgroup is the
    // longest of an accepted step's Schur launch).  The rounds' sums are formed and added in the same order as before.
    constexpr int NU = 2;                                    // (rounds in flight: a list of ~3600 edges is one round of sixteen workgroups, config 5's 11 k are three)
    for (int q0 = q_lo; q0 < q_hi; q0 += 256 * PS * NU) {             // workgroup-uniform trip count
        int e[NU], k[NU], act[NU]; float uv[NU][2]; double pk[NU][3]; bool on[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int q = q0 + u * 256 * PS + (int)threadIdx.x;
            on[u] = q < q_hi;
            const int qc = on[u] ? q : q_lo;                                // (an idle lane reads the list's first entry: a valid address, nothing of it is used)
            e[u] = B.ps_edges[qc]; k[u] = B.ps_pt ? B.ps_pt[qc] : 0;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(e[u]), "+v"(k[u]));
        if (!B.ps_pt) {
#pragma unroll
            for (int u = 0; u < NU; ++u) k[u] = B.e_pt[e[u]];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            act[u] = B.active[e[u]];
            const float2 uvf = *reinterpret_cast<const float2*>(B.e_uv + 2 * (size_t)e[u]);
            uv[u][0] = uvf.x; uv[u][1] = uvf.y;
            pk[u][0] = pts_c[3 * (size_t)k[u]]; pk[u][1] = pts_c[3 * (size_t)k[u] + 1]; pk[u][2] = pts_c[3 * (size_t)k[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(act[u]), "+v"(uv[u][0]), "+v"(uv[u][1]), "+v"(pk[u][0]), "+v"(pk[u][1]), "+v"(pk[u][2]));
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (q0 + u * 256 * PS >= q_hi) break;                       // (workgroup-uniform: the list ended in an earlier round)
            double r[2] = {0, 0}, w = 0, rho0, Jp[2][6], Jl[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 6; ++c) Jp[a][c] = 0;
            if (on[u] && act[u]) ba_edge(cam, T, pk[u], uv[u], robust, delta, r, w, rho0, Jp, Jl);
            else w = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {                            // upper-triangle entries 0..15, then 16..20 and the 6 of b_p (21..26)
                double x[16], o4[4];
                int c = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = a; b < 6; ++b, ++c) if (c / 16 == half) x[c % 16] = w * (Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b]);
                if (half) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) x[5 + a] = -(w * (Jp[0][a] * r[0] + Jp[1][a] * r[1]));
#pragma unroll
                    for (int i = 11; i < 16; ++i) x[i] = 0.0;
                }
                (void)o4;
                const double tsum = vo_wave_reduce16t(x);
                if ((lane & 15) < 4) s_part[wave * 32 + 16 * half + 4 * VO_R16T_K(lane) + VO_R32_SLOT(lane >> 4)] += tsum;
            }
        }
    }
    __syncthreads();
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

// lambda_0 = 1e-5 x the largest diagonal entry of the linearised system (g2o's computeLambdaInit; scal[4]).  The point blocks' entries are at hand in
// the workgroup that sums them; H_pp's are complete only when every pose workgroup has added its share, so every workgroup of the launch takes a
// ticket behind its atomics (each wave drains its own first: ADVICE r3) and the last one reads H_pp's diagonal past its caches.  (Round 4 had a
// launch of its own for this, k_ba_maxdiag2: 6 us + a boundary, twice per BA.)
// (Round 6: the point workgroups' sums -- the two chi2 and the largest point-block diagonal entry -- go to per-workgroup partials, B.partU, and the last
// workgroup adds them up: they were ~2000 same-address atomics per launch, which the memory side performs one after the other: ~15 of the launch's 26 us.)
__device__ __forceinline__ void p2_lin_finish(const BaDev& B, BaCtl* ctl_, int expected, int stage) {
    __shared__ int s_lin_last;
    __shared__ double s_fin[3 * 4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_lin_last = __hip_atomic_fetch_add(&ctl_->lin_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1;
    __syncthreads();
    if (!s_lin_last) return;
    double a = 0, b = 0, v = 0;
    for (int i = threadIdx.x; i < B.gp; i += 256) { a += pb_ld(B.partU + 3 * (size_t)i); b += pb_ld(B.partU + 3 * (size_t)i + 1); v = fmax(v, pb_ld(B.partU + 3 * (size_t)i + 2)); }
    for (int i = threadIdx.x; i < B.D; i += 256) v = fmax(v, fabs(pb_ld(&B.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)])));
    a = vo_wave_sum_f64(a); b = vo_wave_sum_f64(b);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; s_fin[w] = a; s_fin[4 + w] = b; s_fin[8 + w] = v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (s_fin[0] + s_fin[1]) + (s_fin[2] + s_fin[3]); b = (s_fin[4] + s_fin[5]) + (s_fin[6] + s_fin[7]); v = fmax(fmax(s_fin[8], s_fin[9]), fmax(s_fin[10], s_fin[11]));
        if (a != 0.0) atomicAdd(&B.scal[0], a);
        if (b != 0.0 && stage == 0) atomicAdd(&B.scal[5], b);      // (the first-generation path has k_ba_chi for this)
        if (v > 0) atomicMax((unsigned long long*)&B.scal[4], (unsigned long long)__double_as_longlong(v));
        ctl_->lin_ticket = 0;
    }
}

// Round 6: (a) the control block comes in one vector load (fields by readlane) and the descriptor as a copy (one batch of scalar loads) instead of a field per
// dependent trip; (b) the poses are staged in LDS (pose_cap of them fit the launch's dynamic LDS; more: read where they lie), so an edge's pose is not a trip of
// its own; (c) a lane asks for everything two of its edges need -- activity byte, pose number, observation -- in one batch instead of one edge and one field at a
// time: a point with 21 edges was 6 rounds x 4 dependent trips, now 3 x 1.  The sums are taken in the same order: same values.
__global__ __launch_bounds__(256) void k_ba_lin2(BaBatch Q, int pose_cap, int ps) {
    BA_PROBLEM_COPY(Q)
    static_assert(sizeof(BaCtl) <= 112 && sizeof(BaCtl) % 4 == 0, "the head load takes the control block as sizeof / 4 <= 28 words");
    const int hw_ = reinterpret_cast<const int*>(ctl_)[min((int)(threadIdx.x & 63), (int)sizeof(BaCtl) / 4 - 1)];
    auto h_i = [&](size_t byte_off) { return __builtin_amdgcn_readlane(hw_, (int)(byte_off / 4)); };
    if (h_i(offsetof(BaCtl, finished)) || B.D > BA_FOLD_D) return;
    if (!(h_i(offsetof(BaCtl, need_lin)) && h_i(offsetof(BaCtl, first)))) return;      // only the first step of a round: later linearisations come out of k_ba_upchi2
    const int gp = B.gp;
    const int buf_ = h_i(offsetof(BaCtl, buf)), robust = h_i(offsetof(BaCtl, robust)), lbuf = h_i(offsetof(BaCtl, lbuf)), stage = h_i(offsetof(BaCtl, stage));
    const double* const poses_c = buf_ ? B.posesB : B.posesA; const double* const pts_c = buf_ ? B.ptsB : B.ptsA;
    __shared__ double s_part[4 * 32];
    extern __shared__ double s_T[];                          // [12 pose_cap]
    const int expected = gp + B.n_free * ps;               // the workgroups of this problem that do anything (a launch over several problems may be wider)
    if ((int)blockIdx.x >= gp) {
        if ((int)blockIdx.x - gp < B.n_free * ps) {
            if (ps == PSPLIT_LONE) p2_lin_poses_body<PSPLIT_LONE>(B.cam, B, robust, B.delta, blockIdx.x - gp, poses_c, pts_c, s_part);
            else p2_lin_poses_body<PSPLIT>(B.cam, B, robust, B.delta, blockIdx.x - gp, poses_c, pts_c, s_part);
            p2_lin_finish(B, ctl_, expected, stage);
        }
        return;
    }
    double* const rec = p2_rec(B, lbuf);
    double* const Wt = p2_w(B, lbuf);
    const int k = blockIdx.x * 64 + (threadIdx.x >> 2), sub = threadIdx.x & 3;
    const bool in_lds = B.n_poses <= pose_cap;
    if (in_lds) for (int i = threadIdx.x; i < 12 * B.n_poses; i += 256) s_T[i] = poses_c[i];
    double chi[2] = {0.0, 0.0};                            // robust chi2 of the linearisation; plain chi2 (the initial state's, reported as chi2_initial)
    double H[6] = {0, 0, 0, 0, 0, 0}, b3[3] = {0, 0, 0}, p[3] = {0, 0, 0};
    int q0 = 0, q1 = 0;
    if (k < B.n_points) {
        p[0] = pts_c[3 * (size_t)k]; p[1] = pts_c[3 * (size_t)k + 1]; p[2] = pts_c[3 * (size_t)k + 2];
        q0 = B.pt_start[k]; q1 = B.pt_start[k + 1];
    }
    __syncthreads();                                        // the poses are in LDS
    const double* const Tb = in_lds ? s_T : poses_c;
    auto one = [&](int e, int act, int jp, const float (&uv)[2]) {
        if (!act) return;
        double r[2], w, rho0, Jp[2][6], Jl[2][3];
        ba_edge(B.cam, Tb + 12 * (size_t)jp, p, uv, robust, B.delta, r, w, rho0, Jp, Jl);
        Wt[e] = w;
        chi[0] += rho0; chi[1] += r[0] * r[0] + r[1] * r[1];
        b3[0] -= w * (Jl[0][0] * r[0] + Jl[1][0] * r[1]); b3[1] -= w * (Jl[0][1] * r[0] + Jl[1][1] * r[1]); b3[2] -= w * (Jl[0][2] * r[0] + Jl[1][2] * r[1]);
        H[0] += w * (Jl[0][0] * Jl[0][0] + Jl[1][0] * Jl[1][0]); H[1] += w * (Jl[0][0] * Jl[0][1] + Jl[1][0] * Jl[1][1]); H[2] += w * (Jl[0][0] * Jl[0][2] + Jl[1][0] * Jl[1][2]);
        H[3] += w * (Jl[0][1] * Jl[0][1] + Jl[1][1] * Jl[1][1]); H[4] += w * (Jl[0][1] * Jl[0][2] + Jl[1][1] * Jl[1][2]); H[5] += w * (Jl[0][2] * Jl[0][2] + Jl[1][2] * Jl[1][2]);
    };
    for (int q = q0 + sub; q < q1; q += 8) {
        const bool two = q + 4 < q1;
        int ea = B.edges_by_point ? q : B.pt_edges[q], eb = two ? (B.edges_by_point ? q + 4 : B.pt_edges[q + 4]) : ea;
        if (!B.edges_by_point) asm volatile("" : "+v"(ea), "+v"(eb));
        int aa = B.active[ea], ab = B.active[eb], ja = B.e_pose[ea], jb = B.e_pose[eb];
        const float2 ua = *reinterpret_cast<const float2*>(B.e_uv + 2 * (size_t)ea), ub = *reinterpret_cast<const float2*>(B.e_uv + 2 * (size_t)eb);
        float uva[2] = {ua.x, ua.y}, uvb[2] = {ub.x, ub.y};
        asm volatile("" : "+v"(aa), "+v"(ab), "+v"(ja), "+v"(jb), "+v"(uva[0]), "+v"(uva[1]), "+v"(uvb[0]), "+v"(uvb[1]));      // (one batch)
        one(ea, aa, ja, uva);
        if (two) one(eb, ab, jb, uvb);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) H[i] = ba_quad_sum(H[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b3[i] = ba_quad_sum(b3[i]);
    if (k < B.n_points && sub == 0) p2_rec_store(rec, k, H, b3, p);
    const double vdiag = k < B.n_points ? fmax(fabs(H[0]), fmax(fabs(H[3]), fabs(H[5]))) : 0.0;      // the point block's diagonal (what the record holds)
    ba_fold_zero(B, blockIdx.x, gp);
    double red[3] = {chi[0], chi[1], 0.0};
    {   // (the workgroup's largest diagonal entry: a wave maximum each, joined behind the block reduction's barriers)
        double vm = vdiag;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vm = fmax(vm, __shfl_xor(vm, o, 64));
        __shared__ double s_vm[4];
        if ((threadIdx.x & 63) == 0) s_vm[threadIdx.x >> 6] = vm;
        ba_block_reduce<2>(red, s_part);                    // (its barriers also publish s_vm)
        if (threadIdx.x == 0) red[2] = fmax(fmax(s_vm[0], s_vm[1]), fmax(s_vm[2], s_vm[3]));
    }
    if (threadIdx.x == 0) { pb_st(B.partU + 3 * (size_t)blockIdx.x, red[0]); pb_st(B.partU + 3 * (size_t)blockIdx.x + 1, red[1]); pb_st(B.partU + 3 * (size_t)blockIdx.x + 2, red[2]); }
    p2_lin_finish(B, ctl_, expected, stage);
}

// pb_jac in two halves: the camera-frame point (X, Y, 1 / Z), and the pose Jacobian rebuilt from it (the same expressions: the values are pb_jac's)
__device__ __forceinline__ void p2_cam_point(const double (&T)[12], const double (&p)[3], double& X, double& Y, double& Zi) {
#pragma clang fp contract(fast)
    X = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[9]; Y = T[3] * p[0] + T[4] * p[1] + T[5] * p[2] + T[10];
    const double Z = T[6] * p[0] + T[7] * p[1] + T[8] * p[2] + T[11];
    Zi = 1.0 / (Z + 1e-18);
}
__device__ __forceinline__ void p2_jp(const BaCam& cam, double X, double Y, double Zi, double (&Jp)[2][6]) {
#pragma clang fp contract(fast)
    const double Zi2 = Zi * Zi, fx = cam.fx, fy = cam.fy;
    Jp[0][0] = -fx * Zi; Jp[0][1] = 0; Jp[0][2] = fx * X * Zi2; Jp[0][3] = fx * X * Y * Zi2; Jp[0][4] = -fx - fx * X * X * Zi2; Jp[0][5] = fx * Y * Zi;
    Jp[1][0] = 0; Jp[1][1] = -fy * Zi; Jp[1][2] = fy * Y * Zi2; Jp[1][3] = fy + fy * Y * Y * Zi2; Jp[1][4] = -fy * X * Y * Zi2; Jp[1][5] = -fy * X * Zi;
}
__device__ __forceinline__ void p2_jl(const double (&Jp)[2][6], const double (&T)[12], double (&Jl)[2][3]) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Jl[a][c] = Jp[a][0] * T[c] + Jp[a][1] * T[3 + c] + Jp[a][2] * T[6 + c];
}
// One slice (<= BA_SLICE pairs) of a 6x6 block.  A thread holds ONE pair at a time and nothing across pairs: the pair's 36 (+ 6) products leave in three
// 16-value wavefront reductions (two columns of the block each, + two entries of b_s on a diagonal block) whose results the row leaders add up in
// LDS.  Between the passes a lane keeps only the two camera-frame points, the 2x2 middle factor M and g: the 2x6 pose Jacobians are rebuilt from
// the camera-frame points in every pass (a dozen multiplications) instead of occupying 40 registers.  (Round 3 first kept 42 f64 accumulators
// per thread: 226 VGPRs, two waves per SIMD, and a launch over several problems ran its workgroups in as many rounds as it had problems.)
template <bool DIAG>
__device__ __forceinline__ void p2_schur_slice(const BaDev& B, const BaBlock blk, double lambda, const double* poses_c, const double* rec, const double* Wt, double* s_part, double* s_tot) {
#ifdef P2_STAMPS
    long long ts_[5]; ts_[0] = wall_clock64();
#endif
    double T1[12], T2[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { T2[i] = p2_uniform(poses_c[12 * (size_t)blk.j2 + i]); T1[i] = DIAG ? 0.0 : p2_uniform(poses_c[12 * (size_t)blk.j1 + i]); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < blk.count; base += 256) {                   // workgroup-uniform trip count: every lane takes part in the reductions
        bool on = base + (int)threadIdx.x < blk.count;
        const int pi = blk.start + base + (int)threadIdx.x;
        // two batches of loads, each asked for before anything of it is looked at (without the pins the compiler moves every load behind the test of the one
        // before it); an idle lane of a partial slice asks for nothing (a third of the slices are partial: read-and-drop cost 8 / 16 streams 2 %)
        int2 pr = make_int2(0, 0);
        int pt_l = 0;
        if (on) { pr = B.pairs[pi]; if (B.pair_pt) pt_l = B.pair_pt[pi]; }      // (the pair's point arrives with the pair where the plan wrote it: one load level less)
        asm volatile("" : "+v"(pr.x), "+v"(pr.y), "+v"(pt_l));
        if (on && !B.pair_pt) pt_l = B.e_pt[pr.x];
        int a12 = 0;
        double Hh[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0}, p[3] = {0, 0, 0}, w1 = 0, w2 = 0;
        if (on) {                                           // both activity bytes, the point's record, the two weights (an inactive pair's are read and dropped)
            const int a1 = B.active[pr.x], a2 = B.active[pr.y];
            w2 = Wt[pr.y]; w1 = DIAG ? w2 : Wt[pr.x];
            p2_rec_load(rec, pt_l, Hh, bl, p);
            a12 = a1 | (a2 << 8);
        }
        asm volatile("" : "+v"(a12), "+v"(w1), "+v"(w2), "+v"(Hh[0]), "+v"(Hh[2]), "+v"(Hh[4]), "+v"(bl[0]), "+v"(bl[2]), "+v"(p[1]));
        on = on && (a12 & 0xff) && (DIAG || (a12 >> 8));
        double X1 = 0, Y1 = 0, Zi1 = 0, X2 = 0, Y2 = 0, Zi2 = 0, M[2][2] = {{0, 0}, {0, 0}}, g[2] = {0, 0};      // an idle lane: M = g = 0, every product is 0
        if (on) {
            double h[9];
            const double Hs[9] = {Hh[0], Hh[1], Hh[2], Hh[1], Hh[3], Hh[4], Hh[2], Hh[4], Hh[5]};
            ba_inv3_damped(Hs, lambda, h);
            double Jp2[2][6], Jl2[2][3], G2[2][3];
            p2_cam_point(T2, p, X2, Y2, Zi2);
            p2_jp(B.cam, X2, Y2, Zi2, Jp2);
            p2_jl(Jp2, T2, Jl2);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) G2[a][c] = Jl2[a][0] * h[c] + Jl2[a][1] * h[3 + c] + Jl2[a][2] * h[6 + c];
            if (DIAG) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) M[a][b] = w2 * w2 * (G2[a][0] * Jl2[b][0] + G2[a][1] * Jl2[b][1] + G2[a][2] * Jl2[b][2]);
                g[0] = w2 * (G2[0][0] * bl[0] + G2[0][1] * bl[1] + G2[0][2] * bl[2]); g[1] = w2 * (G2[1][0] * bl[0] + G2[1][1] * bl[1] + G2[1][2] * bl[2]);
            } else {
                double Jp1[2][6], Jl1[2][3];
                p2_cam_point(T1, p, X1, Y1, Zi1);
                p2_jp(B.cam, X1, Y1, Zi1, Jp1);
                p2_jl(Jp1, T1, Jl1);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) M[a][b] = w1 * w2 * (Jl1[a][0] * G2[b][0] + Jl1[a][1] * G2[b][1] + Jl1[a][2] * G2[b][2]);
            }
        }
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {                                   // columns 2 cp, 2 cp + 1 of the block: x[6 cc + r]; x[12 + cc]: entry 2 cp + cc of b_s
            // (the empty asm makes the camera-frame points new values to the compiler in every pass: it must not keep the Jacobians of one pass for the next)
            asm volatile("" : "+v"(X2), "+v"(Y2), "+v"(Zi2));
            if (!DIAG) asm volatile("" : "+v"(X1), "+v"(Y1), "+v"(Zi1));
            double Jp1[2][6], Jp2[2][6];
            p2_jp(B.cam, X2, Y2, Zi2, Jp2);
            if (DIAG) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 6; ++c) Jp1[a][c] = Jp2[a][c];
            } else p2_jp(B.cam, X1, Y1, Zi1, Jp1);
            double x[16], o4[4];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = 2 * cp + cc;
                const double n0 = M[0][0] * Jp2[0][c] + M[0][1] * Jp2[1][c], n1 = M[1][0] * Jp2[0][c] + M[1][1] * Jp2[1][c];
#pragma unroll
                for (int r6 = 0; r6 < 6; ++r6) x[6 * cc + r6] = Jp1[0][r6] * n0 + Jp1[1][r6] * n1;
                x[12 + cc] = DIAG ? Jp2[0][c] * g[0] + Jp2[1][c] * g[1] : 0.0;
            }
            x[14] = x[15] = 0.0;
            (void)o4;
            const double tsum = vo_wave_reduce16t(x);
            if ((lane & 15) < 4) {
                double* d = s_part + wave * 48 + 16 * cp + 4 * VO_R16T_K(lane) + VO_R32_SLOT(lane >> 4);
                *d = base ? *d + tsum : tsum;
            }
        }
    }
#ifdef P2_STAMPS
    ts_[1] = wall_clock64();
#endif
    __syncthreads();
    if (threadIdx.x < 48) s_tot[threadIdx.x] = s_part[threadIdx.x] + s_part[48 + threadIdx.x] + s_part[96 + threadIdx.x] + s_part[144 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 48) {
        const int cp = threadIdx.x >> 4, idx = threadIdx.x & 15;
        const double val = s_tot[threadIdx.x];
        if (idx < 12) {
            const int r = idx % 6, c = 2 * cp + idx / 6;
            if (DIAG) { if (c <= r) atomicAdd(&B.S[ba_sidx(B, 6 * blk.j1 + r, 6 * blk.j1 + c)], -val); }
            else atomicAdd(&B.S[ba_sidx(B, 6 * blk.j2 + c, 6 * blk.j1 + r)], -val);      // j1 < j2: the block below the diagonal is the one the Cholesky reads
        } else if (DIAG && idx < 14) {
            atomicAdd(&B.bs[6 * blk.j1 + 2 * cp + idx - 12], -val);
        }
    }
#ifdef P2_STAMPS
    ts_[2] = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.z == 0 && (blockIdx.x % 97) == 5 && B.n_points > 1000) printf("[schur2 wg %d, %d pairs, diag %d] 10 ns ticks: loads + compute %lld | reduce + atomics issued %lld\n", (int)blockIdx.x, blk.count, (int)DIAG, ts_[1] - ts_[0], ts_[2] - ts_[1]);
#endif
}
// Everything a workgroup needs before its first pair comes in ONE vector load: lane i < 28 takes word i of the control block, lanes 32 .. 47 the words of
// scal[0 .. 7], lane 48 the slice count, lanes 52 .. 55 the slice's descriptor (its index clamped: a workgroup behind the last slice reads the last one's and
// leaves); the fields come out by readlane.  Written field by field -- as up to round 5, and again when the fields were merely read at the top of the function:
// the compiler moves every load to its first use, behind the branches -- a slice workgroup made nine dependent trips to L2 (~0.5 us each, forty launches per BA)
// before it asked for its pairs.
template <int PS>
__device__ __forceinline__ void ba_schur2_body(const BaDev& B, BaCtl* ctl_, double* s_part, double* s_tot) {
    static_assert(sizeof(BaCtl) <= 112 && sizeof(BaCtl) % 4 == 0 && sizeof(BaBlock) == 16, "the head load's lane map");
    const int n_pose_blk = B.n_free * PS;
    const int sl = (int)blockIdx.x - n_pose_blk;
    const int lane_ = threadIdx.x & 63;
    const int* hp_ = reinterpret_cast<const int*>(ctl_) + min(lane_, (int)sizeof(BaCtl) / 4 - 1);
    if (lane_ >= 32 && lane_ < 48) hp_ = reinterpret_cast<const int*>(B.scal) + (lane_ - 32);
    if (lane_ == 48 && B.n_slices) hp_ = B.n_slices;
    if (lane_ >= 52 && lane_ < 56 && B.blocks) hp_ = reinterpret_cast<const int*>(B.blocks + min(max(sl, 0), max(B.n_blocks - 1, 0))) + (lane_ - 52);
    if (lane_ >= 56 && lane_ < 58 && sl < 0) hp_ = B.ps_start + (int)blockIdx.x / PS + (lane_ - 56);      // (a pose workgroup: its list's bounds)
    const int hw_ = *hp_;
    auto h_i = [&](size_t byte_off) { return __builtin_amdgcn_readlane(hw_, (int)(byte_off / 4)); };
    const int finished = h_i(offsetof(BaCtl, finished)), buf_ = h_i(offsetof(BaCtl, buf)), need_lin = h_i(offsetof(BaCtl, need_lin)), first = h_i(offsetof(BaCtl, first)),
              robust = h_i(offsetof(BaCtl, robust)), lbuf = h_i(offsetof(BaCtl, lbuf));
    const double lambda_c = __hiloint2double(h_i(offsetof(BaCtl, lambda) + 4), h_i(offsetof(BaCtl, lambda)));
    const double scal4 = __hiloint2double(__builtin_amdgcn_readlane(hw_, 32 + 9), __builtin_amdgcn_readlane(hw_, 32 + 8));
    const int nsl_v = __builtin_amdgcn_readlane(hw_, 48);
    BaBlock blk;
    blk.j1 = __builtin_amdgcn_readlane(hw_, 52); blk.j2 = __builtin_amdgcn_readlane(hw_, 53); blk.start = __builtin_amdgcn_readlane(hw_, 54); blk.count = __builtin_amdgcn_readlane(hw_, 55);
    if (finished || B.D > BA_FOLD_D) return;
    double* const poses_c = buf_ ? B.posesB : B.posesA; double* const pts_c = buf_ ? B.ptsB : B.ptsA;
    if ((int)blockIdx.x < n_pose_blk) {
        // H_pp / b_p of the accepted state (zeroed by the step that accepted it).  On the first step of a round k_ba_lin2 has done it (lambda_0
        // needs the diagonal before this launch); after a rejected step the sums of the unchanged state are still there.
        if (need_lin && !first) p2_lin_poses_body<PS>(B.cam, B, robust, B.delta, blockIdx.x, poses_c, pts_c, s_part, __builtin_amdgcn_readlane(hw_, 56), __builtin_amdgcn_readlane(hw_, 57));
        return;
    }
    if (sl >= B.n_blocks || (B.n_slices && sl >= nsl_v)) return;
    const double lambda = p2_uniform((need_lin && first) ? 1e-5 * scal4 : lambda_c);      // as k_ba_chol16 derives it (the control block is updated there)
    const int lb = lbuf;
    const double* const rec = p2_rec(B, lb);
    const double* const Wt = p2_w(B, lb);
    if (blk.j1 == blk.j2) p2_schur_slice<true>(B, blk, lambda, poses_c, rec, Wt, s_part, s_tot);
    else p2_schur_slice<false>(B, blk, lambda, poses_c, rec, Wt, s_part, s_tot);
}
__global__ __launch_bounds__(256) void k_ba_schur2(BaBatch Q) {
    BA_PROBLEM_COPY(Q)
    __shared__ double s_part[4 * 48];
    __shared__ double s_tot[48];
    ba_schur2_body<PSPLIT>(B, ctl_, s_part, s_tot);
}
// A lone problem (the engine's usual case with one stream): the descriptor rides in the kernel's arguments -- scalar loads from the argument
// segment, which is there when the wave starts -- instead of being fetched from the descriptor table first: one dependent trip to L2 less at the
// head of each of a BA's forty step launches.  Same body, same results.
__global__ __launch_bounds__(256) void k_ba_schur2_one(BaDev B, BaCtl* ctl_) {
    __shared__ double s_part[4 * 48];
    __shared__ double s_tot[48];
    ba_schur2_body<PSPLIT_LONE>(B, ctl_, s_part, s_tot);
}

// exp(d) * T for a pose increment d = [translation, rotation] (g2o_types.h:56-60).  LM increments are small rotations: below 0.25 rad the
// coefficients sin(th)/th, (1 - cos th)/th^2, (th - sin th)/th^3 come from their Taylor series to th^14 (error < 1e-19) instead of the
// sin / cos call chains (~400 instructions, and every workgroup of k_ba_upchi2 rebuilds all trial poses)
__device__ __forceinline__ void p2_exp_mul(const double* d, const double* T, double (&Tn)[12]) {
#pragma clang fp contract(fast)
    const double w[3] = {d[3], d[4], d[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, Bc, C;
    if (th2 < 0.0625) {
        A = 1.0 - th2 * (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0) * (1.0 - th2 * (1.0 / 156.0) * (1.0 - th2 * (1.0 / 210.0)))))));
        Bc = 0.5 * (1.0 - th2 * (1.0 / 12.0) * (1.0 - th2 * (1.0 / 30.0) * (1.0 - th2 * (1.0 / 56.0) * (1.0 - th2 * (1.0 / 90.0) * (1.0 - th2 * (1.0 / 132.0) * (1.0 - th2 * (1.0 / 182.0) * (1.0 - th2 * (1.0 / 240.0))))))));
        C = (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0) * (1.0 - th2 * (1.0 / 156.0) * (1.0 - th2 * (1.0 / 210.0) * (1.0 - th2 * (1.0 / 272.0))))))));
    } else { const double th = sqrt(th2); A = sin(th) / th; Bc = (1.0 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double Wm[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9], R[9], V[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) W2[3 * i + c] = Wm[3 * i] * Wm[c] + Wm[3 * i + 1] * Wm[3 + c] + Wm[3 * i + 2] * Wm[6 + c];
#pragma unroll
    for (int i = 0; i < 9; ++i) { const double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + A * Wm[i] + Bc * W2[i]; V[i] = I + Bc * Wm[i] + C * W2[i]; }
    const double tx = V[0] * d[0] + V[1] * d[1] + V[2] * d[2], ty = V[3] * d[0] + V[4] * d[1] + V[5] * d[2], tz = V[6] * d[0] + V[7] * d[1] + V[8] * d[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) Tn[3 * i + c] = R[3 * i] * T[c] + R[3 * i + 1] * T[3 + c] + R[3 * i + 2] * T[6 + c];
    Tn[9] = R[0] * T[9] + R[1] * T[10] + R[2] * T[11] + tx;
    Tn[10] = R[3] * T[9] + R[4] * T[10] + R[5] * T[11] + ty;
    Tn[11] = R[6] * T[9] + R[7] * T[10] + R[8] * T[11] + tz;
}

// Row r (0..2) of exp(d) * T: the three rotation entries and translation entry r -- the same expressions as p2_exp_mul, evaluated by three lanes per
// pose instead of one lane per pose (k_ba_upchi2: the pose prelude was its register peak and ran on one wave in eight)
__device__ __forceinline__ void p2_exp_mul_row(const double* d, const double* T, int r, double (&row)[3], double& tr) {
#pragma clang fp contract(fast)
    const double w[3] = {d[3], d[4], d[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, Bc, C;
    if (th2 < 0.0625) {
        A = 1.0 - th2 * (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0) * (1.0 - th2 * (1.0 / 156.0) * (1.0 - th2 * (1.0 / 210.0)))))));
        Bc = 0.5 * (1.0 - th2 * (1.0 / 12.0) * (1.0 - th2 * (1.0 / 30.0) * (1.0 - th2 * (1.0 / 56.0) * (1.0 - th2 * (1.0 / 90.0) * (1.0 - th2 * (1.0 / 132.0) * (1.0 - th2 * (1.0 / 182.0) * (1.0 - th2 * (1.0 / 240.0))))))));
        C = (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0) * (1.0 - th2 * (1.0 / 156.0) * (1.0 - th2 * (1.0 / 210.0) * (1.0 - th2 * (1.0 / 272.0))))))));
    } else { const double th = sqrt(th2); A = sin(th) / th; Bc = (1.0 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double Wm[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    const double m0 = r == 0 ? Wm[0] : (r == 1 ? Wm[3] : Wm[6]), m1 = r == 0 ? Wm[1] : (r == 1 ? Wm[4] : Wm[7]), m2 = r == 0 ? Wm[2] : (r == 1 ? Wm[5] : Wm[8]);
    const double mr[3] = {m0, m1, m2};
    double Rr[3], Vr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double w2 = m0 * Wm[c] + m1 * Wm[3 + c] + m2 * Wm[6 + c];
        const double I = (c == r) ? 1.0 : 0.0;
        Rr[c] = I + A * mr[c] + Bc * w2; Vr[c] = I + Bc * mr[c] + C * w2;
    }
    const double tt = Vr[0] * d[0] + Vr[1] * d[1] + Vr[2] * d[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) row[c] = Rr[0] * T[c] + Rr[1] * T[3 + c] + Rr[2] * T[6 + c];
    tr = Rr[0] * T[9] + Rr[1] * T[10] + Rr[2] * T[11] + tt;
}

// trial state and its robust chi2, then (last workgroup) the LM decision of k_ba_chi_control
#define UPC_T 512
// FUSED: the workgroup runs in the same launch as the Cholesky (k_ba_cholup): everything that does not depend on the solution -- the
// point records, the first round of edges, the current poses -- is requested first, then the workgroup waits for the word the solver
// workgroup sets behind its results (which it has written through to memory), and reads them past its own caches.
// LDS of an update workgroup behind the poses and the solution (doubles): the round's points for the overflow lanes, their pass-1 / pass-2 accumulators, the overflow list
#define UPC_OVF_CAP 1024                                     // overflow list entries (a workgroup's edges beyond the first four of each point: 60-130 in the bench's windows)
#define UPC_OVF_MINE 4                                       // list entries a lane can own (its point's edges 5 .. 20 spread over the point's four lanes; beyond: the lane walks them itself)
#define UPC_LDS_EXTRA (3 * (UPC_T / 4) + 9 * UPC_OVF_CAP + UPC_OVF_CAP / 2)
template <bool FUSED>
__device__ __forceinline__ void ba_upchi2_body(const BaDev& B, BaCtl* ctl_, int rep, const int bx, const int bz) {
    // 128 points per workgroup and round (4 lanes each), `rep` rounds: with several problems per launch every workgroup's fixed costs (the trial
    // poses, the partial sums and the ticket) are spread over more points -- a lone problem keeps rep = 1 and the most workgroups.
    // Round 6: (a) workgroup gp owns NO points: it does what workgroup 0 did beside its points -- the trial poses to global memory, the pose part of the gain
    // ratio (atomics + a drain of every wave) -- so that the workgroup that takes the last ticket is no longer the one with the most work;
    // (b) a point's edges beyond its first four are not walked by its four lanes in rounds of four (the bench's windows: largest degree per workgroup
    // 10 on average, 21 at most: 3-6 dependent rounds per pass where most lanes had one) but put on a workgroup-wide list and taken ONE PER LANE,
    // their contributions written to the entry's own LDS cell and added up by the lane that listed them, in the order it listed them (no atomics: two runs of
    // one solve add in the same order): two rounds per pass whatever the degrees are (scripts/ba_degree_stats.py).
    // points per workgroup and round: 128 (four lanes each) -- or fewer for a lone problem's fused launch (BaDev::upc_ppw, a multiple of 16 = whole wavefronts), where
    // every workgroup has a compute unit to itself anyway (the solver's LDS size) and the chip has more compute units than 128-point workgroups: pass 1 and pass 2 are
    // bound by the double-precision issue of the two wavefronts a SIMD holds, and 96 points leave two of the eight wavefronts without work
    const int P = (FUSED && rep == 1 && B.upc_ppw > 0) ? B.upc_ppw : UPC_T / 4;
    const int gp = (B.n_points + rep * P - 1) / (rep * P);
    if (bx > gp) return;
    const bool pose_wg = bx == gp;
    extern __shared__ double s_dyn[];
    double* const s_T = s_dyn;                             // trial poses [n_poses][12]
    double* const s_dp = s_dyn + 12 * (size_t)B.n_poses;   // pose increments [D]
    double* const s_Tc = s_dp + B.D;                       // current poses [n_poses][12] (pass 1 reads them per edge)
    double* const s_pp = s_Tc + 12 * (size_t)B.n_poses;    // [128][3] the round's points: current, then trial (what an overflow lane knows of a point that is not its own)
    double* const s_c = s_pp + 3 * (UPC_T / 4);            // [UPC_OVF_CAP][9] an overflow entry's contribution: pass 1 rhs (3), pass 2 H_ll (6) + b_l (3)
    int* const s_list = reinterpret_cast<int*>(s_c + 9 * UPC_OVF_CAP);      // [UPC_OVF_CAP] (local point << 24 | edge position)
    __shared__ double s_w[3 * (UPC_T / 64)];
    __shared__ double s_hb[9 * (UPC_T / 4)];               // H_ll (6) and b_l (3) of the round's 128 points, parked during pass 1
    __shared__ int s_last, s_accept, s_nov;
    // workgroup-uniform values that arrive through vector loads (the control block is written by this kernel's last workgroup, so the compiler
    // may not use scalar loads): moved to scalar registers by hand, or they and everything derived from them (four record / weight base pointers)
    // occupy vector registers for the whole kernel
    double lambda = FUSED ? 0.0 : p2_uniform(ctl_->lambda);
    const int robust = __builtin_amdgcn_readfirstlane(ctl_->robust);
    bool ok = FUSED ? true : __builtin_amdgcn_readfirstlane((int)(B.scal[3] != 0.0)) != 0;
#ifdef P2_STAMPS
    long long tq_[8], tw_ = 0; int nq_ = 0; const bool stamp_ = (bx == 7 || bx == gp - 1) && threadIdx.x == 0 && ctl_->it == 4 && bz == 0;
#define P2_STAMP() { if (nq_ < 8) tq_[nq_++] = wall_clock64(); }
#else
#define P2_STAMP()
#endif
    P2_STAMP()
    BA_STATE(B)
    const int lb = __builtin_amdgcn_readfirstlane(ctl_->lbuf);
    const double* const rec = p2_rec(B, lb);
    const double* const Wt = p2_w(B, lb);
    double* const rec_n = p2_rec(B, lb ^ 1);
    double* const Wn = p2_w(B, lb ^ 1);
    int k = bx * rep * P + (threadIdx.x >> 2);
    const int sub = threadIdx.x & 3, pl = threadIdx.x >> 2;
    // (the list packs an edge position into 24 bits; and a graph with so many poses that the list and the cells do not fit LDS beside them -- upc_ovf = 0, the
    // launch then carries no such region -- walks its edges in rounds of four as before)
    // Only in the FUSED form (a lone problem, one workgroup per CU because of the solver's LDS: registers and LDS are free there): with several problems per launch the
    // list's state costs the kernel its second workgroup per CU (149 against 128 VGPRs: 8 / 16 streams 6015 / 7088 against 6184 / 7382 frames/s, measured).
    const bool ovf_ok = FUSED && B.upc_ovf != 0 && B.n_edges < (1 << 24);
    bool live = false;
    double H[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0}, p[3] = {0, 0, 0}, rhs[3] = {0, 0, 0};
    int q0 = 0, q1 = 0;
    // edge A: the lane's own (its point's sub-th edge); edge B: the lane's first item of the overflow list.  Everything of both that does not depend on the
    // solution is requested -- and, FUSED, linearised at the current state -- before the solution is waited for.
    int eA = -1, jA = 0; uint8_t actA = 0; double wA = 0; float2 uvA = make_float2(0.f, 0.f);
    int eB = -1, jB = 0, plB = 0; uint8_t actB = 0; double wB = 0; float2 uvB = make_float2(0.f, 0.f);
    double JpA[2][6], JlA[2][3], JpB[2][6], JlB[2][3];
    bool preA = false, preB = false, serial = false;
    int nov = 0, mine[UPC_OVF_MINE], n_mine = 0, q_ser = 0;      // mine: the list entries this lane made; q_ser: its point's edges from here on it walks itself
    auto prep = [&](bool pre_jac) {
        live = ok && !pose_wg && k < B.n_points && (int)(threadIdx.x >> 2) < P;
        eA = -1; actA = 0; q0 = q1 = 0; eB = -1; actB = 0; preA = preB = false;
        if (FUSED && threadIdx.x == 0) s_nov = 0;
        if (live) {
            q0 = B.pt_start[k]; q1 = B.pt_start[k + 1];
            p2_rec_load(rec, k, H, bl, p);
            if (q0 + sub < q1) {
                const int e = B.edges_by_point ? q0 + sub : B.pt_edges[q0 + sub];
                actA = B.active[e]; eA = e; jA = B.e_pose[e]; wA = Wt[e]; uvA = reinterpret_cast<const float2*>(B.e_uv)[e];
            }
        }
        n_mine = 0; q_ser = q0 + sub + 4;
        if (!FUSED) { serial = true; nov = 0; return; }     // (several problems per launch: no overflow list, see ovf_ok)
        __syncthreads();                                    // s_nov is zero; the previous round's readers of s_pp / s_c / s_list are done
        if (live) {
            if (sub == 0 && ovf_ok) {
#pragma unroll
                for (int c = 0; c < 3; ++c) s_pp[3 * pl + c] = p[c];
            }
        }
        if (live && ovf_ok) {
#pragma unroll
            for (int m = 0; m < UPC_OVF_MINE; ++m) {
                mine[m] = 0;
                if (q_ser < q1) { const int pos = atomicAdd(&s_nov, 1); mine[m] = pos; if (pos < UPC_OVF_CAP) s_list[pos] = (pl << 24) | q_ser; q_ser += 4; n_mine = m + 1; }
            }
        }
        __syncthreads();
        nov = __builtin_amdgcn_readfirstlane(s_nov);
        serial = !ovf_ok || nov > UPC_OVF_CAP;              // (a workgroup whose points have more than 1024 further edges: rounds of four as before)
        if (serial) { nov = 0; n_mine = 0; q_ser = q0 + sub + 4; }
        if ((int)threadIdx.x < nov) {
            const int it = s_list[threadIdx.x], q = it & 0xFFFFFF;
            plB = it >> 24;
            const int e = B.edges_by_point ? q : B.pt_edges[q];
            actB = B.active[e]; eB = e; jB = B.e_pose[e]; wB = Wt[e]; uvB = reinterpret_cast<const float2*>(B.e_uv)[e];
        }
        if (pre_jac) {
            if (live && eA >= 0 && actA && jA < B.n_free) {
                double T0[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) T0[i] = poses_c[12 * (size_t)jA + i];
                pb_jac(B.cam, T0, p, JpA, JlA);
                preA = true;
            }
            if (eB >= 0 && actB && jB < B.n_free) {
                double T0[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) T0[i] = poses_c[12 * (size_t)jB + i];
                const double pb[3] = {s_pp[3 * plB], s_pp[3 * plB + 1], s_pp[3 * plB + 2]};
                pb_jac(B.cam, T0, pb, JpB, JlB);
                preB = true;
            }
        }
    };
    prep(FUSED);
    __shared__ double s_ctl[4];                             // FUSED: lambda, ok, cur, ni as the solver workgroup published them
    int c_steps = 0, c_qmax = 0, c_it = 0, c_max_it = 0, c_iters_done = 0;      // (thread 0, FUSED)
    if (FUSED) {
        // Nothing that can be done without the solution is left behind the wait: the current poses go to LDS, both of the lane's edges are linearised at the
        // current state (pass 1: rhs -= W_e^T dp_j needs only those Jacobians and dp), and behind the solver's word the solution, lambda, ok, cur and
        // ni arrive in ONE batch of loads (the solver stores them side by side: dl[0 .. D + 3]).  (Measured and dropped: every value as two tagged
        // 64-bit words that the lanes poll themselves -- no word, no drain in the solver -- was 1.7 % SLOWER end to end: twice the bytes from 171
        // workgroups at the same instant, on one memory channel.)
        for (int i = threadIdx.x; i < 12 * B.n_poses; i += UPC_T) s_Tc[i] = poses_c[i];
        __shared__ int s_flag;
        if (threadIdx.x == 0) {
            // (the bookkeeping fields the LM decision will update: read here, while the solver works -- the decision, if this workgroup takes the last ticket, then
            // stores new values without a trip for the old ones; none of them is written by anybody else during this launch)
            c_steps = ctl_->steps; c_qmax = ctl_->qmax; c_it = ctl_->it; c_max_it = ctl_->max_it; c_iters_done = ctl_->iters_done;
            const int wseq = c_steps + 1;                  // (steps: the last workgroup of the previous step's launch wrote it)
            int f = 0;
            for (int it = 0; it < (1 << 21) && !f; ++it) { f = __hip_atomic_load(&ctl_->chol_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == wseq; if (!f) __builtin_amdgcn_s_sleep(2); }
            s_flag = f;
        }
        __syncthreads();
#ifdef P2_STAMPS
        tw_ = wall_clock64();
#endif
        const bool got = s_flag != 0;
        if (got && (int)threadIdx.x < B.D + 4) {
            const double v = pb_ld(B.dl + threadIdx.x);
            if ((int)threadIdx.x < B.D) s_dp[threadIdx.x] = v; else s_ctl[threadIdx.x - B.D] = v;
        }
        const int seen = __syncthreads_and(got ? 1 : 0);    // (also publishes s_Tc, s_dp and s_ctl to the workgroup)
        lambda = p2_uniform(s_ctl[0]);
        ok = seen != 0 && __builtin_amdgcn_readfirstlane((int)(s_ctl[1] != 0.0)) != 0;      // (a wait that ran out: the step counts as failed)
        live = live && ok;
        if (!ok) { eA = -1; actA = 0; q0 = q1 = 0; preA = preB = false; eB = -1; actB = 0; nov = 0; }
    } else {
        for (int i = threadIdx.x; i < B.D; i += UPC_T) s_dp[i] = B.dl[i];       // the solution (k_ba_chol16, phase2 = 1)
    }
    // current poses -> s_Tc; trial poses -> s_T: copies for the fixed ones (and for all of them after a failed factorisation), exp(dp) * T for the free
    // ones with three lanes per pose, a row each (one lane per pose made this prelude the kernel's register peak, on one wave in eight)
    const int n_exp = ok ? B.n_free : 0;
    for (int i = threadIdx.x; i < 12 * B.n_poses; i += UPC_T) {
        const double v = FUSED ? s_Tc[i] : poses_c[i];      // (FUSED: this lane's own entries of a moment ago)
        if (!FUSED) s_Tc[i] = v;
        if (i >= 12 * n_exp) { s_T[i] = v; if (pose_wg) poses_t[i] = v; }
    }
    for (int t = threadIdx.x; t < 3 * n_exp; t += UPC_T) {
        const int j = t / 3, r = t - 3 * j;
        const double* d = FUSED ? s_dp + 6 * j : B.dl + 6 * j;
        double row[3], tr;
        p2_exp_mul_row(d, FUSED ? s_Tc + 12 * j : poses_c + 12 * (size_t)j, r, row, tr);
        if (pose_wg && r == 0) {                   // the pose part of the gain ratio and of the step size, once
            double sc = 0, mx = 0;
            for (int a = 0; a < 6; ++a) { sc += d[a] * (lambda * d[a] + B.bp[6 * j + a]); mx = fmax(mx, fabs(d[a])); }
            atomicAdd(&B.scal[2], sc);
            atomicMax((unsigned long long*)&B.scal[7], (unsigned long long)__double_as_longlong(mx));
        }
        double* o = s_T + 12 * j;
        o[3 * r] = row[0]; o[3 * r + 1] = row[1]; o[3 * r + 2] = row[2]; o[9 + r] = tr;
        if (pose_wg) { double* g = poses_t + 12 * (size_t)j; g[3 * r] = row[0]; g[3 * r + 1] = row[1]; g[3 * r + 2] = row[2]; g[9 + r] = tr; }
    }
    // the pose workgroup's atomics on scal[2] / scal[7] above come from lanes of several waves; the fence-free ticket at the end of this kernel has only
    // thread 0's wave drain vmcnt before it takes its ticket, and a workgroup-scope barrier does not drain it: every wave of that workgroup
    // drains here, so the atomics are performed before it can take a ticket (k_ba_round got the same fix in ac73ae7)
    if (pose_wg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    P2_STAMP()
    double chi = 0, sc = 0, mx = 0;
    for (int r = 0; r < rep; ++r, k += P) {
        if (r) { prep(false); rhs[0] = rhs[1] = rhs[2] = 0; }
        if (!actA) eA = -1;
        if (!actB) eB = -1;
        // rhs -= W_e^T dp_j for one edge of point `pt` to free pose j: into acc (the lane's own point) or, for an overflow edge, into the point's LDS cell
        auto pass1 = [&](int j, double w, const double (&pt)[3], double (&acc)[3]) {
            double T[12], Jp[2][6], Jl[2][3];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = s_Tc[12 * j + i];
            pb_jac(B.cam, T, pt, Jp, Jl);
            const double* d6 = s_dp + 6 * j;
            double t0 = 0, t1 = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) { t0 += Jp[0][a] * d6[a]; t1 += Jp[1][a] * d6[a]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] -= w * (Jl[0][c] * t0 + Jl[1][c] * t1);
        };
        auto pass1_pre = [&](int j, double w, const double (&Jp)[2][6], const double (&Jl)[2][3], double (&acc)[3]) {      // the same from Jacobians built before the wait
            const double* d6 = s_dp + 6 * j;
            double t0 = 0, t1 = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) { t0 += Jp[0][a] * d6[a]; t1 += Jp[1][a] * d6[a]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] -= w * (Jl[0][c] * t0 + Jl[1][c] * t1);
        };
        // H_ll and b_l are not needed while pass 1 runs (the kernel's register peak): lane 0 of the point parks them in LDS, all four lanes take them
        // back afterwards (same wavefront: program order is enough)
        double* const park = s_hb + (threadIdx.x >> 2);
        if (live && sub == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) park[i * (UPC_T / 4)] = H[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) park[(6 + i) * (UPC_T / 4)] = bl[i];
        }
        if (live) {
            if (sub == 0) { rhs[0] = bl[0]; rhs[1] = bl[1]; rhs[2] = bl[2]; }
            if (preA && r == 0) pass1_pre(jA, wA, JpA, JlA, rhs);
            else if (eA >= 0 && jA < B.n_free) pass1(jA, wA, p, rhs);
#pragma nounroll
            for (int q = q_ser; q < q1; q += 4) {         // (nothing, usually: the edges the list did not take)
                const int e = B.edges_by_point ? q : B.pt_edges[q], j = B.e_pose[e];
                if (!B.active[e] || j >= B.n_free) continue;
                pass1(j, Wt[e], p, rhs);
            }
        }
        // overflow edges, one per lane (the first one was fetched, and FUSED linearised, in prep)
        for (int i = threadIdx.x; i < nov; i += UPC_T) {
            int e = eB, j = jB, plq = plB; double w = wB; bool pre = preB && r == 0;
            if (i >= UPC_T) {
                const int it = s_list[i], q = it & 0xFFFFFF;
                plq = it >> 24; e = B.edges_by_point ? q : B.pt_edges[q];
                if (!B.active[e]) e = -1; else { j = B.e_pose[e]; w = Wt[e]; }
                pre = false;
            }
            double acc[3] = {0, 0, 0};
            if (e >= 0 && j < B.n_free) {
                if (pre) pass1_pre(j, w, JpB, JlB, acc);
                else { const double pt[3] = {s_pp[3 * plq], s_pp[3 * plq + 1], s_pp[3 * plq + 2]}; pass1(j, w, pt, acc); }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) s_c[9 * i + c] = acc[c];
        }
        __syncthreads();                                    // the entries' shares of rhs are in their cells (and nobody reads s_pp's CURRENT points any more)
#pragma unroll
        for (int m = 0; m < UPC_OVF_MINE; ++m)
            if (m < n_mine) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rhs[c] += s_c[9 * mine[m] + c];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) rhs[c] = ba_quad_sum(rhs[c]);
        P2_STAMP()
        double Hn[6] = {0, 0, 0, 0, 0, 0}, bn[3] = {0, 0, 0}, pn[3] = {0, 0, 0};
        // second pass over the point's edges at the TRIAL state: robust chi2 (the LM decision) and the whole linearisation (Huber weight,
        // H_ll, b_l) that the next step needs if this one is accepted -- k_ba_lin2's point part at the price of a few dozen FMAs per edge
        auto pass2 = [&](int e, int j, const float* uv, const double (&pt)[3], double (&Ha)[6], double (&ba)[3]) {
            double rr[2], w, rho0, Jp[2][6], Jl[2][3];
            ba_edge(B.cam, s_T + 12 * j, pt, uv, robust, B.delta, rr, w, rho0, Jp, Jl);
            chi += rho0;
            Wn[e] = w;
            ba[0] -= w * (Jl[0][0] * rr[0] + Jl[1][0] * rr[1]); ba[1] -= w * (Jl[0][1] * rr[0] + Jl[1][1] * rr[1]); ba[2] -= w * (Jl[0][2] * rr[0] + Jl[1][2] * rr[1]);
            Ha[0] += w * (Jl[0][0] * Jl[0][0] + Jl[1][0] * Jl[1][0]); Ha[1] += w * (Jl[0][0] * Jl[0][1] + Jl[1][0] * Jl[1][1]); Ha[2] += w * (Jl[0][0] * Jl[0][2] + Jl[1][0] * Jl[1][2]);
            Ha[3] += w * (Jl[0][1] * Jl[0][1] + Jl[1][1] * Jl[1][1]); Ha[4] += w * (Jl[0][1] * Jl[0][2] + Jl[1][1] * Jl[1][2]); Ha[5] += w * (Jl[0][2] * Jl[0][2] + Jl[1][2] * Jl[1][2]);
        };
        if (live) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double Hq[6], bq[3];
#pragma unroll
            for (int i = 0; i < 6; ++i) Hq[i] = park[i * (UPC_T / 4)];
#pragma unroll
            for (int i = 0; i < 3; ++i) bq[i] = park[(6 + i) * (UPC_T / 4)];
            const double Hs[9] = {Hq[0], Hq[1], Hq[2], Hq[1], Hq[3], Hq[4], Hq[2], Hq[4], Hq[5]};
            double h[9];
            ba_inv3_damped(Hs, lambda, h);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double d = h[3 * a] * rhs[0] + h[3 * a + 1] * rhs[1] + h[3 * a + 2] * rhs[2];
                pn[a] = p[a] + d;
                if (sub == 0) { pts_t[3 * (size_t)k + a] = pn[a]; if (ovf_ok) s_pp[3 * pl + a] = pn[a]; sc += d * (lambda * d + bq[a]); mx = fmax(mx, fabs(d)); }
            }
        }
        __syncthreads();                                    // s_pp holds the TRIAL points
        if (live) {
            if (eA >= 0) { const float uvv[2] = {uvA.x, uvA.y}; pass2(eA, jA, uvv, pn, Hn, bn); }
#pragma nounroll
            for (int q = q_ser; q < q1; q += 4) {
                const int e = B.edges_by_point ? q : B.pt_edges[q];
                if (!B.active[e]) continue;
                pass2(e, B.e_pose[e], B.e_uv + 2 * (size_t)e, pn, Hn, bn);
            }
        }
        for (int i = threadIdx.x; i < nov; i += UPC_T) {
            int e = eB, j = jB, plq = plB; float2 uv = uvB;
            if (i >= UPC_T) {
                const int it = s_list[i], q = it & 0xFFFFFF;
                plq = it >> 24; e = B.edges_by_point ? q : B.pt_edges[q];
                if (!B.active[e]) e = -1; else { j = B.e_pose[e]; uv = reinterpret_cast<const float2*>(B.e_uv)[e]; }
            }
            double Ha[6] = {0, 0, 0, 0, 0, 0}, ba[3] = {0, 0, 0};
            if (e >= 0) {
                const double pt[3] = {s_pp[3 * plq], s_pp[3 * plq + 1], s_pp[3 * plq + 2]};
                const float uvv[2] = {uv.x, uv.y};
                pass2(e, j, uvv, pt, Ha, ba);
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) s_c[9 * i + c] = Ha[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) s_c[9 * i + 6 + c] = ba[c];
        }
        __syncthreads();                                    // the entries' shares of H_ll / b_l are in their cells
#pragma unroll
        for (int m = 0; m < UPC_OVF_MINE; ++m)
            if (m < n_mine) {
#pragma unroll
                for (int c = 0; c < 6; ++c) Hn[c] += s_c[9 * mine[m] + c];
#pragma unroll
                for (int c = 0; c < 3; ++c) bn[c] += s_c[9 * mine[m] + 6 + c];
            }
#pragma unroll
        for (int i = 0; i < 6; ++i) Hn[i] = ba_quad_sum(Hn[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) bn[i] = ba_quad_sum(bn[i]);
        if (live && sub == 0) p2_rec_store(rec_n, k, Hn, bn, pn);
    }
    P2_STAMP()
    chi = vo_wave_sum_f64(chi); sc = vo_wave_sum_f64(sc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    constexpr int NWV = UPC_T / 64;
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; s_w[w] = chi; s_w[NWV + w] = sc; s_w[2 * NWV + w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a_ = 0, b_ = 0, m_ = 0;
        for (int w = 0; w < NWV; ++w) { a_ += s_w[w]; b_ += s_w[NWV + w]; m_ = fmax(m_, s_w[2 * NWV + w]); }
        // The partials leave as write-through (sc1) stores and the ticket follows once they have been performed: no release fence -- on this
        // part a device-scope fence writes the XCD's dirty L2 lines back (this kernel has just produced megabytes of them), 1.5-5 us per
        // workgroup in the clock stamps.  The last workgroup reads the partials past its own L2 (pb_ld).
        pb_st(B.partU + 3 * (size_t)bx, a_);
        pb_st(B.partU + 3 * (size_t)bx + 1, b_);
        pb_st(B.partU + 3 * (size_t)bx + 2, m_);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(&ctl_->arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gp;      // (gp point workgroups + the pose workgroup)
    }
    __syncthreads();
    P2_STAMP()
#ifdef P2_STAMPS
    if (stamp_ && !s_last) { long long tw2_ = 0;
#ifdef P2_STAMPS
        if (FUSED) tw2_ = tw_ % 1000000000ll;
#endif
        printf("[upchi2 wg %d of %d] abs flag seen %lld | prelude %lld pass1 %lld pass2 %lld reduce+ticket %lld  (abs ticks: poses done %lld, end %lld)\n", bx, gp, tw2_, tq_[1] - tq_[0], tq_[2] - tq_[1], tq_[3] - tq_[2], tq_[4] - tq_[3], tq_[1] % 1000000000ll, tq_[4] % 1000000000ll); }
#endif
    if (!s_last) return;
    // block 0's atomics on scal[2] / scal[7] were performed at the memory side (and before block 0 took its ticket); this workgroup's XCD may still hold
    // the line it read scal[3] from at the start: agent-scope loads go past that L2 (there is no acquire fence in this kernel any more).  Requested
    // here, ahead of the partial sums, so that the two round trips overlap
    double sc2 = 0, sc7 = 0;
    if (threadIdx.x == 0) { sc2 = pb_ld(B.scal + 2); sc7 = pb_ld(B.scal + 7); }
    {   // the last workgroup: the partials of every workgroup, then g2o's gain-ratio test and lambda policy (as k_ba_chi_control)
        const double* pu = B.partU;
        double a = 0, b = 0, m = 0;
        for (int i = threadIdx.x; i <= gp; i += UPC_T) { a += pb_ld(pu + 3 * i); b += pb_ld(pu + 3 * i + 1); m = fmax(m, pb_ld(pu + 3 * i + 2)); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); m = fmax(m, __shfl_xor(m, o, 64)); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; s_w[w] = a; s_w[NWV + w] = b; s_w[2 * NWV + w] = m; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        BaCtl* c = ctl_;
        c->arrived = 0;
        double s1 = 0, sB = 0, mB = 0;
        for (int w = 0; w < NWV; ++w) { s1 += s_w[w]; sB += s_w[NWV + w]; mB = fmax(mB, s_w[2 * NWV + w]); }
        const double s2 = sc2 + sB;
        const double m7 = fmax(sc7, mB);
        const double tmp = ok ? s1 : DBL_MAX;
        const double scale = (ok ? s2 : 0.0) + 1e-3;
        // (FUSED: the solver workgroup of this launch may have written cur, lambda and ni -- past this XCD's L2, which may hold the line from before)
        const double cur = FUSED ? s_ctl[2] : c->cur, lam = FUSED ? s_ctl[0] : c->lambda, ni = FUSED ? s_ctl[3] : c->ni;      // (FUSED: the values the solver workgroup of this launch published)
        const double rho = (cur - tmp) / scale;
        bool converged = false;
        int accept = 0;
        if (rho > 0 && isfinite(tmp)) {
            double a = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
            a = fmin(a, 2.0 / 3.0);
            c->lambda = lam * fmax(1.0 / 3.0, a); c->ni = 2; c->cur = tmp;
            c->buf = buf_ ^ 1; c->need_lin = 1; accept = 1; // trial state becomes the current state,
            c->lbuf = lb ^ 1;                               // the linearisation this launch wrote at it becomes the current one,
            B.scal[0] = tmp; B.scal[4] = 0;                 // and its chi2 is the trial chi2 (k_ba_chol16 takes it over as `cur`)
        } else { c->lambda = lam * ni; c->ni = 2 * ni; }
        if (ok) converged = m7 < 1e-10;
#ifdef P2_STAMPS
        if (B.n_points < 1000) printf("[upchi2 ctl] stage %d it %d qmax %d ok %d cur %.12e trial %.12e rho %.3e scale %.3e m7 %.3e (pose part %.3e) lambda %.3e\n", c->stage, c->it, c->qmax, (int)ok, cur, tmp, rho, scale, m7, sc7, c->lambda);
#endif
        if (!FUSED) { c_steps = c->steps; c_qmax = c->qmax; c_it = c->it; c_max_it = c->max_it; c_iters_done = c->iters_done; }
        c_qmax += 1; c->steps = c_steps + 1;
        if (!(rho < 0 && c_qmax < 10 && !converged)) {      // this LM iteration is over
            c->iters_done = c_iters_done + 1;
            if (c_qmax == 10 || rho == 0 || converged || c_it + 1 >= c_max_it) c->finished = 1;
            c->it = c_it + 1; c_qmax = 0;
        }
        c->qmax = c_qmax;
        s_accept = accept;
    }
    __syncthreads();
    if (s_accept) {
        for (int i = threadIdx.x; i < 36 * B.n_free; i += UPC_T) B.Hpp[i] = 0;
        for (int i = threadIdx.x; i < B.D; i += UPC_T) B.bp[i] = 0;
    }
    P2_STAMP()
#ifdef P2_STAMPS
    if (threadIdx.x == 0 && bz == 0 && (ctl_->it == 5 || ctl_->it == 4)) printf("[upchi2 last wg abs end %lld] ", tq_[5] % 1000000000ll);
    if (threadIdx.x == 0 && bz == 0 && ctl_->it == 5) printf("[upchi2 last wg %d of %d] prelude %lld pass1 %lld pass2 %lld reduce+ticket %lld control %lld\n", bx, gp, tq_[1] - tq_[0], tq_[2] - tq_[1], tq_[3] - tq_[2], tq_[4] - tq_[3], tq_[5] - tq_[4]);
#endif
}

__global__ __launch_bounds__(UPC_T) void k_ba_upchi2(BaBatch Q, int rep) {      // (two workgroups per CU: 128 VGPRs -- at 129 the kernel ran one)
    BA_PROBLEM_COPY(Q)
    if (ctl_->finished || B.D > BA_FOLD_D) return;
    ba_upchi2_body<false>(B, ctl_, rep, (int)blockIdx.x, (int)blockIdx.z);
}
