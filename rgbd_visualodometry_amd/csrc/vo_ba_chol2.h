// vo_ba_chol2.h -- the reduced system's Cholesky + solve for D <= CH2_MAXD (and, with the last row block outside LDS, D = 177 .. 191: "SPILL" in the body), second generation (k_ba_chol16v2, and workgroup 0 of k_ba_cholup):
// the arithmetic of k_ba_chol16 (16-column panels, in-register block factor, f64-MFMA tiles) run as a DATAFLOW inside one workgroup instead
// of as barrier-separated phases, on a TILE-MAJOR matrix, with every panel solve turned into a product with the block's inverse.
// Included by vo_ba.hip behind the first generation's DPP helpers.  Reference: the linear solver of src/backend.cpp:23-27 (g2o's dense
// Cholesky on the Schur complement), one call per LM iteration of :140-159.
//
// Layout.  The lower triangle of the augmented matrix [S b; b^T 0] is kept as 16x16 tiles, tile (i, j <= i) at (i (i + 1) / 2 + j) * 272
// doubles, entry (r, c) of a tile at 17 r + c (ch2_sidx).  The row stride of 17 makes every access pattern of the solve free of bank
// conflicts (16 lanes reading one column of 16 rows: banks 34 r + 2 c mod 64, all different; 16 lanes reading one row: contiguous), and
// the tile order makes every address "tile base (a scalar) + a per-lane constant + an immediate": a trailing-tile update is ~30
// instructions, where the packed-row layout of the first generation spent ~150 on index arithmetic.  The Schur kernel writes S in the
// same layout in global memory (BaDev::s_tiles), so the matrix still arrives by straight global -> LDS DMA; the waves that fetched it
// clear it behind the load.  The right-hand side rides along as row D, so the forward substitution falls out of the factorisation.
//
// Dataflow.  The factorisation is a chain of 16x16 block factorisations (POTRF) that nothing can shorten: block k+1 needs block k's panel
// solve (TRSM) of its own 16 rows and that tile's product with itself (SYRK).  The first generation put two workgroup barriers into every
// link of that chain.  Here the waves have ROLES and hand work to each other through words in LDS; after the load nobody executes s_barrier:
//   wave 0       the chain: factors diagonal block k in registers (a row per lane, DPP row broadcasts), publishing every finished COLUMN at
//                once (16 values + the pivot's inverse into a column buffer, then the word `prog`); then takes tiles (k+1, k) and (k+1, k+1)
//                from the solver that owns row block k+1 (`rdy`), solves the first as a transposed MFMA product with W_k -- the result
//                registers are the SYRK's operands -- and applies it to the second.  At the end it runs the backward substitution.
//   wave 1       the inverter, alone on its SIMD: W_k = L_kk^-T, the first generation's DPP panel solve applied to the identity, streamed one
//                column behind the factorisation and stored column by column into the diagonal tile (which nobody needs as L_kk again);
//                the word `inv`.  With W_k every panel solve below the block is X = A W_k on the matrix cores (4 MFMAs per tile) and
//                the backward substitution needs no triangular solve either.
//   waves 2,3,6,7  solvers, two per remaining SIMD; row block i belongs to solver (i - 1) mod 4, which alone writes its tiles (except what wave 0
//                takes).  LEFT-LOOKING with a fixed schedule: at stage k a solver brings its tiles of block column k up to date with
//                all k panels in ONE read-modify-write (two accumulators per tile, the next panel's operands in flight behind the MFMAs)
//                while block k is being factored, waits for W_k, solves them.  The solver that owns row block k+2 is on the chain
//                wave 0 -> W_k -> L(k+2, k) -> tiles (k+2, k+1), (k+2, k+2) -> wave 0: it has given those two tiles their panels 0 .. k-1 one
//                stage ahead, solves the critical tile first and alone (as a transposed product, so that the result registers are the
//                operands of what follows), adds panel k to the two tiles and raises `rdy`; `rowdone[i]` = stages solved for row block i.
//   waves 4, 5   help to load; wave 4 clears S in global memory; wave 5 owns the last row block of a system that does not fit (SPILL: the row's tiles in its registers).  (An MFMA, a DPP operation and a plain FMA in double precision all draw on
//                the same 16 lanes per clock of their SIMD -- a v_mfma_f64_16x16x4 holds them for ~64 clocks -- so work placed beside
//                wave 0 or beside the inverter slows the chain by more than it relieves the solvers: measured)
// Words are monotone and never reset.  Producers store data, then the word, in program order (LDS executes a wave's instructions in order;
// the asm memory clobbers keep the compiler from reordering); consumers poll the word, then load.  Every wait is bounded: after
// CH2_SPIN_LIMIT polls a wave raises `abort`, stops waiting and the solve is reported as failed -- a logic error can produce a wrong
// (flagged) result, not a hung GPU.
#pragma once
#include <type_traits>
#ifndef CH2_EXP
#define CH2_EXP 0                           // tools/chol_bench: timing experiments that leave work out (results are wrong); 0 in the library
#endif

#define CH2_T 512
#define CH2_NW (CH2_T / 64)
#define CH2_NC (CH2_NW - 1)                 // waves that load the matrix: everyone but wave 0
#define CH2_NS 4                            // solvers: waves 2, 3, 6, 7 (two per SIMD); wave 1 inverts the blocks with SIMD 1 to itself; wave 4 clears S
#define CH2_SPIN_LIMIT (1 << 21)
#define CH2_RS 17                           // row stride of a tile (doubles)
#define CH2_TS 272                          // tile stride (16 rows x 17)
#define CH2_HEAD 528                        // doubles in front of the tiles: the published columns [2][16][16], 16 that take stores nobody reads
#define CH2_MAXD 174                        // 11 row blocks of the augmented matrix: 66 tiles = 140 KiB of LDS
#define CH2_SPILL_T 12                      // D = 177 .. 191 (12 row blocks, the right-hand side inside the last): row block 11 stays in GLOBAL memory (round 6, see "spill" in the body)

#ifdef CH2_STAMPS
#define CH2_STAMP(slot) { if (lane == 0) { const int s_ = (slot); if (s_ < 128) ((long long*)B.W)[wave * 128 + s_] = clock64(); } }
#else
#define CH2_STAMP(slot)
#endif
struct Ch2Flags { int prog, dma, init, ok, abort_, inv, bsdone; int rowdone[16], rdy[16]; };      // bsdone: the backward substitution has read the spilled row block (spill only)

__host__ __device__ inline int ch2_tix(int i, int j) { return i * (i + 1) / 2 + j; }
__host__ __device__ inline size_t ch2_sidx(int r, int c) { return (size_t)ch2_tix(r >> 4, c >> 4) * CH2_TS + (size_t)((r & 15) * CH2_RS + (c & 15)); }
__host__ __device__ inline size_t ch2_s_doubles(int D) { const int T = (D + 15) >> 4; return (size_t)T * (T + 1) / 2 * CH2_TS; }      // S in global memory: the row blocks of rows < D
__host__ __device__ inline bool ch2_fits(int D) { return D <= CH2_MAXD || (((D + 15) >> 4) == CH2_SPILL_T && ((D + 16) >> 4) == CH2_SPILL_T); }      // sizes the second generation takes
__host__ __device__ inline size_t ch2_lds_bytes(int D) {
    const int T = (D + 16) >> 4;
    const size_t tiles = D > CH2_MAXD ? (size_t)(T - 1) * T / 2 + 2 : (size_t)T * (T + 1) / 2;      // spill: row blocks 0 .. T-2 and the last row block's two right-most tiles
    return sizeof(double) * (CH2_HEAD + tiles * CH2_TS);
}

// the words and the published columns are read and written through explicit LDS pointers: a volatile access through a generic pointer stays a
// flat_load / flat_store (the address-space inference leaves volatile accesses alone)
typedef __attribute__((address_space(3))) int ch2_lds_int;
typedef __attribute__((address_space(3))) double ch2_lds_f64;
__device__ __forceinline__ int ch2_peek(const int* p) { return __builtin_amdgcn_readfirstlane(*(const volatile ch2_lds_int*)p); }
__device__ __forceinline__ void ch2_wait_ge(int* p, int target, int* abort_) {
    if (ch2_peek(p) < target) {
        int i = 0;
        for (; i < CH2_SPIN_LIMIT; ++i) {
#ifndef CH2_NOSLEEP
            __builtin_amdgcn_s_sleep(1);
#endif
            if (ch2_peek(p) >= target) break;
            if ((i & 1023) == 1023 && ch2_peek(abort_)) break;
        }
        if (i >= CH2_SPIN_LIMIT) *(volatile ch2_lds_int*)abort_ = 1;
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ch2_set(int* p, int v) { asm volatile("" ::: "memory"); *(volatile ch2_lds_int*)p = v; }
__device__ __forceinline__ void ch2_inc(int* p) { asm volatile("" ::: "memory"); (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// column J of the in-register block factorisation (ChCol of the first generation) + publication: lane c of the column buffer's row J
// gets L[c][J] (c > J) or 1 / L[J][J] (c == J); lanes c < J hold leftovers nobody reads
template <int J> struct Ch2Col {
    static __device__ __forceinline__ void run(double (&a)[CH_NB], double* s_colp, int* s_prog, int progbase, int r16) {
        // 1 / sqrt(d) = y (1 + e u), y = rsq(d), e = 1 - d y^2, u = 1/2 + 3/8 e (ba_rsqrt_parts); the column is a[J] y (1 + e u) = l0 + (l0 e) u with
        // l0 e beside u: the dependent chain from pivot to column is bcast, rsq, d y, e, u | l0 e, l -- one operation shorter than q = e u first
        const double d = ch_bcast_v<J>(a[J]);
        const double y = __builtin_amdgcn_rsq(d);
        const double l0 = a[J] * y;
        const double e = fma(-(d * y), y, 1.0);
        const double u = fma(e, 0.375, 0.5), l0e = l0 * e, ye = y * e;
        const double l = ch_fma_for_dpp(l0e, u, l0);
        const double pinv = fma(ye, u, y);
        a[J] = l;
        ChRank1<J, J + 1>::run(a, l);
        s_colp[J * CH_NB + r16] = r16 == J ? pinv : l;
        ch2_set(s_prog, progbase + J + 1);
        Ch2Col<J + 1>::run(a, s_colp, s_prog, progbase, r16);
    }
};
template <> struct Ch2Col<CH_NB> { static __device__ __forceinline__ void run(double (&)[CH_NB], double*, int*, int, int) {} };

template <int KK, int C> struct Ch2StreamRow {
    static __device__ __forceinline__ void run(double (&x)[CH_NB], double Lk) { ch_fnma_bcast<C>(x[C], Lk, x[KK]); Ch2StreamRow<KK, C + 1>::run(x, Lk); }
};
template <int KK> struct Ch2StreamRow<KK, CH_NB> { static __device__ __forceinline__ void run(double (&)[CH_NB], double) {} };
// one column step of the streamed solve: x[K] *= 1 / L[K][K], x[C] -= L[C][K] x[K] (C > K)
template <int K> struct Ch2Stream {
    // `f` / Lk[K]: the word and column K as read one step ago (speculatively: valid iff f says so).  Column K+1 is requested before column
    // K is used, so that a wave that is behind the factorisation pays no LDS round trip per step and catches up.
    // `w`: row r16 of the result's home; x[K] is final after step K and leaves at once, so that behind the
    // last column one store remains
    static __device__ __forceinline__ void run(double (&x)[CH_NB], double (&Lk)[CH_NB], int f, const double* s_colp, int* s_prog, int progbase, int r16, int* abort_, double* w) {
        int fn = 0;
        if (K + 1 < CH_NB) {
            fn = __builtin_amdgcn_readfirstlane(*(const volatile ch2_lds_int*)s_prog);
            Lk[K + 1 < CH_NB ? K + 1 : K] = *(const volatile ch2_lds_f64*)(s_colp + (K + 1 < CH_NB ? K + 1 : K) * CH_NB + r16);
        }
        if (f < progbase + K + 1) {                             // column K was not out yet when it was read: wait, read again
            ch2_wait_ge(s_prog, progbase + K + 1, abort_);
            Lk[K] = *(const volatile ch2_lds_f64*)(s_colp + K * CH_NB + r16);
        }
        double l = Lk[K];
        ch_exec_settle(l);
        x[K] = ch_mul_bcast<K>(l, x[K]);
        w[K] = x[K];
        Ch2StreamRow<K, K + 1>::run(x, l);
        Ch2Stream<K + 1>::run(x, Lk, fn, s_colp, s_prog, progbase, r16, abort_, w);
    }
};
template <> struct Ch2Stream<CH_NB> { static __device__ __forceinline__ void run(double (&)[CH_NB], double (&)[CH_NB], int, const double*, int*, int, int, int*, double*) {} };

// wait until arr[i] >= need for every i of the 16-bit set `mask`: ONE load per poll, lane i reads word i (an LDS round trip costs ~130
// clocks whether it fetches one word or sixteen)
__device__ __forceinline__ void ch2_wait_set(int* arr, unsigned mask, int need, int* abort_) {
    const int li = threadIdx.x & 15;
    const bool want = (mask >> li) & 1u;
    for (int i = 0; i < CH2_SPIN_LIMIT; ++i) {
        const int v = *(const volatile ch2_lds_int*)(arr + li);
        if (__ballot(want && v < need) == 0ull) break;
        if ((i & 1023) == 1023 && ch2_peek(abort_)) break;
        if (i == CH2_SPIN_LIMIT - 1) *(volatile ch2_lds_int*)abort_ = 1;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}


// N trailing tiles at once: tile (ti, tj) -= sum over panels k0 .. k1 of L(ti, k) L(tj, k)^T, 4 MFMAs per tile and panel, interleaved over the tiles.
// Lane l holds A[l & 15][l >> 4 + 4 q], B likewise, C: row (l >> 4) + 4 q, column l & 15.  ONE read-modify-write of the tile for all its panels.
template <int N>
__device__ __forceinline__ void ch2_tiles_mp(double* s_L, const int (&ti)[3], const int (&tj)[3], int k0, int k1, int o_op, int o_c) {
    // two accumulators per tile and the next panel's operands requested before this panel's MFMAs (an LDS round trip is 130+ clocks)
    double c[N][4];
    double* pc[N];
    const double* pa[N]; const double* pb[N];
    f64x4 acc0[N], acc1[N];
    double a[N][4], b[N][4];
#pragma unroll
    for (int t = 0; t < N; ++t) {
        pc[t] = s_L + ch2_tix(ti[t], tj[t]) * CH2_TS + o_c;
        pa[t] = s_L + ch2_tix(ti[t], k0) * CH2_TS + o_op;
        pb[t] = s_L + ch2_tix(tj[t], k0) * CH2_TS + o_op;
        acc0[t] = f64x4{0.0, 0.0, 0.0, 0.0}; acc1[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[t][q] = pa[t][4 * q]; b[t][q] = pb[t][4 * q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) c[t][q] = pc[t][4 * CH2_RS * q];
    }
    for (int k = k0; k <= k1; ++k) {
        double an[N][4], bn[N][4];
        const int step = k < k1 ? CH2_TS : 0;                   // tiles (i, k) and (i, k + 1) are neighbours in memory; the last panel re-reads itself
#pragma unroll
        for (int t = 0; t < N; ++t) {
            pa[t] += step; pb[t] += step;
#pragma unroll
            for (int q = 0; q < 4; ++q) { an[t][q] = pa[t][4 * q]; bn[t][q] = pb[t][4 * q]; }
        }
#pragma unroll
        for (int t = 0; t < N; ++t) {
            acc0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][0], b[t][0], acc0[t], 0, 0, 0);
            acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][1], b[t][1], acc1[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < N; ++t) {
            acc0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][2], b[t][2], acc0[t], 0, 0, 0);
            acc1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][3], b[t][3], acc1[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < N; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[t][q] = an[t][q]; b[t][q] = bn[t][q]; }
    }
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) pc[t][4 * CH2_RS * q] = c[t][q] - (acc0[t][q] + acc1[t][q]);
}

// x_out receives the solution, B.scal[3] whether the system was positive definite (and no wait ran out).
// PUB (k_ba_cholup: the update workgroups of the SAME launch read the results): everything this workgroup hands on -- the control block's
// fields, the cleared sums, the solution, scal[3] -- leaves as write-through stores, and ctl->chol_seq = steps + 1 follows once all of
// them have been performed (the same fence-free publication as k_ba_upchi2's ticket).
__device__ __forceinline__ void ch2_pub_i(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ch2_pub_d(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool PUB>
__device__ __forceinline__ void ba_chol16v2_body(const BaDev& B, BaCtl* ctl_, double* s_mem, double* x_out, bool clear_after_load, const bool check_finished = false) {
    // the wave number goes through readfirstlane: as a per-thread value every role branch below would be compiled as divergent (EXEC masks,
    // loop counters and tile indices in vector registers)
    const int D = B.D, DA = D + 1, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
    const double* const A = B.S;
    const int nblk = (D + CH_NB - 1) / CH_NB;                  // column blocks = row blocks of S
    const int T = (DA + CH_NB - 1) / CH_NB;                    // row blocks of the augmented matrix
    const int iD = D >> 4, rD = D & 15;                        // where the right-hand-side row lives
    double* const s_col = s_mem;                               // [2][16][16] published columns of the block being factored (stage parity); x during the backward substitution
    double* const s_L = s_mem + CH2_HEAD;                           // the tiles; the diagonal tile of a factored block holds W_k = L_kk^-T (not L_kk, which nobody needs again)
    const int o_row = CH2_RS * r16;                            // row r16 of a tile (a row per lane)
    const int o_op = CH2_RS * r16 + kq;                        // MFMA operand: row r16, columns kq + 4 q
    const int o_c = CH2_RS * kq + r16;                         // MFMA result: rows kq + 4 q, column r16
    // SPILL (D = 177 .. 191: 78 tiles, 66 + 2 fit beside the update role's static LDS).  The last row block R = nblk - 1 -- up to 15 rows of S and the
    // right-hand side -- stays in global memory, where the Schur launch left it, except its two right-most tiles (R, R-1) and (R, R), which wave 0 takes
    // over at stage R-1 like any row block's and which live in two LDS slots behind row block R-1.  Wave 5 (idle otherwise) owns the row: it holds the
    // tiles L(R, p) in REGISTERS, in the MFMA operand layout, from the moment they are final -- every product it forms is the transposed one (result
    // registers = operand registers, as in the solvers' critical step), so nothing is ever re-laid out -- and writes each to global memory once, for
    // the backward substitution.  The solver waves see Ts = T - 1 row blocks.
    const bool spill = D > CH2_MAXD;
    const int Rs = spill ? nblk - 1 : -1, Ts = spill ? T - 1 : T;
    double* const Ag = const_cast<double*>(A);
    auto tix = [&](int i, int j) { return i == Rs ? Rs * (Rs + 1) / 2 + (j - (Rs - 1)) : ch2_tix(i, j); };      // where tile (i, j) lies in LDS (row block Rs: j >= Rs - 1 only)
    auto sidx = [&](int r, int c) { return (size_t)tix(r >> 4, c >> 4) * CH2_TS + (size_t)((r & 15) * CH2_RS + (c & 15)); };
    auto g_ld = [&](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };      // (past this compute unit's vector cache: another wave wrote it)
    // The head: ONE trip to L2 for everything this workgroup needs of the control block and of scal[] -- lane i < 28 takes word i of the control block, lanes
    // 32 .. 47 the words of scal[0 .. 7]; the fields come out by readlane.  (Up to round 6 a field at a time behind a branch each: need_lin, first, lambda or
    // scal[4] -- and then thread 0's take-over of the fresh linearisation, another three dependent trips in front of block 0's factorisation.  A vector load, not
    // scalar ones: the barrier below waits for LDS and scalar memory together, a vector load stays in flight across it, so the tiles' DMA goes out while the head
    // is still on its way.)
    static_assert(sizeof(BaCtl) <= 112 && sizeof(BaCtl) % 4 == 0, "the head load takes the control block as sizeof / 4 <= 28 words");
    const int hw_ = *(lane < 32 ? reinterpret_cast<const int*>(ctl_) + min(lane, (int)sizeof(BaCtl) / 4 - 1) : reinterpret_cast<const int*>(B.scal) + ((lane - 32) & 15));
    auto h_i = [&](size_t byte_off) { return __builtin_amdgcn_readlane(hw_, (int)(byte_off / 4)); };
    auto h_d = [&](size_t byte_off) { return __hiloint2double(__builtin_amdgcn_readlane(hw_, (int)(byte_off / 4) + 1), __builtin_amdgcn_readlane(hw_, (int)(byte_off / 4))); };
    auto h_scal = [&](int i) { return __hiloint2double(__builtin_amdgcn_readlane(hw_, 32 + 2 * i + 1), __builtin_amdgcn_readlane(hw_, 32 + 2 * i)); };
    __shared__ Ch2Flags F;
    if (tid < 16) { F.rowdone[tid] = 0; F.rdy[tid] = -1; }
    if (tid == 0) { F.prog = 0; F.dma = 0; F.init = 0; F.ok = 1; F.abort_ = 0; F.inv = 0; F.bsdone = 0; }
    const double* const Hpp = B.Hpp;
    __syncthreads();                                           // the words are zero
    CH2_STAMP(127)
    // (values of the head: evaluated where they are first needed -- behind the DMA's issue for the loading waves, behind block 0's row loads for wave 0)
#define CH2_LAMBDA() ((h_i(offsetof(BaCtl, need_lin)) && h_i(offsetof(BaCtl, first))) ? 1e-5 * h_scal(4) : h_d(offsetof(BaCtl, lambda)))      /* as k_ba_init_S derives it */
    // The fresh linearisation is taken over and the trial sums are cleared at the END (take_over, wave 0 lane 0, in front of the solution's publication): nothing in this
    // workgroup reads those fields again, and the update workgroups look at them only behind the solver's word.
    auto take_over = [&](double lambda) {
        BaCtl* c = ctl_;
        const int need_lin0 = h_i(offsetof(BaCtl, need_lin)), first0 = h_i(offsetof(BaCtl, first));
        const double cur_v = need_lin0 ? h_scal(0) : h_d(offsetof(BaCtl, cur)), ni_v = (need_lin0 && first0) ? 2.0 : h_d(offsetof(BaCtl, ni));
        if (PUB) {
            // what the update workgroups need of the control block goes out beside the solution (dl[D]: lambda, [D + 1]: ok, [D + 2]: cur, [D + 3]: ni): they fetch
            // all of it in ONE batch of loads behind the word (three dependent trips before round 6)
            ch2_pub_d(B.dl + D, lambda); ch2_pub_d(B.dl + D + 2, cur_v); ch2_pub_d(B.dl + D + 3, ni_v);
            if (need_lin0) {
                ch2_pub_d(&c->cur, cur_v);
                if (first0) { ch2_pub_d(&c->lambda, lambda); ch2_pub_d(&c->ni, 2.0); ch2_pub_i(&c->first, 0); }
                ch2_pub_i(&c->need_lin, 0);
            }
            ch2_pub_d(B.scal + 1, 0.0); ch2_pub_d(B.scal + 2, 0.0); ch2_pub_d(B.scal + 7, 0.0);
        } else {
            if (need_lin0) {
                c->cur = cur_v;
                if (first0) { c->lambda = lambda; c->ni = 2; c->first = 0; }
                c->need_lin = 0;
            }
            B.scal[1] = 0; B.scal[2] = 0; B.scal[7] = 0;
        }
    };

    // W_k = L_kk^-T: the first generation's panel solve (a row per lane, DPP row broadcasts of the published columns) applied to the rows
    // of the identity -- ONE such pass per stage, by one wave, while wave 0 waits for it (the double-precision DPP pipe is shared by the
    // whole compute unit); every panel solve below the block is then X = A W_k on the matrix cores.
    auto invert_block = [&](int k) {
        const double* s_colp = s_col + (k & 1) * 256;
        double x[CH_NB], Lk[CH_NB];
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) { x[c] = c == r16 ? 1.0 : 0.0; Lk[c] = 0.0; }
        // streamed: a column step per published column, so that W_k is complete one step behind the factorisation (a wave that comes late
        // takes what is there in one batch)
        const int f0 = __builtin_amdgcn_readfirstlane(*(const volatile ch2_lds_int*)&F.prog);
        Lk[0] = *(const volatile ch2_lds_f64*)(s_colp + r16);
        // Lane r16 writes row r16 of tile (k, k), whose rows wave 0 took into registers before it published the block's first column.  (Four DPP
        // rows hold the same sixteen rows: same values to the same addresses.)  Behind a partial last block row D - 16 k of the tile is the
        // right-hand side: that lane's stores go to the 16 doubles nobody reads; the rows below it are nobody's.
        // (no branch: an EXEC write between the DPP operations of the chain needs wait states the compiler does not see)
        double* const w = r16 != D - CH_NB * k ? s_L + tix(k, k) * CH2_TS + o_row : s_mem + 512;
        Ch2Stream<0>::run(x, Lk, f0, s_colp, &F.prog, CH_NB * k, r16, &F.abort_, w);
        ch2_set(&F.inv, k + 1);
    };
    // tiles (ti[t], k) <- tile W_k, N at once, in place: 4 MFMAs each on accumulators of their own (lane l: A[l & 15][(l >> 4) + 4 q] = the
    // tile's row, B[(l >> 4) + 4 q][l & 15] = W_k's row; result rows (l >> 4) + 4 q, column l & 15)
    auto solve_tiles = [&](auto nconst, const int (&ti)[3], int k) {
        constexpr int N = decltype(nconst)::value;
        const double* pw = s_L + ch2_tix(k, k) * CH2_TS + o_c;
        double bw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bw[q] = pw[4 * CH2_RS * q];
        double a[N][4];
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const double* pa = s_L + ch2_tix(ti[t], k) * CH2_TS + o_op;
#pragma unroll
            for (int q = 0; q < 4; ++q) a[t][q] = pa[4 * q];
        }
        f64x4 acc[N];
#pragma unroll
        for (int t = 0; t < N; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < N; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][q], bw[q], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < N; ++t) {
            double* pc = s_L + ch2_tix(ti[t], k) * CH2_TS + o_c;
#pragma unroll
            for (int q = 0; q < 4; ++q) pc[4 * CH2_RS * q] = acc[t][q];
        }
    };

    // Backward substitution L^T x = y on ONE wave (wave 0), everything between two blocks in registers: no word, no hand-off.
    // Lane (c = r16, g = kq): the residual of block p lives in DPP row p & 3 (register rr[p >> 2], lane c = entry c); x_q, once known,
    // is copied into all four DPP rows.  Step q (from the bottom): x_q = W_q r_q (16 DPP multiply-adds, four sums; row c of W_q per
    // lane; valid in DPP row q & 3) -> all rows (one ds_bpermute pair) -> r_p -= L(q, p)^T x_q for the blocks p < q, four at a time (DPP
    // row g takes block 4 j + g: column c of its tile per lane, x_q by row broadcast), the four that contain block q-1 first.  Rows of a
    // tile beyond the matrix (the right-hand-side row, what lies below it) meet x entries that are forced to zero.
    // Measured at D = 144 (tools/chol_bench, whole kernel): three waves with a row per lane, x_p through LDS and a word per block 36.2 us;
    // this 34.2 us; the same on the matrix cores (vectors in the MFMA's k layout, 4 loads + 4 MFMAs per tile, no lane-to-lane movement
    // at all) 38.1 us -- a v_mfma_f64_16x16x4 holds the SIMD's double-precision pipe for 64 clocks whether or not its accumulator is
    // free, 216 of them are 14 k clocks; without any backward substitution 29.6 us.
    auto backsub1 = [&]() {
        ch2_wait_set(F.rowdone, 1u << iD, nblk, &F.abort_);
        for (int i = 1; i < T; ++i) if (i != iD) ch2_wait_ge(&F.rowdone[i], i, &F.abort_);
        ch2_wait_ge(&F.inv, nblk, &F.abort_);
        CH2_STAMP(61)
#ifdef CH2_DEBUG
        for (int i = lane; i < T * (T + 1) / 2 * CH2_TS; i += 64) B.W[i] = s_L[i];      // the augmented factor (W_k on the diagonal), for tools/chol_bench
#endif
        double* const s_xb = s_col;                             // x, for the coalesced store at the end
        const int g = kq, c = r16;
        double rr[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {                           // y: row D of the augmented factor
            const int col = 16 * (4 * j + g) + c, cc = min(col, D - 1);
            const bool in_g = !(CH2_EXP & 16) && spill && (cc >> 4) <= Rs - 2;      // (spill: iD == Rs; the tiles left of (Rs, Rs-1) are in global memory)
            const double vl = s_L[tix(iD, in_g ? Rs - 1 : cc >> 4) * CH2_TS + CH2_RS * rD + (cc & 15)];
            const double vg = in_g ? g_ld(Ag + (size_t)ch2_tix(iD, cc >> 4) * CH2_TS + CH2_RS * rD + (cc & 15)) : 0.0;
            const double v = in_g ? vg : vl;
            rr[j] = col < D ? v : 0.0;
        }
        auto load_w = [&](int q, double (&w)[CH_NB]) {          // row c of W_q (the last block may be partial: its row nb is the right-hand side, the lane's result is dropped)
            const double* t = s_L + tix(q, q) * CH2_TS + CH2_RS * c;
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) w[j] = t[j];
        };
        auto load_t = [&](int q, int j, double (&t)[CH_NB]) {   // column c of tile (q, 4 j + g), clamped to the tiles left of the diagonal (the result of a clamped lane lands in a residual nobody reads again)
            const int jj = min(4 * j + g, max(q - 1, 0));
            if (q == Rs && !(CH2_EXP & 16)) {                   // spill: tiles (Rs, jj <= Rs - 2) are wave 5's, in global memory (a lane group per tile: both sources are read, one is taken)
                const bool in_g = jj <= Rs - 2;
                const double* pl = s_L + tix(q, Rs - 1) * CH2_TS + c;
                const double* pg = Ag + (size_t)ch2_tix(q, in_g ? jj : 0) * CH2_TS + c;
#pragma unroll
                for (int m = 0; m < CH_NB; ++m) { const double vl = pl[CH2_RS * m], vg = g_ld(pg + CH2_RS * m); t[m] = in_g ? vg : vl; }
                return;
            }
            const double* pt = s_L + ch2_tix(q, jj) * CH2_TS + c;
#pragma unroll
            for (int m = 0; m < CH_NB; ++m) t[m] = pt[CH2_RS * m];
        };
        auto dot_bcast = [&](const double (&v)[CH_NB], double src) -> double {      // sum_m v[m] * src[lane m of the DPP row]
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            ch_exec_settle(src);
            ch_fnma_bcast<0>(s0, src, v[0]); ch_fnma_bcast<1>(s1, src, v[1]); ch_fnma_bcast<2>(s2, src, v[2]); ch_fnma_bcast<3>(s3, src, v[3]);
            ch_fnma_bcast<4>(s0, src, v[4]); ch_fnma_bcast<5>(s1, src, v[5]); ch_fnma_bcast<6>(s2, src, v[6]); ch_fnma_bcast<7>(s3, src, v[7]);
            ch_fnma_bcast<8>(s0, src, v[8]); ch_fnma_bcast<9>(s1, src, v[9]); ch_fnma_bcast<10>(s2, src, v[10]); ch_fnma_bcast<11>(s3, src, v[11]);
            ch_fnma_bcast<12>(s0, src, v[12]); ch_fnma_bcast<13>(s1, src, v[13]); ch_fnma_bcast<14>(s2, src, v[14]); ch_fnma_bcast<15>(s3, src, v[15]);
            return -((s0 + s1) + (s2 + s3));
        };
        double w[CH_NB], t[CH_NB];
        load_w(nblk - 1, w);
        if (nblk >= 2) load_t(nblk - 1, (nblk - 2) >> 2, t);
        auto step = [&](int q, double& rq) {
            const int nb = min(CH_NB, D - CH_NB * q);
            double xs = dot_bcast(w, rq);                       // valid in DPP row q & 3
            xs = c < nb ? xs : 0.0;
            const double xq = __shfl(xs, 16 * (q & 3) + c, 64);
            s_xb[CH_NB * q + c] = xq;                           // (four lanes, one value)
            if (q == 0) return;
            const int jc = (q - 1) >> 2;
            // the round with block q-1: its operands were requested a step ago
            const double d = dot_bcast(t, xq);
            if (jc == 2) rr[2] -= d; else if (jc == 1) rr[1] -= d; else rr[0] -= d;
            // the next step's operands, then the other rounds of this one
            load_w(q - 1, w);
            double t2[CH_NB];
            for (int j = jc - 1; j >= 0; --j) {
                load_t(q, j, t2);
                const double d2 = dot_bcast(t2, xq);
                if (j == 1) rr[1] -= d2; else rr[0] -= d2;
            }
            if (q >= 2) load_t(q - 1, (q - 2) >> 2, t);
        };
        for (int q = nblk - 1; q >= 8; --q) step(q, rr[2]);
        for (int q = min(nblk - 1, 7); q >= 4; --q) step(q, rr[1]);
        for (int q = min(nblk - 1, 3); q >= 0; --q) step(q, rr[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < D; i += 64) { const double v = s_xb[i]; if (PUB) ch2_pub_d(x_out + i, v); else x_out[i] = v; }
        if (spill) ch2_set(&F.bsdone, 1);                       // (the spilled row block has been read: wave 5 clears it for the next step's atomics)
    };

#ifdef P2_STAMPS
    const long long ts0_ = wall_clock64(); long long ts1_ = 0;
#endif
    if (wave == 0) {
        // ================= P: the chain of diagonal blocks =====================================================================
        // (check_finished: the caller has NOT looked at ctl->finished -- that would be a trip to L2 in front of everything; the flag comes with the head, and a
        // finished problem's workgroup leaves here, before it has stored anything, while the other waves leave behind their loads)
        if (check_finished && h_i(offsetof(BaCtl, finished))) return;
        __builtin_amdgcn_s_setprio(3);
        bool ok = true;
        double lambda = 0.0;
        for (int k = 0; k < nblk; ++k) {
            const int j0 = CH_NB * k, nb = min(CH_NB, D - j0);
            const bool mine = r16 < nb;
            double* const tkk = s_L + tix(k, k) * CH2_TS + o_row;
            double a[CH_NB];
            if (k == 0) {
                // block 0 straight from global memory into registers (row r16 per lane; H_pp + lambda I join here): the factorisation starts
                // while the other waves are still bringing the tiles into LDS
                const int jb = r16 / 6, ab = r16 - 6 * jb;
                const double* grow = A + o_row;
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) {
                    const int b = c - 6 * jb;
                    const bool inb = mine && b >= 0 && b <= ab;
                    const double h = inb ? Hpp[36 * jb + 6 * ab + b] : 0.0;
                    a[c] = grow[c] + h;
                }
                lambda = CH2_LAMBDA();                           // (the head's first use on this wave: behind the row's loads in program order)
                a[r16] += lambda;                                // (r16 is a lane value: the compiler selects per register)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // S is cleared behind the loads: these count as one of them
                if (lane == 0) ch2_inc(&F.dma);
            } else {
#pragma unroll
                for (int c = 0; c < CH_NB; ++c) a[c] = tkk[c];          // written by this wave at the end of the previous stage
            }
            CH2_STAMP(3 * k + 1)
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) a[c] = (mine && c <= r16) ? a[c] : (c == r16 ? 1.0 : 0.0);
            double* const s_colp = s_col + (k & 1) * 256;
            ch_exec_settle(a[0]);
#if CH2_EXP & 8
            for (int c = 0; c < CH_NB; ++c) s_colp[c * CH_NB + r16] = 1.0;
            ch2_set(&F.prog, CH_NB * k + CH_NB);
#else
            Ch2Col<0>::run(a, s_colp, &F.prog, CH_NB * k, r16);
#endif
            CH2_STAMP(3 * k + 2)
            const double myinv = s_colp[r16 * CH_NB + r16];
            ok = ok && ch_pivots_ok(myinv);
            if (k == 0) ch2_wait_ge(&F.init, CH2_NC, &F.abort_);        // the DMA of the tiles has landed: block 0's rows may be overwritten
            if (k + 1 < T) {
                // ---- the rest of the critical loop stays on this wave.  What a wave can issue is ~1 instruction per 10 clocks here
                // whatever the instruction (tools/lds_probe: 3 tiles x 1 panel = 75 instructions = 1200 clocks; a 136-instruction DPP
                // solve of 16 rows 1400), so the loop is kept SHORT rather than clever: tile (k+1, k) is solved as a product with
                // W_k = L_kk^-T (wave 1 inverts the block while it is being factored) -- computed TRANSPOSED, X^T = W_k^T A^T, so that the
                // result registers are at once the MFMA operands of tile (k+1, k+1) -= X X^T (lane (kq, r): X[r][kq + 4 q]) -- 4 + 4
                // MFMAs, 12 LDS reads, 8 writes, then the 16 reads that bring the next block's rows into the row-per-lane form.  The
                // owner of row block k+1 has both tiles ready through panel k-1 (rdy) -- normally long before.
                CH2_STAMP(3 * k + 3)
#if !(CH2_EXP & 4)
                ch2_wait_ge(&F.rdy[k + 1], k, &F.abort_);
#endif
                CH2_STAMP(80 + (k & 7))
                double* const px = s_L + tix(k + 1, k) * CH2_TS + o_op;
                double* const pd = s_L + tix(k + 1, k + 1) * CH2_TS + o_c;
                double at[4], cc[4], w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { at[q] = px[4 * q]; cc[q] = pd[4 * CH2_RS * q]; }
#if !(CH2_EXP & 2)
                ch2_wait_ge(&F.inv, k + 1, &F.abort_);
#endif
                CH2_STAMP(3 * k + 4 >= 3 * nblk ? 99 : 90 + (k & 7))
                const double* pw = s_L + tix(k, k) * CH2_TS + o_c;
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = pw[4 * CH2_RS * q];
                // X^T[r][c] = sum_m W^T[r][m] A^T[m][c]: "A" operand W^T[r][kq + 4 q] = W[kq + 4 q][r] (the result pattern of the tile that holds W),
                // "B" operand A^T[kq + 4 q][c] = A[c][kq + 4 q] (the operand pattern of tile (k+1, k)); lane (kq, c) gets X^T[kq + 4 q][c] = X[c][kq + 4 q]
                f64x4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = x0;
                x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[0], at[0], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[1], at[1], x1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[2], at[2], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[3], at[3], x1, 0, 0, 0);
                double xo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { xo[q] = x0[q] + x1[q]; px[4 * q] = xo[q]; }      // L(k+1, k)
                if (k + 1 < nblk) {
                    f64x4 s0 = {0.0, 0.0, 0.0, 0.0}, s1 = s0;
                    s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xo[0], xo[0], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xo[1], xo[1], s1, 0, 0, 0);
                    s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xo[2], xo[2], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xo[3], xo[3], s1, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) pd[4 * CH2_RS * q] = cc[q] - (s0[q] + s1[q]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ch2_set(&F.rowdone[k + 1], k + 1);
            }
            if (nb < CH_NB) {
                // a partial last block: the right-hand side is row nb of this very tile; its forward substitution y_c = (b_c - sum_{m<c} L[c][m] y_m)
                // / L[c][c] runs here, in registers (lane c: b_c and row c of the block, y_m by DPP broadcast: ChFwd of the first generation)
                double* prhs = s_L + tix(k, k) * CH2_TS + CH2_RS * nb;
                double Lr[CH_NB];
#pragma unroll
                for (int m = 0; m < CH_NB; ++m) Lr[m] = (mine && m < r16) ? a[m] : 0.0;
                double bc = prhs[min(r16, nb - 1)], yf = 0.0;
                bc = mine ? bc : 0.0;
                const double inv = mine ? myinv : 0.0;
                ch_exec_settle(bc);
                ChFwd<0>::run(Lr, bc, inv, yf, r16);
                if (lane < nb) prhs[lane] = yf;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ch2_set(&F.rowdone[iD], nblk);
            }
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef P2_STAMPS
        ts1_ = wall_clock64();
#endif
        // ================= backward substitution: this wave takes rows 0 .. 63 ==================================================
        CH2_STAMP(60)
#if !(CH2_EXP & 1)
        backsub1();
#endif
        CH2_STAMP(62)
        ok = ok && ch2_peek(&F.abort_) == 0;
        if (PUB) {
            if (lane == 0) { F.ok = ok ? 1 : 0; take_over(lambda); ch2_pub_d(B.scal + 3, ok ? 1.0 : 0.0); ch2_pub_d(B.dl + D + 1, ok ? 1.0 : 0.0); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) ch2_pub_i(&ctl_->chol_seq, h_i(offsetof(BaCtl, steps)) + 1);
#ifdef P2_STAMPS
            if (lane == 0 && ctl_->it == 4) printf("[solver D %d] abs: start %lld factor done %lld published %lld\n", D, ts0_ % 1000000000ll, ts1_ % 1000000000ll, (long long)wall_clock64() % 1000000000ll);
#endif
        } else if (lane == 0) { F.ok = ok ? 1 : 0; take_over(lambda); B.scal[3] = ok ? 1.0 : 0.0; }
    } else {
        // ================= everyone else: bring the system into LDS ==============================================================
        const int nrb = spill ? nblk - 1 : nblk;                // row blocks that come into LDS whole
        const int ndbl = nrb * (nrb + 1) / 2 * CH2_TS;          // the tiles of S (rows < D), in the order and layout of LDS
        const int nsl = spill ? 2 * CH2_TS : 0;                 // spill: tiles (Rs, Rs-1) and (Rs, Rs) -- neighbours in global memory -- go into the two slots behind them
        {
            typedef __attribute__((address_space(3))) void lds_void;
            typedef __attribute__((address_space(1))) const void glb_void;
            const int npiece = (ndbl + 127) >> 7;               // 128 doubles = 1 KiB per piece (272 is a multiple of 16: the last piece ends on a lane pair)
            for (int pc = wave - 1; pc < npiece; pc += CH2_NC)
                if (pc * 128 + 2 * lane < ndbl)
                    __builtin_amdgcn_global_load_lds((glb_void*)(A + (size_t)pc * 128 + 2 * lane), (lds_void*)(s_L + (size_t)pc * 128), 16, 0, 0);
            const double* const Asl = A + (size_t)ch2_tix(nblk - 1, nblk - 2) * CH2_TS;
            for (int pc = wave - 1; pc * 128 < nsl; pc += CH2_NC)
                if (pc * 128 + 2 * lane < nsl)
                    __builtin_amdgcn_global_load_lds((glb_void*)(Asl + (size_t)pc * 128 + 2 * lane), (lds_void*)(s_L + (size_t)ndbl + (size_t)pc * 128), 16, 0, 0);
        }
        const int ct = tid - 64;                                // 0 .. 64 * CH2_NC - 1
        const double rhs_v = ct < D ? B.bs[ct] + B.bp[ct] : 0.0;      // D <= 174 < 448
        if (T > nblk) {                                         // D is a multiple of 16: the right-hand side is a row block of its own, which no DMA fills
            double* z = s_L + ch2_tix(T - 1, 0) * CH2_TS;
            for (int i = ct; i < T * CH2_TS; i += 64 * CH2_NC) z[i] = 0.0;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (check_finished && h_i(offsetof(BaCtl, finished))) return;
        if (lane == 0) ch2_inc(&F.dma);
        ch2_wait_ge(&F.dma, CH2_NC + 1, &F.abort_);
        const int nb0 = min(CH_NB, D);
        const double lambda = CH2_LAMBDA();
        for (int i = ct; i < 36 * (D / 6); i += 64 * CH2_NC) {  // the 6x6 diagonal blocks of H_pp (lower halves) and lambda on the diagonal; rows of block 0 are wave 0's
            const int j = i / 36, a = (i % 36) / 6, b = i % 6, row = 6 * j + a;
            if (b <= a && row >= nb0) s_L[sidx(row, 6 * j + b)] += Hpp[i] + (a == b ? lambda : 0.0);      // (spill: a pose's 6x6 block in row block Rs lies in tile column >= Rs - 1: the slots)
        }
        if (ct < D) {
            if (spill && (ct >> 4) <= Rs - 2) Ag[(size_t)ch2_tix(Rs, ct >> 4) * CH2_TS + CH2_RS * rD + (ct & 15)] = rhs_v;      // (the right-hand side's entries under the spilled tiles: wave 5 reads them past the cache, behind `init`)
            else s_L[sidx(D, ct)] = rhs_v;
        }
        if (ct == 0) s_L[sidx(D, D)] = 0.0;
        if (clear_after_load && ct < D) B.bs[ct] = 0.0;         // b_s and (below) S are the targets of the next step's Schur atomics (the kernel boundary publishes the stores)
        if (spill) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) ch2_inc(&F.init);
        CH2_STAMP(126)
        if (clear_after_load && wave == 4) {                    // every copy of S has left global memory (dma): wave 4, which has no other work, clears it
            double2* g = reinterpret_cast<double2*>(const_cast<double*>(A));
            for (int i = lane; 2 * i < ndbl; i += 64) g[i] = make_double2(0.0, 0.0);
            if (spill) { double2* gs = reinterpret_cast<double2*>(Ag + (size_t)ch2_tix(Rs, Rs - 1) * CH2_TS); for (int i = lane; 2 * i < nsl; i += 64) gs[i] = make_double2(0.0, 0.0); }
        }
        if (spill && wave == 5) {
            // ================= the spilled row block (see "SPILL" above): R = 11, tiles (R, 0 .. 9) in registers, (R, 10) and (R, 11) in the slots =====================
            constexpr int R = CH2_SPILL_T - 1, NK = R - 1;      // NK tiles are kept in registers
            ch2_wait_ge(&F.init, CH2_NC, &F.abort_);
            double xr[NK][4];                                   // L(R, k) in the operand layout (row r16, columns kq + 4 q) once stage k is through; before: the tile as it came
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const double* t = Ag + (size_t)ch2_tix(R, k) * CH2_TS + o_op;
#pragma unroll
                for (int q = 0; q < 4; ++q) xr[k][q] = g_ld(t + 4 * q);
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                // tile (R, k) -= sum_{p < k} L(R, p) L(k, p)^T, formed as the TRANSPOSED product (A operand: L(k, p), B operand: L(R, p)): the result registers hold
                // [kq + 4 q][r16] of the transpose = [r16][kq + 4 q] of the tile: the operand layout again
                if (k >= 1) {
                    ch2_wait_ge(&F.rowdone[k], k, &F.abort_);   // L(k, 0 .. k-1) are final
                    f64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = a0;
#pragma unroll
                    for (int p = 0; p < k; ++p) {
                        const double* pb = s_L + ch2_tix(k, p) * CH2_TS + o_op;
                        double b[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) b[q] = pb[4 * q];
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[0], xr[p][0], a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[1], xr[p][1], a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[2], xr[p][2], a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[3], xr[p][3], a1, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) xr[k][q] -= a0[q] + a1[q];
                }
                // X = A W_k as X^T = W_k^T A^T (wave 0's form): result = the operand layout of X
                ch2_wait_ge(&F.inv, k + 1, &F.abort_);
                const double* pw = s_L + ch2_tix(k, k) * CH2_TS + o_c;
                double w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = pw[4 * CH2_RS * q];
                f64x4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = x0;
                x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[0], xr[k][0], x0, 0, 0, 0); x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[1], xr[k][1], x1, 0, 0, 0);
                x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[2], xr[k][2], x0, 0, 0, 0); x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[3], xr[k][3], x1, 0, 0, 0);
                double* const tg = Ag + (size_t)ch2_tix(R, k) * CH2_TS + o_op;
#pragma unroll
                for (int q = 0; q < 4; ++q) { xr[k][q] = x0[q] + x1[q]; tg[4 * q] = xr[k][q]; }      // (to global memory for the backward substitution)
            }
            // the two tiles wave 0 takes at stage R-1, through panel R-2: (R, R-1) -= L(R, p) L(R-1, p)^T (transposed form, read and written in the operand
            // layout), (R, R) -= L(R, p) L(R, p)^T (both operands in registers; result layout)
            ch2_wait_ge(&F.rowdone[R - 1], R - 1, &F.abort_);
            {
                double* const p1 = s_L + tix(R, R - 1) * CH2_TS + o_op;
                double* const p2 = s_L + tix(R, R) * CH2_TS + o_c;
                double c1[4], c2[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { c1[q] = p1[4 * q]; c2[q] = p2[4 * CH2_RS * q]; }
                f64x4 u0 = {0.0, 0.0, 0.0, 0.0}, u1 = u0, v0 = u0, v1 = u0;
#pragma unroll
                for (int p = 0; p < NK; ++p) {
                    const double* pb = s_L + ch2_tix(R - 1, p) * CH2_TS + o_op;
                    double b[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[q] = pb[4 * q];
                    u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[0], xr[p][0], u0, 0, 0, 0); v0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[p][0], xr[p][0], v0, 0, 0, 0);
                    u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[1], xr[p][1], u1, 0, 0, 0); v1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[p][1], xr[p][1], v1, 0, 0, 0);
                    u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[2], xr[p][2], u0, 0, 0, 0); v0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[p][2], xr[p][2], v0, 0, 0, 0);
                    u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b[3], xr[p][3], u1, 0, 0, 0); v1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[p][3], xr[p][3], v1, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { p1[4 * q] = c1[q] - (u0[q] + u1[q]); p2[4 * CH2_RS * q] = c2[q] - (v0[q] + v1[q]); }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the tiles in global memory too: the backward substitution reads them behind rowdone[R])
            if (lane == 0) *(volatile ch2_lds_int*)&F.rdy[R] = R - 1;
            // ... and once the backward substitution has read the row block, its tiles in global memory are cleared for the next step's atomics
            if (clear_after_load) {
                ch2_wait_ge(&F.bsdone, 1, &F.abort_);
                double2* gs = reinterpret_cast<double2*>(Ag + (size_t)ch2_tix(R, 0) * CH2_TS);
                for (int i = lane; 2 * i < NK * CH2_TS; i += 64) gs[i] = make_double2(0.0, 0.0);
            }
        }

        if (wave == 1) {
            // ================= the inverter: W_k = L_kk^-T, one column step behind the factorisation ================================
            ch2_wait_ge(&F.init, CH2_NC, &F.abort_);            // W_0 goes where the DMA wrote tile (0, 0)
            for (int k = 0; k < nblk; ++k) { invert_block(k); CH2_STAMP(k) }
        } else if (wave != 4 && wave != 5) {
            // ================= solvers ==========================================================================================
            // Solver g owns the row blocks i = g + 1, g + 1 + CH2_NS, ...: it alone writes their tiles below block row 0 -- except tile
            // (i, i-1), which wave 0 solves, and the last panel of tile (i, i), which wave 0 applies.  The schedule is fixed (no
            // bookkeeping): LEFT-LOOKING, tile (i, j) gets all its panels 0 .. j-1 in one read-modify-write at stage j, right before it is
            // solved against block j (X = A W_j).  The two tiles that wave 0 takes from the owner of row block k+1 at stage k -- (k+1, k) and
            // (k+1, k+1) -- are the exception: their panels 0 .. k-2 go in one stage ahead (D), and panel k-1, which needs wave 0's L(k, k-1),
            // is ONE fused step whose own operands are in registers before the wait: it sits between the end of wave 0's stage k-1 and
            // the end of its block k.  Foreign inputs: W_k (inv) and L(j, .) of other solvers' / wave 0's row block j (rowdone[j] = stages
            // solved for row block j).
            // waves 2, 3, 6, 7 -- or 2, 3, 7, 6 in the twelve-row-block (spilled) form: the owners' work grows with the square of the row block's number (solver 1 of 4 has row
            // blocks 2, 6, 10), and with 6 and 7 swapped the two SIMDs the solvers live on issue 107 : 113 of the tile products instead of 95 : 125.  At that size the
            // matrix pipe of the busier SIMD is what the last stages wait for (tools/chol_bench: D = 168 ... 186 -1.0 ... -1.4 us; below, where it is not, consecutive
            // row blocks on different SIMDs are worth more: D = 132 ... 156 +0.4 ... +0.6 us with the swap -- and with eleven row blocks the bench lost 1 % at 300 steps)
            const int g = wave <= 3 ? wave - 2 : (spill ? 9 - wave : wave - 4);
            auto first_row = [&](int lo) { return lo + ((g + 1 - lo) % CH2_NS + CH2_NS) % CH2_NS; };      // my first row block >= lo
            auto mark_rows = [&](int k, int i0, int n) {        // row blocks i0, i0 + CH2_NS, .. (n of them) are solved through stage k
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < n) *(volatile ch2_lds_int*)&F.rowdone[i0 + CH2_NS * lane] = k + 1;
            };
            auto rows_update = [&](int i0, int j, int p0, int p1) {      // tiles (i0, j), (i0 + CH2_NS, j), ..: panels p0 .. p1
                if (i0 >= Ts || p1 < p0) return;
                const int ti[3] = {i0, i0 + CH2_NS, i0 + 2 * CH2_NS}, tj[3] = {j, j, j};      // T <= 11: at most three of my row blocks
                if (i0 + 2 * CH2_NS < Ts) ch2_tiles_mp<3>(s_L, ti, tj, p0, p1, o_op, o_c);
                else if (i0 + CH2_NS < Ts) ch2_tiles_mp<2>(s_L, ti, tj, p0, p1, o_op, o_c);
                else ch2_tiles_mp<1>(s_L, ti, tj, p0, p1, o_op, o_c);
            };
            ch2_wait_ge(&F.init, CH2_NC, &F.abort_);
            if (g == 0) ch2_set(&F.rdy[1], 0);                  // tiles (1, 0) and (1, 1) have no panel to wait for
            for (int k = 0; k < nblk; ++k) {
                if (k + 1 >= Ts) break;                          // nothing below the last block (its right-hand-side row is wave 0's)
                const bool own1 = k % CH2_NS == g;               // row block k+1 is mine: its tiles (k+1, k) and (k+1, k+1) are finished (end of my stage k-1) and wave 0's
                CH2_STAMP(6 * k)
                // (A) block column k of my rows, up to date
                if (k >= 1) {
                    ch2_wait_ge(&F.rowdone[k], k, &F.abort_);
                    rows_update(own1 ? k + 1 + CH2_NS : first_row(k + 1), k, 0, k - 1);
                }
                CH2_STAMP(6 * k + 1)
                // (D) one stage ahead, while block k is still being factored: the two tiles wave 0 will take from me at stage k+1, through panel k-1
                const int i0 = first_row(k + 2);
                const bool crit = i0 == k + 2 && i0 < Ts;        // row block k+2 is mine: the chain wave 0 -> W_k -> L(k+2, k) -> tiles (k+2, k+1), (k+2, k+2) -> wave 0 runs through me
                if (crit && k >= 1) {
                    ch2_wait_ge(&F.rowdone[k + 1], k, &F.abort_);
                    const int ti[3] = {k + 2, k + 2, 0}, tj[3] = {k + 1, k + 2, 0};
                    ch2_tiles_mp<2>(s_L, ti, tj, 0, k - 1, o_op, o_c);
                }
                if (i0 >= Ts) continue;
                int i1 = i0;                                     // first row block of the ordinary solve below
                if (crit) {
                    // The critical tile first and alone, as the TRANSPOSED product X^T = W_k^T A^T: the result registers are then the operand
                    // registers of the update that follows (row r16, columns kq + 4 q) -- no trip through LDS between the two.  Then panel k
                    // of the two tiles wave 0 is waiting for (everything of mine is in registers before the waits), then the rest.
                    double* const px = s_L + ch2_tix(k + 2, k) * CH2_TS + o_op;
                    double* const pc1 = s_L + ch2_tix(k + 2, k + 1) * CH2_TS + o_c;
                    double* const pc2 = s_L + ch2_tix(k + 2, k + 2) * CH2_TS + o_c;
                    double at[4], w[4], b[4], c1[4], c2[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { at[q] = px[4 * q]; c1[q] = pc1[4 * CH2_RS * q]; c2[q] = pc2[4 * CH2_RS * q]; }
                    ch2_wait_ge(&F.inv, k + 1, &F.abort_);
                    CH2_STAMP(6 * k + 2)
                    __builtin_amdgcn_s_setprio(2);
                    const double* pw = s_L + ch2_tix(k, k) * CH2_TS + o_c;
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[q] = pw[4 * CH2_RS * q];
                    f64x4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = x0;
                    x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[0], at[0], x0, 0, 0, 0); x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[1], at[1], x1, 0, 0, 0);
                    x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[2], at[2], x0, 0, 0, 0); x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[3], at[3], x1, 0, 0, 0);
                    double x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { x[q] = x0[q] + x1[q]; px[4 * q] = x[q]; }
                    ch2_wait_ge(&F.rowdone[k + 1], k + 1, &F.abort_);      // L(k+1, k): wave 0, right behind W_k
                    const double* pb = s_L + ch2_tix(k + 1, k) * CH2_TS + o_op;
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[q] = pb[4 * q];
                    f64x4 u0 = {0.0, 0.0, 0.0, 0.0}, u1 = u0, v0 = u0, v1 = u0;
                    u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], b[0], u0, 0, 0, 0); v0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], x[0], v0, 0, 0, 0);
                    u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], b[1], u1, 0, 0, 0); v1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], x[1], v1, 0, 0, 0);
                    u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], b[2], u0, 0, 0, 0); v0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], x[2], v0, 0, 0, 0);
                    u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], b[3], u1, 0, 0, 0); v1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], x[3], v1, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { pc1[4 * CH2_RS * q] = c1[q] - (u0[q] + u1[q]); pc2[4 * CH2_RS * q] = c2[q] - (v0[q] + v1[q]); }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) { *(volatile ch2_lds_int*)&F.rdy[k + 2] = k + 1; *(volatile ch2_lds_int*)&F.rowdone[k + 2] = k + 1; }
                    __builtin_amdgcn_s_setprio(0);
                    i1 = i0 + CH2_NS;
                } else {
                    // (B) W_k (wave 1's)
                    ch2_wait_ge(&F.inv, k + 1, &F.abort_);
                    CH2_STAMP(6 * k + 2)
                }
                // (C) my other tiles of block column k: X = A W_k
                if (i1 < Ts) {
                    const int ti[3] = {i1, i1 + CH2_NS, i1 + 2 * CH2_NS};
                    const int n = (Ts - 1 - i1) / CH2_NS + 1;
                    if (n == 3) solve_tiles(std::integral_constant<int, 3>{}, ti, k);
                    else if (n == 2) solve_tiles(std::integral_constant<int, 2>{}, ti, k);
                    else solve_tiles(std::integral_constant<int, 1>{}, ti, k);
                    mark_rows(k, i1, n);
                }
                CH2_STAMP(6 * k + 3)
            }
        }
    }
}
