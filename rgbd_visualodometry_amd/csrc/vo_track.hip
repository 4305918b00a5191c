// vo_track.hip -- per-frame tracking chain on gfx950:
//
//   k_frustum       frustum / view-angle filter (reference src/frame.cpp:70-91, loop src/frontend.cpp:171-184): one lane per active
//                   map point, candidates appended with ONE atomic per wavefront; the candidates' descriptors are copied into a
//                   slab in candidate order (the lane's match-record buffer, free until k_match_emit)
//   k_match         exact brute-force Hamming 1-NN of every candidate against the frame's descriptors (replaces
//                   cv::FlannBasedMatcher + LSH, src/frontend.cpp:33,:187).  One workgroup = 64 candidates x a quarter of the
//                   keypoints (gridDim.y slices): 8 wavefronts, each with a 64-descriptor tile in LDS per round (wave-wide
//                   broadcast reads), 8 chained v_bcnt per pair; the partial minima meet in LDS and ONE atomicMin per candidate
//                   and workgroup updates best[q] = (dist << 22) | keypoint (first minimum wins)
//   k_match_mfma    the same search on the matrix cores for larger frames (>= 8 Mi map-point x feature pairs): descriptor bits as +-1 bytes,
//                   v_mfma_i32_16x16x64_i8 leaves 2 x Hamming distance; same packed key, bit-identical results
//   k_match_gate    min distance, gate max(min * ratio, 30) (src/frontend.cpp:190-211), ORDER-PRESERVING compaction
//   k_match_emit    match records and the float32 3-D / 2-D pairs (:225-230) by output position
//   k_ransac_hyp    one lane per hypothesis: counter-based 4-sample, Grunert P3P, 4th point disambiguates
//   k_ransac_score  1024-thread workgroups keep up to 4096 correspondences in registers and stream a tile of hypotheses (scalar loads)
//                   past them: reprojection test, __ballot + popcount; beyond 256 hypotheses in two stages around k_ransac_peek
//   k_ransac_peek   after the first 128 hypotheses: how far the adaptive stop lets the sequential scan go (the rest is not scored)
//   k_ransac_select adaptive-stop scan (solvePnPRansac semantics, src/frontend.cpp:238-241) replayed over the records (running maxima,
//                   found in parallel), inlier list of the winner (order-preserving)
//   k_pose_lm       whole 2 x 10-iteration Levenberg-Marquardt with Huber kernel, one workgroup per lane -- or several for large
//                   inlier sets (replicated 6x6 solve, partial sums exchanged through write-through stores)
//                   (g2o semantics, src/frontend.cpp:257-329; Jacobian include/myslam/g2o_types.h:86-100)
//
// The chain never returns to the host between stages: counts live in TrackDev, grids are sized
// by capacity and trimmed on the device.  All double arithmetic mirrors oracle/o_track.cpp
// operation by operation (file is compiled with -ffp-contract=off).
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstring>

#include <type_traits>
#include "vo_internal.h"
#include "vo_reduce.h"

#ifdef VO_LM_STAMPS
__device__ long long g_dbg[8];
__device__ long long g_dbg2[8];
#endif

// every kernel of the chain: blockIdx.z = lane; the lane's pointers come from its descriptor (uniform loads)
#define LANE_PTRS(lanes) \
    const LaneDesc& ld_ = lanes[blockIdx.z]; \
    TrackDev* tr = ld_.tr; uint32_t* best = ld_.best; int32_t* cand = ld_.cand; vo_match* matches = ld_.matches; \
    float* cxyz = ld_.cxyz; float* cuv = ld_.cuv; double* hyp_pose = ld_.hyp_pose; int* hyp_cnt = ld_.hyp_cnt; \
    int32_t* inliers = ld_.inliers; uint8_t* mask = ld_.mask; const uint32_t* fdesc = ld_.fdesc; const int* nkp_p = ld_.nkp; \
    const vo_keypoint* kps = ld_.kps; const CamD cam = ld_.cam; \
    (void)tr; (void)best; (void)cand; (void)matches; (void)cxyz; (void)cuv; (void)hyp_pose; (void)hyp_cnt; (void)inliers; (void)mask; (void)fdesc; (void)nkp_p; (void)kps; (void)cam;

// ------------------------------------------------------------------------------------------
// small double-precision helpers (same operation order as oracle/o_math.h)
// ------------------------------------------------------------------------------------------
struct D3 { double x, y, z; };
__device__ __forceinline__ D3 mk(double a, double b, double c) { D3 r; r.x = a; r.y = b; r.z = c; return r; }
__device__ __forceinline__ D3 sub(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 scl(double s, D3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross3(D3 a, D3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ D3 nrm3(D3 a) { double n = sqrt(dot3(a, a)); return mk(a.x / n, a.y / n, a.z / n); }
__device__ __forceinline__ D3 rot(const double* R, D3 v) {
    return mk(R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z, R[6] * v.x + R[7] * v.y + R[8] * v.z);
}
__device__ __forceinline__ D3 xform(const double* T, D3 p) { D3 r = rot(T, p); return mk(r.x + T[9], r.y + T[10], r.z + T[11]); }
__device__ __forceinline__ double comp(D3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// ------------------------------------------------------------------------------------------
// K10 + K11
// ------------------------------------------------------------------------------------------
#define MQ 64
#define MT 64
#define MATCH_NONE 0xFFFFFFFFu      // best[q] = (distance << 22) | keypoint index: one 32-bit atomicMin keeps the first minimum
#define MATCH_GRID_X 512            // workgroups (candidate tiles) per lane; more candidates are taken in a stride loop
// visibility filter: every active map point gets best[q] = ~0; visible ones are appended to the
// (unordered) candidate list -- order is restored later because results are indexed by q.
__global__ __launch_bounds__(256) void k_frustum(const LaneDesc* __restrict__ lanes) {
    LANE_PTRS(lanes)
    const double* __restrict__ map_pos = ld_.map_pos; const double* __restrict__ map_nrm = ld_.map_nrm;
    const uint8_t* __restrict__ map_flags = ld_.map_flags; const int32_t* __restrict__ active = ld_.active; const int n_active = ld_.n_active;
    const int q = blockIdx.x * 256 + threadIdx.x;
    bool vis = false;
    if (q < n_active) {
        best[q] = MATCH_NONE;
        const int mi = active[q];
        if (!(map_flags[mi] & VO_MAP_FLAG_OUTLIER)) {
            double T[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = tr->T[i];
            const D3 pw = mk(map_pos[3 * (size_t)mi], map_pos[3 * (size_t)mi + 1], map_pos[3 * (size_t)mi + 2]);
            const D3 pc = xform(T, pw);
            if (pc.z > 0) {
                const double u = cam.fx * pc.x / pc.z + cam.cx, v = cam.fy * pc.y / pc.z + cam.cy;
                if (!(u < 0 || u >= cam.W || v < 0 || v >= cam.H)) {
                    // camera centre C = -R^T t
                    const D3 C = mk(-(T[0] * T[9] + T[3] * T[10] + T[6] * T[11]), -(T[1] * T[9] + T[4] * T[10] + T[7] * T[11]), -(T[2] * T[9] + T[5] * T[10] + T[8] * T[11]));
                    const D3 dir = nrm3(sub(pw, C));
                    const double dd = dir.x * map_nrm[3 * (size_t)mi] + dir.y * map_nrm[3 * (size_t)mi + 1] + dir.z * map_nrm[3 * (size_t)mi + 2];
                    vis = !(dd < 0.8660254037844387);
                }
            }
        }
    }
    // append: ONE atomic per wavefront on the lane's counter (pad0 = live candidate counter, reset by k_match_gate); a
    // per-candidate atomic on that single address serialises in L2 (~10 k of them per frame)
    const unsigned long long m = __ballot(vis);
    if (m) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&tr->pad0, __popcll(m));
        base = __shfl(base, leader, 64);
        if (vis) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            cand[pos] = q;
            // the candidate's descriptor goes into a slab in candidate order (the lane's match-record buffer is free until
            // k_match_emit): k_match then reads 64 consecutive 32-byte rows per workgroup instead of gathering 32-byte rows of the
            // map through active[] -- four times over, once per keypoint slice -- at 64..128 bytes of fabric traffic per row
            if (2 * (long long)n_active <= (long long)ld_.cap) {
                const uint4* src = (const uint4*)(ld_.map_desc + (size_t)active[q] * 8);
                uint4* dst = (uint4*)matches + 2 * (size_t)pos;
                dst[0] = src[0]; dst[1] = src[1];
            }
        }
    }
}

// One workgroup = 64 candidates x a slice of the lane's keypoints: 8 wavefronts, wave w takes keypoint tile w of every
// round of 8 tiles (64 descriptors each, staged in LDS by the whole workgroup, read as wave-wide broadcasts) and keeps the
// running argmin of its tiles in a register; the 8 partial results meet in LDS and the workgroup issues ONE atomicMin
// per candidate.  gridDim.y workgroups share a candidate tile (keypoint rounds y, y + gridDim.y, ...): 4 atomics per
// candidate instead of round 1's one per 64-keypoint tile (32 per candidate, 4.5 MB of L2 atomics per frame).
#define MW 8                        // wavefronts per workgroup = keypoint tiles in flight
__global__ __launch_bounds__(64 * MW) void k_match(const LaneDesc* __restrict__ lanes) {
    LANE_PTRS(lanes)
    const uint32_t* __restrict__ map_desc = ld_.map_desc; const int32_t* __restrict__ active = ld_.active;
    __shared__ uint4 s_train[MW][MT * 2];
    __shared__ uint32_t s_part[MW][MQ];
    const int nkp = *nkp_p, ncand = tr->pad0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool slab = 2 * (long long)ld_.n_active <= (long long)ld_.cap;      // k_frustum left the candidates' descriptors in candidate order
    // the candidate count is only known on the device: a fixed number of workgroups strides over the candidate tiles
    for (int c0 = blockIdx.x * MQ; c0 < ncand; c0 += gridDim.x * MQ) {
        const int ci = min(c0 + lane, ncand - 1);          // clamped: every lane computes, only valid ones store
        const int q = cand[ci];
        const uint4* qd = slab ? (const uint4*)matches + 2 * (size_t)ci : (const uint4*)(map_desc + (size_t)active[q] * 8);
        const uint4 qa = qd[0], qb = qd[1];
        uint32_t bk = MATCH_NONE;                          // (dist << 22) | keypoint
        for (int r0 = blockIdx.y * MW * MT; r0 < nkp; r0 += gridDim.y * MW * MT) {
            const int t0 = r0 + wave * MT;                  // this wave's tile of the round
            const int nt = max(0, min(MT, nkp - t0));
            if (nt > 0) {                                   // wave-uniform
                const uint4* src = (const uint4*)(fdesc + (size_t)t0 * 8);
                if (lane < nt) { s_train[wave][2 * lane] = src[2 * lane]; s_train[wave][2 * lane + 1] = src[2 * lane + 1]; }
            }
            __syncthreads();
            // distance and tile-local index travel as one key (dist << 6 | t): a single v_min keeps the first minimum;
            // the eight popcounts are chained through v_bcnt's accumulate operand (two pairs in flight hide the chain)
            uint32_t tk = 0xFFFFFFFFu;
#pragma unroll 2
            for (int t = 0; t < nt; ++t) {
                const uint4 a = s_train[wave][2 * t], b = s_train[wave][2 * t + 1];
                uint32_t h;
                asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(h) : "v"(qa.x ^ a.x), "v"((uint32_t)t));      // seeds the sum with t
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(qa.y ^ a.y));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(qa.z ^ a.z));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(qa.w ^ a.w));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(qb.x ^ b.x));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(qb.y ^ b.y));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(qb.z ^ b.z));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h) : "v"(qb.w ^ b.w));
                // h = dist + t  ->  key = (dist << 6) | t = ((h - t) << 6) + t = (h << 6) - 63 t
                tk = min(tk, (h << 6) - 63u * (uint32_t)t);
            }
            if (nt > 0) bk = min(bk, ((tk >> 6) << 22) | (uint32_t)(t0 + (int)(tk & 63u)));    // tiles ascend: first minimum wins
            __syncthreads();
        }
        s_part[wave][lane] = bk;
        __syncthreads();
        if (wave == 0) {
            uint32_t m = s_part[0][lane];
#pragma unroll
            for (int w = 1; w < MW; ++w) m = min(m, s_part[w][lane]);
            if (c0 + lane < ncand && m != MATCH_NONE) atomicMin(&best[q], m);
        }
        __syncthreads();
    }
}

// ---- the same search on the matrix cores ------------------------------------------------------------------------------------------
// Hamming distance as a dot product: with every descriptor bit written as a byte (+1 / -1 for a keypoint, -1 / +1 for a candidate) the
// int8 dot product over the 256 bits is  -(256 - 2 h) ; with the accumulator started at 256 the MFMA leaves D = 2 h.  One
// v_mfma_i32_16x16x64_i8 covers 16 candidates x 16 keypoints x 64 bits: four of them per 16 x 16 tile -- 256 pairs per 128 cycles of one
// wave's matrix pipe against 64 pairs per ~90 cycles of its vector ALU in k_match.  The result is the same packed key
// (distance << 22 | keypoint index, unsigned minimum = first minimum), so both kernels are interchangeable bit for bit; this one pays
// off once the expansion of a 64-keypoint tile into bytes (every workgroup redoes it for the tiles it visits) is shared by enough
// candidates: it is used above MMF_MIN_PAIRS candidate-keypoint pairs (host launcher).
// A workgroup = 4 waves x 64 candidates (four 16-row A operand sets per wave, built once) x a slice of the keypoints; keypoint tiles of 64
// are expanded into LDS as sixteen PLANES of 16-byte k-chunks ([chunk 0..15][keypoint 0..63][16 B]): the bank of an operand read then depends on
// the keypoint row only, and each of the four 16-lane groups a ds_read_b128 is served in ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...:
// MI355X_MICROARCH.md, LDS table) holds every row 0..15 once -- conflict-free.  (Round 3's 272-byte rows assumed groups of 16 consecutive
// lanes: one two-way conflict per group, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 41 %.)  The expansion writes a keypoint per lane: the 8
// consecutive lanes of a ds_write_b128 group hit 8 consecutive 16-byte slots.
// Operand maps (checked with exact integer data by the parity tests: tests/test_gpu_parity.py): lane l supplies, for k-step s,
// bits 64 s + 16 (l >> 4) .. + 15 of row / column (l & 15); D: column = l & 15, rows 4 (l >> 4) + 0..3.
#define MMF_NA 4                    // 16-candidate operand sets per wave
#define MMF_CAND (64 * MMF_NA)
#define MMF_MIN_PAIRS (8ll << 20)             // active map points x features per frame from which the launcher takes this kernel: 4.6 k x 500 -> k_match (14.8 vs 20.1 us), 21 k x 2000 -> here (50.6 vs 73.1 us), 41 k x 8000 -> here (235 vs 520 us)
#define MMF_PLANE (MT * 16)             // bytes of one k-chunk plane
typedef int mmf_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mmf_expand4(uint32_t x, uint32_t flip) {       // 4 bits -> 4 bytes: bit set -> +1 (flip = 0xFFFFFFFF) / -1 (flip = 0x01010101)
    const uint32_t y = (x * 0x00204081u) & 0x01010101u;      // bit i -> byte i (no two partial products share a bit position)
    return (y * 0xFEu) ^ flip;                                 // byte 0 -> flip byte, byte 1 -> 0xFE ^ flip byte
}
__device__ __forceinline__ mmf_v4i mmf_expand16(uint32_t bits, uint32_t flip) {
    mmf_v4i v;
    v[0] = (int)mmf_expand4(bits & 15u, flip); v[1] = (int)mmf_expand4((bits >> 4) & 15u, flip);
    v[2] = (int)mmf_expand4((bits >> 8) & 15u, flip); v[3] = (int)mmf_expand4((bits >> 12) & 15u, flip);
    return v;
}
__global__ __launch_bounds__(256) void k_match_mfma(const LaneDesc* __restrict__ lanes) {
    LANE_PTRS(lanes)
    const uint32_t* __restrict__ map_desc = ld_.map_desc; const int32_t* __restrict__ active = ld_.active;
    __shared__ __attribute__((aligned(16))) uint8_t s_kp[16 * MMF_PLANE];
    const int nkp = *nkp_p, ncand = tr->pad0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    const bool slab = 2 * (long long)ld_.n_active <= (long long)ld_.cap;
    for (int c0 = blockIdx.x * MMF_CAND; c0 < ncand; c0 += gridDim.x * MMF_CAND) {
        // ---- A operands: candidates c0 + 16 MMF_NA wave + 16 a + r16; per k-step s one 16-byte operand (bits 64 s + 16 g .. + 15)
        mmf_v4i A[MMF_NA][4];
#pragma unroll
        for (int a = 0; a < MMF_NA; ++a) {
            const int ci = min(c0 + 16 * MMF_NA * wave + 16 * a + r16, ncand - 1);      // clamped: every lane computes, only valid ones store
            const uint32_t* qd = slab ? (const uint32_t*)((const uint4*)matches + 2 * (size_t)ci) : map_desc + (size_t)active[cand[ci]] * 8;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const uint32_t w = qd[2 * s4 + (g >> 1)];
                A[a][s4] = mmf_expand16((w >> (16 * (g & 1))) & 0xFFFFu, 0x01010101u);         // candidate side carries the minus sign
            }
        }
        uint32_t best_k[MMF_NA][4];
        int32_t out_q[MMF_NA][4];                          // where the sixteen results of lane (g, 0) go: requested now, used behind the tiles (the load used to sit between the minimum and its atomic: clock stamps, 6 of a workgroup's 14 us)
#pragma unroll
        for (int a = 0; a < MMF_NA; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                best_k[a][q] = MATCH_NONE;
                const int ci = c0 + 16 * MMF_NA * wave + 16 * a + 4 * g + q;
                out_q[a][q] = (r16 == 0 && ci < ncand) ? cand[ci] : -1;
            }
        // the descriptor words of a tile are requested one tile ahead: the trip to L2 (~1.5 us, eight tiles per workgroup at the bench workload) runs
        // behind the previous tile's MFMAs instead of between two barriers
        const int kp_l = threadIdx.x & 63, qtr_l = threadIdx.x >> 6;
        uint2 w2n = make_uint2(0u, 0u);
        if ((int)(blockIdx.y * MT) < nkp) w2n = *reinterpret_cast<const uint2*>(fdesc + (size_t)min((int)(blockIdx.y * MT) + kp_l, nkp - 1) * 8 + 2 * qtr_l);
        for (int t0 = blockIdx.y * MT; t0 < nkp; t0 += gridDim.y * MT) {
            __syncthreads();                                // the previous tile has been consumed
            {   // expand keypoints t0 .. t0 + 63: lane -> keypoint, wave -> quarter of the descriptor (2 words = 64 bytes = chunks 4 qtr .. 4 qtr + 3)
                const int kp = kp_l, qtr = qtr_l;
                const uint2 w2 = w2n;
                const int tn = t0 + gridDim.y * MT;
                if (tn < nkp) w2n = *reinterpret_cast<const uint2*>(fdesc + (size_t)min(tn + kp, nkp - 1) * 8 + 2 * qtr);
                uint8_t* dst = s_kp + (size_t)(4 * qtr) * MMF_PLANE + kp * 16;
                *reinterpret_cast<mmf_v4i*>(dst) = mmf_expand16(w2.x & 0xFFFFu, 0xFFFFFFFFu); *reinterpret_cast<mmf_v4i*>(dst + MMF_PLANE) = mmf_expand16(w2.x >> 16, 0xFFFFFFFFu);
                *reinterpret_cast<mmf_v4i*>(dst + 2 * MMF_PLANE) = mmf_expand16(w2.y & 0xFFFFu, 0xFFFFFFFFu); *reinterpret_cast<mmf_v4i*>(dst + 3 * MMF_PLANE) = mmf_expand16(w2.y >> 16, 0xFFFFFFFFu);
            }
            __syncthreads();
            // (the test for columns past the end sits in a workgroup-uniform branch: only the last tile of a lane's keypoints pays for it,
            // and no MFMA runs under a divergent EXEC mask)
            auto tile = [&](auto partial_c) {
                constexpr bool PARTIAL = decltype(partial_c)::value;
#pragma unroll
                for (int sub = 0; sub < MT / 16; ++sub) {
                    const uint8_t* row = s_kp + g * MMF_PLANE + (16 * sub + r16) * 16;       // k-step q: chunk g + 4 q
                    const mmf_v4i b0 = *reinterpret_cast<const mmf_v4i*>(row), b1 = *reinterpret_cast<const mmf_v4i*>(row + 4 * MMF_PLANE),
                                  b2 = *reinterpret_cast<const mmf_v4i*>(row + 8 * MMF_PLANE), b3 = *reinterpret_cast<const mmf_v4i*>(row + 12 * MMF_PLANE);
                    const uint32_t kpi = (uint32_t)(t0 + 16 * sub + r16);
                    const bool kvalid = !PARTIAL || (int)kpi < nkp;
                    // the four operand sets' accumulators advance side by side: a dependent MFMA waits for the whole pass of its predecessor
                    mmf_v4i acc[MMF_NA];
#pragma unroll
                    for (int a = 0; a < MMF_NA; ++a) acc[a] = mmf_v4i{256, 256, 256, 256};
#pragma unroll
                    for (int a = 0; a < MMF_NA; ++a) acc[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a][0], b0, acc[a], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < MMF_NA; ++a) acc[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a][1], b1, acc[a], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < MMF_NA; ++a) acc[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a][2], b2, acc[a], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < MMF_NA; ++a) acc[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a][3], b3, acc[a], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < MMF_NA; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {       // acc = 2 h: key = (h << 22) | kp = (acc << 21) + kp (one v_lshl_add_u32)
                            const uint32_t key = (((uint32_t)acc[a][q]) << 21) + kpi;
                            best_k[a][q] = min(best_k[a][q], PARTIAL ? (kvalid ? key : MATCH_NONE) : key);
                        }
                }
            };
            if (t0 + MT > nkp) tile(std::true_type{}); else tile(std::false_type{});
        }
        // ---- register q of lane (g, r16) holds candidate 4 g + q of set a, minimum over the keypoint columns r16 + 16 k it has seen:
        // the 16 lanes of a DPP row finish the minimum, lane r16 == 0 publishes
#pragma unroll
        for (int a = 0; a < MMF_NA; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t v = best_k[a][q];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
                best_k[a][q] = v;
            }
        if (r16 == 0) {
#pragma unroll
            for (int a = 0; a < MMF_NA; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (out_q[a][q] >= 0 && best_k[a][q] != MATCH_NONE) atomicMin(&best[out_q[a][q]], best_k[a][q]);
                }
        }
    }
}

__device__ __forceinline__ int block_excl_scan_flag(bool flag, int* s_w, int& total) {
    // order-preserving position of `flag` lanes inside a 1024-thread workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < 16; ++i) { int c = s_w[i]; if (i < wave) off += c; tot += c; }
    __syncthreads();
    total = tot;
    return off + before;
}

// Distance gate + ORDER-PRESERVING compaction.  The (dist, kp) results of all active points are first
// staged in LDS with coalesced loads (packed to 32 bit); then every thread owns one contiguous
// segment of the active list, so the output order is the list order with a single block scan.
#define GATE_LDS_MAX 36864          // entries (144 KiB)
__global__ __launch_bounds__(1024) void k_match_gate(const LaneDesc* __restrict__ lanes, float ratio, float floor_dist) {
    LANE_PTRS(lanes)
    const int n_active = ld_.n_active, cap = ld_.cap, use_lds = ld_.gate_lds;
    extern __shared__ uint32_t s_pk[];
    __shared__ int s_w[16];
    __shared__ int s_min;
    if (threadIdx.x == 0) s_min = 1 << 30;
#ifdef VO_LM_STAMPS
    long long gt_[6]; int gi_ = 0; gt_[gi_++] = clock64();
#define GATE_STAMP() gt_[gi_++] = clock64();
#else
#define GATE_STAMP()
#endif
    uint32_t mn = MATCH_NONE;                               // distance sits in the top bits: min over the packed words
    if (use_lds) {
        const uint4* b4 = reinterpret_cast<const uint4*>(best);     // 4 results per load (lane buffers are 16-byte aligned)
        uint4* s4 = reinterpret_cast<uint4*>(s_pk);
        const int n4 = (n_active + 3) >> 2;
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += 1024) {
            uint4 v = b4[i];
            const int q = 4 * i;
            if (q + 1 >= n_active) v.y = MATCH_NONE;
            if (q + 2 >= n_active) v.z = MATCH_NONE;
            if (q + 3 >= n_active) v.w = MATCH_NONE;
            mn = min(min(mn, v.x), min(v.y, min(v.z, v.w)));
            s4[i] = v;
        }
    } else {
        for (int q = threadIdx.x; q < n_active; q += 1024) mn = min(mn, best[q]);
    }
    __syncthreads();
    GATE_STAMP()
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(&s_min, mn == MATCH_NONE ? (1 << 30) : (int)(mn >> 22));
    __syncthreads();
    const int gmin = s_min;
    const float max_dis = fmaxf((float)gmin * ratio, floor_dist);
    const int seg = (n_active + 1023) / 1024;
    const int q0 = min(n_active, (int)threadIdx.x * seg), q1 = min(n_active, q0 + seg);
    auto fetch = [&](int q, int& d, int& kp) -> bool {
        const uint32_t pk = use_lds ? s_pk[q] : best[q];
        if (pk == MATCH_NONE) return false;
        d = (int)(pk >> 22); kp = (int)(pk & 0x3FFFFFu);
        return true;
    };
    GATE_STAMP()
    int keep = 0;
    for (int q = q0; q < q1; ++q) { int d, kp; if (fetch(q, d, kp) && (float)d <= max_dis) ++keep; }
    GATE_STAMP()
    int incl = keep;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int off = 0, total = 0;
    for (int i = 0; i < 16; ++i) { const int c = s_w[i]; if (i < wave) off += c; total += c; }
    int pos = off + incl - keep;
    GATE_STAMP()
    // The kept queries go, in list order, to a compact index list (the candidate buffer is free again); k_match_emit
    // writes the records by position with many workgroups, so the dependent gathers (active -> map position,
    // keypoint) of different matches overlap instead of queueing up inside one thread's segment.
    for (int q = q0; q < q1; ++q) {
        int d, kp;
        if (fetch(q, d, kp) && (float)d <= max_dis) { if (pos < cap) cand[pos] = q; ++pos; }
    }
#ifdef VO_LM_STAMPS
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.z == 0) { GATE_STAMP() printf("[gate] n_active %d ncand %d | stage %lld min %lld count %lld scan %lld write %lld clocks\n", n_active, tr->pad0, gt_[1] - gt_[0], gt_[2] - gt_[1], gt_[3] - gt_[2], gt_[4] - gt_[3], gt_[5] - gt_[4]); }
#endif
    if (threadIdx.x == 0) {
        const int ncand = tr->pad0;
        tr->n_cand = ncand; tr->pad0 = 0;                 // counter ready for the next pass
        tr->n_match = min(total, cap); tr->min_dist = ncand ? gmin : -1;
        if (total > cap) tr->status = VO_E_OVERFLOW;
    }
}

// match records + float32 correspondence pairs of the kept queries (list written by k_match_gate), by output position
__global__ __launch_bounds__(256) void k_match_emit(const LaneDesc* __restrict__ lanes) {
    LANE_PTRS(lanes)
    const int32_t* __restrict__ active = ld_.active; const double* __restrict__ map_pos = ld_.map_pos;
    const int nout = tr->n_match;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < nout; p += gridDim.x * 256) {
        const int q = cand[p];
        const uint32_t pk = best[q];
        const int d = (int)(pk >> 22), kp = (int)(pk & 0x3FFFFFu), mi = active[q];
        vo_match m; m.map_index = mi; m.kp_index = kp; m.distance = d; m.flags = 0;
        matches[p] = m;
        cxyz[3 * p] = (float)map_pos[3 * (size_t)mi]; cxyz[3 * p + 1] = (float)map_pos[3 * (size_t)mi + 1]; cxyz[3 * p + 2] = (float)map_pos[3 * (size_t)mi + 2];
        cuv[2 * p] = kps[kp].x; cuv[2 * p + 1] = kps[kp].y;
    }
}

// ------------------------------------------------------------------------------------------
// quartic / P3P / sampler (mirrors oracle/o_track.cpp)
// ------------------------------------------------------------------------------------------
__device__ int solve_quartic_dev(double a4, double a3, double a2, double a1, double a0, double roots[4]) {
    if (!(fabs(a4) > 1e-300)) return 0;
    const double b3 = a3 / a4, b2 = a2 / a4, b1 = a1 / a4, b0 = a0 / a4;
    const double p = b2 - 3.0 * b3 * b3 / 8.0;
    const double q = b1 - b2 * b3 / 2.0 + b3 * b3 * b3 / 8.0;
    const double r = b0 - b1 * b3 / 4.0 + b2 * b3 * b3 / 16.0 - 3.0 * b3 * b3 * b3 * b3 / 256.0;
    double ys[4]; int n = 0;
    const double qtol = 1e-13 * (1.0 + fabs(p) * sqrt(fabs(p)) + fabs(r));
    if (fabs(q) <= qtol) {
        double disc = p * p - 4.0 * r;
        if (disc < 0) return 0;
        double sd = sqrt(disc);
        double z0 = (-p + sd) / 2.0, z1 = (-p - sd) / 2.0;
        if (z0 >= 0) { double y = sqrt(z0); ys[n++] = y; ys[n++] = -y; }
        if (z1 >= 0) { double y = sqrt(z1); ys[n++] = y; ys[n++] = -y; }
    } else {
        const double c2 = 8.0 * p, c1 = 2.0 * p * p - 8.0 * r, c0 = -q * q;
        double lo = 0.0, hi = 1.0;
        int guard = 0;
        while (!((((8.0 * hi + c2) * hi + c1) * hi + c0) > 0.0) && guard < 600) { hi *= 2.0; ++guard; }
        if (guard >= 600) return 0;
        for (int it = 0; it < 64; ++it) {
            double mid = 0.5 * (lo + hi);
            if ((((8.0 * mid + c2) * mid + c1) * mid + c0) > 0.0) hi = mid; else lo = mid;
        }
        double m = 0.5 * (lo + hi);
        for (int it = 0; it < 4; ++it) {
            double gd = (24.0 * m + 2.0 * c2) * m + c1;
            if (gd == 0.0) break;
            double mn = m - (((8.0 * m + c2) * m + c1) * m + c0) / gd;
            if (mn > 0.0) m = mn;
        }
        if (!(m > 0.0)) return 0;
        const double s = sqrt(2.0 * m), h = q / (2.0 * s), base = p / 2.0 + m;
        double d1 = s * s - 4.0 * (base + h);
        if (d1 >= 0) { double sd = sqrt(d1); ys[n++] = (s + sd) / 2.0; ys[n++] = (s - sd) / 2.0; }
        double d2 = s * s - 4.0 * (base - h);
        if (d2 >= 0) { double sd = sqrt(d2); ys[n++] = (-s + sd) / 2.0; ys[n++] = (-s - sd) / 2.0; }
    }
    for (int i = 0; i < 4; ++i) {
        if (i >= n) break;
        double x = ys[i] - b3 / 4.0;
        for (int it = 0; it < 2; ++it) {
            double f = (((x + b3) * x + b2) * x + b1) * x + b0;
            double fd = ((4.0 * x + 3.0 * b3) * x + 2.0 * b2) * x + b1;
            if (fd != 0.0) x -= f / fd;
        }
        roots[i] = x;
    }
    return n;
}

__device__ __forceinline__ void frame_of_dev(D3 A, D3 B, D3 Cc, D3 e[3]) {
    e[0] = nrm3(sub(B, A));
    e[2] = nrm3(cross3(e[0], sub(Cc, A)));
    e[1] = cross3(e[2], e[0]);
}

// returns number of solutions; poses as 12 doubles each
__device__ int p3p_grunert_dev(const D3 P[3], const D3 f[3], double Rt[4][12]) {
    const D3 d12 = sub(P[1], P[2]), d02 = sub(P[0], P[2]), d01 = sub(P[0], P[1]);
    const double a2 = dot3(d12, d12), b2 = dot3(d02, d02), c2 = dot3(d01, d01);
    if (!(a2 > 1e-18 && b2 > 1e-18 && c2 > 1e-18)) return 0;
    const double ca = dot3(f[1], f[2]), cb = dot3(f[0], f[2]), cg = dot3(f[0], f[1]);
    const double q = (a2 - c2) / b2, ac = (a2 + c2) / b2;
    const double A4 = (q - 1.0) * (q - 1.0) - 4.0 * c2 / b2 * ca * ca;
    const double A3 = 4.0 * (q * (1.0 - q) * cb - (1.0 - ac) * ca * cg + 2.0 * c2 / b2 * ca * ca * cb);
    const double A2 = 2.0 * (q * q - 1.0 + 2.0 * q * q * cb * cb + 2.0 * ((b2 - c2) / b2) * ca * ca -
                             4.0 * ac * ca * cb * cg + 2.0 * ((b2 - a2) / b2) * cg * cg);
    const double A1 = 4.0 * (-q * (1.0 + q) * cb + 2.0 * a2 / b2 * cg * cg * cb - (1.0 - ac) * ca * cg);
    const double A0 = (1.0 + q) * (1.0 + q) - 4.0 * a2 / b2 * cg * cg;
    double vs[4];
    const int nr = solve_quartic_dev(A4, A3, A2, A1, A0, vs);
    int ns = 0;
    D3 ep[3];
    frame_of_dev(P[0], P[1], P[2], ep);
    for (int i = 0; i < 4; ++i) {
        if (i >= nr) break;
        const double v = vs[i];
        if (!(v > 0.0)) continue;
        const double den = 2.0 * (cg - v * ca);
        if (fabs(den) < 1e-12) continue;
        const double u = ((q - 1.0) * v * v - 2.0 * q * cb * v + 1.0 + q) / den;
        if (!(u > 0.0)) continue;
        const double s1sq = b2 / (1.0 + v * v - 2.0 * v * cb);
        if (!(s1sq > 0.0)) continue;
        const double s1 = sqrt(s1sq), s2 = u * s1, s3 = v * s1;
        const D3 Q0 = scl(s1, f[0]), Q1 = scl(s2, f[1]), Q2 = scl(s3, f[2]);
        D3 eq[3];
        frame_of_dev(Q0, Q1, Q2, eq);
        double* o = Rt[ns];
        for (int r_ = 0; r_ < 3; ++r_)
            for (int c_ = 0; c_ < 3; ++c_)
                o[3 * r_ + c_] = comp(eq[0], r_) * comp(ep[0], c_) + comp(eq[1], r_) * comp(ep[1], c_) + comp(eq[2], r_) * comp(ep[2], c_);
        const D3 rp = rot(o, P[0]);
        o[9] = Q0.x - rp.x; o[10] = Q0.y - rp.y; o[11] = Q0.z - rp.z;
        if (!(isfinite(o[9]) && isfinite(o[10]) && isfinite(o[11]))) continue;
        ++ns;
    }
    return ns;
}

__device__ __forceinline__ uint64_t mix64_dev(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t rng_draw_dev(uint64_t seed, uint64_t hyp, uint64_t j) { return mix64_dev(seed ^ mix64_dev(hyp * 0x100000001B3ull + j)); }

__device__ void sample4_dev(uint64_t seed, int hyp, int n, int idx[4]) {
    uint64_t j = 0;
    for (int k = 0; k < 4; ++k) {
        for (;;) {
            int c = (int)(rng_draw_dev(seed, (uint64_t)hyp, j++) % (uint64_t)n);
            bool dup = false;
            for (int i = 0; i < k; ++i) dup |= (idx[i] == c);
            if (!dup) { idx[k] = c; break; }
            if (j > 64) {
                for (c = 0; c < n; ++c) { dup = false; for (int i = 0; i < k; ++i) dup |= (idx[i] == c); if (!dup) break; }
                idx[k] = c; break;
            }
        }
    }
}

__device__ __forceinline__ bool reproj_ok_dev(const CamD& cam, const double* T, const float* X, const float* z, double thr2) {
    const D3 pc = xform(T, mk((double)X[0], (double)X[1], (double)X[2]));
    if (!(pc.z > 0)) return false;
    const double du = cam.fx * pc.x + (cam.cx - (double)z[0]) * pc.z, dv = cam.fy * pc.y + (cam.cy - (double)z[1]) * pc.z;
    return du * du + dv * dv <= thr2 * (pc.z * pc.z);
}

__global__ __launch_bounds__(64) void k_ransac_hyp(const LaneDesc* __restrict__ lanes, int n_hyp, int pass) {
    LANE_PTRS(lanes)
    const uint64_t seed = ld_.seed + (uint64_t)pass;
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= n_hyp) return;
    const int n = tr->n_match;
    if (n < 4) { hyp_cnt[h] = -1; return; }
    int id[4];
    sample4_dev(seed, h, n, id);
    D3 P[3], f[3];
    for (int k = 0; k < 3; ++k) {
        P[k] = mk((double)cxyz[3 * id[k]], (double)cxyz[3 * id[k] + 1], (double)cxyz[3 * id[k] + 2]);
        f[k] = nrm3(mk(((double)cuv[2 * id[k]] - cam.cx) / cam.fx, ((double)cuv[2 * id[k] + 1] - cam.cy) / cam.fy, 1.0));
    }
    double Rt[4][12];
    const int ns = p3p_grunert_dev(P, f, Rt);
    int bi = -1;
    double be = DBL_MAX;
    for (int s = 0; s < 4; ++s) {
        if (s >= ns) break;
        const D3 pc = xform(Rt[s], mk((double)cxyz[3 * id[3]], (double)cxyz[3 * id[3] + 1], (double)cxyz[3 * id[3] + 2]));
        if (!(pc.z > 0)) continue;
        const double du = cam.fx * pc.x / pc.z + cam.cx - (double)cuv[2 * id[3]], dv = cam.fy * pc.y / pc.z + cam.cy - (double)cuv[2 * id[3] + 1];
        const double e = du * du + dv * dv;
        if (e < be) { be = e; bi = s; }
    }
    if (bi < 0) { hyp_cnt[h] = -1; return; }
    for (int i = 0; i < 12; ++i) hyp_pose[(size_t)12 * h + i] = Rt[bi][i];
    hyp_cnt[h] = 0;
}

// Scoring: a workgroup keeps up to RS_R x 1024 correspondences in REGISTERS (read once, coalesced) and streams a tile of `ht`
// hypotheses past them: a hypothesis is 12 uniform doubles, i.e. scalar loads, so the vector side only does the reprojection test
// and one ballot + popcount per wavefront.  blockIdx.x = hypothesis tile, blockIdx.y = chunk of RS_R x 1024 correspondences;
// the chunks' counts meet in hyp_cnt[h] (integer atomics: exact in any order).  With one workgroup per hypothesis every one of
// them re-read all K correspondences (round 2: 10x the algorithmic bytes; 800 MB of L2 reads per pass at 2048 x 19.6 k).
#define RS_R 4
__global__ __launch_bounds__(1024) void k_ransac_score(const LaneDesc* __restrict__ lanes, int h_first, int n_hyp, int ht, double thr2, int rank, int world) {
    LANE_PTRS(lanes)
    __shared__ int s_cnt[64];
    const int n = tr->n_match;
    const int k0 = blockIdx.y * (RS_R * 1024);
    if (k0 >= n) return;                                      // the grid is sized for the largest lane
    const double* __restrict__ hp = ld_.hyp_pose;
    // (Keeping a lane's workgroups on one XCD -- an 8x oversized grid whose other workgroups leave at once -- did not lower the fabric
    // traffic of this kernel (8.7x its algorithmic bytes, 4 MB per launch) and cost 4 us of empty waves: measured, removed.)
    const int h0 = h_first + blockIdx.x * ht;
    // second stage (h_first > 0): the scan over the first h_first counts (k_ransac_peek) has bounded the number of hypotheses the
    // sequential RANSAC loop would ever look at; the ones beyond it are not scored (their count stays 0: never a record)
    if (h_first > 0 && h0 >= tr->pad1) return;
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    float X[RS_R][3], Z[RS_R][2];
    bool live[RS_R];
#pragma unroll
    for (int r = 0; r < RS_R; ++r) {
        const int k = k0 + r * 1024 + (int)threadIdx.x, kc = min(k, n - 1);
        live[r] = k < n;
        X[r][0] = cxyz[3 * kc]; X[r][1] = cxyz[3 * kc + 1]; X[r][2] = cxyz[3 * kc + 2]; Z[r][0] = cuv[2 * kc]; Z[r][1] = cuv[2 * kc + 1];
    }
    __syncthreads();
    for (int hi = 0; hi < ht; ++hi) {
        const int h = h0 + hi;
        if (h >= n_hyp) break;
        if (world > 1 && h % world != rank) { if (blockIdx.y == 0 && threadIdx.x == 0) hyp_cnt[h] = 0; continue; }      // another rank scores it: 0 goes into the sum
        if (hyp_cnt[h] < 0) continue;                         // degenerate sample (k_ransac_hyp); counts only grow from 0, so the test is stable
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = hp[(size_t)12 * h + i];
        int c = 0;
#pragma unroll
        for (int r = 0; r < RS_R; ++r) c += __popcll(__ballot(live[r] && reproj_ok_dev(cam, T, X[r], Z[r], thr2)));
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt[hi], c);
    }
    __syncthreads();
    if ((int)threadIdx.x < ht) {
        const int h = h0 + threadIdx.x;
        if (h < n_hyp && s_cnt[threadIdx.x]) atomicAdd(&hyp_cnt[h], s_cnt[threadIdx.x]);
    }
}

__device__ __forceinline__ int ransac_update_iters_dev(double conf, int n_pts, int n_inl, int max_iters) {
    const double w = (double)n_inl / (double)n_pts;
    const double qf = 1.0 - w * w * w * w, target = 1.0 - conf;
    if (!(qf > 0.0)) return 0;
    // The product loop below defines the result (the CPU restatement runs the same loop).  With a poor hypothesis qf is close to 1
    // and the loop would run all max_iters dependent multiplications (13 us at 2048) only to return max_iters: a logarithm
    // estimate with a 2 % margin -- far beyond the k * 2^-53 rounding drift of the product -- settles that case at once.
    if (log(target) / log(qf) > 1.02 * (double)max_iters + 4.0) return max_iters;
    double acc = 1.0; int k = 0;
    while (acc > target && k < max_iters) { acc *= qf; ++k; }
    return k;
}

// The sequential scan of cv::RANSACPointSetRegistrator (best so far, adaptive iteration count) only acts at RECORDS: hypotheses
// whose count exceeds every earlier one.  They are found in parallel (running maximum over the hypothesis order), compacted in
// order, and one thread replays the scan over the handful of records instead of walking 100 .. 2048 counts in global memory.
// Returns (thread 0 only): best hypothesis, its count, the final iteration bound, the index the sequential loop stops at.
__device__ __forceinline__ void ransac_scan_dev(const int* hyp_cnt, int n_hyp, int n, double conf, int* s_w, int* s_pmax, int* s_rec_h, int* s_rec_c, int* s_nrec,
                                                int& bh, int& best_cnt, int& niters, int& used) {
    if (threadIdx.x == 0) *s_nrec = 0;
    __syncthreads();
    int run_max = 3;                                          // a model needs more than 3 inliers (best_cnt starts at 3)
    for (int base = 0; base < n_hyp && n >= 4; base += 1024) {
        const int h = base + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int cnt = h < n_hyp ? hyp_cnt[h] : -1;
        int pm = cnt;                                         // inclusive prefix maximum inside the wavefront
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pm, o, 64); if (lane >= o) pm = max(pm, t); }
        if (lane == 63) s_pmax[wave] = pm;
        __syncthreads();
        int before = run_max;                                 // maximum of everything in front of this lane
        for (int w = 0; w < wave; ++w) before = max(before, s_pmax[w]);
        const int excl = __shfl_up(pm, 1, 64);
        if (lane > 0) before = max(before, excl);
        const bool rec = cnt > before;
        int tot;
        const int pos = block_excl_scan_flag(rec, s_w, tot);
        const int nr0 = *s_nrec;
        if (rec && nr0 + pos < 64) { s_rec_h[nr0 + pos] = h; s_rec_c[nr0 + pos] = cnt; }
        int allmax = run_max;
        for (int w = 0; w < 16; ++w) allmax = max(allmax, s_pmax[w]);
        run_max = allmax;
        __syncthreads();
        if (threadIdx.x == 0) *s_nrec = min(64, nr0 + tot);   // counts grow at least by one per record: the replay below stops long before 64 records
        __syncthreads();
    }
    bh = -1; best_cnt = 3; niters = n_hyp; used = 0;
    if (threadIdx.x == 0 && n >= 4) {
        for (int i = 0; i < *s_nrec; ++i) {
            const int h = s_rec_h[i], cnt = s_rec_c[i];
            if (h >= niters) break;                           // the sequential scan would have stopped before this hypothesis
            best_cnt = cnt; bh = h; niters = min(niters, ransac_update_iters_dev(conf, n, cnt, niters));
        }
        used = max(niters, bh + 1);                           // the scan leaves its loop at the first index >= niters, and it has looked at index bh
    }
}

// after the first stage of the scoring: how far the sequential scan can still get (tr->pad1), given the counts of hypotheses [0, h_first)
__global__ __launch_bounds__(1024) void k_ransac_peek(const LaneDesc* __restrict__ lanes, int h_first, int n_hyp, double conf) {
    LANE_PTRS(lanes)
    __shared__ int s_w[16], s_pmax[16], s_rec_h[64], s_rec_c[64], s_nrec;
    int bh, best_cnt, niters, used;
    ransac_scan_dev(hyp_cnt, h_first, tr->n_match, conf, s_w, s_pmax, s_rec_h, s_rec_c, &s_nrec, bh, best_cnt, niters, used);
    // the bound after h_first hypotheses, for a scan over n_hyp of them: update_iters(.., max = n_hyp) clipped by what the records gave
    if (threadIdx.x == 0) {
        int lim = n_hyp;
        if (tr->n_match >= 4) for (int i = 0; i < s_nrec; ++i) { if (s_rec_h[i] >= lim) break; lim = min(lim, ransac_update_iters_dev(conf, tr->n_match, s_rec_c[i], lim)); }
        tr->pad1 = lim;
    }
}

__global__ __launch_bounds__(1024) void k_ransac_select(const LaneDesc* __restrict__ lanes, int n_hyp, double thr2, double conf) {
    LANE_PTRS(lanes)
    __shared__ int s_w[16];
    __shared__ int s_best;
    __shared__ int s_pmax[16];
    __shared__ int s_rec_h[64], s_rec_c[64], s_nrec;
    const int n = tr->n_match;
    {
        int bh, best_cnt, niters, used;
        ransac_scan_dev(hyp_cnt, n_hyp, n, conf, s_w, s_pmax, s_rec_h, s_rec_c, &s_nrec, bh, best_cnt, niters, used);
        if (threadIdx.x == 0) {
            s_best = bh;
            tr->best_hyp = bh; tr->iters_used = used; tr->best_cnt = bh >= 0 ? best_cnt : 0;
            if (bh >= 0) for (int i = 0; i < 12; ++i) tr->T[i] = hyp_pose[(size_t)12 * bh + i];
            for (int i = 0; i < 12; ++i) tr->T_ransac[i] = tr->T[i];
        }
    }
    __syncthreads();
    const int bsel = s_best;
    if (bsel < 0) { if (threadIdx.x == 0) tr->n_inl = 0; return; }
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = hyp_pose[(size_t)12 * bsel + i];
    int outn = 0;
    for (int base = 0; base < n; base += 1024) {
        const int k = base + threadIdx.x;
        const bool ok = (k < n) && reproj_ok_dev(cam, T, &cxyz[3 * k], &cuv[2 * k], thr2);
        int tot;
        const int pos = outn + block_excl_scan_flag(ok, s_w, tot);
        if (ok) inliers[pos] = k;
        outn += tot;
    }
    if (threadIdx.x == 0) tr->n_inl = outn;
}

// ------------------------------------------------------------------------------------------
// K14 pose-only LM in one workgroup
// ------------------------------------------------------------------------------------------
#define LM_T 512
#define LM_W (LM_T / 64)
#define LM_NV 28            // 21 (upper H) + 6 (b) + 1 (chi)
#define LM_LDS_MAX 6144     // inlier correspondences staged in LDS (20 B each)

__device__ __forceinline__ void so3_exp_dev(const double w[3], double R[9]) {
#pragma clang fp contract(fast)
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double A, B;
    if (th2 < 0.0625) {     // |theta| < 0.25: Taylor to theta^14 (error < 1e-19), avoids the sin/cos call chains
        // reciprocal constants: a division by a literal is still a full IEEE divide (~10 dependent operations) on the GPU
        A = 1.0 - th2 * (1.0 / 6.0) * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0) * (1.0 - th2 * (1.0 / 156.0) * (1.0 - th2 * (1.0 / 210.0)))))));
        B = 0.5 * (1.0 - th2 * (1.0 / 12.0) * (1.0 - th2 * (1.0 / 30.0) * (1.0 - th2 * (1.0 / 56.0) * (1.0 - th2 * (1.0 / 90.0) * (1.0 - th2 * (1.0 / 132.0) * (1.0 - th2 * (1.0 / 182.0) * (1.0 - th2 * (1.0 / 240.0))))))));
    } else { A = sin(th) / th; B = (1.0 - cos(th)) / th2; }
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += W[3 * i + k] * W[3 * k + j]; W2[3 * i + j] = s; }
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + A * W[i] + B * W2[i];
}

// Tn = exp(d) * T   (tangent = [translation, rotation], g2o_types.h:56-60)
__device__ void se3_exp_mul_dev(const double d[6], const double* T, double* Tn) {
#pragma clang fp contract(fast)
    const double w[3] = {d[3], d[4], d[5]};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double B, C;
    if (th2 < 0.0625) {
        B = 0.5 * (1.0 - th2 * (1.0 / 12.0) * (1.0 - th2 * (1.0 / 30.0) * (1.0 - th2 * (1.0 / 56.0) * (1.0 - th2 * (1.0 / 90.0) * (1.0 - th2 * (1.0 / 132.0) * (1.0 - th2 * (1.0 / 182.0) * (1.0 - th2 * (1.0 / 240.0))))))));
        C = 1.0 / 6.0 * (1.0 - th2 * (1.0 / 20.0) * (1.0 - th2 * (1.0 / 42.0) * (1.0 - th2 * (1.0 / 72.0) * (1.0 - th2 * (1.0 / 110.0) * (1.0 - th2 * (1.0 / 156.0) * (1.0 - th2 * (1.0 / 210.0) * (1.0 - th2 * (1.0 / 272.0))))))));
    } else { B = (1.0 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9], V[9], R[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += W[3 * i + k] * W[3 * k + j]; W2[3 * i + j] = s; }
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + B * W[i] + C * W2[i];
    so3_exp_dev(w, R);
    const double tx = V[0] * d[0] + V[1] * d[1] + V[2] * d[2], ty = V[3] * d[0] + V[4] * d[1] + V[5] * d[2], tz = V[6] * d[0] + V[7] * d[1] + V[8] * d[2];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += R[3 * i + k] * T[3 * k + j]; Tn[3 * i + j] = s; }
    Tn[9] = R[0] * T[9] + R[1] * T[10] + R[2] * T[11] + tx;
    Tn[10] = R[3] * T[9] + R[4] * T[10] + R[5] * T[11] + ty;
    Tn[11] = R[6] * T[9] + R[7] * T[10] + R[8] * T[11] + tz;
}

// 6x6 Cholesky solve; one reciprocal per pivot instead of a division per element
__device__ __forceinline__ bool chol6_dev(double* A, double* b) {
#pragma clang fp contract(fast)
    double inv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0.0)) return false;
        inv[j] = vo_rsqrt_f64(d);
#pragma unroll
        for (int i = j + 1; i < 6; ++i) { double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s * inv[j]; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) { double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= A[i * 6 + k] * b[k];
        b[i] = s * inv[i]; }
#pragma unroll
    for (int i = 5; i >= 0; --i) { double s = b[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= A[k * 6 + i] * b[k];
        b[i] = s * inv[i]; }
    return true;
}

// ---- several workgroups per lane (large inlier sets) ------------------------------------------------------------------------
// Above a few thousand inliers one edge pass costs more than a hand-off between workgroups (config 5: 18 k inliers, 20 us per
// pass in one workgroup), so the lane's inlier list is cut into `nwg` contiguous shares, one workgroup each.  After its own
// reduction every workgroup publishes its 28 sums (write-through stores), arrives at the lane's counter and, once all have
// arrived, adds up ALL partials in the same order: every workgroup then holds the same H, b, chi2 and takes the same LM
// decisions -- the serial part is replicated, nothing is broadcast.  Hand-off recipe: cdna_hip_programming.md Guideline 16 (R1) with
// agent-scope (sc1) loads on the consuming side.  The partials are double buffered by episode parity: a workgroup can be at
// most one episode ahead of the slowest one.  Every spin is bounded (a timeout marks the lane's result and lets everybody go on).
#define LM_KMAX 8
#define LM_X_DOUBLES (32 + 2 * LM_KMAX * 32)       // per lane: [counter line | exit line] (256 B) + 2 x LM_KMAX x 32 partial sums
struct LmX { unsigned* cnt; unsigned* exit_cnt; double* part; int nwg, wg; unsigned epi; int* status; int* s_flag; bool dead; };      // dead: a hand-off has run out (sticky: the launch's remaining ones are skipped, its result carries VO_E_DEVICE)
__device__ __forceinline__ void lm_xchg(LmX& X, double* tot, int nv) {      // tot: LDS, [0, nv) valid in this workgroup -> sums over the lane's workgroups
    const int par = X.epi & 1;
    if (X.dead) { ++X.epi; return; }                  // (workgroup-uniform: every thread read the same s_flag)
    if ((int)threadIdx.x < nv) __hip_atomic_store(X.part + (size_t)(par * LM_KMAX + X.wg) * 32 + threadIdx.x, tot[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++X.epi;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(X.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)X.nwg * X.epi;
        unsigned spins = 0; unsigned long long t0 = 0; int ok = 1;
        while (__hip_atomic_load(X.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (!t0) t0 = now; else if (now - t0 > 20000000ull) { ok = 0; break; }       // 0.2 s: a peer workgroup never became resident
            }
        }
        if (!ok) *X.status = VO_E_DEVICE;
        *X.s_flag = ok;
    }
    __syncthreads();
    if (!*X.s_flag) { X.dead = true; return; }        // the sums stay this workgroup's own; no further hand-off waits 0.2 s again
    if ((int)threadIdx.x < nv) {
        double sum = 0;
        for (int w = 0; w < X.nwg; ++w) sum += __hip_atomic_load(X.part + (size_t)(par * LM_KMAX + w) * 32 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[threadIdx.x] = sum;
    }
    __syncthreads();
}

// One pass over the active edges at pose T: robust chi2, H (upper triangle) and b, reduced over the whole
// workgroup into the LDS buffer `tot` ([0..20] H, [21..26] b, [27] chi) that every thread reads afterwards.
__device__ void lm_pass_dev(const CamD& cam, const float* cxyz, const float* cuv, const int32_t* edges,
                            const uint8_t* mask, int n, int round, const double* T, double delta,
                            double* s_part, double* tot, int& phase, const uint16_t* idx, int nidx, LmX* X) {
#pragma clang fp contract(fast)
    const bool robust = round == 0;
#ifdef VO_LM_STAMPS
    const long long ts0 = clock64();
#endif
    double a32[32];                                    // [0..20] H upper triangle, [21..26] b, [27] chi2, [28..31] padding of the reduction
#pragma unroll
    for (int i = 0; i < 32; ++i) a32[i] = 0;
    // Per edge: the weighted outer products of a = [J0, e0] and b = [J1, e1] give H, J^T e and chi2 at once;
    // J0[1] = J1[0] = 0 (g2o_types.h:97-99) removes a third of the products.
    double g[6] = {0, 0, 0, 0, 0, 0};
    // second round with the correspondences in LDS: the surviving edges come as a compact index list, so every lane has
    // work (a masked loop leaves more than half of the lanes idle in every iteration)
    const int cnt = idx ? nidx : n;
    for (int i = threadIdx.x; i < cnt; i += LM_T) {
        int k;
        if (idx) k = idx[i];
        else {
            if (round == 1 && !(mask[i] & 2)) continue;
            k = edges ? edges[i] : i;                  // edges == nullptr: correspondences already gathered (LDS)
        }
        // R p + t with fused multiply-adds (the shared xform() helper is compiled without contraction for the kernels whose
        // integer decisions must match the oracle bit for bit; here only the tolerance-checked LM sums depend on it)
        const double px = (double)cxyz[3 * k], py = (double)cxyz[3 * k + 1], pz = (double)cxyz[3 * k + 2];
        D3 pc;
        pc.x = fma(T[0], px, fma(T[1], py, fma(T[2], pz, T[9])));
        pc.y = fma(T[3], px, fma(T[4], py, fma(T[5], pz, T[10])));
        pc.z = fma(T[6], px, fma(T[7], py, fma(T[8], pz, T[11])));
        const double zz = pc.z + 1e-18;
        // 1 / z: v_rcp_f64 (2^-26) + two Newton steps, ~1 ulp; the IEEE sequence (div_scale / fmas / fixup) is twice as long
        double Zi = __builtin_amdgcn_rcp(zz);
        Zi = fma(fma(-zz, Zi, 1.0), Zi, Zi);
        Zi = fma(fma(-zz, Zi, 1.0), Zi, Zi);
        const double fx = cam.fx, fy = cam.fy, xz = pc.x * Zi, yz = pc.y * Zi;
        const double e0 = (double)cuv[2 * k] - (fx * xz + cam.cx), e1 = (double)cuv[2 * k + 1] - (fy * yz + cam.cy);   // g2o_types.h:83
        const double e2 = e0 * e0 + e1 * e1;
        double r1 = 1.0;
        if (robust && e2 > delta * delta) { const double se = sqrt(e2); a32[27] += 2.0 * se * delta - delta * delta; r1 = delta / se; } else a32[27] += e2;
        const double fxz = fx * Zi, fyz = fy * Zi;
        // J0 = [-fx/Z, 0, fx X/Z^2, fx XY/Z^2, -fx - fx X^2/Z^2, fx Y/Z],  J1 = [0, -fy/Z, fy Y/Z^2, fy + fy Y^2/Z^2, -fy XY/Z^2, -fy X/Z]
        const double a0 = -fxz, a2 = fxz * xz, a3 = a2 * pc.y, a4 = -fx - a2 * pc.x, a5 = fxz * pc.y;
        const double b1 = -fyz, b2 = fyz * yz, b3 = fy + b2 * pc.y, b4 = -b2 * pc.x, b5 = -fyz * pc.x;
        const double wa0 = r1 * a0, wa2 = r1 * a2, wa3 = r1 * a3, wa4 = r1 * a4, wa5 = r1 * a5;
        const double wb1 = r1 * b1, wb2 = r1 * b2, wb3 = r1 * b3, wb4 = r1 * b4, wb5 = r1 * b5;
        // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
        a32[0] += wa0 * a0; a32[2] += wa0 * a2; a32[3] += wa0 * a3; a32[4] += wa0 * a4; a32[5] += wa0 * a5;
        a32[6] += wb1 * b1; a32[7] += wb1 * b2; a32[8] += wb1 * b3; a32[9] += wb1 * b4; a32[10] += wb1 * b5;
        a32[11] += wa2 * a2 + wb2 * b2; a32[12] += wa2 * a3 + wb2 * b3; a32[13] += wa2 * a4 + wb2 * b4; a32[14] += wa2 * a5 + wb2 * b5;
        a32[15] += wa3 * a3 + wb3 * b3; a32[16] += wa3 * a4 + wb3 * b4; a32[17] += wa3 * a5 + wb3 * b5;
        a32[18] += wa4 * a4 + wb4 * b4; a32[19] += wa4 * a5 + wb4 * b5;
        a32[20] += wa5 * a5 + wb5 * b5;
        g[0] += wa0 * e0; g[1] += wb1 * e1; g[2] += wa2 * e0 + wb2 * e1; g[3] += wa3 * e0 + wb3 * e1; g[4] += wa4 * e0 + wb4 * e1; g[5] += wa5 * e0 + wb5 * e1;
    }
#pragma unroll
    for (int a_ = 0; a_ < 6; ++a_) a32[21 + a_] = -g[a_];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef VO_LM_STAMPS
    const long long ts1 = clock64();
#endif
    const double rsum = vo_wave_reduce32t(a32);        // lane l of row r: wavefront sum of a32[4 VO_R32T_K(l) + VO_R32_SLOT(r)]
#ifdef VO_LM_STAMPS
    const long long ts2 = clock64();
#endif
    // cross-wave: partials and totals are double buffered by `phase`, so two barriers per pass are enough (a buffer
    // is rewritten two passes later, after every reader has crossed two barriers)
    double* part = s_part + phase * (LM_W * 32);
    if ((lane & 15) < 8) part[wave * 32 + 4 * VO_R32T_K(lane) + VO_R32_SLOT(lane >> 4)] = rsum;
    __syncthreads();
    if (threadIdx.x < LM_NV) {
        double sum = 0;
#pragma unroll
        for (int w = 0; w < LM_W; ++w) sum += part[w * 32 + threadIdx.x];
        tot[threadIdx.x] = sum;
    }
    __syncthreads();                                   // tot[0..27] (LDS) is now valid for every thread
    phase ^= 1;
    if (X) lm_xchg(*X, tot, LM_NV);                    // several workgroups per lane: sums over all of them
#ifdef VO_LM_STAMPS
    if (threadIdx.x == 0) { g_dbg[4] += ts1 - ts0; g_dbg[5] += ts2 - ts1; g_dbg[6] += clock64() - ts2; }
#endif
}

// g2o Levenberg-Marquardt (lambda/rho policy of OptimizationAlgorithmLevenberg).  The trial pass also
// linearises at the trial pose, so an accepted step needs no second pass (same numbers g2o recomputes).
__device__ int lm_optimize_dev(const CamD& cam, const float* cxyz, const float* cuv, const int32_t* edges, const uint8_t* mask, int n,
                               int round, double* T, double delta, int max_it, double* s_part, double* s_tot, int& phase,
                               LmX* X, const uint16_t* idx = nullptr, int nidx = 0) {
    // The current linearisation and the trial one live in two LDS buffers (s_tot + 32 cur, s_tot + 32 (cur ^ 1)): an
    // accepted step flips `cur`, nothing is copied and no thread keeps 2 x 28 doubles in registers across a pass.
    int cur = 0;
#ifdef VO_LM_STAMPS
    long long t_pass = 0, t_serial = 0, n_pass = 1, t0 = clock64();
#endif
    lm_pass_dev(cam, cxyz, cuv, edges, mask, n, round, T, delta, s_part, s_tot, phase, idx, nidx, X);
#ifdef VO_LM_STAMPS
    t_pass += clock64() - t0;
#endif
    double lambda, ni = 2;
    {
        const int dg[6] = {0, 6, 11, 15, 18, 20};
        double md = 0;
        for (int i = 0; i < 6; ++i) md = fmax(md, fabs(s_tot[dg[i]]));
        lambda = 1e-5 * md;
    }
    int it = 0;
    for (; it < max_it; ++it) {
        const double* cv = s_tot + 32 * cur;
        double H[36], bvec[6];
        { int c = 0; for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { const double h = cv[c]; H[a * 6 + b] = h; H[b * 6 + a] = h; ++c; } }
        for (int a = 0; a < 6; ++a) bvec[a] = cv[21 + a];
        double chi_cur = cv[27];
        double rho = 0; int qmax = 0; bool converged = false;
        do {
            double A[36], x[6], Tn[12];
#ifdef VO_LM_STAMPS
            long long t1 = clock64();
#endif
            for (int i = 0; i < 36; ++i) A[i] = H[i];
            for (int i = 0; i < 6; ++i) { A[i * 7] += lambda; x[i] = bvec[i]; }
            const bool ok = chol6_dev(A, x);
            if (ok) se3_exp_mul_dev(x, T, Tn);
            else for (int i = 0; i < 12; ++i) Tn[i] = T[i];
#ifdef VO_LM_STAMPS
            long long t2 = clock64(); t_serial += t2 - t1;
#endif
            // block-uniform control flow: every thread holds the same H, b, lambda
            const double* tv = s_tot + 32 * (cur ^ 1);
            lm_pass_dev(cam, cxyz, cuv, edges, mask, n, round, Tn, delta, s_part, s_tot + 32 * (cur ^ 1), phase, idx, nidx, X);
#ifdef VO_LM_STAMPS
            t_pass += clock64() - t2; ++n_pass;
#endif
            const double tmp = ok ? tv[27] : DBL_MAX;
            rho = chi_cur - tmp;
            double scale = 1e-3;
            if (ok) for (int i = 0; i < 6; ++i) scale += x[i] * (lambda * x[i] + bvec[i]);
            rho /= scale;
            if (rho > 0 && isfinite(tmp)) {
                double a = 1.0 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                a = fmin(a, 2.0 / 3.0);
                lambda *= fmax(1.0 / 3.0, a); ni = 2; chi_cur = tmp;
                for (int i = 0; i < 12; ++i) T[i] = Tn[i];
                cur ^= 1;                                  // the trial linearisation becomes the current one
            } else { lambda *= ni; ni *= 2; }
            if (ok) {
                double mx = 0;
                for (int i = 0; i < 6; ++i) mx = fmax(mx, fabs(x[i]));
                converged = mx < 1e-10;
            }
            ++qmax;
        } while (rho < 0 && qmax < 10 && !converged);
        if (qmax == 10 || rho == 0 || converged) { ++it; break; }
    }
#ifdef VO_LM_STAMPS
    if (threadIdx.x == 0) { g_dbg[0] += t_pass; g_dbg[1] += t_serial; g_dbg[2] += n_pass; g_dbg[3] += it; }
#endif
    return it;
}

__device__ __forceinline__ double edge_chi2_dev(const CamD& cam, const double* T, const float* X, const float* z) {
    const D3 pc = xform(T, mk((double)X[0], (double)X[1], (double)X[2]));
    const double Zi = 1.0 / (pc.z + 1e-18);
    const double e0 = (double)z[0] - (cam.fx * pc.x * Zi + cam.cx), e1 = (double)z[1] - (cam.fy * pc.y * Zi + cam.cy);
    return e0 * e0 + e1 * e1;
}

__global__ __launch_bounds__(LM_T) void k_pose_lm(const LaneDesc* __restrict__ lanes, double delta, double cut, int it_r, int it_p, int write_flags) {
    LANE_PTRS(lanes)
    if (!write_flags) matches = nullptr;
#ifdef VO_LM_STAMPS
    const long long t_kernel0 = clock64();
#endif
    extern __shared__ float s_corr[];                  // gathered inlier correspondences: n x 3 then n x 2 floats
    __shared__ double s_part[2 * LM_W * 32];
    __shared__ double s_out[2 * 32];
    __shared__ double s_x[32];
    int phase = 0;
    __shared__ int s_cnt[2];
    __shared__ int s_xflag;
    // this workgroup's share of the lane's inlier list: [lo, lo + n) of n_all (gridDim.x workgroups per lane)
    const int n_all = tr->n_inl, nwg = gridDim.x, wg = blockIdx.x;
    const int lo = (int)((long long)n_all * wg / nwg), n = (int)((long long)n_all * (wg + 1) / nwg) - lo;
    LmX X_; LmX* X = nullptr;
    if (nwg > 1) {
        unsigned* cw = reinterpret_cast<unsigned*>(ld_.lm_x);
        X_.cnt = cw; X_.exit_cnt = cw + 32; X_.part = ld_.lm_x + 32; X_.nwg = nwg; X_.wg = wg; X_.epi = 0; X_.status = &tr->status; X_.s_flag = &s_xflag; X_.dead = false;
        X = &X_;
    }
    const int32_t* edges = inliers + lo;
    const int32_t* edges_g = edges;
    uint8_t* const mask_l = mask + lo;
    double T[12];
    for (int i = 0; i < 12; ++i) T[i] = tr->T[i];
    if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    if (n <= LM_LDS_MAX) {
        for (int i = threadIdx.x; i < n; i += LM_T) {
            const int k = edges[i];
            s_corr[3 * i] = cxyz[3 * k]; s_corr[3 * i + 1] = cxyz[3 * k + 1]; s_corr[3 * i + 2] = cxyz[3 * k + 2];
            s_corr[3 * n + 2 * i] = cuv[2 * k]; s_corr[3 * n + 2 * i + 1] = cuv[2 * k + 1];
        }
        cxyz = s_corr; cuv = s_corr + 3 * n; edges = nullptr;
    }
    __syncthreads();
    int iters = 0;
    if (n_all > 0) iters += lm_optimize_dev(cam, cxyz, cuv, edges, mask_l, n, 0, T, delta, it_r, s_part, s_out, phase, X);
    // edges whose chi2 exceeds the cut leave the second round (frontend.cpp:294-306)
    int loc = 0;
    __shared__ uint16_t s_keep[LM_LDS_MAX];            // LDS path: ordered list of the surviving edges
    __shared__ int s_wk[LM_W];
    const bool lds_path = edges == nullptr;
    int run = 0;
    for (int base = 0; base < n; base += LM_T) {
        const int i = base + threadIdx.x;
        bool keep = false;
        if (i < n) {
            const int k = edges ? edges[i] : i;
            keep = !(edge_chi2_dev(cam, T, &cxyz[3 * k], &cuv[2 * k]) > cut);
            mask_l[i] = keep ? 2 : 0;
            loc += keep ? 1 : 0;
        }
        if (lds_path) {                                // block-uniform
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const unsigned long long m = __ballot(keep);
            if (lane == 0) s_wk[wave] = __popcll(m);
            __syncthreads();
            int before = __popcll(m & ((1ull << lane) - 1ull)), tot = 0;
#pragma unroll
            for (int w = 0; w < LM_W; ++w) { const int cw = s_wk[w]; if (w < wave) before += cw; tot += cw; }
            if (keep) s_keep[run + before] = (uint16_t)i;
            run += tot;
            __syncthreads();
        }
    }
    if (loc) atomicAdd(&s_cnt[0], loc);
    __syncthreads();
    int kept_all = s_cnt[0];
    if (X) {                                           // the lane's total (every workgroup takes the same branch below)
        if (threadIdx.x == 0) s_x[0] = (double)s_cnt[0];
        __syncthreads();
        lm_xchg(*X, s_x, 1);
        kept_all = (int)s_x[0];
    }
    if (kept_all > 0) iters += lm_optimize_dev(cam, cxyz, cuv, edges, mask_l, n, 1, T, delta, it_p, s_part, s_out, phase, X, lds_path ? s_keep : nullptr, run);
    loc = 0;
    for (int i = threadIdx.x; i < n; i += LM_T) {
        const int k = edges ? edges[i] : i;
        const bool in = !(edge_chi2_dev(cam, T, &cxyz[3 * k], &cuv[2 * k]) > cut);
        mask_l[i] = (mask_l[i] & 2) | (in ? 1 : 0);
        if (matches) matches[edges_g[i]].flags = VO_MATCH_RANSAC_INLIER | (in ? VO_MATCH_LM_INLIER : 0);     // frontend.cpp:242,:317-329
        loc += in ? 1 : 0;
    }
    if (loc) atomicAdd(&s_cnt[1], loc);
    __syncthreads();
    int inl_all = s_cnt[1];
    if (X) {
        if (threadIdx.x == 0) s_x[0] = (double)s_cnt[1];
        __syncthreads();
        lm_xchg(*X, s_x, 1);
        inl_all = (int)s_x[0];
        // the last workgroup out clears the counters for the next launch (nobody polls them any more)
        if (threadIdx.x == 0 && __hip_atomic_fetch_add(X->exit_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg - 1) {
            __hip_atomic_store(X->cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(X->exit_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0 && wg == 0) {
        for (int i = 0; i < 12; ++i) tr->T[i] = T[i];
        tr->lm_iters += iters; tr->n_lm_inl = inl_all;
#ifdef VO_LM_STAMPS
        for (int i = 0; i < 7; ++i) { tr->dbg[i] = g_dbg[i]; g_dbg[i] = 0; }
        tr->dbg[7] = clock64() - t_kernel0;
#endif
    }
}

__global__ void k_set_nmatch(TrackDev* tr, int n) { tr->n_match = n; tr->n_inl = 0; }

// device-map update: scatter packed host records to their slots (vo_map_upsert)
// desc: one row per point, or -- with kp != nullptr -- the descriptor table of a frame slot, row kp[i] (point created from that keypoint)
__global__ void k_map_scatter(int n, const int32_t* __restrict__ idx, const double* __restrict__ xyz, const double* __restrict__ nrm,
                              const uint32_t* __restrict__ desc, const int32_t* __restrict__ kp, const uint8_t* __restrict__ flags, double* __restrict__ mpos,
                              double* __restrict__ mnrm, uint32_t* __restrict__ mdesc, uint8_t* __restrict__ mflags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t k = (size_t)idx[i];
    if (kp) { const uint32_t* src = desc + 8 * (size_t)kp[i]; for (int w = 0; w < 8; ++w) mdesc[8 * k + w] = src[w]; desc = nullptr; }
    if (xyz) { mpos[3 * k] = xyz[3 * (size_t)i]; mpos[3 * k + 1] = xyz[3 * (size_t)i + 1]; mpos[3 * k + 2] = xyz[3 * (size_t)i + 2]; }
    if (nrm) { mnrm[3 * k] = nrm[3 * (size_t)i]; mnrm[3 * k + 1] = nrm[3 * (size_t)i + 1]; mnrm[3 * k + 2] = nrm[3 * (size_t)i + 2]; }
    if (desc) for (int w = 0; w < 8; ++w) mdesc[8 * k + w] = desc[8 * (size_t)i + w];
    if (flags) mflags[k] = flags[i];
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
int vo_track_set_attrs() {
    HIP_TRY(hipFuncSetAttribute((const void*)k_match_gate, hipFuncAttributeMaxDynamicSharedMemorySize, GATE_LDS_MAX * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)k_pose_lm, hipFuncAttributeMaxDynamicSharedMemorySize, LM_LDS_MAX * 20));
    return VO_OK;
}

static CamD cam_of(const vo_ctx* c) { CamD k; k.fx = c->p.fx; k.fy = c->p.fy; k.cx = c->p.cx; k.cy = c->p.cy; k.W = c->p.width; k.H = c->p.height; return k; }

// Descriptor of lane `lane` of context c, tracking the frame in slot `slot`; d_tr = where the lane's result header lives.
void vo_lane_fill(vo_ctx* c, int lane, int slot, uint64_t seed, TrackDev* d_tr, LaneDesc* o) {
    const size_t M = c->lane_stride, L = (size_t)lane;
    o->tr = d_tr; o->best = c->d_best + L * M; o->cand = c->d_mcand + L * M; o->matches = c->d_matches + L * M;
    o->cxyz = c->d_corr_xyz + 3 * L * M; o->cuv = c->d_corr_uv + 2 * L * M;
    o->hyp_pose = c->d_hyp_pose + (size_t)12 * L * c->p.max_hypotheses; o->hyp_cnt = c->d_hyp_cnt + L * c->p.max_hypotheses;
    o->inliers = c->d_inliers + L * M; o->mask = c->d_lm_mask + L * M; o->lm_x = c->d_lm_x + L * LM_X_DOUBLES;
    o->fdesc = (const uint32_t*)(c->d_desc + (size_t)slot * c->plan.nfeat * 32); o->nkp = c->d_nkp + slot; o->kps = c->d_kps + (size_t)slot * c->plan.nfeat;
    o->map_pos = c->d_map_pos; o->map_nrm = c->d_map_nrm; o->map_flags = c->d_map_flags; o->map_desc = c->d_map_desc; o->active = c->d_active;
    o->n_active = c->n_active; o->cap = c->corr_cap; o->max_hyp = c->p.max_hypotheses; o->gate_lds = c->n_active <= GATE_LDS_MAX ? 1 : 0;
    o->seed = seed; o->cam = cam_of(c);
}

int vo_track_match_launch(vo_ctx* prof, hipStream_t st, const LaneDesc* dl, int nl, ChainDims dims, float ratio, float floor_dist) {
    const int na = dims.max_active;
    if (na > 0) {
        { ProfScope ps(prof, "k_frustum", st);
          hipLaunchKernelGGL(k_frustum, dim3((na + 255) / 256, 1, nl), dim3(256), 0, st, dl); }
        const char* mmf_s = getenv("VO_MATCH_MFMA");        // tests / experiments: 0 = never, 1 = always (read per launch)
        const int mmf_env = mmf_s ? atoi(mmf_s) : -1;
        const bool use_mfma = mmf_env >= 0 ? mmf_env != 0 : (long long)na * dims.max_feat >= MMF_MIN_PAIRS;
        ProfScope ps(prof, use_mfma ? "k_match_mfma" : "k_match", st);
        // keypoint rounds (8 tiles = 512 keypoints) are spread over up to 4 workgroups only while the launch stays within ~4 workgroups per compute
        // unit: every extra slice re-reads the candidate tile (usually through another XCD's L2) and adds an atomicMin per candidate
        if (use_mfma) {
            // candidate tiles of MMF_CAND; the keypoint tiles of a lane are spread over as many workgroups as keep the grid within ~4 per compute unit
            const int ct = std::min((na + MMF_CAND - 1) / MMF_CAND, MATCH_GRID_X), kt = (dims.max_feat + MT - 1) / MT;
            const int ksplit = std::max(1, std::min(kt, 1024 / std::max(1, ct * std::max(1, nl))));
            hipLaunchKernelGGL(k_match_mfma, dim3(ct, ksplit, nl), dim3(256), 0, st, dl);
        } else {
        const int tiles = std::min((na + MQ - 1) / MQ, MATCH_GRID_X) * std::max(1, nl);
        const int ksplit = std::max(1, std::min(std::min(4, (dims.max_feat + MW * MT - 1) / (MW * MT)), 1024 / std::max(1, tiles)));
        hipLaunchKernelGGL(k_match, dim3(std::min((na + MQ - 1) / MQ, MATCH_GRID_X), ksplit, nl), dim3(64 * MW), 0, st, dl);
        }
    }
    { ProfScope ps(prof, "k_match_gate", st);
      const size_t lds = sizeof(uint32_t) * (((size_t)std::min(na, GATE_LDS_MAX) + 3) & ~(size_t)3);
      hipLaunchKernelGGL(k_match_gate, dim3(1, 1, nl), dim3(1024), lds, st, dl, ratio, floor_dist); }
    if (na > 0) { ProfScope ps(prof, "k_match_emit", st); hipLaunchKernelGGL(k_match_emit, dim3(32, 1, nl), dim3(256), 0, st, dl); }
    HIP_TRY(hipGetLastError());
    return VO_OK;
}

int vo_corr_from_host(vo_ctx* c, const float* xyz, const float* uv, int n) {
    if (n > c->corr_cap) return VO_E_OVERFLOW;
    hipStream_t st = c->stream;
    if (n > 0) {
        float* stage = (float*)vo_stage(c, sizeof(float) * 5 * (size_t)n);
        if (!stage) return VO_E_NOMEM;
        HIP_TRY(hipStreamSynchronize(st));                  // the staging buffer may still feed an earlier vo_map_upsert
        memcpy(stage, xyz, sizeof(float) * 3 * (size_t)n);
        memcpy(stage + 3 * (size_t)n, uv, sizeof(float) * 2 * (size_t)n);
        HIP_TRY(hipMemcpyAsync(c->d_corr_xyz, stage, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->d_corr_uv, stage + 3 * (size_t)n, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_set_nmatch, dim3(1), dim3(1), 0, st, c->d_track, n);
    HIP_TRY(hipStreamSynchronize(st));      // staging buffer is reused by later calls
    c->corr_external = true;
    return VO_OK;
}

int vo_track_ransac_launch(vo_ctx* prof, hipStream_t st, const LaneDesc* dl, int nl, int n_hyp, float reproj_px, float conf, int pass, int stage, int rank, int world, int corr_hint, bool all_counts) {
    const double thr2 = (double)reproj_px * (double)reproj_px;
    if (stage & 1) {
      { ProfScope ps(prof, "k_ransac_hyp", st);
        hipLaunchKernelGGL(k_ransac_hyp, dim3((n_hyp + 63) / 64, 1, nl), dim3(64), 0, st, dl, n_hyp, pass); }
      { ProfScope ps(prof, "k_ransac_score", st);
        // hypotheses per workgroup: ~1/24 of the set, up to 32; correspondence chunks from the host's hint
        const int chunks = std::max(1, (corr_hint + RS_R * 1024 - 1) / (RS_R * 1024));
        // Beyond 256 hypotheses (and unless the hypotheses are sharded over ranks: every rank must then produce all of its counts for the
        // exchange) the scoring runs in two stages: the first 128, a peek at how far the adaptive stop lets the scan go, then the
        // rest -- whose workgroups leave at once when their hypotheses lie beyond that bound.  The result is the full scan's.
        const int h_split = (n_hyp > 256 && world == 1 && !all_counts) ? 128 : n_hyp;     // all_counts: the caller reads every hypothesis' count (vo_pnp_ransac)
        const int ht1 = std::max(1, std::min(32, (h_split + 15) / 24));      // default.yaml's 100 hypotheses: 4 per workgroup -- 19 us and a quarter of the re-read traffic, against 25 us with one each
        hipLaunchKernelGGL(k_ransac_score, dim3((h_split + ht1 - 1) / ht1, chunks, nl), dim3(1024), 0, st, dl, 0, h_split, ht1, thr2, rank, world);
        if (h_split < n_hyp) {
            hipLaunchKernelGGL(k_ransac_peek, dim3(1, 1, nl), dim3(1024), 0, st, dl, h_split, n_hyp, (double)conf);
            const int ht2 = 32;
            hipLaunchKernelGGL(k_ransac_score, dim3((n_hyp - h_split + ht2 - 1) / ht2, chunks, nl), dim3(1024), 0, st, dl, h_split, n_hyp, ht2, thr2, rank, world);
        } }
    }
    if (stage & 2) { ProfScope ps(prof, "k_ransac_select", st);
      hipLaunchKernelGGL(k_ransac_select, dim3(1, 1, nl), dim3(1024), 0, st, dl, n_hyp, thr2, (double)conf); }
    HIP_TRY(hipGetLastError());
    return VO_OK;
}

int vo_track_lm_launch(vo_ctx* prof, hipStream_t st, const LaneDesc* dl, int nl, double delta, double cut, int it_r, int it_p, bool write_flags, int inlier_hint) {
    ProfScope ps(prof, "k_pose_lm", st);
    // workgroups per lane: one up to ~6 k inliers (a pass is then cheaper than a hand-off), one per 3 k beyond; all workgroups of the
    // launch must be resident at once, so the split is only taken while the grid stays far below the chip (2 per compute unit fit)
    const char* const force_s = getenv("VO_LM_WGS");       // tests: force the workgroups per lane (read per launch; tests/test_gpu_parity.py runs the hand-off form at small sizes)
    const int force = force_s ? atoi(force_s) : 0;
    int nwg = force > 0 ? force : (inlier_hint > 6000 ? (inlier_hint + 2999) / 3000 : 1);
    nwg = std::max(1, std::min(std::min(nwg, LM_KMAX), 128 / std::max(1, nl)));
    hipLaunchKernelGGL(k_pose_lm, dim3(nwg, 1, nl), dim3(LM_T), LM_LDS_MAX * 20, st, dl, delta, cut, it_r, it_p, write_flags ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return VO_OK;
}

int vo_map_scatter_launch(vo_ctx* c, int n, const int32_t* d_idx, const double* d_xyz, const double* d_nrm, const uint32_t* d_desc, const int32_t* d_kp, const uint8_t* d_flags) {
    hipLaunchKernelGGL(k_map_scatter, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, d_idx, d_xyz, d_nrm, d_desc, d_kp, d_flags,
                       c->d_map_pos, c->d_map_nrm, c->d_map_desc, c->d_map_flags);
    HIP_TRY(hipGetLastError());
    return VO_OK;
}
