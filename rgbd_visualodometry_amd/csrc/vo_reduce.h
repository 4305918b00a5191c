// vo_reduce.h -- wavefront-wide (64 lanes) f64 sum on gfx950 with DPP row operations instead of
// ds_bpermute: 6 steps of (2 x v_mov_b32_dpp + v_add_f64), no LDS traffic, result broadcast through
// v_readlane (SGPR).  hipcc lowers __shfl_xor on doubles to 2 ds_bpermute_b32 per step, which made the
// 28-value reductions of the LM kernels LDS-crossbar bound.
#pragma once
#include <hip/hip_runtime.h>

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double vo_dpp_mov_f64(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi2, lo2);
}

// sum over the 64 lanes of the wavefront, returned in every lane
__device__ __forceinline__ double vo_wave_sum_f64(double x) {
    x += vo_dpp_mov_f64<0xB1, 0xF>(x);      // quad_perm [1,0,3,2]
    x += vo_dpp_mov_f64<0x4E, 0xF>(x);      // quad_perm [2,3,0,1]
    x += vo_dpp_mov_f64<0x141, 0xF>(x);     // row_half_mirror
    x += vo_dpp_mov_f64<0x140, 0xF>(x);     // row_mirror: every lane holds its row's sum
    x += vo_dpp_mov_f64<0x142, 0xA>(x);     // row_bcast15 into rows 1 and 3
    x += vo_dpp_mov_f64<0x143, 0xC>(x);     // row_bcast31 into rows 2 and 3: lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

// ---- 32 values at once -------------------------------------------------------------------------------------------
// Reducing k values one by one costs k x 6 butterfly steps.  With 32 values per lane the first two steps can halve the
// register count instead: v_permlane32_swap / v_permlane16_swap (gfx950) exchange lane halves / odd-even 16-lane rows of
// TWO registers, so one add folds a pair of values and each survivor keeps a different value per half / row.  After
// 16 + 8 adds there are 8 registers holding 4 values each (one per DPP row); 4 rotate-and-add steps inside the rows
// finish: 56 adds instead of 192.
__device__ __forceinline__ void vo_swap_halves_f64(double& a, double& b) {     // a lanes 32..63 <-> b lanes 0..31
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void vo_swap_rows_f64(double& a, double& b) {       // a rows 1, 3 <-> b rows 0, 2
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// Index of the value that DPP row `row` (lane >> 4) of out[k] holds after vo_wave_reduce32: 4 k + VO_R32_SLOT(row).
#define VO_R32_SLOT(row) ((((row) & 1) << 1) | ((row) >> 1))
// v[0..31]: per-lane partials.  out[k], k = 0..7: every lane of row r holds the wavefront sum of v[4 k + VO_R32_SLOT(r)].
__device__ __forceinline__ void vo_wave_reduce32(double (&v)[32], double (&out)[8]) {
    double p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { double a = v[2 * i], b = v[2 * i + 1]; vo_swap_halves_f64(a, b); p[i] = a + b; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { double a = p[2 * k], b = p[2 * k + 1]; vo_swap_rows_f64(a, b); out[k] = a + b; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double x = out[k];
        x += vo_dpp_mov_f64<0x128, 0xF>(x);     // row_ror:8
        x += vo_dpp_mov_f64<0x124, 0xF>(x);     // row_ror:4
        x += vo_dpp_mov_f64<0x122, 0xF>(x);     // row_ror:2
        x += vo_dpp_mov_f64<0x121, 0xF>(x);     // row_ror:1
        out[k] = x;
    }
}
// 16 values: v[0..15] per-lane partials; out[k], k = 0..3: every lane of row r holds the wavefront sum of v[4 k + VO_R32_SLOT(r)] (28 adds)
__device__ __forceinline__ void vo_wave_reduce16(double (&v)[16], double (&out)[4]) {
    double p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { double a = v[2 * i], b = v[2 * i + 1]; vo_swap_halves_f64(a, b); p[i] = a + b; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { double a = p[2 * k], b = p[2 * k + 1]; vo_swap_rows_f64(a, b); out[k] = a + b; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double x = out[k];
        x += vo_dpp_mov_f64<0x128, 0xF>(x);
        x += vo_dpp_mov_f64<0x124, 0xF>(x);
        x += vo_dpp_mov_f64<0x122, 0xF>(x);
        x += vo_dpp_mov_f64<0x121, 0xF>(x);
        out[k] = x;
    }
}

// The same first two stages, then the rows are finished TRANSPOSED: instead of four butterflies of four steps (16 move + add pairs), the four
// values fold onto each other -- lane pairs (l, l ^ 1) split them two and two, lane pairs (l, l ^ 2) one and one, two rotations finish -- 5
// adds, 5 moves and 6 selects.  Result: ONE value per lane; lane l of row r holds the wavefront sum of v[4 VO_R16T_K(l) + VO_R32_SLOT(r)].
#define VO_R16T_K(lane) ((((lane) & 1) << 1) | (((lane) >> 1) & 1))
__device__ __forceinline__ double vo_wave_reduce16t(double (&v)[16]) {
    double p[8], o[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { double a = v[2 * i], b = v[2 * i + 1]; vo_swap_halves_f64(a, b); p[i] = a + b; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { double a = p[2 * k], b = p[2 * k + 1]; vo_swap_rows_f64(a, b); o[k] = a + b; }
    const int lane = threadIdx.x & 63;
    const bool odd = lane & 1, hi = lane & 2;
    // even lanes keep o[0], o[1] and receive them from their odd neighbour, odd lanes o[2], o[3]
    const double r0 = (odd ? o[2] : o[0]) + vo_dpp_mov_f64<0xB1, 0xF>(odd ? o[0] : o[2]);      // quad_perm [1,0,3,2]
    const double r1 = (odd ? o[3] : o[1]) + vo_dpp_mov_f64<0xB1, 0xF>(odd ? o[1] : o[3]);
    // lanes 0, 1 of a quad keep r0 (o[0] / o[2]), lanes 2, 3 keep r1 (o[1] / o[3])
    double t = (hi ? r1 : r0) + vo_dpp_mov_f64<0x4E, 0xF>(hi ? r0 : r1);                        // quad_perm [2,3,0,1]
    t += vo_dpp_mov_f64<0x124, 0xF>(t);     // row_ror:4  (lane & 3 is preserved: the four quads of the row)
    t += vo_dpp_mov_f64<0x128, 0xF>(t);     // row_ror:8
    return t;
}

// 32 values, transposed finish (see vo_wave_reduce16t): 8 adds, 8 moves and 14 selects instead of 32 move + add pairs behind the two swap stages.
// ONE value per lane; lane l of row r holds the wavefront sum of v[4 VO_R32T_K(l) + VO_R32_SLOT(r)].
#define VO_R32T_K(lane) ((((lane) & 1) << 2) | ((lane) & 2) | (((lane) >> 2) & 1))
__device__ __forceinline__ double vo_wave_reduce32t(double (&v)[32]) {
    double p[16], o[8], r[4], q[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { double a = v[2 * i], b = v[2 * i + 1]; vo_swap_halves_f64(a, b); p[i] = a + b; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { double a = p[2 * k], b = p[2 * k + 1]; vo_swap_rows_f64(a, b); o[k] = a + b; }
    const int lane = threadIdx.x & 63;
    const bool odd = lane & 1, hi = lane & 2, b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (odd ? o[4 + i] : o[i]) + vo_dpp_mov_f64<0xB1, 0xF>(odd ? o[i] : o[4 + i]);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < 2; ++i) q[i] = (hi ? r[2 + i] : r[i]) + vo_dpp_mov_f64<0x4E, 0xF>(hi ? r[i] : r[2 + i]);        // quad_perm [2,3,0,1]
    double t = (b2 ? q[1] : q[0]) + vo_dpp_mov_f64<0x124, 0xF>(b2 ? q[0] : q[1]);      // row_ror:4: lane l takes from lane l + 4, whose bit 2 is the other one
    t += vo_dpp_mov_f64<0x128, 0xF>(t);     // row_ror:8
    return t;
}

// 1 / sqrt(d): v_rsq_f64 (2^-24) + one cubic correction e (1/2 + 3/8 e), e = 1 - d y^2 -> 2^-52.7 relative error,
// 6 dependent operations instead of the ~25 of an IEEE sqrt followed by an IEEE divide.
__device__ __forceinline__ double vo_rsqrt_f64(double d) {
#pragma clang fp contract(fast)
    const double y = __builtin_amdgcn_rsq(d);
    const double e = 1.0 - (d * y) * y;
    return y + y * (e * (0.5 + 0.375 * e));
}
