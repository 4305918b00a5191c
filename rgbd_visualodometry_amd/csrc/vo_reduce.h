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
