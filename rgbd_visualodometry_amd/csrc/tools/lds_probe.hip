// lds_probe.hip -- developer tool: what the building blocks of vo_ba_chol2.h cost on one wave with nothing else running
#include "../vo_ba.hip"
void vo_prof_begin(vo_ctx*, const char*) {}
void vo_prof_end(vo_ctx*) {}
void* vo_stage(vo_ctx*, size_t) { return nullptr; }
int vo_kf_host_pairs(vo_ctx*, int**, int**, int*, int**, double**) { return VO_E_DEVICE; }
int vo_scratch(vo_ctx*, size_t) { return VO_E_DEVICE; }
int vo_prof_begin(vo_ctx*, const char*, hipStream_t) { return -1; }
void vo_prof_end(vo_ctx*, int) {}

__global__ __launch_bounds__(512) void k_probe(double* out, int nw_active) {
    extern __shared__ double s_mem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
    double* s_L = s_mem + 1024;
    for (int i = tid; i < 66 * CH2_TS; i += 512) s_L[i] = 1.0 / (1 + (i % 97));
    __syncthreads();
    if (wave >= nw_active) return;
    const int o_op = CH2_RS * r16 + kq, o_c = CH2_RS * kq + r16;
    long long t[12];
    // (1) dependent LDS round trips: ds_read_b32 chain
    int idx = lane;
    t[0] = clock64();
#pragma unroll
    for (int i = 0; i < 16; ++i) idx = ((volatile int*)s_L)[idx & 1023] & 1023;
    asm volatile("" : "+v"(idx));
    t[1] = clock64();
    // (2) 3 tiles x 1 panel
    { const int ti[3] = {3 + wave, 6, 9}, tj[3] = {1, 1, 1};
      for (int r = 0; r < 8; ++r) ch2_tiles_mp<3>(s_L, ti, tj, 0, 0, o_op, o_c); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[2] = clock64();
    { const int ti[3] = {3 + wave, 6, 9}, tj[3] = {2, 2, 2};
      for (int r = 0; r < 8; ++r) ch2_tiles_mp<3>(s_L, ti, tj, 0, 1, o_op, o_c); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[3] = clock64();
    { const int ti[3] = {9, 0, 0}, tj[3] = {8, 0, 0};
      for (int r = 0; r < 8; ++r) ch2_tiles_mp<1>(s_L, ti, tj, 0, 7, o_op, o_c); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[4] = clock64();
    // (3) DPP solve of 16 rows (ChSolve) incl. loads and stores
    for (int r = 0; r < 8; ++r) {
        double x[CH_NB], Lk[CH_NB];
        double* prow = s_L + ch2_tix(2, 1) * CH2_TS + CH2_RS * r16;
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) { x[c] = prow[c]; Lk[c] = s_mem[c * 16 + r16]; }
        ch_exec_settle(x[0]);
        ChSolve<0>::run(x, Lk);
#pragma unroll
        for (int c = 0; c < CH_NB; ++c) prow[c] = x[c];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[5] = clock64();
    if (lane == 0) { double* o = out + wave * 8; o[0] = (t[1] - t[0]) / 16.0; o[1] = (t[2] - t[1]) / 8.0; o[2] = (t[3] - t[2]) / 8.0; o[3] = (t[4] - t[3]) / 8.0; o[4] = (t[5] - t[4]) / 8.0; }
    out[64 + tid] = idx;
}
int main() {
    double* d; hipMalloc(&d, 8 * 2048);
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    for (int nw : {1, 2, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_probe, dim3(1), dim3(512), 8 * (1024 + 66 * CH2_TS), 0, d, nw); hipDeviceSynchronize(); }
        double h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%d active wave(s): wave 0: dependent ds_read_b32 round trip %.0f clk | 3 tiles x 1 panel %.0f | 3 tiles x 2 panels %.0f | 1 tile x 8 panels %.0f | DPP solve of 16 rows (load, 136 DPP, store) %.0f", nw, h[0], h[1], h[2], h[3], h[4]);
        if (nw > 1) printf("   last wave: %.0f %.0f %.0f %.0f %.0f", h[8 * (nw - 1)], h[8 * (nw - 1) + 1], h[8 * (nw - 1) + 2], h[8 * (nw - 1) + 3], h[8 * (nw - 1) + 4]);
        printf("\n");
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
