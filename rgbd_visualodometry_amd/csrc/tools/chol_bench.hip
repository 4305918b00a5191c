// chol_bench.hip -- developer tool (not part of the product library): runs the reduced-system Cholesky kernels of
// vo_ba.hip on random SPD systems, checks the solution against a host Cholesky and times them with HIP events.
//   build:  make tools        run on the GPU box:  rgbd_visualodometry_amd/csrc/build/chol_bench
// build with -DCH_STAMPS for the per-phase clock stamps (they perturb the timing: every stamp drains the LDS queue)
#include "../vo_ba.hip"

#include <random>

// context helpers of vo_capi.hip that vo_ba_run references; the tool never calls vo_ba_run
int vo_trace_level(void) { return 0; }
void vo_prof_begin(vo_ctx*, const char*) {}
void vo_prof_end(vo_ctx*) {}
void* vo_stage(vo_ctx*, size_t) { return nullptr; }
int vo_kf_host_pairs(vo_ctx*, int**, int**, int*, int**, double**) { return VO_E_DEVICE; }
int vo_scratch(vo_ctx*, size_t) { return VO_E_DEVICE; }
int vo_prof_begin(vo_ctx*, const char*, hipStream_t) { return -1; }
void vo_prof_end(vo_ctx*, int) {}

static void host_solve(int D, std::vector<double> S, std::vector<double> b, std::vector<double>& x) {
    for (int j = 0; j < D; ++j) {
        double d = S[j * D + j];
        for (int k = 0; k < j; ++k) d -= S[j * D + k] * S[j * D + k];
        d = std::sqrt(d); S[j * D + j] = d;
        for (int i = j + 1; i < D; ++i) { double s = S[i * D + j]; for (int k = 0; k < j; ++k) s -= S[i * D + k] * S[j * D + k]; S[i * D + j] = s / d; }
    }
    for (int i = 0; i < D; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= S[i * D + k] * b[k]; b[i] = s / S[i * D + i]; }
    for (int i = D - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < D; ++k) s -= S[k * D + i] * b[k]; b[i] = s / S[i * D + i]; }
    x = b;
}

__global__ void k_dpp_probe(double* out) {
    const int lane = threadIdx.x;
    double v = 100.0 + lane;
    double m = 2.0;
    asm volatile("s_nop 4");
    const double b = ch_bcast<3>(v);
    double acc = 1000.0;
    double vv = ch_mul_for_dpp(v, 1.0);
    ch_fnma_bcast<5>(acc, vv, m);                 // 1000 - v[lane 5 of row] * 2
    out[lane] = b; out[64 + lane] = acc;
}

__global__ void k_spin(long long clocks, int* sink) {
    const long long t0 = clock64();
    while (clock64() - t0 < clocks) { }
    if (clocks < 0) *sink = 1;
}

__global__ void k_reduce32_probe(double* out) {
    const int lane = threadIdx.x;
    double v[32], r[8];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = (double)((lane * 37 + i * 101) % 1009) + 0.25 * i;
    vo_wave_reduce32(v, r);
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k * 64 + lane] = r[k];
}

__global__ void k_lat_probe(double* out, double seed) {
    double x = seed + threadIdx.x, y = 1.0000001, z = 0.5;
    long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 256; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    long long t1 = clock64();
#pragma unroll
    for (int i = 0; i < 128; ++i) asm volatile("v_mul_f64 %0, %0, %1\n\ts_nop 1\n\tv_fmac_f64_dpp %0, -%0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(z));
    long long t2 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
    long long t3 = clock64();
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y), "v"(z));
    long long t4 = clock64();
#pragma unroll
    for (int i = 0; i < 128; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(reinterpret_cast<int&>(y)) : "v"(i));
    long long t5 = clock64();
    double b0 = x, b1 = x + 1, b2 = x + 2, b3 = x + 3;
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("v_fmac_f64_dpp %0, -%4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, -%4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, -%4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(y), "v"(z));
    long long t6 = clock64();
    if (threadIdx.x == 0) out[5] = (t6 - t5) / 256.0;
    a0 += b0 + b1 + b2 + b3;
    if (threadIdx.x == 0) { out[0] = (t1 - t0) / 256.0; out[1] = (t2 - t1) / 128.0; out[2] = (t3 - t2) / 64.0; out[3] = (t4 - t3) / 256.0; out[4] = (t5 - t4) / 128.0; }
    out[8 + threadIdx.x] = x + a0 + a1 + a2 + a3 + y;
}

__global__ void k_rsq_probe(const double* x, double* y, int n) {
#pragma clang fp contract(fast)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = 1.0 - (d * y0) * y0;
    const double y1 = y0 + y0 * (0.5 * e);                                  // one Newton step
    const double yc = y0 + y0 * (e * (0.5 + 0.375 * e));                    // one cubic (Halley-like) step
    y[i] = y0; y[n + i] = y1; y[2 * n + i] = yc; { double yy, qq; ba_rsqrt_parts(d, yy, qq); y[3 * n + i] = yy + yy * qq; }
}

int main(int argc, char** argv) {
    {   // do small kernels on different streams overlap?  N streams x 20 spin kernels of ~50 us (1 workgroup of 256 threads each)
        int* sink; hipMalloc(&sink, 4);
        for (int ns : {1, 2, 4, 8, 16}) {
            std::vector<hipStream_t> sts(ns);
            for (auto& x : sts) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < 20; ++it) for (int i = 0; i < ns; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, sts[i], 120000LL, sink);
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("concurrency probe: %2d streams x 20 kernels of ~50 us: %.0f us wall (serial would be %.0f)\n", ns, us, 50.0 * 20 * ns);
            for (auto& x : sts) hipStreamDestroy(x);
        }
        hipFree(sink);
    }
    { double* d; hipMalloc(&d, 512 * 8); hipLaunchKernelGGL(k_reduce32_probe, dim3(1), dim3(64), 0, 0, d); std::vector<double> h(512); hipMemcpy(h.data(), d, 512 * 8, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int k = 0; k < 8; ++k) for (int lane = 0; lane < 64; ++lane) {
          const int i = 4 * k + VO_R32_SLOT(lane >> 4);
          double ex = 0; for (int l = 0; l < 64; ++l) ex += (double)((l * 37 + i * 101) % 1009) + 0.25 * i;
          if (h[k * 64 + lane] != ex) { if (bad < 4) printf("reduce32 probe: out[%d] lane %d = %.2f, expected %.2f\n", k, lane, h[k * 64 + lane], ex); ++bad; }
      }
      printf("reduce32 probe: %d mismatches\n", bad); hipFree(d); }
    { double* d; hipMalloc(&d, 128 * 8); hipLaunchKernelGGL(k_lat_probe, dim3(1), dim3(64), 0, 0, d, 1.0); double h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("latency probe (clocks): dependent v_fma_f64 %.1f | mul+nop+fmac_dpp pair %.1f | dependent v_rsq_f64 %.1f | independent v_fma_f64 %.1f | dependent v_cndmask_b32 %.1f | independent v_fmac_f64_dpp %.1f\n", h[0], h[1], h[2], h[3], h[4], h[5]); hipFree(d); }
    { const int n = 1 << 16; std::vector<double> hx(n), hy(4 * n); std::mt19937_64 rg(7); std::uniform_real_distribution<double> ud(-30.0, 30.0);
      for (auto& v : hx) v = std::exp2(ud(rg));
      double *dx, *dy; hipMalloc(&dx, n * 8); hipMalloc(&dy, 4 * n * 8); hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_rsq_probe, dim3(n / 256), dim3(256), 0, 0, dx, dy, n); hipMemcpy(hy.data(), dy, 4 * n * 8, hipMemcpyDeviceToHost);
      const char* nm[4] = {"v_rsq_f64", "+1 newton", "+1 cubic", "ba_rsqrt_parts"};
      for (int v = 0; v < 4; ++v) { long double mx = 0; for (int i = 0; i < n; ++i) { long double ex = 1.0L / sqrtl((long double)hx[i]); long double er = fabsl((long double)hy[v * n + i] - ex) / ex; if (er > mx) mx = er; }
          printf("rsq probe %-20s max rel err %.3Le (2^%.1Lf)\n", nm[v], mx, log2l(mx)); }
      hipFree(dx); hipFree(dy); }

    { double* d; hipMalloc(&d, 128 * 8); hipLaunchKernelGGL(k_dpp_probe, dim3(1), dim3(64), 0, 0, d); double h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("bcast<3>: "); for (int i = 0; i < 64; i += 5) printf("[%d]=%.0f ", i, h[i]); printf("\nfnma<5>: "); for (int i = 0; i < 64; i += 5) printf("[%d]=%.0f ", i, h[64 + i]); printf("\n"); hipFree(d); }

    std::vector<int> sizes = {6, 12, 18, 24, 30, 48, 60, 96, 120, 126, 132, 138, 144, 156, 168, 174, 180, 186, 192, 198, 216};
    if (argc > 1 && !strcmp(argv[1], "big")) sizes = {216, 366, 480, 510, 516, 540, 600, 720, 900, 960};      // k_ba_chol16g up to the 160 free keyframes the ABI admits
    hipFuncSetAttribute((const void*)k_ba_chol16, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void*)k_ba_chol16g, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void*)k_ba_chol16v2, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);

    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int bad = 0;
    for (int D : sizes) {
        std::mt19937_64 rng(1234 + D);
        std::normal_distribution<double> nd(0.0, 1.0);
        std::vector<double> M((size_t)D * D), S((size_t)D * D), b(D), x;
        for (auto& v : M) v = nd(rng);
        for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) { double s = 0; for (int k = 0; k < D; ++k) s += M[i * D + k] * M[j * D + k]; S[i * D + j] = s * 1e3 + (i == j ? 1e3 * D : 0.0); }
        for (auto& v : b) v = nd(rng) * 1e2;
        host_solve(D, S, b, x);
        BaDev B; memset(&B, 0, sizeof(B));
        B.D = D;
        double* d_b0; double* d_S0;
        hipMalloc(&B.S, sizeof(double) * D * D); hipMalloc(&B.bs, sizeof(double) * D); hipMalloc(&d_b0, sizeof(double) * D);
        hipMalloc(&B.Hpp, sizeof(double) * 36 * (D / 6 + 1)); hipMemset(B.Hpp, 0, sizeof(double) * 36 * (D / 6 + 1));      // the kernel adds blockdiag(H_pp) + lambda I, b_p on load
        hipMalloc(&B.bp, sizeof(double) * D); hipMemset(B.bp, 0, sizeof(double) * D);
        hipMalloc(&B.scal, 64); hipMalloc(&B.dl, 8 * (D + 16)); hipMemset(B.dl, 0, 8 * (D + 16)); hipMalloc(&B.ctl, sizeof(BaCtl));
        hipMemset(B.scal, 0, 64); hipMemset(B.ctl, 0, sizeof(BaCtl));
        hipMalloc(&d_S0, sizeof(double) * D * D);
#if defined(CH2_DEBUG) || defined(CH2_STAMPS)
        hipMalloc(&B.W, sizeof(double) * ((D + 2) * (D + 2) + 8192)); hipMemset(B.W, 0, sizeof(double) * ((D + 2) * (D + 2) + 8192));
#endif
        {   // k_ba_chol16 takes the packed lower triangle (ba_tri), k_ba_chol16g the full matrix
            std::vector<double> Sp((size_t)D * D, 0.0);
            if (D <= 192) { for (int r = 0; r < D; ++r) for (int c = 0; c <= r; ++c) Sp[(size_t)r * (r + 1) / 2 + c] = S[(size_t)r * D + c]; }
            else Sp = S;
            hipMemcpy(d_S0, Sp.data(), sizeof(double) * D * D, hipMemcpyHostToDevice);
        }
        double* d_S0t = nullptr; const size_t nt_dbl = ba_tile_doubles(D) + 128;      // the same lower triangle as 16x16 tiles (vo_ba_chol2.h)
        {
            std::vector<double> St(nt_dbl, 0.0);
            for (int r = 0; r < D; ++r) for (int c = 0; c <= r; ++c) St[ba_tile_idx(r, c)] = S[(size_t)r * D + c];
            hipMalloc(&d_S0t, sizeof(double) * nt_dbl); hipMemcpy(d_S0t, St.data(), sizeof(double) * nt_dbl, hipMemcpyHostToDevice);
        }
        double* S_packed = B.S; double* S_tiles; hipMalloc(&S_tiles, sizeof(double) * nt_dbl);
        hipMemcpy(d_b0, b.data(), sizeof(double) * D, hipMemcpyHostToDevice);
        BaDev* d_B; hipMalloc(&d_B, sizeof(BaDev)); hipMemcpy(d_B, &B, sizeof(BaDev), hipMemcpyHostToDevice);
        BaBatch Q; memset(&Q, 0, sizeof(Q)); Q.Bs = d_B; Q.ctls = B.ctl; Q.n = 1;
        for (int variant = 1; variant < (ch2_fits(D) ? 3 : 2); ++variant) {          // 1: first generation (barrier phases, packed rows), 2: second generation (roles + LDS words, tiles: vo_ba_chol2.h)
            B.s_tiles = variant == 2; B.S = variant == 2 ? S_tiles : S_packed;
            hipMemcpy(d_B, &B, sizeof(BaDev), hipMemcpyHostToDevice);
            float tot = 0; const int reps = 50;
            for (int it = 0; it < reps + 5; ++it) {
                hipMemcpyAsync(B.bs, d_b0, sizeof(double) * D, hipMemcpyDeviceToDevice, st);
                if (variant == 2) hipMemcpyAsync(B.S, d_S0t, sizeof(double) * nt_dbl, hipMemcpyDeviceToDevice, st);
                else hipMemcpyAsync(B.S, d_S0, sizeof(double) * D * D, hipMemcpyDeviceToDevice, st);      // the global-resident kernels factor S in place
                hipEventRecord(e0, st);
                if (variant == 2) hipLaunchKernelGGL(k_ba_chol16v2, dim3(1), dim3(CH2_T), ch2_lds_bytes(D), st, Q);
                else if (D <= 192) hipLaunchKernelGGL(k_ba_chol16, dim3(1), dim3(CH_THREADS), sizeof(double) * (CH_NB * CH_NB + (size_t)(D + 1) * (D + 2) / 2 + 2 * (size_t)D), st, Q, 0, 0);
                else hipLaunchKernelGGL(k_ba_chol16g, dim3(1), dim3(CH_THREADS), sizeof(double) * (CH_NB * CH_NB + (size_t)(CH_NB + 1) * (D + 1) + 2 * (size_t)D), st, Q);
                hipEventRecord(e1, st);
                hipStreamSynchronize(st);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 5) tot += ms;
            }
            std::vector<double> got(D); double ok;
            hipMemcpy(got.data(), variant == 2 ? B.dl : B.bs, sizeof(double) * D, hipMemcpyDeviceToHost);
            hipMemcpy(&ok, B.scal + 3, 8, hipMemcpyDeviceToHost);
            double err = 0, ref = 0;
            for (int i = 0; i < D; ++i) { err = std::max(err, std::fabs(got[i] - x[i])); ref = std::max(ref, std::fabs(x[i])); }
            printf("D %3d  %-13s  %8.2f us  ok %.0f  max|dx|/max|x| %.3e%s\n", D, variant == 2 ? "k_ba_chol16v2" : (D <= 192 ? "k_ba_chol16" : "k_ba_chol16g"), 1e3 * tot / reps, ok, err / ref,
                   (ok == 1.0 && err / ref < 1e-10) ? "" : "   <-- MISMATCH");
            if (!(ok == 1.0 && err / ref < 1e-10)) ++bad;
#ifdef CH2_DEBUG
            if (variant == 2) {                                  // the augmented factor [L 0; y^T .] against the host's
                const int DA = D + 1;
                const int Tt = (DA + 15) / 16;
                std::vector<double> Lh((size_t)DA * DA, 0.0), Ld((size_t)Tt * (Tt + 1) / 2 * CH2_TS);
                for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) Lh[(size_t)i * DA + j] = S[(size_t)i * D + j];
                for (int j = 0; j < D; ++j) Lh[(size_t)D * DA + j] = b[j];
                for (int j = 0; j < D; ++j) {
                    double d = Lh[(size_t)j * DA + j]; for (int k = 0; k < j; ++k) d -= Lh[(size_t)j * DA + k] * Lh[(size_t)j * DA + k];
                    d = std::sqrt(d); Lh[(size_t)j * DA + j] = d;
                    for (int i = j + 1; i < DA; ++i) { double s = Lh[(size_t)i * DA + j]; for (int k = 0; k < j; ++k) s -= Lh[(size_t)i * DA + k] * Lh[(size_t)j * DA + k]; Lh[(size_t)i * DA + j] = s / d; }
                }
                hipMemcpy(Ld.data(), B.W, sizeof(double) * Ld.size(), hipMemcpyDeviceToHost);
                int nbad = 0; int tb[16][16]; memset(tb, 0, sizeof(tb));
                for (int i = 0; i < DA; ++i) for (int j = 0; j <= i && j < D; ++j) {
                    if (i / 16 == j / 16 && i < D) continue;   // the diagonal tiles hold W_k = L_kk^-T at the end, not L_kk (the solution check covers them)
                    const double g = Ld[ch2_sidx(i, j)], h = Lh[(size_t)i * DA + j];
                    if (!(std::fabs(g - h) <= 1e-9 * (1.0 + std::fabs(h)))) { if (nbad < 6) printf("        L[%d][%d] = %.6e, host %.6e (tile %d,%d)\n", i, j, g, h, i / 16, j / 16); ++nbad; tb[i / 16][j / 16]++; }
                }
                if (nbad) { printf("        %d factor entries differ; per tile (row block: counts per column block):\n", nbad);
                    for (int i = 0; i * 16 < DA; ++i) { printf("          %2d:", i); for (int j = 0; j <= i; ++j) printf(" %3d", tb[i][j]); printf("\n"); } }
            }
#endif
#ifdef CH2_STAMPS
            if (variant == 2 && (D == 144 || D == 48)) {
                std::vector<long long> ts(8 * 128);
                hipMemcpy(ts.data(), B.W, sizeof(long long) * ts.size(), hipMemcpyDeviceToHost);
                const long long t0 = ts[127];
                auto T = [&](int w, int s) { return (double)(ts[w * 128 + s] - t0); };
                const int nblk = (D + 15) / 16;
                printf("        v2 timeline (clocks since the first barrier), D = %d\n", D);
                printf("        wave 0: "); for (int k = 0; k < nblk; ++k) printf("[k%d potrf %.0f..%.0f rdy-wait from %.0f] ", k, T(0, 3 * k + 1), T(0, 3 * k + 2), T(0, 3 * k + 3)); printf("| bwd wait %.0f..%.0f end %.0f\n", T(0, 60), T(0, 61), T(0, 62));
                printf("        wave 0 rdy seen:      "); for (int k = 0; k < 8; ++k) printf("k%d %.0f  ", k, T(0, 80 + k)); printf("\n");
                printf("        wave 1 W_k published: "); for (int k = 0; k < 9; ++k) printf("k%d %.0f  ", k, T(1, k)); printf("\n");
                printf("        wave 0 inv seen:      "); for (int k = 0; k < 8; ++k) printf("k%d %.0f  ", k, T(0, 90 + k)); printf("\n");
                for (int w : {4, 5}) { printf("        feeder %d: ", w); for (int k = w - 4; k < nblk; k += 2) printf("[k%d %.0f pre ..%.0f inputs ..%.0f rdy %.0f] ", k, T(w, 4 * k), T(w, 4 * k + 1), T(w, 4 * k + 2), T(w, 4 * k + 3)); printf("\n"); }
                for (int w : {2, 3, 6, 7}) { printf("        wave %d: init %.0f ", w, T(w, 126)); for (int k = 0; k < nblk; ++k) printf("[k%d %.0f A ..%.0f D,W ..%.0f C ..%.0f] ", k, T(w, 6 * k), T(w, 6 * k + 1), T(w, 6 * k + 2), T(w, 6 * k + 3)); printf("\n"); }
            }
#endif
#ifdef CH_STAMPS
            if (variant == 1 && D <= 192) { double tt[12]; hipMemcpy(tt, B.dl, 96, hipMemcpyDeviceToHost);
                printf("        clocks: load+block0 %.0f | solve+barrier %.0f | wave0: tile00 %.0f, load rows %.0f, factor %.0f, barrier wait %.0f | bwd: loads %.0f, push+chain %.0f, barrier %.0f, head/tail %.0f+%.0f\n",
                       tt[0], tt[1], tt[5], tt[6], tt[7], tt[2], tt[4], tt[8], tt[9], tt[3], tt[10]); }
#endif
        }
        hipFree(B.Hpp); hipFree(B.bp); hipFree(d_B); hipFree(d_S0); hipFree(d_S0t); hipFree(S_tiles); hipFree(S_packed); hipFree(B.bs); hipFree(d_b0); hipFree(B.scal); hipFree(B.ctl);
    }
    hipError_t e = hipGetLastError();
    printf("last error: %s, mismatches: %d\n", hipGetErrorString(e), bad);
    return bad != 0;
}
