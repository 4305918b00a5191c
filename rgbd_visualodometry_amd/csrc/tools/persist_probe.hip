// persist_probe -- what does a PERSISTENT one-workgroup solver on a second stream buy a chain of LM steps?  (developer tool)
// A model of the local BA's step with the durations of the bench workload and no arithmetic:
//   mode 0  today's two launches per step: S (1275 workgroups of ~4.5 us, "Schur slices") then CU (workgroup 0 spins 38 us and raises a word,
//           170 workgroups wait for it and then work 12 us: "k_ba_cholup")
//   mode 1  the solver lives through all steps in a launch of its own on a second stream: the slices of S raise one of ten column counters when
//           they are done, the solver starts a step when the first three columns are complete and needs column k + 2 before stage k (ten
//           stages of 3.8 us), then publishes a tagged word; the step's second launch on the first stream is only the 170 waiting workgroups.
// Prints us per step for both, and whether the two streams really ran side by side (the persistent form deadlocks -- bounded -- otherwise).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ void spin_us(double us) { const long long t0 = wall_clock64(), n = (long long)(us * 100.0); while (wall_clock64() - t0 < n) __builtin_amdgcn_s_sleep(1); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define CSTR 64      // ints between two column counters: 256 bytes, so that they live in different memory channels (adjacent counters: +7 us per step)
struct Words { int col[16 * CSTR]; int tag; int abort_; int pad[14]; };

__global__ __launch_bounds__(256) void k_S(Words* w, int G, int ncol, double us, int counted) {
    spin_us(us);
    __syncthreads();
    if (counted && threadIdx.x == 0) { const int c = (int)(((long long)blockIdx.x * ncol) / G); __hip_atomic_fetch_add(&w->col[c * CSTR], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
// mode 0: workgroup 0 = solver, the others wait for its word; mode 1 (solver < 0): everybody waits
__global__ __launch_bounds__(512) void k_CU(Words* w, int step, double us_solve, double us_tail, int has_solver) {
    __shared__ int seen;
    if (has_solver && blockIdx.x == 0) {
        spin_us(us_solve);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&w->tag, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < (1 << 20) && !s; ++i) { s = ld_agent(&w->tag) >= step + 1; if (!s) __builtin_amdgcn_s_sleep(2); }
        seen = s;
    }
    __syncthreads();
    if (seen) spin_us(us_tail);
}
__global__ __launch_bounds__(512) void k_P(Words* w, int nsteps, int G, int ncol, double us_stage, double us_back, int overlap) {
    // wave 1 = the loader: polls the column counters (global memory, agent scope) ahead of the stages and raises a word in LDS; the other waves
    // only ever wait for that word (an LDS poll costs ~100 clocks, a global one 1-2 us)
    __shared__ volatile int colrdy, bad;
    if (threadIdx.x == 0) { colrdy = 0; bad = 0; }
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if (wave == 1) {
        if (threadIdx.x == 64) {
            for (int s = 0; s < nsteps && !bad; ++s)
                for (int c = 0; c < ncol && !bad; ++c) {
                    const int per = (int)(((long long)(c + 1) * G + ncol - 1) / ncol) - (int)(((long long)c * G + ncol - 1) / ncol);
                    const int want = (s + 1) * per;
                    const long long t0 = wall_clock64();
                    while (ld_agent(&w->col[c * CSTR]) < want) { __builtin_amdgcn_s_sleep(1); if (wall_clock64() - t0 > 2000000) { bad = 1; break; } }      // 20 ms
                    colrdy = s * ncol + c + 1;
                }
            if (bad) { w->abort_ = 1; __hip_atomic_store(&w->tag, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        return;
    }
    for (int s = 0; s < nsteps; ++s) {
        for (int k = 0; k < ncol; ++k) {
            const int need = s * ncol + ((overlap && k + 3 < ncol) ? k + 3 : ncol);
            while (colrdy < need && !bad) __builtin_amdgcn_s_sleep(1);
            if (bad) return;
            spin_us(us_stage);
        }
        spin_us(us_back);
        if (threadIdx.x == 0) __hip_atomic_store(&w->tag, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_idle(double us) { spin_us(us); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int nsteps = 20, G = 1275, ncol = 10, reps = argc > 1 ? atoi(argv[1]) : 30;
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t a, b; CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
    Words* w; CK(hipMalloc((void**)&w, sizeof(Words)));
    CK(hipFuncSetAttribute((const void*)k_S, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024));
    CK(hipFuncSetAttribute((const void*)k_P, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    CK(hipFuncSetAttribute((const void*)k_CU, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    // G per column must match k_S's mapping: column c gets the workgroups with floor(b * ncol / G) == c
    for (int mode = 0; mode < 5; ++mode) {
        double best = 1e30, sum = 0;
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipMemsetAsync(w, 0, sizeof(Words), a)); CK(hipStreamSynchronize(a));
            const double t0 = now_us();
            if (mode == 3) hipLaunchKernelGGL(k_idle, dim3(1), dim3(512), 0, b, 1400.0);
            if (mode == 1 || mode == 2) hipLaunchKernelGGL(k_P, dim3(1), dim3(512), 140 * 1024, b, w, nsteps, G, ncol, 3.8, 4.5, mode == 1);
            for (int s = 0; s < nsteps; ++s) {
                hipLaunchKernelGGL(k_S, dim3(G), dim3(256), 60 * 1024, a, w, G, ncol, 4.5, mode == 1 || mode == 2 || mode == 4);
                if (mode == 0 || mode >= 3) hipLaunchKernelGGL(k_CU, dim3(171), dim3(512), 140 * 1024, a, w, s, 38.0 + 4.5, 12.0, 1);
                else hipLaunchKernelGGL(k_CU, dim3(170), dim3(512), 16 * 1024, a, w, s, 0.0, 12.0, 0);
            }
            CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
            const double dt = now_us() - t0;
            if (r >= 3) { sum += dt; if (dt < best) best = dt; }
        }
        Words h; CK(hipMemcpy(&h, w, sizeof(h), hipMemcpyDeviceToHost));
        printf("mode %d (%s): %.1f us per step (best %.1f), 20 steps; abort %d\n", mode, mode == 0 ? "two launches per step" : (mode == 1 ? "persistent solver on a second stream, columns streamed" : (mode == 2 ? "persistent solver, waits for the whole of S" : (mode == 3 ? "two launches per step beside an idle 1.4 ms kernel on the other stream" : "two launches per step, slices count"))), sum / reps / nsteps, best / nsteps, h.abort_);
    }
    return 0;
}
