// Does hipExtLaunchKernel's hipExtAnyOrderLaunch let a kernel start while its predecessor IN THE SAME STREAM is still running (no barrier between the
// two packets)?  Kernel A spins (bounded) on a word that only kernel B sets; B is launched behind A with the flag.  (Developer probe, not part of the product.)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_a(int* flag, long long* out) {
    const long long t0 = wall_clock64();
    int seen = 0;
    for (int i = 0; i < (1 << 22) && !seen; ++i) { seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (!seen) __builtin_amdgcn_s_sleep(8); }
    out[0] = seen; out[1] = wall_clock64() - t0;
}
__global__ void k_b(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main() {
    int* flag; long long* out; long long h[2];
    hipMalloc(&flag, 4); hipMalloc(&out, 16);
    hipStream_t st; hipStreamCreate(&st);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemsetAsync(flag, 0, 4, st); hipStreamSynchronize(st);
        hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, st, flag, out);
        hipExtLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, st, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, flag);
        hipError_t e = hipStreamSynchronize(st);
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("flags %d: A saw B's word while running: %lld (A ran %.1f us) [%s]\n", mode, h[0], h[1] * 0.01, hipGetErrorString(e));
    }
    return 0;
}
