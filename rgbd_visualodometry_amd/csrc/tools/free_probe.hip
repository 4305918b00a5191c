// Does hipFree wait for kernels that are running on OTHER streams?  A kernel spins ~100 ms on stream 1 (it touches nothing but a clock); the host frees an unrelated
// buffer meanwhile and times the call.  (Developer probe: decides whether a buffer may be freed while kernels that read it are still queued -- vo_ba_engine_drain.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_spin(long long ticks, long long* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64); out[0] = 1; }
int main() {
    long long* out; void* x;
    hipMalloc(&out, 8); hipMalloc(&x, 64 << 20);
    hipStream_t s1; hipStreamCreate(&s1);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 10000000ll, out);      // 100 ms of 10 ns ticks
    const auto t0 = std::chrono::steady_clock::now();
    hipFree(x);
    const double ms_free = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hipStreamSynchronize(s1);
    const double ms_all = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("hipFree of an unrelated 64 MB buffer while another stream's kernel runs for 100 ms: %.2f ms (the kernel was done after %.2f ms)\n", ms_free, ms_all);
    return 0;
}
