// queue_map -- which HIP streams share a hardware queue?  (developer tool)
// A 300 us spin kernel goes to stream i, an empty kernel to stream j right behind it: if the empty kernel only finishes when the spin does, the
// two streams are served by the same hardware queue.  Prints the sharing classes of 12 default-class streams and of 8 highest-class streams,
// and again after the first four of each have been destroyed and re-created.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ void k_spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
__global__ void k_empty() {}
static double us_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool shares(hipStream_t a, hipStream_t b) {
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, 30000LL);          // 100 MHz wall clock: 300 us
    const double t0 = us_now();
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, b);
    (void)hipStreamSynchronize(b);
    const double dt = us_now() - t0;
    (void)hipDeviceSynchronize();
    return dt > 200.0;
}
static void classes(const char* what, std::vector<hipStream_t>& s) {
    const int n = (int)s.size();
    std::vector<int> cls(n, -1); int nc = 0;
    for (int i = 0; i < n; ++i) {
        if (cls[i] >= 0) continue;
        cls[i] = nc;
        for (int j = i + 1; j < n; ++j) if (cls[j] < 0 && shares(s[i], s[j])) cls[j] = nc;
        ++nc;
    }
    printf("%-58s", what);
    for (int i = 0; i < n; ++i) printf(" %d", cls[i]);
    printf("   (%d queues)\n", nc);
}
int main() {
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); (void)hipDeviceSynchronize(); }
    std::vector<hipStream_t> nrm(12), high(8);
    for (auto& s : nrm) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    classes("12 default-class streams, in creation order:", nrm);
    for (auto& s : high) (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
    classes("8 highest-class streams:", high);
    { std::vector<hipStream_t> mix{nrm[0], nrm[1], nrm[2], nrm[3], high[0], high[1], high[2], high[3]}; classes("default 0..3 + highest 0..3:", mix); }
    for (int i = 0; i < 4; ++i) { (void)hipStreamDestroy(nrm[i]); (void)hipStreamCreateWithFlags(&nrm[i], hipStreamNonBlocking); }
    classes("default class after re-creating streams 0..3:", nrm);
    std::vector<hipStream_t> more(4);
    for (auto& s : more) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    { std::vector<hipStream_t> all(nrm); all.insert(all.end(), more.begin(), more.end()); classes("the 12 + 4 more default-class streams:", all); }
    std::vector<hipStream_t> low(4);
    for (auto& s : low) (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo);
    { std::vector<hipStream_t> mix{nrm[0], nrm[1], nrm[2], nrm[3], low[0], low[1], low[2], low[3]}; classes("default 0..3 + lowest 0..3:", mix); }
    return 0;
}
