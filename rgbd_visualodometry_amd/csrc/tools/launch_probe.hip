// launch_probe: what a kernel launch costs the host thread, and what a chain of tiny dependent kernels costs on the GPU (developer tool)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <atomic>
#include <thread>
struct Big { int v[160]; };
__global__ void k_empty(int* p) { if (p && threadIdx.x == 999999) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 999999) *p = b.v[3]; }
__global__ void k_touch(int* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 1 << 22);
    hipMemset(d, 0, 1 << 22);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d);
        double t1 = now(); hipStreamSynchronize(st); double t2 = now();
        printf("1000 empty kernels, queue kept full: host %.2f us per launch, end to end %.2f us per kernel\n", (t1 - t0) / 1000, (t2 - t0) / 1000);
    }
    Big b; for (int i = 0; i < 160; ++i) b.v[i] = i;
    { double t0 = now(); for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, b, d); double t1 = now(); hipStreamSynchronize(st);
      printf("1000 kernels with 644 bytes of arguments: host %.2f us per launch\n", (t1 - t0) / 1000); }
    // a chain launched into an EMPTY queue (the resident cut's situation): 16 small dependent kernels, then the host waits
    for (int rep = 0; rep < 3; ++rep) {
        double acc = 0;
        for (int it = 0; it < 100; ++it) {
            hipStreamSynchronize(st);
            double t0 = now();
            for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, st, d, 16384);
            hipStreamSynchronize(st);
            acc += now() - t0;
        }
        printf("16 small dependent kernels into an empty queue + wait: %.1f us per chain (%.2f per kernel)\n", acc / 100, acc / 1600);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, st, d, 16384);
    hipStreamEndCapture(st, &g);
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
        for (int rep = 0; rep < 3; ++rep) {
            double acc = 0;
            for (int it = 0; it < 100; ++it) { hipStreamSynchronize(st); double t0 = now(); hipGraphLaunch(ge, st); hipStreamSynchronize(st); acc += now() - t0; }
            printf("the same 16 kernels as a hipGraph + wait: %.1f us per chain\n", acc / 100);
        }
    } else printf("hipGraphInstantiate failed\n");
    // round 6: the same chain with every node's parameters set anew before each launch (a graph whose arguments and grids change per use), alone and
    // beside a second thread that keeps launching into its own stream (the tracker beside the cut)
    {
        hipGraph_t g2; hipGraphExec_t ge2; hipGraphCreate(&g2, 0);
        std::vector<hipGraphNode_t> nodes(16);
        int* dp = d; int nn = 16384;
        void* args[2] = {&dp, &nn};
        hipKernelNodeParams kp{}; kp.func = (void*)k_touch; kp.gridDim = dim3(64); kp.blockDim = dim3(256); kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
        bool ok = true;
        for (int i = 0; i < 16 && ok; ++i) ok = hipGraphAddKernelNode(&nodes[i], g2, i ? &nodes[i - 1] : nullptr, i ? 1 : 0, &kp) == hipSuccess;
        ok = ok && hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0) == hipSuccess;
        if (!ok) { printf("explicit graph failed\n"); return 0; }
        std::atomic<int> stop{0}, go{0};
        auto chains = [&](const char* tag) {
            for (int variant = 0; variant < 3; ++variant) {
                double acc = 0, host = 0;
                for (int it = 0; it < 200; ++it) {
                    hipStreamSynchronize(st);
                    double t0 = now();
                    if (variant == 0) for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(k_touch, dim3(64 + (it & 3)), dim3(256), 0, st, d, 16384);
                    else {
                        if (variant == 2) for (int i = 0; i < 16; ++i) { kp.gridDim = dim3(64 + (it & 3)); nn = 16384 - i; if (hipGraphExecKernelNodeSetParams(ge2, nodes[i], &kp) != hipSuccess) { printf("SetParams failed\n"); return; } }
                        hipGraphLaunch(ge2, st);
                    }
                    double t1 = now();
                    hipStreamSynchronize(st);
                    acc += now() - t0; host += t1 - t0;
                }
                printf("%s: %s: %.1f us per chain of 16 (host %.1f us until the last call returned)\n", tag, variant == 0 ? "plain launches" : variant == 1 ? "graph launch" : "graph, 16 x SetParams + launch", acc / 200, host / 200);
            }
        };
        chains("alone");
        std::thread other([&]() {
            hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
            int* d2; hipMalloc(&d2, 1 << 20);
            go = 1;
            while (!stop.load()) { for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_touch, dim3(8), dim3(256), 0, s2, d2, 2048); hipStreamSynchronize(s2); }
            hipStreamSynchronize(s2);
        });
        while (!go.load()) {}
        chains("beside a launching thread");
        stop = 1; other.join();
    }
    return 0;
}
