// Host-side launch cost on this box: eager launches vs hipGraph replay of dependent tiny kernels.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(float* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void tiny12(float* a, int b, const float* c, int d, const float* e, float* f, const float* g, const float* h,
                       const float* i, float* j, int k, int l) { if (threadIdx.x == 0 && blockIdx.x == 0) j[0] += 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* p; hipMalloc(&p, 1024); hipMemset(p, 0, 1024);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int mode = 0; mode < 2; ++mode) {
        hipStream_t st = mode ? s : nullptr;
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, p, 1);
        hipStreamSynchronize(st);
        const int N = 20000;
        double t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, p, 1);
        double t1 = now();
        hipStreamSynchronize(st);
        double t2 = now();
        printf("%s stream: eager tiny: host %.2f us/launch, total %.2f us/launch\n", mode ? "side" : "null", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
        t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny12, dim3(256), dim3(256), 0, st, p, 1, p, 2, p, p, p, p, p, p, 3, 4);
        t1 = now();
        hipStreamSynchronize(st);
        t2 = now();
        printf("%s stream: eager 12-arg: host %.2f us/launch, total %.2f us/launch\n", mode ? "side" : "null", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    }
    for (int nodes : {8, 56}) {
        hipGraph_t g; hipGraphExec_t e;
        hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, s, p, 1);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
        for (int i = 0; i < 10; ++i) hipGraphLaunch(e, s);
        hipStreamSynchronize(s);
        const int N = 500;
        double t0 = now();
        for (int i = 0; i < N; ++i) hipGraphLaunch(e, s);
        double t1 = now();
        hipStreamSynchronize(s);
        double t2 = now();
        printf("graph of %d tiny kernels: host %.1f us/replay (%.2f per node), total %.1f us/replay (%.2f per node)\n", nodes,
               (t1 - t0) / N * 1e6, (t1 - t0) / N * 1e6 / nodes, (t2 - t0) / N * 1e6, (t2 - t0) / N * 1e6 / nodes);
    }
    return 0;
}
