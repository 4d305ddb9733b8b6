// How fast can ONE short kernel read a 31-44 MB weight matrix that was read by the previous launch too
// (MALL-warm, L2-cold)?  Floor for the Tacotron2 LSTM GEMV launches.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ w, int64_t n4, float* out) {
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x);
    const int64_t stride = (int64_t)gridDim.x * 256;
    float4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int64_t j = base + i * stride;
        const float4* p = w + (j < n4 ? j : 0);
        if (NT) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
            v[i] = make_float4(t.x, t.y, t.z, t.w);
        } else v[i] = *p;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (s == 12345.678f) out[0] = s;
}

template <int NL, bool NT>
void run(const char* name, const float4* w, int64_t n4, float* out, int blocks_per_cu_hint) {
    const int grid = (int)((n4 + 256LL * NL - 1) / (256LL * NL));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((read_kernel<NL, NT>), dim3(grid), dim3(256), 0, 0, w, n4, out);
    hipEventRecord(a);
    const int it = 200;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((read_kernel<NL, NT>), dim3(grid), dim3(256), 0, 0, w, n4, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s grid %6d  %.2f us/launch  %.2f TB/s\n", name, grid, ms * 1e3 / it, n4 * 16.0 / (ms * 1e-3 / it) / 1e12);
}

int main() {
    for (double mb : {31.5, 44.0, 75.5, 300.0}) {
        const int64_t n4 = (int64_t)(mb * 1e6 / 16);
        float4* w; float* out;
        hipMalloc(&w, n4 * 16); hipMalloc(&out, 4);
        hipMemset(w, 0, n4 * 16);
        printf("---- %.1f MB\n", mb);
        run<1, false>("1 float4/thread", w, n4, out, 0);
        run<2, false>("2 float4/thread", w, n4, out, 0);
        run<4, false>("4 float4/thread", w, n4, out, 0);
        run<8, false>("8 float4/thread", w, n4, out, 0);
        run<16, false>("16 float4/thread", w, n4, out, 0);
        run<4, true>("4 float4/thread nt", w, n4, out, 0);
        run<8, true>("8 float4/thread nt", w, n4, out, 0);
        hipFree(w); hipFree(out);
    }
    return 0;
}
