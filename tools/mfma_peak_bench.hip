// What the matrix pipe of this chip sustains with nothing else going on: every SIMD of every CU issues back-to-back independent MFMAs on
// register operands (no LDS, no memory), long enough for the power management to settle.  Prints TFLOP/s and the shader clock the run
// settled at (clock64 vs the 100 MHz wall clock), for v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32, with random operands (the same
// pair for every MFMA, or 32 pairs in rotation) and with zeros: switching activity, hence power, depends on the data.  The guide's peaks (2.5 PFLOP/s bf16, 157.3 TFLOP/s fp32) assume 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_peak_bench.hip -o tools/bin/mfma_peak_bench
// Run:   mfma_peak_bench [waves_per_simd = 1] [ms_per_case = 400]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 h8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f16v;

// ROT: every MFMA takes another (A, B) register pair than the one before it (4 A fragments x 8 B fragments in registers), as a
// kernel that streams operands does; without it the same two fragments feed every MFMA and the operand buses never toggle
template <bool BF16, bool ROT>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
    const int tid = threadIdx.x;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    uint4 a = in[tid], b = in[256 + tid];
    uint4 av[4], bv[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) av[u] = in[(tid + 37 * u) & 511];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = in[(tid + 61 * j + 256) & 511];
    f16v acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint4 aa = ROT ? av[u] : a, bb = ROT ? bv[j] : b;
                if constexpr (BF16)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(h8, aa), __builtin_bit_cast(h8, bb), acc[j], 0, 0, 0);
                else
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, aa.x), __builtin_bit_cast(float, bb.x), acc[j], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <bool BF16, bool ROT>
static void run(const char* name, const uint4* in, float* out, unsigned long long* clk, int blocks, double target_ms) {
    const double flop_per_mfma = BF16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {                      // calibrate, settle, measure
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<BF16, ROT>), dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) iters = (int)(iters * target_ms / ms) + 1;
    }
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 32 * flop_per_mfma;
    printf("%-44s %8.1f TFLOP/s   %.2f GHz   (%.0f ms, %d blocks of 4 waves)\n", name, flops / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0), ms, blocks);
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const double target_ms = argc > 2 ? atof(argv[2]) : 400.0;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * wps;
    std::vector<uint16_t> h(512 * 8);
    uint4 *rnd, *zero; float* out; unsigned long long* clk;
    hipMalloc(&rnd, 512 * 16); hipMalloc(&zero, 512 * 16); hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 16);
    srand(1);
    for (auto& v : h) { float f = ((float)rand() / RAND_MAX - 0.5f) * 2.f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    hipMemcpy(rnd, h.data(), 512 * 16, hipMemcpyHostToDevice);
    hipMemset(zero, 0, 512 * 16);
    printf("%d CUs, %d wave(s) per SIMD\n", prop.multiProcessorCount, wps);
    run<true, true>("v_mfma_f32_32x32x16_bf16, random, 32 operand pairs", rnd, out, clk, blocks, target_ms);
    run<true, false>("v_mfma_f32_32x32x16_bf16, random, one operand pair", rnd, out, clk, blocks, target_ms);
    run<true, false>("v_mfma_f32_32x32x16_bf16, zero operands", zero, out, clk, blocks, target_ms);
    run<false, true>("v_mfma_f32_32x32x2_f32, random, 32 operand pairs", rnd, out, clk, blocks, target_ms);
    run<false, false>("v_mfma_f32_32x32x2_f32, random, one operand pair", rnd, out, clk, blocks, target_ms);
    run<false, false>("v_mfma_f32_32x32x2_f32, zero operands", zero, out, clk, blocks, target_ms);
    return 0;
}
