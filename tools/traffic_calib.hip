// Known-bytes kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access patterns of the conv
// engine (profiles/README.md "traffic calibration").  Every kernel moves exactly N = 1 GiB (well past the 256 MB
// Infinity Cache) once:
//   calib_read16   float4 per lane, streaming                         (weight staging, row epilogue residual reads)
//   calib_read4    one dword per lane from 4 rows two channels apart  (activation staging: TTS_LOAD_JOB in conv_mfma.hip)
//   calib_write16  float4 per lane                                    (row epilogue stores)
//   calib_write4s  one dword per lane at stride 8 dwords              (polyphase upsampler epilogue, y_ts = 8)
// Build: hipcc --offload-arch=gfx950 -O3 tools/traffic_calib.hip -o tools/bin/traffic_calib
// Run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace ...  and again with WRITE_SIZE; profiles/traffic_calib.py divides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t N_BYTES = (size_t)1 << 30;

__global__ __launch_bounds__(256) void calib_read16(const float4* __restrict__ src, float* __restrict__ sink, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345e-30f) sink[0] = acc;     // never true: keeps the loads alive
}

// rows of `row` floats; a block reads 256 consecutive columns of 8 rows (4 per kk half: rows c, c+2, c+4, c+6), like the
// activation staging of one octet
__global__ __launch_bounds__(256) void calib_read4(const float* __restrict__ src, float* __restrict__ sink, int row, int n_oct) {
    const int col0 = blockIdx.x * 128;
    const int t = threadIdx.x & 127, kk = threadIdx.x >> 7;
    float acc = 0.f;
    for (int o = blockIdx.y; o < n_oct; o += gridDim.y) {
        const float* p = src + ((size_t)o * 8 + kk) * row + col0 + t;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc += p[(size_t)2 * c * row];
    }
    if (acc == 1.2345e-30f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void calib_write16(float4* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// phase-strided dword stores: block (x, phase) writes element q*8 + phase for 256 consecutive q; the 8 phases of a line are
// written by 8 different blocks (grid.y), as the polyphase transposed conv does
__global__ __launch_bounds__(256) void calib_write4s(float* __restrict__ dst, size_t n) {
    const int phase = blockIdx.y;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q * 8 + phase < n; q += (size_t)gridDim.x * 256)
        dst[q * 8 + phase] = 1.f;
}

int main() {
    float *a, *sink;
    if (hipMalloc(&a, N_BYTES) != hipSuccess || hipMalloc(&sink, 256) != hipSuccess) return 1;
    (void)hipMemset(a, 0, N_BYTES);
    const size_t n4 = N_BYTES / 16, n = N_BYTES / 4;
    const int row = 1 << 20, n_oct = (int)(n / row / 8);          // 1 Mi columns x (8 x n_oct) rows
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read16, dim3(8192), dim3(256), 0, 0, (const float4*)a, sink, n4);
        hipLaunchKernelGGL(calib_read4, dim3(row / 128, 4), dim3(256), 0, 0, a, sink, row, n_oct);
        hipLaunchKernelGGL(calib_write16, dim3(8192), dim3(256), 0, 0, (float4*)a, n4);
        hipLaunchKernelGGL(calib_write4s, dim3(4096, 8), dim3(256), 0, 0, a, n);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("traffic_calib: every kernel moved %zu bytes per launch\n", N_BYTES);
    return 0;
}
