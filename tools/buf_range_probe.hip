// Probe: how does gfx950 range-check a raw (stride 0) buffer_load_dwordx4 that straddles num_records, and one whose voffset is negative
// (wraps) but whose later dwords would land at offsets >= 0?   hipcc --offload-arch=gfx950 tools/buf_range_probe.hip -o tools/bin/buf_range_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ i4 ld16(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__global__ void probe(const float* x, int nrec_bytes, float* out) {
    i4 r;
    const unsigned long long a = (unsigned long long)x;
    r.x = (int)(a & 0xffffffffu); r.y = (int)((a >> 32) & 0xffffu); r.z = nrec_bytes; r.w = 0x00020000;
    const int lane = threadIdx.x;
    // lane l loads 16 bytes at byte offset 4 * (l - 8): lanes 0..7 start negative, lanes 5..7 straddle 0; lanes near the end straddle num_records
    // (inline asm: the intrinsic's result, used component by component, is narrowed by hipcc into four dword loads of the SAME address)
    i4 v;
    const int vo = 4 * (lane - 8);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(vo), "s"(r) : "memory");
    out[4 * lane + 0] = __int_as_float(v.x); out[4 * lane + 1] = __int_as_float(v.y);
    out[4 * lane + 2] = __int_as_float(v.z); out[4 * lane + 3] = __int_as_float(v.w);
}
int main() {
    float h[64], *d, *o, ho[256];
    for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    const int nrec = 4 * 37;            // 37 valid floats
    probe<<<1, 64>>>(d, nrec, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d first elem %3d: %6.0f %6.0f %6.0f %6.0f\n", l, l - 8, ho[4 * l], ho[4 * l + 1], ho[4 * l + 2], ho[4 * l + 3]);
    return 0;
}
