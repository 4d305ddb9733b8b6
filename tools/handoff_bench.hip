// Latency of a block-to-block hand-off through memory inside one resident grid on MI355X (256 CUs in 8 XCDs): one block stores a word,
// its partner polls for it and answers -- the primitive of the Tacotron2 persistent decoder's dataflow schedule (csrc/tacotron2.hip).
// Partners on DIFFERENT XCDs (blocks i, i ^ 1) against partners on the SAME XCD (i, i ^ 8: workgroups go to the XCDs round-robin), and the
// cache-policy bits of the store and of the polling load (aux: 1 = sc0, 2 = nt, 16 = sc1; sc1 = device scope, what the decoder uses).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/handoff_bench.hip -o tools/bin/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int bi4 __attribute__((ext_vector_type(4)));
__device__ float hb_load(bi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void hb_store(float v, bi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ inline bi4 hb_rsrc(const void* p, int bytes) {
    const unsigned long long a = (unsigned long long)p;
    bi4 r;
    r.x = (int)(unsigned)a; r.y = (int)(unsigned)(a >> 32); r.z = bytes; r.w = 0x00020000;
    return r;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SAUX, int LAUX, bool INV>
__global__ __launch_bounds__(64) void pingpong(float* box, int mask, int iters, int active, unsigned long long* ticks, unsigned* xcc, int* fails) {
    const int i = blockIdx.x, partner = i ^ mask;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) xcc[i] = x;
    const int pid = ((i / (2 * mask)) * mask) + (i & (mask - 1));              // pair number 0 .. 127 (i with bit `mask` removed)
    if (threadIdx.x != 0 || pid % (128 / active) != 0) return;                   // `active` pairs ping-pong at the same time
    const bi4 rs = hb_rsrc(box, 256 * 128);
    const bool leader = (i & mask) == 0;
    const unsigned long long t0 = wall_clock64();
    int bad = 0;
    for (int it = 1; it <= iters && !bad; ++it) {
        if (leader) hb_store((float)it, rs, partner * 128, 0, SAUX);
        int spin = 0;
        for (;;) {
            if (INV) asm volatile("buffer_inv sc0" ::: "memory");
            const float v = hb_load(rs, i * 128, 0, LAUX);
            if (v == (float)it) break;
            if (++spin > (1 << 16)) { bad = 1; break; }
        }
        if (!leader) hb_store((float)it, rs, partner * 128, 0, SAUX);
    }
    const unsigned long long t1 = wall_clock64();
    ticks[i] = t1 - t0;
    if (bad) atomicAdd(fails, 1);
}

template <int SAUX, int LAUX, bool INV>
static void run(const char* name, int mask, int active, float* box, unsigned long long* ticks, unsigned* xcc, int* fails) {
    const int iters = 2000;
    CK(hipMemset(box, 0, 256 * 128));
    CK(hipMemset(fails, 0, 4));
    CK(hipMemset(ticks, 0, 256 * 8));
    hipLaunchKernelGGL((pingpong<SAUX, LAUX, INV>), dim3(256), dim3(64), 0, 0, box, mask, iters, active, ticks, xcc, fails);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> ht(256);
    std::vector<unsigned> hx(256);
    int hf = 0;
    CK(hipMemcpy(ht.data(), ticks, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hf, fails, 4, hipMemcpyDeviceToHost));
    double sum = 0; int n = 0, same = 0, pairs = 0;
    for (int i = 0; i < 256; ++i) {
        if (ht[i]) { sum += (double)ht[i]; ++n; }
        if ((i & mask) == 0) { ++pairs; same += hx[i] == hx[i ^ mask]; }
    }
    printf("%-34s partner i^%d (%3d of %3d pairs on one XCD) %3d pairs active: %s one hand-off %.2f us\n", name, mask, same, pairs, active,
           hf ? "TIMED OUT (the polled word never became visible);" : "", n ? sum / n / 100.0 / iters / 2.0 : 0.0);
}

int main() {
    float* box; unsigned long long* ticks; unsigned* xcc; int* fails;
    CK(hipMalloc(&box, 256 * 128)); CK(hipMalloc(&ticks, 256 * 8)); CK(hipMalloc(&xcc, 256 * 4)); CK(hipMalloc(&fails, 4));
    for (int active : {1, 128}) {
        for (int mask : {1, 8}) {
            run<16, 16, false>("store sc1, load sc1", mask, active, box, ticks, xcc, fails);
            run<17, 17, false>("store sc0 sc1, load sc0 sc1", mask, active, box, ticks, xcc, fails);
            run<0, 1, false>("store plain, load sc0", mask, active, box, ticks, xcc, fails);
            run<0, 0, true>("store plain, buffer_inv sc0 + load", mask, active, box, ticks, xcc, fails);
            run<16, 1, false>("store sc1, load sc0", mask, active, box, ticks, xcc, fails);
            run<0, 16, false>("store plain, load sc1", mask, active, box, ticks, xcc, fails);
        }
    }
    return 0;
}
