// Cost of a grid-wide barrier inside one persistent kernel on MI355X (256 CUs in 8 XCDs, one block per CU), with and
// without a data exchange across it: every block publishes 32 floats, the barrier, every block reads all 8192 floats
// (the shape of the Tacotron2 decoder step: 8 utterances x 1024 hidden units produced by 256 blocks, consumed by all).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/grid_barrier_bench.hip -o tools/bin/grid_barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// (clang's __builtin_amdgcn_raw_buffer_load_b128 of ROCm 7.2 lowers to ONE dword load splatted over the vector: declare the LLVM intrinsics)
typedef int bi4 __attribute__((ext_vector_type(4)));
typedef float bf4 __attribute__((ext_vector_type(4)));
__device__ bf4 bench_buffer_load_f4(bi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void bench_buffer_store_f1(float v, bi4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ inline bi4 bench_rsrc(const void* p, int bytes) {
    const unsigned long long a = (unsigned long long)p;
    bi4 r;
    r.x = (int)(unsigned)a; r.y = (int)(unsigned)(a >> 32); r.z = bytes; r.w = 0x00020000;
    return r;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, int* err) {
    __threadfence();                                            // every thread's stores reach device scope
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE);          // agent scope by default in HIP device code
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *err = 1; ok = false; break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // drop stale lines of this CU's L1 / this XCD's L2
    return ok;
}

// variant 1: no read-modify-write at all.  Block b publishes its epoch in slots[b]; wave 0 of every block polls all slots
// with one 1 KB load (lane l reads slots 4l..4l+3).
__device__ __forceinline__ bool grid_barrier_slots(unsigned* slots, unsigned epoch, int nb, int* err) {
    __threadfence();
    __syncthreads();
    bool ok = true;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(slots + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int l = threadIdx.x;
        int spins = 0;
        for (;;) {
            bool all = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sidx = 4 * l + i;
                const unsigned v = sidx < nb ? __hip_atomic_load(slots + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
                all = all && (int)(v - epoch) >= 0;
            }
            if (__all(all)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 20)) { *err = 1; ok = false; break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

// variant 2: two levels.  The blocks of one XCD meet on a counter in THEIR L2 (workgroup-scope atomics execute in the
// XCD's L2), the last arriver publishes the XCD's epoch at agent scope and polls the 8 XCD slots, then releases its XCD
// through an L2 flag that the others poll with an L2 read-modify-write (a plain load could be served by a stale L1 line).
struct XcdBar { unsigned cnt[8][32]; unsigned flag[8][32]; unsigned gslot[8 * 32]; unsigned pop[8]; };
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ bool grid_barrier_xcd(XcdBar* bar, unsigned epoch, unsigned xcc, unsigned pop, unsigned live_mask, int* err, unsigned zero) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(&bar->cnt[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        if (old + 1 == pop * epoch) {
            __hip_atomic_store(&bar->gslot[xcc * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                bool all = true;
#pragma unroll
                for (int x = 0; x < 8; ++x)
                    if (live_mask >> x & 1u) {
                        const unsigned v = __hip_atomic_load(&bar->gslot[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        all = all && (int)(v - epoch) >= 0;
                    }
                if (all) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 20)) { *err = 1; ok = false; break; }
            }
            __hip_atomic_exchange(&bar->flag[xcc][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            while ((int)(__hip_atomic_fetch_add(&bar->flag[xcc][0], zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 20)) { *err = 1; ok = false; break; }
            }
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    return ok;
}

// variants 3 / 4: the slot barrier WITHOUT cache maintenance (no buffer_wbl2 / buffer_inv): the exchanged data itself is
// written and read with cache-bypassing buffer accesses (aux 16 = sc1, agent scope; 17 = sc0 sc1, system scope), a plain
// s_waitcnt vmcnt(0) orders the data before the slot
__device__ __forceinline__ bool grid_barrier_nofence(unsigned* slots, unsigned epoch, int nb, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(slots + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int l = threadIdx.x;
        int spins = 0;
        for (;;) {
            bool all = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sidx = 4 * l + i;
                const unsigned v = sidx < nb ? __hip_atomic_load(slots + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
                all = all && (int)(v - epoch) >= 0;
            }
            if (__all(all)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 20)) { *err = 1; ok = false; break; }
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    return ok;
}

template <int MODE>   // 0: barrier only; 1: publish 32 floats + barrier + read 8192 floats
__global__ __launch_bounds__(256) void barrier_loop(unsigned* cnt, int iters, float* xch, float* out, int* err, int variant,
                                                    unsigned* slots, XcdBar* bar) {
    const unsigned nb = gridDim.x;
    float acc = 0.f;
    unsigned xcc = 0, pop = 0, live = 0;
    if (variant == 2 || variant == 5) {                     // census: how many blocks each XCD got (one slow barrier)
        xcc = xcc_id();
        if (threadIdx.x == 0) atomicAdd(&bar->pop[xcc], 1u);
        if (!grid_barrier(cnt, nb, err)) return;
        pop = __hip_atomic_load(&bar->pop[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int x = 0; x < 8; ++x)
            if (__hip_atomic_load(&bar->pop[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) live |= 1u << x;
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 && variant >= 2) {
            const bi4 rs = bench_rsrc(xch + (size_t)(it & 1) * 8192, 8192 * 4);
            const float v = (float)((it & 15) + blockIdx.x);
            if (threadIdx.x < 32) {
                if (variant != 4) bench_buffer_store_f1(v, rs, (blockIdx.x * 32 + threadIdx.x) * 4, 0, 16);
                else bench_buffer_store_f1(v, rs, (blockIdx.x * 32 + threadIdx.x) * 4, 0, 17);
            }
        } else if (MODE == 1) {
            float* dst = xch + (size_t)(it & 1) * 8192;
            if (threadIdx.x < 32) dst[blockIdx.x * 32 + threadIdx.x] = (float)((it & 15) + blockIdx.x);
        }
        const bool okb = variant == 0 ? grid_barrier(cnt, nb * (unsigned)(it + 1), err)
                       : variant == 1 ? grid_barrier_slots(slots, (unsigned)(it + 1), (int)nb, err)
                       : variant == 3 || variant == 4 ? grid_barrier_nofence(slots, (unsigned)(it + 1), (int)nb, err)
                                      : grid_barrier_xcd(bar, (unsigned)(it + 1), xcc, pop, live, err, (unsigned)variant - 2u - (variant >= 5 ? 3u : 0u));
        if (!okb) return;
        if (MODE == 1 && variant >= 2) {
            const bi4 rs = bench_rsrc(xch + (size_t)(it & 1) * 8192, 8192 * 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf4 r = variant != 4 ? bench_buffer_load_f4(rs, (threadIdx.x + 256 * i) * 16, 0, 16)
                                           : bench_buffer_load_f4(rs, (threadIdx.x + 256 * i) * 16, 0, 17);
                acc += r.x + r.y + r.z + r.w;
            }
        } else if (MODE == 1) {
            const float* src = xch + (size_t)(it & 1) * 8192;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(src + (threadIdx.x + 256 * i) * 4);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (MODE == 1) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned* cnt; float *xch, *out; int* err;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&xch, 2 * 8192 * 4)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&err, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* slots; XcdBar* bar;
    CK(hipMalloc(&slots, 1024)); CK(hipMalloc(&bar, sizeof(XcdBar)));
    for (int variant = 1; variant < 5; ++variant)
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(slots, 0, 1024)); CK(hipMemset(bar, 0, sizeof(XcdBar)));
            CK(hipMemset(cnt, 0, 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(xch, 0, 2 * 8192 * 4));
            int it = variant == 0 ? iters / 10 : iters;
            void* args[] = {&cnt, &it, &xch, &out, &err, &variant, &slots, &bar};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel(mode ? (void*)barrier_loop<1> : (void*)barrier_loop<0>, dim3(nb), dim3(256), args, 0, nullptr));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int eh; CK(hipMemcpy(&eh, err, 4, hipMemcpyDeviceToHost));
            std::vector<float> oh(256 * 256);
            CK(hipMemcpy(oh.data(), out, oh.size() * 4, hipMemcpyDeviceToHost));
            bool same = true;                       // every block must have read every other block's value of every iteration
            for (int t = 0; t < 256 && mode; ++t) {
                float want = 0.f;
                for (int i2 = 0; i2 < it; ++i2)
                    for (int i = 0; i < 8; ++i) {
                        const int f = (t + 256 * i) * 4, b = f / 32;
                        const float v = b < nb ? (float)((i2 & 15) + b) : 0.f;
                        want += v + v + v + v;
                    }
                for (int b = 0; b < nb; ++b) same = same && oh[b * 256 + t] == want;
            }
            printf("variant %d mode %d  blocks %d  %d barriers: %.3f ms  -> %.2f us per barrier%s%s\n", variant, mode, nb, it, ms, ms * 1000.0 / it,
                   eh ? "  SPIN TIMEOUT" : "", mode ? (same ? "  (exchange consistent)" : "  (EXCHANGE MISMATCH)") : "");
        }
    return 0;
}
