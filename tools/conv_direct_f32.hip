// EXPERIMENT (round 3, not part of the product; built into tools/conv_bench.hip, TTSAMD_DIRECT=0/1 switches it): measured
// 94-126 TF against conv1d_mfma_f32's 106-125 TF on the production shapes -- equal on C = 128 k = 7 / 11, 4-12 % slower
// elsewhere (profiles/r3/conv_direct_ab.txt): the barrier-per-chunk LDS ring of conv_mfma.hip is not what bounds the fp32
// engine.
// Exact-fp32 Conv1d for the large layers, second generation: same arithmetic as conv_mfma.hip
// (v_mfma_f32_32x32x2_f32, operands as float4 = four channel pairs per LDS entry, channel-first fp32 tensors in HBM), but
//   * the WEIGHTS never go through LDS: every wave streams the A fragments of its own 32 output rows from L2 straight
//     into registers (one float4 per lane and (octet, tap) step = 32 MFMAs = 2048 cycles of matrix-pipe time, fetched one
//     step ahead), in the packed order pack_conv_weight already stores them in;
//   * the ACTIVATIONS go through LDS in slabs of 16 input channels (2 octets x (256 + (K-1) d) columns, 20 KB), double
//     buffered: one barrier per slab, and the loads of slab s + 1 fly during the 2 K steps (64 K MFMAs per wave) of slab s;
//   * a wave owns 32 rows x 256 columns (8 accumulators); the residual (and the running ResBlock sum) is preloaded into
//     the accumulators in the MFMA C layout and the result leaves in that layout (128-byte row segments per half wave).
// Block = 4 waves = WM row slabs x WN column slabs (128 x 256 or 64 x 512 outputs), 39 KB of LDS, two blocks per CU
// (registers: 128 accumulators per lane).  Replaces conv1d_mfma_f32 where the grid fills the chip (launch_conv, conv_mfma.hip); reference ops as there:
// vocoder/hifigan/models.py:46-53, models/fastpitch/fastpitch/transformer.py:72-90.
#include <cstdlib>
#include <cstring>

#include "../tts-arabic-pytorch_amd/csrc/conv_mfma_common.hpp"

namespace ttsamd {

typedef int dir_i4 __attribute__((ext_vector_type(4)));
typedef float dir_f4 __attribute__((ext_vector_type(4)));
__device__ dir_f4 dir_ld16(dir_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float dir_ld4(dir_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void dir_st4(float v, dir_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
constexpr int DIR_OOB = 0x7ffffff0;

__device__ __forceinline__ dir_i4 dir_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    dir_i4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)(unsigned)(a >> 32);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}

template <int K, int WM, int WN>
struct DirGeo {
    static constexpr int NT = 8;                             // 32-column tiles per wave
    static constexpr int NCOLS = WN * NT * 32;
    static constexpr int SO = WN >= 2 ? 1 : 2;               // octets per slab (16 / 8 input channels: 20 staging registers per thread)
    static constexpr int WS = NCOLS + (K - 1) * DMAX;        // entries per (octet, kk) row
    static constexpr int NE = 2 * SO * WS;                   // entries per slab
    static constexpr int NXI = (NE + 255) / 256;             // ... per thread
    static constexpr size_t LDS = (size_t)2 * NE * 16;       // two slabs
};

template <int K, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv1d_direct_f32(const ConvParams p) {
    using G = DirGeo<K, WM, WN>;
    constexpr int NT = G::NT, WS = G::WS, NXI = G::NXI, SO = G::SO, NE = G::NE;
    extern __shared__ __attribute__((aligned(16))) float4 Xs[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5, l31 = lane & 31;
    const int wm = wid / WN, wn = wid % WN;
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * G::NCOLS;
    const int co0 = blockIdx.y * (32 * WM) + 32 * wm;        // this wave's first output row
    int n_out = p.Nout;
    if (p.lens_out) n_out = min(n_out, (int)p.lens_out[b] * p.len_out_mul);
    if (q0 >= n_out) return;
    int in_len = p.Lin;
    if (p.lens_in) in_len = min(in_len, (int)p.lens_in[b] * p.len_in_mul);
    const int dil = p.dil;
    const int W1 = G::NCOLS + (K - 1) * dil;                 // staged columns actually used
    const int x0 = q0 - p.pad;                               // input position of staged column 0
    const int x_cs = p.x_cs, CoutP = p.CoutP, Cout = p.Cout;
    const int n_slabs = p.Cin / (8 * SO);
    const float in_slope = p.in_slope;
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    const int cw = wn * (NT * 32) + l31;                     // this lane's column in tile 0

    // ---- accumulators start from the residual (+ the running sum in the accumulate modes): C layout, one buffer load per
    // element with a per-lane column offset and a scalar row offset (as conv_mfma.hip's EPI 3)
    f32x16 acc[NT];
    int vq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = q0 + cw + 32 * j;
        vq[j] = q < n_out ? q * 4 : DIR_OOB;
    }
    {
        const bool has_res = p.res != nullptr, has_prev = p.mode != 0;
        const int r_cs = has_res ? p.r_cs : p.y_cs;
        const dir_i4 rrs = dir_rsrc(has_res ? (const void*)(p.res + (int64_t)b * p.r_bs) : (const void*)(p.y + (int64_t)b * p.y_bs),
                                    (unsigned)Cout * r_cs * 4);
        const dir_i4 yrs0 = dir_rsrc(p.y + (int64_t)b * p.y_bs, (unsigned)Cout * p.y_cs * 4);
        const int rk = 4 * kk * r_cs * 4, yk = 4 * kk * p.y_cs * 4;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = co0 + (r & 3) + 8 * (r >> 2);
                float v = dir_ld4(rrs, has_res ? vq[j] + rk : DIR_OOB, row * r_cs * 4, 0);
                if (has_prev) v = dir_ld4(yrs0, vq[j] + yk, row * p.y_cs * 4, 0) + v;
                acc[j][r] = v;
            }
    }

    // ---- staging: entry e = (row r8 = (octet, kk) of the slab, column): channels 8 o + kk + {0, 2, 4, 6} at one position
    float sx[4 * NXI];
    int st_off[NXI];
    bool st_ok[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int e = tid + 256 * i;
        const int r8 = min(e / WS, 2 * SO - 1), col = e - (e / WS) * WS;
        const int pos = x0 + col;
        st_ok[i] = e < NE && col < W1 && pos >= 0 && pos < in_len;
        st_off[i] = (8 * (r8 >> 1) + (r8 & 1)) * x_cs + min(max(pos, 0), max(in_len - 1, 0));
    }
#define DIR_LOAD(S)                                                                                     \
    {                                                                                                   \
        const float* __restrict__ xs_ = xb + (int64_t)(S) * (8 * SO) * x_cs;                           \
        _Pragma("unroll") for (int i = 0; i < NXI; ++i)                                                 \
            _Pragma("unroll") for (int pc = 0; pc < 4; ++pc) sx[4 * i + pc] = xs_[st_off[i] + 2 * pc * x_cs]; \
    }
#define DIR_LRELU(v) ((v) > 0.f ? (v) : (v) * in_slope)
#define DIR_WRITE(BUF)                                                                                  \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NXI; ++i) {                                               \
            const int e = tid + 256 * i;                                                                \
            if (e < NE)                                                                                 \
                Xs[(BUF) * NE + e] = st_ok[i] ? make_float4(DIR_LRELU(sx[4 * i]), DIR_LRELU(sx[4 * i + 1]), \
                                                            DIR_LRELU(sx[4 * i + 2]), DIR_LRELU(sx[4 * i + 3])) \
                                              : make_float4(0.f, 0.f, 0.f, 0.f);                        \
        }                                                                                               \
    }
    DIR_LOAD(0)
    DIR_WRITE(0)
    __syncthreads();

    const dir_i4 wrs = dir_rsrc(p.w, (unsigned)(p.Cin / 8) * K * 2 * CoutP * 16);
    const int wv = (kk * CoutP + co0 + l31) * 16;            // this lane's A fragment inside an (octet, tap) step
    const int wstep = 2 * CoutP * 16;
    dir_f4 a_next = dir_ld16(wrs, wv, 0, 0);
    for (int s = 0; s < n_slabs; ++s) {
        if (s + 1 < n_slabs) DIR_LOAD(s + 1)                 // lands during this slab's 4 K steps
        const float4* sB = Xs + (s & 1) * NE + kk * WS + cw;
        float4 Bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) Bf[j] = sB[j * 32];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int o = 0; o < SO; ++o) {
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const dir_f4 a = a_next;
                {   // A fragment of the next step (clamped at the very end: a harmless re-load)
                    const int step_next = min((s * SO + o) * K + t + 1, n_slabs * SO * K - 1);
                    a_next = dir_ld16(wrs, wv, step_next * wstep, 0);
                }
                const float4* cur = sB + o * 2 * WS + t * dil;
                // first B operands of the next step: next tap, next octet of this slab, or (clamped) this step again
                const float4* nx = (t + 1 < K) ? cur + dil : (o + 1 < SO ? sB + (o + 1) * 2 * WS : cur);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float4 bq = Bf[j & 3];
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) {
                        const float av = pq == 0 ? a.x : (pq == 1 ? a.y : (pq == 2 ? a.z : a.w));
                        const float bv = pq == 0 ? bq.x : (pq == 1 ? bq.y : (pq == 2 ? bq.z : bq.w));
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                        // the slot is refilled behind the last MFMA that reads it: tile j + 4 of this step, or tile j - 4 of the next
                        if (pq == 3) Bf[j & 3] = (j < 4) ? cur[(j + 4) * 32] : nx[(j - 4) * 32];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (s + 1 < n_slabs) DIR_WRITE((s + 1) & 1)          // the other buffer was last read in slab s - 1
        __syncthreads();
    }
#undef DIR_LOAD
#undef DIR_WRITE
#undef DIR_LRELU

    // ---- epilogue: + bias [, ReLU] [, / div], stores in the C layout (per store instruction two 128-byte row segments)
    {
        const dir_i4 yrs = dir_rsrc(p.y + (int64_t)b * p.y_bs, (unsigned)Cout * p.y_cs * 4);
        const int yk = 4 * kk * p.y_cs * 4;
        const bool do_div = p.mode == 2, relu = p.relu_out == 1;
        const float div = p.div;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = p.bias ? p.bias[min(co0 + (r & 3) + 8 * (r >> 2) + 4 * kk, Cout - 1)] : 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[j][r] + bv[r];
                if (relu) v = fmaxf(v, 0.f);
                if (do_div) v = v / div;
                dir_st4(v, yrs, vq[j] + yk, (co0 + (r & 3) + 8 * (r >> 2)) * p.y_cs * 4, 0);
            }
    }
}

template <int K, int WM, int WN>
static int32_t launch_direct_cfg(const ConvParams& p, hipStream_t stream) {
    using G = DirGeo<K, WM, WN>;
    static bool attr_set[16] = {};
    int dev_id = 0;
    TTS_CHECK_HIP(hipGetDevice(&dev_id));
    dev_id &= 15;
    if (!attr_set[dev_id]) {
        TTS_CHECK_HIP(hipFuncSetAttribute((const void*)conv1d_direct_f32<K, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS));
        attr_set[dev_id] = true;
    }
    dim3 grid((p.Nout + G::NCOLS - 1) / G::NCOLS, (p.CoutP + 32 * WM - 1) / (32 * WM), p.batch);
    hipLaunchKernelGGL((conv1d_direct_f32<K, WM, WN>), grid, dim3(256), G::LDS, stream, p);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// the layers this kernel takes over from conv1d_mfma_f32: plain "same" convs whose grid fills the chip
bool direct_supported(const ConvParams& p) {
    if (p.precision != 0 || p.n_phase != 1 || p.y_ts != 1 || p.relu_out > 1 || p.scale != nullptr || p.x_packed || p.y_packed) return false;
    if (!(p.K == 3 || p.K == 7 || p.K == 11)) return false;
    if (p.dil < 1 || p.dil > DMAX || p.pad != (p.K - 1) * p.dil / 2 || p.Nout != p.Lin) return false;
    if (p.Cout % 64 != 0 || p.CoutP != p.Cout || p.Cin % 16 != 0) return false;
    if ((int64_t)p.Cout * std::max(p.y_cs, p.res ? p.r_cs : 0) * 4 >= ((int64_t)1 << 31)) return false;
    const int rb = p.Cout >= 128 ? 128 : 64, ncols = 256 * (128 / rb);
    const int64_t blocks = (int64_t)((p.Nout + ncols - 1) / ncols) * ((p.Cout + rb - 1) / rb) * p.batch;
    const char* mb = getenv("TTSAMD_DIRECT_MIN_BLOCKS");          // tests force the kernel onto small problems
    return blocks >= (mb ? atoi(mb) : 512);
}

int32_t launch_direct(const ConvParams& p, hipStream_t stream) {
#define DIR_CASE(KK)                                                                     \
    if (p.K == KK) {                                                                     \
        if (p.Cout >= 128) return launch_direct_cfg<KK, 4, 1>(p, stream);                \
        return launch_direct_cfg<KK, 2, 2>(p, stream);                                   \
    }
    DIR_CASE(3) DIR_CASE(7) DIR_CASE(11)
#undef DIR_CASE
    set_error("direct conv: kernel size %d not instantiated", p.K);
    return TTSAMD_EINVAL;
}

}  // namespace ttsamd
