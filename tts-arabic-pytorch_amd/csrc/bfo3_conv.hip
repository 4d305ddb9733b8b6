// Split-bf16 mode of the octet engine (bfo3.hpp): the layers around the fused ResBlock pairs --
//   bfo3_conv1d   Conv1d with a C-in slab loop (conv_pre 80 -> 512 k7; the C = 256 ResBlock convs of HiFi-GAN stage 1;
//                 FastPitch's conv-FF / projections with the fp32 channel-first output)
//   bfo3_convt    ConvTranspose1d(stride u, kernel 2u, padding u/2) as u polyphase 2-tap convs over ONE staged window
//   bfo3_pack / bfo3_unpack   fp32 channel-first <-> x3 tensor
//   bfo3_conv_post            leaky_relu(0.01) -> Conv1d(C -> 1, k7) -> tanh on an x3 tensor
// and the host-side weight packers.  Reference ops: vocoder/hifigan/models.py:46-53 (ResBlock1), :96-99,114-115
// (upsamplers), :112 (conv_pre), :123-125 (conv_post); models/fastpitch/fastpitch/transformer.py:72-90.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bfo3.hpp"
#include "kernels.hpp"

namespace ttsamd {

// =====================================================================================================================
// Conv1d: block = 4 waves, wave = 32 rows x (NT x 32) columns, WM row slabs x WN column slabs; the input goes through LDS
// in slabs of SH 16-channel groups as a hi and a lo plane (<= 80 KB: two blocks per CU), weights stream from L2 (bfo3_mma).
// =====================================================================================================================
template <int K, int WM, int WN, int NT_, int SH_>
struct Bfo3ConvGeo {
    static constexpr int NT = NT_, SH = SH_;
    static constexpr int NCOLS = WN * NT * 32;
    static constexpr int WS = NCOLS + (K - 1) * BFO_DMAX;
    static constexpr int NE = 2 * SH * WS;                   // entries of ONE plane of a slab
    static constexpr int NXI = (2 * NE + 255) / 256;         // 16-byte pieces per thread and slab
    static constexpr int PH = K <= 3 ? 2 : 1;
    static constexpr size_t LDS = (size_t)NE * 32;
    static_assert(LDS <= 80 * 1024, "two blocks per CU");
};

template <int K, int WM, int WN, int NT_, int SH_, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void bfo3_conv1d(const BfoConvParams p) {
    using G = Bfo3ConvGeo<K, WM, WN, NT_, SH_>;
    constexpr int NT = G::NT, WS = G::WS, NXI = G::NXI, SH = G::SH, NE = G::NE;
    extern __shared__ __attribute__((aligned(16))) uint4 Xs[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5, l31 = lane & 31;
    const int wm = wid / WN, wn = wid % WN;
    int b = blockIdx.z;
    int q0 = blockIdx.x * G::NCOLS;
    if (p.compact) {   // ragged batch: dead blocks last (common.hpp: live_tile)
        int tile = 0;
        if (!live_tile(p.lens, p.len_mul, p.Lin, G::NCOLS, p.batch, blockIdx.z * gridDim.x + blockIdx.x, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * G::NCOLS;
    }
    const int co0 = blockIdx.y * (32 * WM) + 32 * wm;        // this wave's first output row
    const int L = p.Lin;
    int len = L;
    if (p.lens) len = min(len, (int)p.lens[b] * p.len_mul);
    const int len_o = p.out_all ? L : len;                   // outputs computed / stored; `len` keeps masking the input
    if (q0 >= len_o) return;
    const int dil = p.dil;
    const int W1 = G::NCOLS + (K - 1) * dil;
    const int x0 = q0 - (K - 1) * dil / 2;
    const int NOI = p.Cin / 8, NHT = (p.Cin + 15) / 16, NOO = p.Cout / 8;
    const int CoutP = (p.Cout + 31) & ~31;
    const bfo_i4 xrs = bfo_rsrc((const char*)p.x + (int64_t)b * NOI * L * 32, (unsigned)NOI * L * 32);
    const bfo_i4 wrs = bfo_rsrc(p.w, (unsigned)NHT * K * 2 * CoutP * 32);
    const int wv = (kk * CoutP + co0 + l31) * 32;
    const int cw = wn * (NT * 32) + l31;
    const uint4* sB = Xs + kk * WS + cw;

    bfo_f16 acc[NT];
    {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = p.bias ? p.bias[min(co0 + 8 * (r >> 2) + 4 * kk + (r & 3), p.Cout - 1)] : 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = bv[r];
    }
    for (int s0 = 0; s0 < NHT; s0 += SH) {
        const int nh = min(SH, NHT - s0);
        if (s0 > 0) __syncthreads();                        // the previous slab has been consumed
        // staged 8 pieces per thread at a time: the accumulators are live here, a whole slab in flight would spill
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                     // opaque: keeps the per-piece index math inside the slab loop
        char* const xb = reinterpret_cast<char*>(Xs);
#pragma unroll
        for (int i0 = 0; i0 < NXI; i0 += 8) {
            bfo_i4 xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = tid_o + 256 * (i0 + i);
                const int e = u >> 1, hk = u & 1;
                const int ol = e / WS, col = e - ol * WS;
                const int o = 2 * s0 + ol, pos = x0 + col;
                const bool ok = i0 + i < NXI && ol < 2 * nh && o < NOI && col < W1 && pos >= 0 && pos < len;
                xv[i] = bfo_ld16(xrs, ok ? ((o * L + pos) * 2 + hk) * 16 : BFO_OOB, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = tid_o + 256 * (i0 + i);
                if (i0 + i < NXI && u < 2 * NE) {
                    bfo_i2 h2, l2;
                    h2.x = xv[i].x; h2.y = xv[i].y; l2.x = xv[i].z; l2.y = xv[i].w;
                    *reinterpret_cast<bfo_i2*>(xb + u * 8) = h2;
                    *reinterpret_cast<bfo_i2*>(xb + NE * 16 + u * 8) = l2;
                }
            }
        }
        __syncthreads();
        bfo3_mma<K, G::PH, NT>(acc, wrs, wv, 2 * CoutP * 32, sB, NE, nh, 2 * WS, dil, s0);
    }

    // ---- epilogue: [+ residual] [+ running sum] [/ div], activation of the consumer
    if (co0 >= p.Cout) return;
    if constexpr (OUT_F32) {
        // fp32 channel-first output (+ fp32 residual): 4 bytes per lane and register, two 128-byte row segments per store
        const bfo_i4 yrs = bfo_rsrc(p.y_f32 + (int64_t)b * p.Cout * L, (unsigned)p.Cout * L * 4);
        const bool has_res = p.res_f32 != nullptr;
        const bfo_i4 rrs = bfo_rsrc(has_res ? p.res_f32 + (int64_t)b * p.Cout * L : p.y_f32, (unsigned)p.Cout * L * 4);
        const float os = p.out_slope;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int q = q0 + cw + 32 * j;
            const int vq = q < len_o ? q * 4 + 4 * kk * L * 4 : BFO_OOB;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = bfo_ld4f(rrs, has_res ? vq : BFO_OOB, (co0 + 8 * (r >> 2) + (r & 3)) * L * 4, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bfo_st4f(bfo_lrelu(acc[j][r] + rv[r], os), yrs, vq, (co0 + 8 * (r >> 2) + (r & 3)) * L * 4, 0);
            if (j & 1) __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    const bfo_i4 yrs = bfo_rsrc((char*)p.y + (int64_t)b * NOO * L * 32, (unsigned)NOO * L * 32);
    const int so0 = (co0 >> 3) * L * 32;
    const float os = p.out_slope;
    const float sc = p.mode == 2 ? 1.f / p.div : 1.f;
    const bool has_res = p.res != nullptr, has_sum = p.mode != 0;
    const bfo_i4 rrs = bfo_rsrc((const char*)(has_res ? p.res : p.y) + (int64_t)b * NOO * L * 32, (unsigned)NOO * L * 32);
    const bfo_i4 srs = bfo_rsrc((const char*)(has_sum ? p.sum_in : p.y) + (int64_t)b * NOO * L * 32, (unsigned)NOO * L * 32);
    const float rinv = has_res ? 1.f / p.res_slope : 1.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = q0 + cw + 32 * j;
        const int vo = q < len_o ? (q * 2 + kk) * 16 : BFO_OOB;
        bfo_i4 rv[4], sv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // branch-free: a masked-off buffer load returns 0 without touching memory
            rv[g] = bfo_ld16(rrs, has_res ? vo : BFO_OOB, so0 + g * L * 32, 0);
            sv[g] = bfo_ld16(srs, has_sum ? vo : BFO_OOB, so0 + g * L * 32, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float r[4], s[4], v[4];
            bfo3_join4(rv[g].x, rv[g].y, rv[g].z, rv[g].w, r);
            bfo3_join4(sv[g].x, sv[g].y, sv[g].z, sv[g].w, s);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc[j][4 * g + e] + bfo_unrelu(r[e], rinv) + s[e]) * sc;
            bfo3_st16(bfo3_act4(v[0], v[1], v[2], v[3], os, -1), yrs, vo, so0 + g * L * 32);
        }
        if (j & 1) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int K, int WM, int WN, int NT, int SH, bool OUT_F32>
static int32_t bfo3_launch_conv_cfg(const BfoConvParams& p_in, hipStream_t stream) {
    BfoConvParams p = p_in;
    using G = Bfo3ConvGeo<K, WM, WN, NT, SH>;
    static std::atomic<uint64_t> lds_done{0};
    TTS_CHECK_HIP(lds_opt_in((const void*)bfo3_conv1d<K, WM, WN, NT, SH, OUT_F32>, (int)G::LDS, lds_done));
    const int CoutP = (p.Cout + 31) & ~31;
    dim3 grid((p.Lin + G::NCOLS - 1) / G::NCOLS, (CoutP + 32 * WM - 1) / (32 * WM), p.batch);
    p.ksplit = 1;
    p.compact = (compact_order(p.lens, p.batch) && !p.out_all) ? 1 : 0;
    hipLaunchKernelGGL((bfo3_conv1d<K, WM, WN, NT, SH, OUT_F32>), grid, dim3(256), G::LDS, stream, p);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int K>
static int32_t bfo3_launch_conv_k(const BfoConvParams& p, hipStream_t stream) {
    // 128-column tiles (registers: 64 accumulators + the A ring of K pairs + 8 B fragments under two waves per SIMD), 64-column
    // tiles for short sequences (FastPitch encoder: 64 tokens per utterance) and for grids under one block per CU
    const int64_t blocks4 = (int64_t)((p.Lin + 127) / 128) * ((p.Cout + 127) / 128) * p.batch;
    const bool narrow = p.Cout >= 128 && (p.Lin <= 96 || blocks4 < 256);
    // slab depth: 4 16-channel groups (45 KB of LDS at k = 11), 8 at k <= 3 (70 KB): half the barriers of the deep-K conv-FF convs
    constexpr int SH = K <= 3 ? 8 : 4;
    if (p.y_f32) {
        if (p.Cout < 128) {
            set_error("bfo3 conv: fp32 output is built for Cout >= 128 (got %d)", p.Cout);
            return TTSAMD_EINVAL;
        }
        if (narrow) return bfo3_launch_conv_cfg<K, 4, 1, 2, SH, true>(p, stream);
        return bfo3_launch_conv_cfg<K, 4, 1, 4, SH, true>(p, stream);
    }
    if (narrow) return bfo3_launch_conv_cfg<K, 4, 1, 2, SH, false>(p, stream);
    if (p.Cout >= 128) return bfo3_launch_conv_cfg<K, 4, 1, 4, SH, false>(p, stream);
    if (p.Cout >= 64) return bfo3_launch_conv_cfg<K, 2, 2, 4, 4, false>(p, stream);
    return bfo3_launch_conv_cfg<K, 1, 4, 4, 2, false>(p, stream);
}

int32_t bfo3_launch_conv(const BfoConvParams& p, hipStream_t stream) {
    TTS_REQUIRE(p.up == 1, "bfo3 conv: use bfo3_launch_convt for transposed convs");
    TTS_REQUIRE(p.Cin % 8 == 0 && p.Cout % 32 == 0, "bfo3 conv: Cin %% 8 and Cout %% 32 must be 0 (Cin=%d, Cout=%d)", p.Cin, p.Cout);
    TTS_REQUIRE(p.dil >= 1 && p.dil <= BFO_DMAX, "bfo3 conv: dilation %d outside [1,%d]", p.dil, BFO_DMAX);
    TTS_REQUIRE((int64_t)std::max(p.Cin, p.Cout) * p.Lin * 4 < ((int64_t)1 << 31), "bfo3 conv: tensor too large for 32-bit offsets");
    TTS_REQUIRE(p.mode == 0 || p.sum_in != nullptr, "bfo3 conv: mode %d needs sum_in", p.mode);
    TTS_REQUIRE(!p.y_f32 || (p.mode == 0 && p.res == nullptr), "bfo3 conv: fp32 output takes res_f32, mode 0");
    TTS_REQUIRE(p.ln_g == nullptr, "bfo3 conv: no LayerNorm epilogue in this mode (launch layernorm_cf)");
    if (p.Lin <= 0) return 0;
    conv_log("bfo3", p.K, p.Cin, p.Cout, p.Lin, p.batch, p.res != nullptr, p.mode, p.len_mul, p.lens != nullptr, 1);
    switch (p.K) {
        case 1: return bfo3_launch_conv_k<1>(p, stream);
        case 3: return bfo3_launch_conv_k<3>(p, stream);
        case 7: return bfo3_launch_conv_k<7>(p, stream);
        case 11: return bfo3_launch_conv_k<11>(p, stream);
        default:
            set_error("bfo3 conv: kernel size %d not instantiated (1,3,7,11)", p.K);
            return TTSAMD_EINVAL;
    }
}

// =====================================================================================================================
// ConvTranspose1d(stride U, kernel 2U, padding U/2):  y[co][q U + rho] = b[co] + sum_ci W[ci][co][ka] x[ci][q + dl]
//                                                                               + W[ci][co][ka + U] x[ci][q + dl - 1],
// ka = (rho + U/2) % U, dl = (rho + U/2) / U  -> per phase rho a 2-tap conv (dil -1) over the same window.
// Block = (32 RT output rows, NQ = 32 NQT input positions, ALL U phases); the 4 waves split the (phase, row tile, column
// group) combos, NT column tiles each.  A lane's result for (octet g, column, phase) is a complete 16-byte half entry, the
// two kk lanes of a column make the 32-byte entry: the stores go straight from the C layout, 32 bytes per output position
// (the 8-byte halves of the plain bf16 engine had to be transposed through LDS first).
// =====================================================================================================================
template <int U, int RT, int NQT, int NT>
__global__ __launch_bounds__(256, 2) void bfo3_convt(const BfoConvParams p) {
    constexpr int NQ = 32 * NQT, CG = NQT / NT, NC = U * RT * CG, NCALL = NC / 4, WSC = NQ + 2;
    static_assert(NC % 4 == 0 && NQT % NT == 0, "combos must split over 4 waves");
    extern __shared__ __attribute__((aligned(16))) uint4 Xs[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5, l31 = lane & 31;
    int b = blockIdx.z;
    int q0 = blockIdx.x * NQ;
    if (p.compact) {   // ragged batch: dead blocks last (common.hpp: live_tile)
        int tile = 0;
        if (!live_tile(p.lens, p.len_mul, p.Lin, NQ, p.batch, blockIdx.z * gridDim.x + blockIdx.x, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * NQ;
    }
    const int co0 = blockIdx.y * (32 * RT);
    const int L = p.Lin, Lo = L * U;
    int len = L;
    if (p.lens) len = min(len, (int)p.lens[b] * p.len_mul);
    if (q0 >= len) return;
    const int NOI = p.Cin / 8, NH = p.Cin / 16, NOO = p.Cout / 8;
    const int CoutP = (p.Cout + 31) & ~31;
    const bfo_i4 xrs = bfo_rsrc((const char*)p.x + (int64_t)b * NOI * L * 32, (unsigned)NOI * L * 32);
    const int ne = NOI * WSC;                                // entries of one plane

    // ---- stage the window: column c = input position q0 - 1 + c
    {
        char* const xb = reinterpret_cast<char*>(Xs);
        for (int u0 = 0; u0 < 2 * ne; u0 += 256 * 8) {
            bfo_i4 xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = u0 + tid + 256 * i;
                const int e = u >> 1, hk = u & 1;
                const int o = e / WSC, col = e - o * WSC;
                const int pos = q0 - 1 + col;
                const bool ok = e < ne && pos >= 0 && pos < len;
                xv[i] = bfo_ld16(xrs, ok ? ((o * L + pos) * 2 + hk) * 16 : BFO_OOB, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int u = u0 + tid + 256 * i;
                if (u < 2 * ne) {
                    bfo_i2 h2, l2;
                    h2.x = xv[i].x; h2.y = xv[i].y; l2.x = xv[i].z; l2.y = xv[i].w;
                    *reinterpret_cast<bfo_i2*>(xb + u * 8) = h2;
                    *reinterpret_cast<bfo_i2*>(xb + ne * 16 + u * 8) = l2;
                }
            }
        }
    }
    __syncthreads();

    const bfo_i4 yrs = bfo_rsrc((char*)p.y + (int64_t)b * NOO * Lo * 32, (unsigned)NOO * Lo * 32);
    const float os = p.out_slope;
#pragma unroll 1
    for (int c = 0; c < NCALL; ++c) {
        const int cb = wid + 4 * c;
        const int rho = cb % U, rt = (cb / U) % RT, cg = cb / (U * RT);
        const int dl = (rho + U / 2) / U;
        const int row0 = co0 + 32 * rt;
        bfo_f16 acc[NT];
        {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = p.bias ? p.bias[min(row0 + 8 * (r >> 2) + 4 * kk + (r & 3), p.Cout - 1)] : 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = bv[r];
        }
        const bfo_i4 wrs = bfo_rsrc((const char*)p.w + (int64_t)rho * NH * 2 * 2 * CoutP * 32, (unsigned)NH * 2 * 2 * CoutP * 32);
        const int wv = (kk * CoutP + row0 + l31) * 32;
        const uint4* sB = Xs + kk * WSC + cg * (NT * 32) + l31 + dl + 1;      // tap 0 reads x[q + dl], tap 1 x[q + dl - 1]
        bfo3_mma<2, 2, NT>(acc, wrs, wv, 2 * CoutP * 32, sB, ne, NH, 2 * WSC, -1);
        if (row0 < p.Cout) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = cg * (NT * 32) + 32 * j + l31;
                const int vo = q0 + n < len ? (((q0 + n) * U + rho) * 2 + kk) * 16 : BFO_OOB;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bfo3_st16(bfo3_act4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], os, -1), yrs, vo,
                             ((row0 >> 3) + g) * Lo * 32);
            }
        }
    }
}

template <int U, int RT, int NQT, int NT>
static int32_t bfo3_launch_convt_cfg(const BfoConvParams& p, hipStream_t stream) {
    constexpr int NQ = 32 * NQT;
    const size_t lds = (size_t)(p.Cin / 8) * (NQ + 2) * 32;
    TTS_REQUIRE(lds <= 80 * 1024, "bfo3 convt: window of %zu bytes does not fit (Cin=%d, u=%d)", lds, p.Cin, U);
    static std::atomic<uint64_t> lds_done{0};
    TTS_CHECK_HIP(lds_opt_in((const void*)bfo3_convt<U, RT, NQT, NT>, 80 * 1024, lds_done));
    dim3 grid((p.Lin + NQ - 1) / NQ, (p.Cout + 32 * RT - 1) / (32 * RT), p.batch);
    BfoConvParams q = p;
    q.compact = compact_order(p.lens, p.batch) ? 1 : 0;
    hipLaunchKernelGGL((bfo3_convt<U, RT, NQT, NT>), grid, dim3(256), lds, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int32_t bfo3_launch_convt(const BfoConvParams& p, hipStream_t stream) {
    TTS_REQUIRE(p.up == 8 || p.up == 2, "bfo3 convt: stride %d not built (8, 2)", p.up);
    TTS_REQUIRE(p.Cin % 16 == 0 && p.Cout % 32 == 0, "bfo3 convt: Cin %% 16 and Cout %% 32 must be 0 (Cin=%d, Cout=%d)", p.Cin, p.Cout);
    TTS_REQUIRE((int64_t)std::max(p.Cin, p.Cout * p.up) * p.Lin * 4 < ((int64_t)1 << 31), "bfo3 convt: tensor too large for 32-bit offsets");
    if (p.Lin <= 0) return 0;
    conv_log("bfo3_convt", 2, p.Cin, p.Cout, p.Lin, p.batch, 0, 0, p.len_mul, p.lens != nullptr, p.up);
    if (p.up == 8) {
        // 64 output rows per block (two row tiles): half the re-staging of the window across grid.y (ups0 228 -> 217 us, ups1 374 -> 365)
        if (p.Cin > 256) return bfo3_launch_convt_cfg<8, 2, 1, 1>(p, stream);      // ups0: 512 -> 256, 32 positions per block (64 x 34 x 32 B)
        return bfo3_launch_convt_cfg<8, 2, 2, 2>(p, stream);                        // ups1: 256 -> 128, 64 positions
    }
    if (p.Cout % 64 == 0) return bfo3_launch_convt_cfg<2, 2, 4, 4>(p, stream);      // ups2: 128 -> 64, 128 positions
    return bfo3_launch_convt_cfg<2, 1, 8, 4>(p, stream);                            // ups3: 64 -> 32, 256 positions
}

// =====================================================================================================================
// layout converters and the HiFi-GAN tail
// =====================================================================================================================
__global__ __launch_bounds__(256) void bfo3_pack_kernel(const float* __restrict__ x, int C, int L, float slope, uint4* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (t >= L) return;
    const float* xr = x + ((int64_t)b * C + 8 * o) * L + t;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (8 * o + e < C) ? bfo_lrelu(xr[(int64_t)e * L], slope) : 0.f;
    uint4* dst = out + (((int64_t)b * gridDim.y + o) * L + t) * 2;
    dst[0] = __builtin_bit_cast(uint4, bfo3_split4(v[0], v[1], v[2], v[3], -1));
    dst[1] = __builtin_bit_cast(uint4, bfo3_split4(v[4], v[5], v[6], v[7], -1));
}

__global__ __launch_bounds__(256) void bfo3_unpack_kernel(const uint4* __restrict__ in, int C, int L, float inv_slope, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (t >= L) return;
    const uint4* src = in + (((int64_t)b * gridDim.y + o) * L + t) * 2;
    const uint4 w0 = src[0], w1 = src[1];
    float v[8];
    bfo3_join4((int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, *reinterpret_cast<float (*)[4]>(v));
    bfo3_join4((int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w, *reinterpret_cast<float (*)[4]>(v + 4));
    float* yr = out + ((int64_t)b * C + 8 * o) * L + t;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (8 * o + e < C) yr[(int64_t)e * L] = (inv_slope >= 1.f ? bfo_unrelu(v[e], inv_slope) : bfo_lrelu(v[e], inv_slope));
}

// LayerNorm over the channel axis of a channel-first fp32 tensor (transformer.py:88,158,174,176; model.py:56) that ALSO writes its
// result as an x3 tensor: the input copy of the conv that follows (FastPitch in this mode).  The arithmetic is that of
// layernorm_cf_octet_kernel (bfo_conv.hip): block = 32 positions x 8 channel groups, group g owns the contiguous channels
// [g C/8, (g+1) C/8) = C/64 whole octets, so its 32-byte entries are complete in one thread.
template <int CPG>
__global__ __launch_bounds__(256) void layernorm_cf_x3_kernel(const float* __restrict__ x, float* __restrict__ y, uint4* __restrict__ yo,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const int64_t* __restrict__ lens, int apply_mask, int C, int S, float eps) {
    __shared__ float red[8][32];
    const int b = blockIdx.y;
    const int tl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + tl;
    const bool ok = t < S;
    const int c0 = g * CPG;
    const float* xb = x + ((int64_t)b * C + c0) * S + (ok ? t : 0);
    float v[CPG];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPG; ++i) {
        v[i] = xb[(int64_t)i * S];
        sum += v[i];
    }
    red[g][tl] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += red[k][tl];
    const float mean = tot / (float)C;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPG; ++i) {
        const float d = v[i] - mean;
        sq = fmaf(d, d, sq);
    }
    red[g][tl] = sq;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += red[k][tl];
    const float rstd = 1.0f / sqrtf(tot / (float)C + eps);
    if (!ok) return;
    float m = 1.f;
    if (apply_mask && lens && t >= (int)lens[b]) m = 0.f;
    float* yb = y + ((int64_t)b * C + c0) * S + t;
#pragma unroll
    for (int i = 0; i < CPG; ++i) {
        v[i] = ((v[i] - mean) * rstd * gamma[c0 + i] + beta[c0 + i]) * m;
        yb[(int64_t)i * S] = v[i];
    }
#pragma unroll
    for (int o = 0; o < CPG / 8; ++o) {
        uint4* dst = yo + (((int64_t)b * (C / 8) + c0 / 8 + o) * S + t) * 2;
        dst[0] = __builtin_bit_cast(uint4, bfo3_split4(v[8 * o], v[8 * o + 1], v[8 * o + 2], v[8 * o + 3], -1));
        dst[1] = __builtin_bit_cast(uint4, bfo3_split4(v[8 * o + 4], v[8 * o + 5], v[8 * o + 6], v[8 * o + 7], -1));
    }
}

int32_t launch_layernorm_cf_x3(const float* x, float* y, void* y_x3, const float* gamma, const float* beta, const int64_t* lens,
                               int32_t apply_mask, int32_t B, int32_t C, int32_t S, hipStream_t s, float eps) {
    TTS_REQUIRE((C == 384 || C == 256 || C == 512) && y_x3, "layernorm (x3): built for 256 / 384 / 512 channels (C=%d)", C);
    if (S <= 0 || B <= 0) return 0;
    dim3 grid((S + 31) / 32, B);
    if (C == 384) hipLaunchKernelGGL(layernorm_cf_x3_kernel<48>, grid, dim3(256), 0, s, x, y, (uint4*)y_x3, gamma, beta, lens, apply_mask, C, S, eps);
    else if (C == 256) hipLaunchKernelGGL(layernorm_cf_x3_kernel<32>, grid, dim3(256), 0, s, x, y, (uint4*)y_x3, gamma, beta, lens, apply_mask, C, S, eps);
    else hipLaunchKernelGGL(layernorm_cf_x3_kernel<64>, grid, dim3(256), 0, s, x, y, (uint4*)y_x3, gamma, beta, lens, apply_mask, C, S, eps);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int32_t bfo3_launch_pack(const float* x, int32_t B, int32_t C, int32_t L, float slope, void* out, hipStream_t s) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    dim3 grid((L + 255) / 256, (C + 7) / 8, B);
    hipLaunchKernelGGL(bfo3_pack_kernel, grid, dim3(256), 0, s, x, C, L, slope, (uint4*)out);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int32_t bfo3_launch_unpack(const void* in, int32_t B, int32_t C, int32_t L, float slope, float* out, hipStream_t s) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    dim3 grid((L + 255) / 256, (C + 7) / 8, B);
    hipLaunchKernelGGL(bfo3_unpack_kernel, grid, dim3(256), 0, s, (const uint4*)in, C, L, 1.f / slope, out);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// wave[b][t] = tanh(b0 + sum_{c,k} w[c][k] a[b][c][t + k - 3]); a = the stage sum, stored activated with slope 0.01.
// HBM-bound (128 B read, 4 B written per sample): 256 samples per block, the (256 + 6) x C/8 entries go through LDS once
// as fp32 (hi + lo joined on the way in).
template <int NO>
__global__ __launch_bounds__(256) void bfo3_conv_post_kernel(const uint4* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const int64_t* __restrict__ lens,
                                                             int len_mul, int L, float* __restrict__ wave, int64_t wave_bs) {
    __shared__ float4 Xs[NO * 2 * 262];                     // [octet][half][column] -> conflict-free 16-byte reads
    const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    int n = L;
    if (lens) n = min(n, (int)lens[b] * len_mul);
    if (t0 >= n) return;
    for (int u = tid; u < NO * 262 * 2; u += 256) {
        const int e = u >> 1, hk = u & 1;
        const int o = e / 262, col = e - o * 262;
        const int pos = t0 - 3 + col;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (pos >= 0 && pos < n) {
            const uint4 r = x[(((int64_t)b * NO + o) * L + pos) * 2 + hk];
            bfo3_join4((int)r.x, (int)r.y, (int)r.z, (int)r.w, v);
        }
        Xs[(o * 2 + hk) * 262 + col] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    const int t = t0 + tid;
    float acc = bias ? bias[0] : 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int hk = 0; hk < 2; ++hk)
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const float4 a = Xs[(o * 2 + hk) * 262 + tid + k];
                const int c = 8 * o + 4 * hk;
                acc = fmaf(w[c * 7 + k], a.x, acc);
                acc = fmaf(w[(c + 1) * 7 + k], a.y, acc);
                acc = fmaf(w[(c + 2) * 7 + k], a.z, acc);
                acc = fmaf(w[(c + 3) * 7 + k], a.w, acc);
            }
    if (t < n) wave[(int64_t)b * wave_bs + t] = tanhf(acc);
}

int32_t bfo3_launch_conv_post(const void* x, const float* w, const float* bias, const int64_t* lens, int32_t len_mul, int32_t B,
                              int32_t C, int32_t L, float* wave, int64_t wave_bs, hipStream_t s) {
    TTS_REQUIRE(C == 32, "bfo3 conv_post: built for 32 input channels (got %d)", C);
    if (B <= 0 || L <= 0) return 0;
    dim3 grid((L + 255) / 256, B);
    hipLaunchKernelGGL(bfo3_conv_post_kernel<4>, grid, dim3(256), 0, s, (const uint4*)x, w, bias, lens, len_mul, L, wave, wave_bs);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// =====================================================================================================================
// host-side weight packers: w = hi + lo, hi = bf16(w) (RNE), lo = bf16(w - hi)
// =====================================================================================================================
static inline uint16_t bfo3_host_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline void bfo3_host_split(float f, uint16_t& hi, uint16_t& lo) {
    hi = bfo3_host_bf16(f);
    const uint32_t hu = (uint32_t)hi << 16;
    float hf;
    std::memcpy(&hf, &hu, 4);
    lo = bfo3_host_bf16(f - hf);
}

int64_t bfo3_packed_conv_elems(int cout, int cin, int k) {
    return (int64_t)((cin + 15) / 16) * k * 2 * ((cout + 31) & ~31) * 16;
}

// out[((((h K + t) 2 + kk) CoutP + co) 2 + plane) 8 + e] = plane(w[co][16 h + 8 kk + e][t])
void bfo3_pack_conv_weight(const float* w, int cout, int cin, int k, uint16_t* out) {
    const int cp = (cout + 31) & ~31, nh = (cin + 15) / 16;
    for (int h = 0; h < nh; ++h)
        for (int t = 0; t < k; ++t)
            for (int kk = 0; kk < 2; ++kk) {
                uint16_t* dst = out + (((int64_t)h * k + t) * 2 + kk) * cp * 16;
                for (int co = 0; co < cp; ++co)
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 16 * h + 8 * kk + e;
                        uint16_t hi = 0, lo = 0;
                        if (co < cout && ci < cin) bfo3_host_split(w[((int64_t)co * cin + ci) * k + t], hi, lo);
                        dst[co * 16 + e] = hi;
                        dst[co * 16 + 8 + e] = lo;
                    }
            }
}

int64_t bfo3_packed_convt_elems(int cin, int cout, int u) {
    return (int64_t)u * (cin / 16) * 2 * 2 * ((cout + 31) & ~31) * 16;
}

// torch ConvTranspose1d weight [Cin][Cout][2u]: phase rho, tap t2 -> kernel index (rho + u/2) % u + t2 u
void bfo3_pack_convt_weight(const float* w, int cin, int cout, int u, uint16_t* out) {
    const int cp = (cout + 31) & ~31, nh = cin / 16, kt = 2 * u, pd = u / 2;
    for (int rho = 0; rho < u; ++rho) {
        const int ka = (rho + pd) % u;
        for (int h = 0; h < nh; ++h)
            for (int t2 = 0; t2 < 2; ++t2)
                for (int kk = 0; kk < 2; ++kk) {
                    uint16_t* dst = out + ((((int64_t)rho * nh + h) * 2 + t2) * 2 + kk) * cp * 16;
                    for (int co = 0; co < cp; ++co)
                        for (int e = 0; e < 8; ++e) {
                            const int ci = 16 * h + 8 * kk + e;
                            uint16_t hi = 0, lo = 0;
                            if (co < cout) bfo3_host_split(w[((int64_t)ci * cout + co) * kt + ka + t2 * u], hi, lo);
                            dst[co * 16 + e] = hi;
                            dst[co * 16 + 8 + e] = lo;
                        }
                }
    }
}

}  // namespace ttsamd
