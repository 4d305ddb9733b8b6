// bf16 octet engine (bfo.hpp): the layers around the fused ResBlock pairs --
//   bfo_conv1d   Conv1d with a C-in slab loop (conv_pre 80 -> 512 k7; the C = 256 ResBlock convs of HiFi-GAN stage 1)
//   bfo_convt    ConvTranspose1d(stride u, kernel 2u, padding u/2) as u polyphase 2-tap convs over ONE staged window,
//                phases interleaved through LDS so that the stores are whole 16-byte entries in position order
//   bfo_pack / bfo_unpack   fp32 channel-first <-> bf16 octet layout
//   bfo_conv_post           leaky_relu(0.01) -> Conv1d(C -> 1, k7) -> tanh on an octet tensor
// and the host-side weight packers.  Reference ops: vocoder/hifigan/models.py:46-53 (ResBlock1), :96-99,114-115
// (upsamplers), :112 (conv_pre), :123-125 (conv_post).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bfo.hpp"
#include "kernels.hpp"

namespace ttsamd {

// =====================================================================================================================
// Conv1d: block = 4 waves, wave = 32 rows x 256 columns (8 accumulators), WM row slabs x WN column slabs; the input goes
// through LDS in slabs of SH 16-channel groups (<= 78 KB: two blocks per CU), weights stream from L2 (bfo_mma).
// =====================================================================================================================
template <int K, int WM, int WN, int NT_>
struct BfoConvGeo {
    static constexpr int NT = NT_;                           // 32-column tiles per wave: 8, or 2 for short sequences (FastPitch encoder)
    static constexpr int NCOLS = WN * NT * 32;
    static constexpr int SH = 8 / WN;                        // 16-channel groups per slab
    static constexpr int WS = NCOLS + (K - 1) * BFO_DMAX;
    static constexpr int NE = 2 * SH * WS;
    static constexpr int NXI = (NE + 255) / 256;
    static constexpr int PH = K <= 3 ? 2 : 1;
    static constexpr size_t LDS = (size_t)NE * 16;
};

template <int K, int WM, int WN, int NT_, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void bfo_conv1d(const BfoConvParams p) {
    using G = BfoConvGeo<K, WM, WN, NT_>;
    constexpr int NT = G::NT, WS = G::WS, NXI = G::NXI, SH = G::SH;
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
    const int nrb = gridDim.y / p.ksplit;                    // row blocks; blockIdx.y = row block + nrb * (split-K slice)
    const int ks = blockIdx.y / nrb;
    const int co0 = (blockIdx.y - ks * nrb) * (32 * WM) + 32 * wm;       // this wave's first output row
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
    const bfo_i4 xrs = bfo_rsrc((const char*)p.x + (int64_t)b * NOI * L * 16, (unsigned)NOI * L * 16);
    const bfo_i4 wrs = bfo_rsrc(p.w, (unsigned)NHT * K * 2 * CoutP * 16);
    const int wv = (kk * CoutP + co0 + l31) * 16;
    const int cw = wn * (NT * 32) + l31;
    const uint4* sB = Xs + kk * WS + cw;

    bfo_f16 acc[NT];
    {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bv[r] = (p.bias && p.ksplit == 1) ? p.bias[min(co0 + 8 * (r >> 2) + 4 * kk + (r & 3), p.Cout - 1)] : 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = bv[r];
    }
    const int n_slabs = (NHT + SH - 1) / SH;
    const int sl_beg = ks * n_slabs / p.ksplit, sl_end = (ks + 1) * n_slabs / p.ksplit;
    for (int s0 = sl_beg * SH; s0 < min(sl_end * SH, NHT); s0 += SH) {
        const int nh = min(SH, NHT - s0);
        if (s0 > sl_beg * SH) __syncthreads();              // the previous slab has been consumed
        // staged 8 entries per thread at a time: the accumulators are live here, a whole slab in flight would spill
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                     // opaque: the per-entry index math must not be hoisted out of the
                                                            // slab loop (50 VGPRs held across the MFMAs otherwise)
#pragma unroll
        for (int i0 = 0; i0 < NXI; i0 += 8) {
            bfo_i4 xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid_o + 256 * (i0 + i);
                const int ol = e / WS, col = e - ol * WS;
                const int o = 2 * s0 + ol, pos = x0 + col;
                const bool ok = i0 + i < NXI && ol < 2 * nh && o < NOI && col < W1 && pos >= 0 && pos < len;
                xv[i] = bfo_ld16(xrs, ok ? (o * L + pos) * 16 : BFO_OOB, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid_o + 256 * (i0 + i);
                if (i0 + i < NXI && e < G::NE) Xs[e] = __builtin_bit_cast(uint4, xv[i]);
            }
        }
        __syncthreads();
        bfo_mma<K, G::PH, NT>(acc, wrs, wv, 2 * CoutP * 16, sB, nh, 2 * WS, dil, s0);
    }

    // ---- epilogue: [+ residual] [+ running sum] [/ div], activation of the consumer, 8-byte stores from the C layout
    if (co0 >= p.Cout) return;
    if (p.ksplit > 1) {
        // raw partial sums, fp32 channel-first [ks][b][Cout][L]; bias / residual / activation happen in bfo_splitk_reduce
        const bfo_i4 prs = bfo_rsrc(p.splitk_ws + ((int64_t)ks * p.batch + b) * p.Cout * L, (unsigned)p.Cout * L * 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int q = q0 + cw + 32 * j;
            const int vq = q < len_o ? q * 4 + 4 * kk * L * 4 : BFO_OOB;
#pragma unroll
            for (int r = 0; r < 16; ++r) bfo_st4f(acc[j][r], prs, vq, (co0 + 8 * (r >> 2) + (r & 3)) * L * 4, 0);
        }
        return;
    }
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
    int vo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = q0 + cw + 32 * j;
        vo[j] = q < len_o ? q * 16 + 8 * kk : BFO_OOB;
    }
    const bfo_i4 yrs = bfo_rsrc((char*)p.y + (int64_t)b * NOO * L * 16, (unsigned)NOO * L * 16);
    const int so0 = (co0 >> 3) * L * 16;
    const float os = p.out_slope;
    const float sc = p.mode == 2 ? 1.f / p.div : 1.f;
    const bool has_res = p.res != nullptr, has_sum = p.mode != 0;
    const bfo_i4 rrs = bfo_rsrc((const char*)(has_res ? p.res : p.y) + (int64_t)b * NOO * L * 16, (unsigned)NOO * L * 16);
    const bfo_i4 srs = bfo_rsrc((const char*)(has_sum ? p.sum_in : p.y) + (int64_t)b * NOO * L * 16, (unsigned)NOO * L * 16);
    const float rinv = has_res ? 1.f / p.res_slope : 1.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bfo_i2 rv[4], sv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // branch-free: a masked-off buffer load returns 0 without touching memory
            rv[g] = bfo_ld8(rrs, has_res ? vo[j] : BFO_OOB, so0 + g * L * 16, 0);
            sv[g] = bfo_ld8(srs, has_sum ? vo[j] : BFO_OOB, so0 + g * L * 16, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4] = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            const float r[4] = {bfo_lo(rv[g].x), bfo_hi(rv[g].x), bfo_lo(rv[g].y), bfo_hi(rv[g].y)};
            const float s[4] = {bfo_lo(sv[g].x), bfo_hi(sv[g].x), bfo_lo(sv[g].y), bfo_hi(sv[g].y)};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] + bfo_unrelu(r[e], rinv) + s[e]) * sc;
            bfo_st8(bfo_act4(v[0], v[1], v[2], v[3], os, -1), yrs, vo[j], so0 + g * L * 16, 0);
        }
        if (j & 1) __builtin_amdgcn_sched_barrier(0);       // two tiles' loads in flight, not all eight (128 registers)
    }
}

// Second half of a split-K conv: thread = (octet, position); y = epilogue(sum_ks partial[ks]) in slice order, with exactly the
// epilogue of bfo_conv1d (bias, fp32 or octet residual, running sum, / div, activation; fp32 channel-first or octet output).
__global__ __launch_bounds__(256) void bfo_splitk_reduce(const BfoConvParams p) {
    const int t = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    const int L = p.Lin;
    int len = L;
    if (p.lens && !p.out_all) len = min(len, (int)p.lens[b] * p.len_mul);
    if (t >= len) return;
    const int NOO = p.Cout / 8;
    const int64_t per = (int64_t)p.batch * p.Cout * L;
    const float* pp = p.splitk_ws + ((int64_t)b * p.Cout + 8 * o) * L + t;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int ks = 0; ks < p.ksplit; ++ks) {                 // slice by slice, the eight channels' loads in flight together
        float tq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tq[e] = pp[ks * per + (int64_t)e * L];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += tq[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += p.bias ? p.bias[8 * o + e] : 0.f;
    if (p.y_f32) {
        float* yp = p.y_f32 + ((int64_t)b * p.Cout + 8 * o) * L + t;
        const float* rp = p.res_f32 ? p.res_f32 + ((int64_t)b * p.Cout + 8 * o) * L + t : nullptr;
#pragma unroll
        for (int e = 0; e < 8; ++e) yp[(int64_t)e * L] = bfo_lrelu(v[e] + (rp ? rp[(int64_t)e * L] : 0.f), p.out_slope);
        return;
    }
    const int64_t ent = ((int64_t)b * NOO + o) * L + t;
    if (p.res) {
        const uint4 r = reinterpret_cast<const uint4*>(p.res)[ent];
        const float rinv = 1.f / p.res_slope;
        const float rr[8] = {bfo_lo((int)r.x), bfo_hi((int)r.x), bfo_lo((int)r.y), bfo_hi((int)r.y),
                             bfo_lo((int)r.z), bfo_hi((int)r.z), bfo_lo((int)r.w), bfo_hi((int)r.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bfo_unrelu(rr[e], rinv);
    }
    if (p.mode != 0) {
        const uint4 r = reinterpret_cast<const uint4*>(p.sum_in)[ent];
        const float ss[8] = {bfo_lo((int)r.x), bfo_hi((int)r.x), bfo_lo((int)r.y), bfo_hi((int)r.y),
                             bfo_lo((int)r.z), bfo_hi((int)r.z), bfo_lo((int)r.w), bfo_hi((int)r.w)};
        const float sc = p.mode == 2 ? 1.f / p.div : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + ss[e]) * sc;
    }
    const bfo_i2 lo = bfo_act4(v[0], v[1], v[2], v[3], p.out_slope, -1), hi = bfo_act4(v[4], v[5], v[6], v[7], p.out_slope, -1);
    uint4 w;
    w.x = (unsigned)lo.x; w.y = (unsigned)lo.y; w.z = (unsigned)hi.x; w.w = (unsigned)hi.y;
    reinterpret_cast<uint4*>(p.y)[ent] = w;
}

constexpr int LNO_MAXV_R = 64;

template <int K, int WM, int WN, int NT, bool OUT_F32>
static int32_t bfo_launch_conv_cfg(const BfoConvParams& p_in, hipStream_t stream) {
    BfoConvParams p = p_in;
    using G = BfoConvGeo<K, WM, WN, NT>;
    static std::atomic<uint64_t> lds_done{0};          // per instantiation: devices already opted in (common.hpp: lds_opt_in)
    TTS_CHECK_HIP(lds_opt_in((const void*)bfo_conv1d<K, WM, WN, NT, OUT_F32>, (int)G::LDS, lds_done));
    const int CoutP = (p.Cout + 31) & ~31;
    dim3 grid((p.Lin + G::NCOLS - 1) / G::NCOLS, (CoutP + 32 * WM - 1) / (32 * WM), p.batch);
    // split K when the tile grid leaves most CUs without a block and the reduction depth allows it (batch 1 / 8)
    p.ksplit = 1;
    p.compact = (compact_order(p.lens, p.batch) && !p.out_all) ? 1 : 0;
    const int64_t blocks = (int64_t)grid.x * grid.y * grid.z;
    const int n_slabs = ((p.Cin + 15) / 16 + G::SH - 1) / G::SH;
    const int64_t per = (int64_t)p.batch * p.Cout * p.Lin;
    const char* ske = opt_str(OPT_BFO_SPLITK);              // 0 disables (A/B and parity runs)
    const char* mse = exp_env("TTSAMD_BFO_SPLITK_MIN_SLABS");
    const int min_slabs = mse ? atoi(mse) : 4;
    const char* mbe = exp_env("TTSAMD_BFO_SPLITK_BLOCKS");
    const int max_blocks = mbe ? atoi(mbe) : 256;   // under one block per CU
    if (p.splitk_ws && blocks < max_blocks && n_slabs >= min_slabs && !(ske && ske[0] == '0')) {
        const char* mk = exp_env("TTSAMD_BFO_SPLITK_MAX");
        int64_t ks = std::min<int64_t>((256 + blocks - 1) / blocks, n_slabs);
        ks = std::min<int64_t>(ks, mk ? atoi(mk) : 4);
        ks = std::min<int64_t>(ks, p.splitk_floats / std::max<int64_t>(per, 1));
        if (ks >= 2) p.ksplit = (int)ks;
    }
    grid.y *= p.ksplit;
    hipLaunchKernelGGL((bfo_conv1d<K, WM, WN, NT, OUT_F32>), grid, dim3(256), G::LDS, stream, p);
    TTS_CHECK_HIP(hipGetLastError());
    const bool ln = p.ln_g != nullptr;
    if (ln) TTS_REQUIRE(OUT_F32 && p.y_f32 && p.ln_b && p.ln_octet && p.Cout % 64 == 0 && p.Cout <= 8 * LNO_MAXV_R,
                        "bfo conv: the LayerNorm epilogue needs the fp32 output and Cout %% 64 == 0, <= %d (Cout=%d)", 8 * LNO_MAXV_R, p.Cout);
    // (a reduction that finishes with the LayerNorm in the same launch -- one launch and one pass fewer, same bits -- was built and
    // measured slower at the batch sizes that split K: bf16 batch 1 2.35 vs 2.40 ms, round 4; removed in round 5)
    if (p.ksplit > 1) {
        dim3 rg((p.Lin + 255) / 256, p.Cout / 8, p.batch);
        hipLaunchKernelGGL(bfo_splitk_reduce, rg, dim3(256), 0, stream, p);
        TTS_CHECK_HIP(hipGetLastError());
    }
    if (ln) {
        return launch_layernorm_cf_octet(p.y_f32, p.y_f32, p.ln_octet, p.ln_g, p.ln_b, p.ln_lens, p.ln_lens != nullptr, p.batch, p.Cout,
                                         p.Lin, stream);
    }
    return 0;
}

template <int K>
static int32_t bfo_launch_conv_k(const BfoConvParams& p, hipStream_t stream) {
    // short sequences (FastPitch encoder: 64 tokens per utterance): 64-column tiles instead of 256
    // ... and for small grids: with 256-column tiles a batch-1 decoder conv is 2 column tiles
    const int64_t blocks8 = (int64_t)((p.Lin + 255) / 256) * ((p.Cout + 127) / 128) * p.batch;
    // (a grid of 256-column tiles under one block per CU -- FastPitch's 1536 -> 384 conv at batch 32 is 192 blocks -- takes the 64-column
    // tiles too: bf16 one-stream step 11.89 -> 11.57 ms, two-stream unchanged; round 3 had the threshold at 48 blocks)
    static const int64_t narrow_blocks = [] { const char* e = exp_env("TTSAMD_BFO_NARROW_BLOCKS"); return e ? (int64_t)atoi(e) : (int64_t)256; }();
    const bool narrow = p.Cout >= 128 && (p.Lin <= 96 || blocks8 < narrow_blocks);
    if (p.y_f32) {
        if (narrow) return bfo_launch_conv_cfg<K, 4, 1, 2, true>(p, stream);
        if (p.Cout >= 128) return bfo_launch_conv_cfg<K, 4, 1, 8, true>(p, stream);
        set_error("bfo conv: fp32 output is built for Cout >= 128 (got %d)", p.Cout);
        return TTSAMD_EINVAL;
    }
    if (narrow) return bfo_launch_conv_cfg<K, 4, 1, 2, false>(p, stream);
    if (p.Cout >= 128) return bfo_launch_conv_cfg<K, 4, 1, 8, false>(p, stream);
    if (p.Cout >= 64) return bfo_launch_conv_cfg<K, 2, 2, 8, false>(p, stream);
    return bfo_launch_conv_cfg<K, 1, 4, 8, false>(p, stream);
}

int32_t bfo_launch_conv(const BfoConvParams& p, hipStream_t stream) {
    TTS_REQUIRE(p.up == 1, "bfo conv: use bfo_launch_convt for transposed convs");
    TTS_REQUIRE(p.Cin % 8 == 0 && p.Cout % 32 == 0, "bfo conv: Cin %% 8 and Cout %% 32 must be 0 (Cin=%d, Cout=%d)", p.Cin, p.Cout);
    TTS_REQUIRE(p.dil >= 1 && p.dil <= BFO_DMAX, "bfo conv: dilation %d outside [1,%d]", p.dil, BFO_DMAX);
    TTS_REQUIRE((int64_t)std::max(p.Cin, p.Cout) * p.Lin * 2 < ((int64_t)1 << 31), "bfo conv: tensor too large for 32-bit offsets");
    TTS_REQUIRE(p.mode == 0 || p.sum_in != nullptr, "bfo conv: mode %d needs sum_in", p.mode);
    TTS_REQUIRE(!p.y_f32 || (p.mode == 0 && p.res == nullptr), "bfo conv: fp32 output takes res_f32, mode 0");
    if (p.Lin <= 0) return 0;
    conv_log("bfo", p.K, p.Cin, p.Cout, p.Lin, p.batch, p.res != nullptr, p.mode, p.len_mul, p.lens != nullptr, 1);
    switch (p.K) {
        case 1: return bfo_launch_conv_k<1>(p, stream);
        case 3: return bfo_launch_conv_k<3>(p, stream);
        case 7: return bfo_launch_conv_k<7>(p, stream);
        case 11: return bfo_launch_conv_k<11>(p, stream);
        default:
            set_error("bfo conv: kernel size %d not instantiated (1,3,7,11)", p.K);
            return TTSAMD_EINVAL;
    }
}

// =====================================================================================================================
// ConvTranspose1d(stride U, kernel 2U, padding U/2):  y[co][q U + rho] = b[co] + sum_ci W[ci][co][ka] x[ci][q + dl]
//                                                                               + W[ci][co][ka + U] x[ci][q + dl - 1],
// ka = (rho + U/2) % U, dl = (rho + U/2) / U  -> per phase rho a 2-tap conv (dil -1) over the same window.
// Block = (32 RT output rows, NQ = 32 NQT input positions, ALL U phases); the 4 waves split the (phase, row tile, column
// group) combos, NT column tiles each.  The results leave through LDS: entry (octet, q, rho) in rows of U + 1 entries
// (the pad entry spreads the 8-byte C-layout writes over the banks), read back in position order -> 16-byte stores,
// 1 KB contiguous per wave instruction (phase-strided 8-byte stores would write 8 bytes per 128-byte line at U = 8).
// =====================================================================================================================
template <int U, int RT, int NQT, int NT>
__global__ __launch_bounds__(256, 2) void bfo_convt(const BfoConvParams p) {
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
    const bfo_i4 xrs = bfo_rsrc((const char*)p.x + (int64_t)b * NOI * L * 16, (unsigned)NOI * L * 16);

    // ---- stage the window: column c = input position q0 - 1 + c
    {
        const int ne = NOI * WSC;
        for (int e0 = 0; e0 < ne; e0 += 256 * 8) {
            bfo_i4 xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + tid + 256 * i;
                const int o = e / WSC, col = e - o * WSC;
                const int pos = q0 - 1 + col;
                const bool ok = e < ne && pos >= 0 && pos < len;
                xv[i] = bfo_ld16(xrs, ok ? (o * L + pos) * 16 : BFO_OOB, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + tid + 256 * i;
                if (e < ne) Xs[e] = __builtin_bit_cast(uint4, xv[i]);
            }
        }
    }
    __syncthreads();

    bfo_f16 acc[NCALL][NT];
#pragma unroll
    for (int c = 0; c < NCALL; ++c) {
        const int cb = wid + 4 * c;
        const int rho = cb % U, rt = (cb / U) % RT, cg = cb / (U * RT);
        const int dl = (rho + U / 2) / U;
        const int row0 = co0 + 32 * rt;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = p.bias ? p.bias[min(row0 + 8 * (r >> 2) + 4 * kk + (r & 3), p.Cout - 1)] : 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][j][r] = bv[r];
        const bfo_i4 wrs = bfo_rsrc((const char*)p.w + (int64_t)rho * NH * 2 * 2 * CoutP * 16, (unsigned)NH * 2 * 2 * CoutP * 16);
        const int wv = (kk * CoutP + row0 + l31) * 16;
        const uint4* sB = Xs + kk * WSC + cg * (NT * 32) + l31 + dl + 1;      // tap 0 reads x[q + dl], tap 1 x[q + dl - 1]
        bfo_mma<2, 2, NT>(acc[c], wrs, wv, 2 * CoutP * 16, sB, NH, 2 * WSC, -1);
    }
    __syncthreads();                                         // the window is dead

    // ---- results -> LDS [octet (4 RT)][q (NQ)][U + 1 entries], activated, bf16
    {
        const float os = p.out_slope;
#pragma unroll
        for (int c = 0; c < NCALL; ++c) {
            const int cb = wid + 4 * c;
            const int rho = cb % U, rt = (cb / U) % RT, cg = cb / (U * RT);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = cg * (NT * 32) + 32 * j + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bfo_i2 w = bfo_act4(acc[c][j][4 * g], acc[c][j][4 * g + 1], acc[c][j][4 * g + 2], acc[c][j][4 * g + 3], os, -1);
                    *reinterpret_cast<bfo_i2*>(reinterpret_cast<char*>(Xs + ((4 * rt + g) * NQ + n) * (U + 1) + rho) + 8 * kk) = w;
                }
            }
        }
    }
    __syncthreads();
    // ---- position-ordered 16-byte stores
    {
        const bfo_i4 yrs = bfo_rsrc((char*)p.y + (int64_t)b * NOO * Lo * 16, (unsigned)NOO * Lo * 16);
        constexpr int NOUT = 4 * RT * NQ * U;
#pragma unroll 4
        for (int i = 0; i < NOUT / 256; ++i) {
            const int idx = tid + 256 * i;
            const int og = idx / (NQ * U), r = idx - og * (NQ * U);
            const int n = r / U, rho = r - n * U;
            const uint4 v = Xs[(og * NQ + n) * (U + 1) + rho];
            const int o = (co0 >> 3) + og;
            const bool ok = q0 + n < len && o < NOO;
            bfo_st16(__builtin_bit_cast(bfo_i4, v), yrs, ok ? (o * Lo + q0 * U + r) * 16 : BFO_OOB, 0, 0);
        }
    }
}

template <int U, int RT, int NQT, int NT>
static int32_t bfo_launch_convt_cfg(const BfoConvParams& p, hipStream_t stream) {
    constexpr int NQ = 32 * NQT;
    const size_t lds = std::max((size_t)(p.Cin / 8) * (NQ + 2) * 16, (size_t)4 * RT * NQ * (U + 1) * 16);
    TTS_REQUIRE(lds <= 80 * 1024, "bfo convt: window of %zu bytes does not fit (Cin=%d, u=%d)", lds, p.Cin, U);
    static std::atomic<uint64_t> lds_done{0};
    TTS_CHECK_HIP(lds_opt_in((const void*)bfo_convt<U, RT, NQT, NT>, 80 * 1024, lds_done));
    dim3 grid((p.Lin + NQ - 1) / NQ, (p.Cout + 32 * RT - 1) / (32 * RT), p.batch);
    BfoConvParams q = p;
    q.compact = compact_order(p.lens, p.batch) ? 1 : 0;
    hipLaunchKernelGGL((bfo_convt<U, RT, NQT, NT>), grid, dim3(256), lds, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int32_t bfo_launch_convt(const BfoConvParams& p, hipStream_t stream) {
    TTS_REQUIRE(p.up == 8 || p.up == 2, "bfo convt: stride %d not built (8, 2)", p.up);
    TTS_REQUIRE(p.Cin % 16 == 0 && p.Cout % 32 == 0, "bfo convt: Cin %% 16 and Cout %% 32 must be 0 (Cin=%d, Cout=%d)", p.Cin, p.Cout);
    TTS_REQUIRE((int64_t)std::max(p.Cin, p.Cout * p.up) * p.Lin * 2 < ((int64_t)1 << 31), "bfo convt: tensor too large for 32-bit offsets");
    if (p.Lin <= 0) return 0;
    conv_log("bfo_convt", 2, p.Cin, p.Cout, p.Lin, p.batch, 0, 0, p.len_mul, p.lens != nullptr, p.up);
    if (p.up == 8) {
        if (p.Cin > 256) return bfo_launch_convt_cfg<8, 1, 2, 2>(p, stream);      // ups0: 512 -> 256, 64 positions per block
        return bfo_launch_convt_cfg<8, 1, 4, 4>(p, stream);                        // ups1: 256 -> 128, 128 positions
    }
    if (p.Cout % 64 == 0) return bfo_launch_convt_cfg<2, 2, 4, 4>(p, stream);      // ups2: 128 -> 64
    return bfo_launch_convt_cfg<2, 1, 8, 4>(p, stream);                            // ups3: 64 -> 32
}

// =====================================================================================================================
// layout converters and the HiFi-GAN tail
// =====================================================================================================================
__global__ __launch_bounds__(256) void bfo_pack_kernel(const float* __restrict__ x, int C, int L, float slope, uint4* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (t >= L) return;
    const float* xr = x + ((int64_t)b * C + 8 * o) * L + t;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (8 * o + e < C) ? bfo_lrelu(xr[(int64_t)e * L], slope) : 0.f;
    uint4 w;
    w.x = (unsigned)bfo_pk(v[0], v[1]); w.y = (unsigned)bfo_pk(v[2], v[3]);
    w.z = (unsigned)bfo_pk(v[4], v[5]); w.w = (unsigned)bfo_pk(v[6], v[7]);
    out[((int64_t)b * gridDim.y + o) * L + t] = w;
}

__global__ __launch_bounds__(256) void bfo_unpack_kernel(const uint4* __restrict__ in, int C, int L, float inv_slope, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (t >= L) return;
    const uint4 w = in[((int64_t)b * gridDim.y + o) * L + t];
    const float v[8] = {bfo_lo((int)w.x), bfo_hi((int)w.x), bfo_lo((int)w.y), bfo_hi((int)w.y),
                        bfo_lo((int)w.z), bfo_hi((int)w.z), bfo_lo((int)w.w), bfo_hi((int)w.w)};
    float* yr = out + ((int64_t)b * C + 8 * o) * L + t;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (8 * o + e < C) yr[(int64_t)e * L] = (inv_slope >= 1.f ? bfo_unrelu(v[e], inv_slope) : bfo_lrelu(v[e], inv_slope));
}

// LayerNorm over the channel axis of a channel-first fp32 tensor (transformer.py:88,158,174,176) that ALSO writes its result as
// octet bf16 entries: the input copy of the conv that follows (FastPitch under config 3).  Block = 32 positions x 8 channel groups;
// group g owns the CONTIGUOUS channels [g C/8, (g+1) C/8) = C/64 whole octets, so its entries are complete in one thread.
constexpr int LNO_MAXV = 64;
// CPG_ = C / 8 as a compile-time constant (48: d_model 384, 32: the predictors' 256), or 0 = run-time bound: then every load sits
// behind its own branch and vmcnt(0) (48 serial round trips: 10-12 us per launch at batch 1 against 5-6)
template <int CPG_>
__global__ __launch_bounds__(256) void layernorm_cf_octet_kernel(const float* __restrict__ x, float* __restrict__ y, uint4* __restrict__ yo,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const int64_t* __restrict__ lens, int apply_mask, int C, int S, float eps) {
    __shared__ float red[8][32];
    const int b = blockIdx.y;
    const int tl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + tl;
    const bool ok = t < S;
    const int cpg = CPG_ ? CPG_ : C / 8, c0 = g * cpg;
    constexpr int NV = CPG_ ? CPG_ : LNO_MAXV;
    const float* xb = x + ((int64_t)b * C + c0) * S + (ok ? t : 0);
    float v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = (i < cpg) ? xb[(int64_t)i * S] : 0.f;
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
    for (int i = 0; i < NV; ++i) {
        const float d = (i < cpg) ? v[i] - mean : 0.f;
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
    for (int i = 0; i < NV; ++i)
        if (i < cpg) {
            v[i] = ((v[i] - mean) * rstd * gamma[c0 + i] + beta[c0 + i]) * m;
            yb[(int64_t)i * S] = v[i];
        }
#pragma unroll
    for (int o = 0; o < NV / 8; ++o)
        if (8 * o < cpg) {
            uint4 w;
            w.x = (unsigned)bfo_pk(v[8 * o], v[8 * o + 1]); w.y = (unsigned)bfo_pk(v[8 * o + 2], v[8 * o + 3]);
            w.z = (unsigned)bfo_pk(v[8 * o + 4], v[8 * o + 5]); w.w = (unsigned)bfo_pk(v[8 * o + 6], v[8 * o + 7]);
            yo[((int64_t)b * (C / 8) + c0 / 8 + o) * S + t] = w;
        }
}

int32_t launch_layernorm_cf_octet(const float* x, float* y, void* y_octet, const float* gamma, const float* beta,
                                  const int64_t* lens, int32_t apply_mask, int32_t B, int32_t C, int32_t S, hipStream_t s, float eps) {
    TTS_REQUIRE(C % 64 == 0 && C <= 8 * LNO_MAXV && y_octet, "layernorm (octet): C %% 64 must be 0 and C <= %d (C=%d)", 8 * LNO_MAXV, C);
    if (S <= 0 || B <= 0) return 0;
    dim3 grid((S + 31) / 32, B);
    if (C == 384) hipLaunchKernelGGL(layernorm_cf_octet_kernel<48>, grid, dim3(256), 0, s, x, y, (uint4*)y_octet, gamma, beta, lens, apply_mask, C, S, eps);
    else if (C == 256) hipLaunchKernelGGL(layernorm_cf_octet_kernel<32>, grid, dim3(256), 0, s, x, y, (uint4*)y_octet, gamma, beta, lens, apply_mask, C, S, eps);
    else hipLaunchKernelGGL(layernorm_cf_octet_kernel<0>, grid, dim3(256), 0, s, x, y, (uint4*)y_octet, gamma, beta, lens, apply_mask, C, S, eps);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int32_t bfo_launch_pack(const float* x, int32_t B, int32_t C, int32_t L, float slope, void* out, hipStream_t s) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    dim3 grid((L + 255) / 256, (C + 7) / 8, B);
    hipLaunchKernelGGL(bfo_pack_kernel, grid, dim3(256), 0, s, x, C, L, slope, (uint4*)out);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int32_t bfo_launch_unpack(const void* in, int32_t B, int32_t C, int32_t L, float slope, float* out, hipStream_t s) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    dim3 grid((L + 255) / 256, (C + 7) / 8, B);
    hipLaunchKernelGGL(bfo_unpack_kernel, grid, dim3(256), 0, s, (const uint4*)in, C, L, 1.f / slope, out);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// wave[b][t] = tanh(b0 + sum_{c,k} w[c][k] a[b][c][t + k - 3]); a = the stage sum, stored activated with slope 0.01.
// HBM-bound (64 B read, 4 B written per sample): 256 samples per block, the (256 + 6) x C/8 entries go through LDS once.
template <int NO>
__global__ __launch_bounds__(256) void bfo_conv_post_kernel(const uint4* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const int64_t* __restrict__ lens,
                                                            int len_mul, int L, float* __restrict__ wave, int64_t wave_bs) {
    __shared__ uint4 Xs[NO * 262];
    const int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    int n = L;
    if (lens) n = min(n, (int)lens[b] * len_mul);
    if (t0 >= n) return;
    for (int e = tid; e < NO * 262; e += 256) {
        const int o = e / 262, col = e - o * 262;
        const int pos = t0 - 3 + col;
        Xs[e] = (pos >= 0 && pos < n) ? x[((int64_t)b * NO + o) * L + pos] : make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int t = t0 + tid;
    float acc = bias ? bias[0] : 0.f;
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const uint4 v = Xs[o * 262 + tid + k];
            const float a[8] = {bfo_lo((int)v.x), bfo_hi((int)v.x), bfo_lo((int)v.y), bfo_hi((int)v.y),
                                bfo_lo((int)v.z), bfo_hi((int)v.z), bfo_lo((int)v.w), bfo_hi((int)v.w)};
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(w[(8 * o + e) * 7 + k], a[e], acc);
        }
    if (t < n) wave[(int64_t)b * wave_bs + t] = tanhf(acc);
}

int32_t bfo_launch_conv_post(const void* x, const float* w, const float* bias, const int64_t* lens, int32_t len_mul, int32_t B,
                             int32_t C, int32_t L, float* wave, int64_t wave_bs, hipStream_t s) {
    TTS_REQUIRE(C == 32, "bfo conv_post: built for 32 input channels (got %d)", C);
    if (B <= 0 || L <= 0) return 0;
    dim3 grid((L + 255) / 256, B);
    hipLaunchKernelGGL(bfo_conv_post_kernel<4>, grid, dim3(256), 0, s, (const uint4*)x, w, bias, lens, len_mul, L, wave, wave_bs);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// =====================================================================================================================
// host-side weight packers
// =====================================================================================================================
static inline uint16_t bfo_host_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int64_t bfo_packed_conv_elems(int cout, int cin, int k) {
    return (int64_t)((cin + 15) / 16) * k * 2 * ((cout + 31) & ~31) * 8;
}

// out[(((h K + t) 2 + kk) CoutP + co) 8 + e] = w[co][16 h + 8 kk + e][t]
void bfo_pack_conv_weight(const float* w, int cout, int cin, int k, uint16_t* out) {
    const int cp = (cout + 31) & ~31, nh = (cin + 15) / 16;
    for (int h = 0; h < nh; ++h)
        for (int t = 0; t < k; ++t)
            for (int kk = 0; kk < 2; ++kk) {
                uint16_t* dst = out + (((int64_t)h * k + t) * 2 + kk) * cp * 8;
                for (int co = 0; co < cp; ++co)
                    for (int e = 0; e < 8; ++e) {
                        const int ci = 16 * h + 8 * kk + e;
                        dst[co * 8 + e] = (co < cout && ci < cin) ? bfo_host_bf16(w[((int64_t)co * cin + ci) * k + t]) : 0;
                    }
            }
}

int64_t bfo_packed_convt_elems(int cin, int cout, int u) {
    return (int64_t)u * (cin / 16) * 2 * 2 * ((cout + 31) & ~31) * 8;
}

// torch ConvTranspose1d weight [Cin][Cout][2u]: phase rho, tap t2 -> kernel index (rho + u/2) % u + t2 u
void bfo_pack_convt_weight(const float* w, int cin, int cout, int u, uint16_t* out) {
    const int cp = (cout + 31) & ~31, nh = cin / 16, kt = 2 * u, pd = u / 2;
    for (int rho = 0; rho < u; ++rho) {
        const int ka = (rho + pd) % u;
        for (int h = 0; h < nh; ++h)
            for (int t2 = 0; t2 < 2; ++t2)
                for (int kk = 0; kk < 2; ++kk) {
                    uint16_t* dst = out + ((((int64_t)rho * nh + h) * 2 + t2) * 2 + kk) * cp * 8;
                    for (int co = 0; co < cp; ++co)
                        for (int e = 0; e < 8; ++e) {
                            const int ci = 16 * h + 8 * kk + e;
                            dst[co * 8 + e] = co < cout ? bfo_host_bf16(w[((int64_t)ci * cout + co) * kt + ka + t2 * u]) : 0;
                        }
                }
    }
}

}  // namespace ttsamd
