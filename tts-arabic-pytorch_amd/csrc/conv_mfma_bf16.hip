// bf16 / split-bf16 MFMA variants of the implicit-GEMM conv engine.  See conv_mfma.hip for the
// structure; this file is derived from it (same tiling, LDS ring, gap interleave).
#include <cstdint>

#include "conv_mfma_common.hpp"

namespace ttsamd {

// ======================================================================================
// bf16 MFMA variants of the same kernel (config 3: bf16 operands, fp32 accumulate; and the
// split-bf16 "3x" mode that keeps fp32-class accuracy).  Identical tiling, staging ring and
// gap interleave; an LDS entry is 4 bf16 (the same 4 channel pairs) and ONE
// v_mfma_f32_32x32x8_bf16_1k consumes it, so a whole octet x tap costs MT*NTL MFMAs of 8
// passes instead of 4*MT*NTL fp32 MFMAs of 16 passes.  Activations stay fp32 in HBM and are
// rounded to bf16 (RNE) on the LDS write; weights are pre-split/packed at create().
// ======================================================================================
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

template <int K, int NT_BLK, int CO_BLK, int NPL>
struct GeoB {
    static constexpr int NOCT = OctsOf<K>::NOCT;
    static constexpr int KC = 8 * NOCT;
    static constexpr int WS = NT_BLK + (K - 1) * DMAX;
    static constexpr int XI = 2 * NOCT * WS;                  // X entries (uint2) per plane per stage
    static constexpr int NXI = (XI + 255) / 256;
    static constexpr int W4 = NOCT * K * 2 * CO_BLK;          // W entries (uint2) per plane per stage
    static constexpr int NW = (W4 + 255) / 256;
    static constexpr int BUF4 = NPL * (XI + W4);              // uint2 per stage
    static constexpr int NSTAGE = (CO_BLK > 32 && 3 * BUF4 * 8 <= 80 * 1024) ? 3 : 2;   // see conv_mfma.hip
    static constexpr int NGRP = NOCT * K;
};

// ACT: GELU / tanh compiled in (Vocos pwconv1, Tacotron2 postnet only): their inline expansions for 64 accumulators
// per lane were most of the 70-100 KB of kernel code (64 KB instruction cache per CU pair)
// EPI 0: row-major float4 epilogue through the dead LDS ring (same as the fp32 kernel's); EPI 2: per-lane epilogue
// (polyphase upsamplers, unaligned rows, packed bf16 output); EPI 3: row epilogue with the residual (+ running sum) PRELOADED
// into the accumulators and a load-free store loop, as conv_mfma.hip's EPI 3
template <int K, int MT, int NTL, int WM, int WN, int NPL, int EPI, bool ACT>
__global__ __launch_bounds__(256, TTS_MINWAVES) void conv1d_mfma_bf16(const ConvParams p) {
    // NPL = 1: plain bf16 operands; NPL = 2: split bf16 (hi + lo planes, 3 MFMAs per product)
    extern __shared__ __attribute__((aligned(16))) uint2 smem4[];
    constexpr int CO_BLK = WM * MT * 32;
    constexpr int NT_BLK = WN * NTL * 32;
    using G = GeoB<K, NT_BLK, CO_BLK, NPL>;
    constexpr int KC = G::KC, WS = G::WS, NXI = G::NXI, NW = G::NW, NGRP = G::NGRP, NSTAGE = G::NSTAGE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // ragged batches: tile-major block order (x = utterance slot, z = time tile), see ConvParams::tile_major
    const int b = p.tile_major ? (int)((blockIdx.x + blockIdx.z) % (unsigned)p.batch) : (int)blockIdx.z;
    const int n_co_tiles = p.CoutP / CO_BLK;
    const int phase = blockIdx.y / n_co_tiles;
    const int co_blk0 = (blockIdx.y % n_co_tiles) * CO_BLK;
    const int q0 = (p.tile_major ? blockIdx.z : blockIdx.x) * NT_BLK;

    int n_out = p.Nout;
    if (p.lens_out) n_out = min(n_out, (int)p.lens_out[b] * p.len_out_mul);
    if (q0 >= n_out) return;
    int in_len = p.Lin;
    if (p.lens_in) in_len = min(in_len, (int)p.lens_in[b] * p.len_in_mul);

    const int dil = p.dil;
    int pad = p.pad;
    if (p.n_phase > 1) pad = -((phase + p.phase_p) / p.n_phase);
    const int span = (K - 1) * (dil < 0 ? -dil : dil);
    const int lo = (dil < 0 ? (K - 1) * dil : 0) - pad;  // first input position relative to q0
    const int W = NT_BLK + span;                          // staged columns actually used (<= WS)

    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    const bool xpk = NPL == 1 && p.x_packed != 0;       // x is the packed bf16 layout (ConvParams::x_packed)
    const uint2* __restrict__ xpb = reinterpret_cast<const uint2*>(p.x) + (int64_t)b * (p.Cin / 4) * p.x_cs;
    // packed bf16 weights: [plane][phase][octet][tap][kk][CoutP][4 bf16]
    const int64_t plane_stride = (int64_t)p.n_phase * (p.Cin / 8) * K * 2 * p.CoutP;     // in uint2
    const uint2* __restrict__ wp4 =
        reinterpret_cast<const uint2*>(p.w_bf16) + (int64_t)phase * (p.Cin / 8) * K * 2 * p.CoutP + co_blk0;
    const float in_slope = p.in_slope;
    const int n_chunks = p.Cin / KC;
    const int x_cs = p.x_cs, CoutP = p.CoutP;

    float ep_bias = 0.f, ep_scale = 1.f;     // see conv_mfma.hip: parked in LDS for the rolled row epilogue
    if (EPI != 2 && tid < CO_BLK) {
        const int co_ = min(co_blk0 + tid, p.Cout - 1);
        if (p.bias) ep_bias = p.bias[co_];
        if (p.scale) ep_scale = p.scale[co_];
    }

    const int qw0 = wn * NTL * 32;
    const bool wave_active = (q0 + qw0) < n_out;   // wave-uniform
    const int kk = lane >> 5, l31 = lane & 31;
    constexpr bool preload = EPI == 3;
    f32x16 acc[MT][NTL];
    // see conv_mfma.hip: buffer loads in the MFMA C layout, one per-lane offset per column tile + a scalar row offset per load
#define TTS_INIT_ACC()                                                                                       \
    if (preload) {                                                                                           \
        const int wm_s = __builtin_amdgcn_readfirstlane(wm);                                                 \
        const int row0 = co_blk0 + wm_s * MT * 32;                                                           \
        int voff[NTL];                                                                                       \
        _Pragma("unroll") for (int j = 0; j < NTL; ++j) {                                                    \
            const int q = q0 + qw0 + j * 32 + l31;                                                           \
            voff[j] = (q < n_out ? q : 0) * 4;                                                               \
        }                                                                                                    \
        {                                                                                                    \
            const int r_cs = p.r_cs;                                                                         \
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res + (int64_t)b * p.r_bs), 0, \
                                                              p.Cout * r_cs * 4, 0x00020000);                \
            const int vk = 4 * kk * r_cs * 4;                                                                \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                   \
                _Pragma("unroll") for (int j = 0; j < NTL; ++j)                                              \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                           \
                        acc[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(       \
                            rs, voff[j] + vk, (row0 + i * 32 + (r & 3) + 8 * (r >> 2)) * r_cs * 4, 0));      \
        }                                                                                                    \
        if (p.mode != 0) {                                                                                   \
            const int y_cs_ = p.y_cs;                                                                        \
            const auto ys = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * p.y_bs, 0, p.Cout * y_cs_ * 4, 0x00020000); \
            const int vk = 4 * kk * y_cs_ * 4;                                                               \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                   \
                _Pragma("unroll") for (int j = 0; j < NTL; ++j) {                                            \
                    f32x16 t;                                                                                \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                           \
                        t[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(               \
                            ys, voff[j] + vk, (row0 + i * 32 + (r & 3) + 8 * (r >> 2)) * y_cs_ * 4, 0));     \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = t[r] + acc[i][j][r];       \
                }                                                                                            \
        }                                                                                                    \
    } else {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                       \
            _Pragma("unroll") for (int j = 0; j < NTL; ++j)                                                  \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;                           \
    }

    // ---- staging registers.  X item it = (oc, kk, col): the 4 channels 8*oc + 2p + kk, p = 0..3,
    // at one input position -> one float4; W is a linear float4 copy.  Loads are unconditional
    // (clamped address + select) so they issue back to back and are waited for only where the
    // chunk is written to LDS.
    float sx[4 * NXI];
    unsigned sw[2 * NPL * NW];
    bool st_ok[NXI], st_in[NXI];
    int st_off[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int it = tid + 256 * i;
        const int ockk = min(it / WS, 2 * G::NOCT - 1), col = it % WS;
        const int pos = q0 + lo + col;
        st_in[i] = it < G::XI;
        st_ok[i] = st_in[i] && (col < W) && (pos >= 0) && (pos < in_len);
        st_off[i] = (xpk ? ockk : ((ockk >> 1) * 8 + (ockk & 1))) * x_cs + min(max(pos, 0), max(in_len - 1, 0));
    }

    // One staging "job" = one memory instruction (+ its VALU).  Jobs are spread one by one
    // over the gaps between MFMAs (an MFMA occupies the matrix pipe for 64 cycles, during
    // which the wave can issue ~50 cycles of other work for free); every gap is pinned with
    // sched_barrier(0).  Load jobs of chunk c+NSTAGE-1 go into the first NGRP-1 operand
    // groups of chunk c, write jobs into its last group.
    constexpr int NLJ = 4 * NXI + NPL * NW;    // load jobs per chunk
    constexpr int NWJ = NXI + NPL * NW;        // write jobs per chunk
    constexpr int NM = (NPL == 2 ? 3 : 1) * MT * NTL;   // MFMAs per operand group
    constexpr int NF = NPL * (MT + NTL);       // ds_read_b64 per operand fetch
    constexpr int GL = (NGRP - 1) * NM;        // gaps carrying load jobs
    static_assert(NGRP >= 2, "operand group layout");
#define TTS_LOAD_JOB(J)                                                                              \
    {                                                                                                \
        if ((J) < 4 * NXI) {                                                                         \
            if (!xpk) {                                                                              \
                sx[(J)] = xc[st_off[(J) / 4] + 2 * ((J) % 4) * x_cs];                                \
            } else if ((J) % 4 == 0) {                                                               \
                const uint2 t2 = xcp[st_off[(J) / 4]];                                               \
                sx[(J)] = __uint_as_float(t2.x); sx[(J) + 1] = __uint_as_float(t2.y);                \
            }                                                                                        \
        } else {                                                                                     \
            const int i_ = ((J)-4 * NXI) % NW, pl_ = ((J)-4 * NXI) / NW;                             \
            const int e = min(tid + 256 * i_, G::W4 - 1);                                            \
            const uint2 t2 = wc[pl_ * plane_stride + (int64_t)(e / CO_BLK) * CoutP + (e % CO_BLK)];  \
            sw[2 * (pl_ * NW + i_)] = t2.x; sw[2 * (pl_ * NW + i_) + 1] = t2.y;                      \
        }                                                                                            \
    }
#define TTS_LRELU(v) ((v) > 0.f ? (v) : (v)*in_slope)
#define TTS_WRITE_JOB(J, SB)                                                                         \
    {                                                                                                \
        if ((J) < NXI) {                                                                             \
            const int i_ = (J);                                                                      \
            if (st_in[i_] && xpk) {                                                                  \
                (SB)[tid + 256 * i_] = st_ok[i_] ? make_uint2(__float_as_uint(sx[4 * i_]), __float_as_uint(sx[4 * i_ + 1])) \
                                                 : make_uint2(0u, 0u);                               \
            } else if (st_in[i_]) {                                                                  \
                float v0 = sx[4 * i_], v1 = sx[4 * i_ + 1], v2 = sx[4 * i_ + 2], v3 = sx[4 * i_ + 3]; \
                v0 = st_ok[i_] ? TTS_LRELU(v0) : 0.f; v1 = st_ok[i_] ? TTS_LRELU(v1) : 0.f;           \
                v2 = st_ok[i_] ? TTS_LRELU(v2) : 0.f; v3 = st_ok[i_] ? TTS_LRELU(v3) : 0.f;           \
                const unsigned h0 = bf16_rne(v0), h1 = bf16_rne(v1), h2 = bf16_rne(v2), h3 = bf16_rne(v3); \
                (SB)[tid + 256 * i_] = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));                 \
                if (NPL == 2) {                                                                      \
                    const unsigned l0 = bf16_rne(v0 - __uint_as_float(h0 << 16)), l1 = bf16_rne(v1 - __uint_as_float(h1 << 16)), \
                                   l2 = bf16_rne(v2 - __uint_as_float(h2 << 16)), l3 = bf16_rne(v3 - __uint_as_float(h3 << 16)); \
                    (SB)[G::XI + tid + 256 * i_] = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));     \
                }                                                                                    \
            }                                                                                        \
        } else {                                                                                     \
            const int i_ = ((J)-NXI) % NW, pl_ = ((J)-NXI) / NW;                                     \
            const int e = tid + 256 * i_;                                                            \
            if (e < G::W4)                                                                           \
                (SB)[NPL * G::XI + pl_ * G::W4 + e] = make_uint2(sw[2 * (pl_ * NW + i_)], sw[2 * (pl_ * NW + i_) + 1]); \
        }                                                                                            \
    }
    // part PART (< NF) of the operand fetch of group GRP of stage STG into register slot SLOT
#define TTS_FETCH_PART(SLOT, STG, GRP, PART)                                                         \
    {                                                                                                \
        const int pl_ = (PART) / (MT + NTL), q_ = (PART) % (MT + NTL);                               \
        if (q_ < MT) {                                                                               \
            const uint2 t2 = sA[(STG)*G::BUF4 + pl_ * G::W4 + (GRP)*2 * CO_BLK + q_ * 32];           \
            a[SLOT][pl_][q_][0] = t2.x; a[SLOT][pl_][q_][1] = t2.y;                                  \
        } else {                                                                                     \
            const int j_ = q_ - MT;                                                                  \
            const uint2 t2 = sB[(STG)*G::BUF4 + pl_ * G::XI + ((GRP) / K) * 2 * WS + ((GRP) % K) * dil + j_ * 32]; \
            bq[SLOT][pl_][j_][0] = t2.x; bq[SLOT][pl_][j_][1] = t2.y;                                \
        }                                                                                            \
    }

    const uint2* sB = smem4 + kk * WS + (qw0 + l31 - pad - lo);
    const uint2* sA = smem4 + NPL * G::XI + kk * CO_BLK + wm * MT * 32 + l31;
    unsigned a[2][NPL][MT][2], bq[2][NPL][NTL][2];

    // prologue: fill NSTAGE-1 stages (bulk), fetch the first operands
    for (int c0 = 0; c0 < NSTAGE - 1 && c0 < n_chunks; ++c0) {
        const float* __restrict__ xc = xb + (int64_t)c0 * KC * x_cs;
        const uint2* __restrict__ xcp = xpb + (int64_t)c0 * (KC / 4) * x_cs;
        const uint2* __restrict__ wc = wp4 + (int64_t)c0 * G::NOCT * K * 2 * CoutP;
        uint2* sbp = smem4 + c0 * G::BUF4;
#pragma unroll
        for (int J = 0; J < NLJ; ++J) TTS_LOAD_JOB(J)
        if (c0 == 0) { TTS_INIT_ACC() }
#pragma unroll
        for (int J = 0; J < NWJ; ++J) TTS_WRITE_JOB(J, sbp)
    }
    if (NSTAGE < 2 || n_chunks < 1) { TTS_INIT_ACC() }
    __syncthreads();
#pragma unroll
    for (int P = 0; P < NF; ++P) TTS_FETCH_PART(0, 0, 0, P)

    int stage = 0;  // c % NSTAGE
    for (int c = 0; c < n_chunks; ++c) {
        // chunk to stage during this chunk (clamped: at the tail the last chunk is re-staged
        // into a dead stage, which keeps the loop body branch-free)
        const int cl = min(c + NSTAGE - 1, n_chunks - 1);
        const float* __restrict__ xc = xb + (int64_t)cl * KC * x_cs;
        const uint2* __restrict__ xcp = xpb + (int64_t)cl * (KC / 4) * x_cs;
        const uint2* __restrict__ wc = wp4 + (int64_t)cl * G::NOCT * K * 2 * CoutP;
        const int stage_next = (stage + 1 == NSTAGE) ? 0 : stage + 1;           // chunk c+1
        const int stage_fill = (stage == 0) ? NSTAGE - 1 : stage - 1;           // chunk c+NSTAGE-1
        uint2* sbf = smem4 + stage_fill * G::BUF4;
        const int sn = (c + 1 < n_chunks) ? stage_next : stage;
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int term = m / (MT * NTL), i = (m / NTL) % MT, j = m % NTL;
                // split bf16: hi*hi + hi*lo + lo*hi (the lo*lo term is below fp32 resolution)
                const int pa = term == 2 ? 1 : 0, pb = term == 1 ? 1 : 0;
                {
                    bf16x4 av, bv;
                    av[0] = (short)(a[cur][pa][i][0] & 0xffff); av[1] = (short)(a[cur][pa][i][0] >> 16);
                    av[2] = (short)(a[cur][pa][i][1] & 0xffff); av[3] = (short)(a[cur][pa][i][1] >> 16);
                    bv[0] = (short)(bq[cur][pb][j][0] & 0xffff); bv[1] = (short)(bq[cur][pb][j][0] >> 16);
                    bv[2] = (short)(bq[cur][pb][j][1] & 0xffff); bv[3] = (short)(bq[cur][pb][j][1] >> 16);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(av, bv, acc[i][j], 0, 0, 0);
                }
                // ---- gap work ----
#pragma unroll
                for (int P = 0; P < NF; ++P)
                    if (P >= m * NF / NM && P < (m + 1) * NF / NM) {
                        if (g + 1 < NGRP) TTS_FETCH_PART(nxt, stage, g + 1, P)
                        else if (NSTAGE == 3) TTS_FETCH_PART(nxt, sn, 0, P)
                    }
#if !defined(TTS_EXP_NOLOAD)
                if (g + 1 < NGRP) {

                    const int t = g * NM + m;
#pragma unroll
                    for (int J = 0; J < NLJ; ++J)
                        if (J >= t * NLJ / GL && J < (t + 1) * NLJ / GL) TTS_LOAD_JOB(J)
                }
#endif
#if !defined(TTS_EXP_NOWRITE)
                if (g + 1 == NGRP) {
#pragma unroll
                    for (int J = 0; J < NWJ; ++J)
                        if (J >= m * NWJ / NM && J < (m + 1) * NWJ / NM) TTS_WRITE_JOB(J, sbf)
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifndef TTS_EXP_NOBARRIER
        __syncthreads();
#endif
        if (NSTAGE == 3) {
            if ((NGRP & 1) != 0) {   // the prefetched group sits in slot 1: next chunk starts from slot 0
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) { a[0][pl][i][0] = a[1][pl][i][0]; a[0][pl][i][1] = a[1][pl][i][1]; }
#pragma unroll
                    for (int j = 0; j < NTL; ++j) { bq[0][pl][j][0] = bq[1][pl][j][0]; bq[0][pl][j][1] = bq[1][pl][j][1]; }
                }
            }
        } else if (c + 1 < n_chunks) {
#pragma unroll
            for (int P = 0; P < NF; ++P) TTS_FETCH_PART(0, stage_next, 0, P)
        }
        stage = stage_next;
    }
#undef TTS_INIT_ACC
#undef TTS_LOAD_JOB
#undef TTS_WRITE_JOB
#undef TTS_FETCH_PART
#undef TTS_LRELU

    if constexpr (EPI != 2) {
        {

            constexpr int LDS_F = NSTAGE * G::BUF4 * 2;                         // floats of LDS this block owns
            constexpr int ROWS_FIT = (LDS_F - 2 * CO_BLK) / NT_BLK;             // whole rows next to the bias/scale vectors
            constexpr int NPASS = (CO_BLK + ROWS_FIT - 1) / ROWS_FIT;           // the tile goes through in NPASS row slabs
            constexpr int ROWS_P = (CO_BLK + NPASS - 1) / NPASS;
            constexpr int LPR = NT_BLK / 4;                                     // lanes per row (one float4 each)
            static_assert(ROWS_FIT >= 1 && (64 % LPR == 0 || LPR % 64 == 0), "epilogue slab");
            float* ep = reinterpret_cast<float*>(smem4);
            float* epb = ep + LDS_F - 2 * CO_BLK;                               // [CO_BLK] bias, [CO_BLK] scale
            float* __restrict__ yb = p.y + (int64_t)b * p.y_bs;
            const float* __restrict__ rb = (p.res && !preload) ? p.res + (int64_t)b * p.r_bs : nullptr;
            const int mode = p.mode, relu_out = p.relu_out, Cout = p.Cout;
            const float div = p.div;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                __syncthreads();                                                // ring stages / previous slab are dead
                if (ps == 0 && tid < CO_BLK) { epb[tid] = ep_bias; epb[CO_BLK + tid] = ep_scale; }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                            if (NPASS == 1 || (row >= ps * ROWS_P && row < (ps + 1) * ROWS_P))
                                ep[(row - ps * ROWS_P) * NT_BLK + qw0 + j * 32 + l31] = acc[i][j][r];
                        }
                __syncthreads();
                constexpr int RPI = LPR >= 64 ? 1 : 64 / LPR;                     // rows per wave instruction
                constexpr int CPL = LPR >= 64 ? LPR / 64 : 1;                     // float4 columns groups per lane
                constexpr int NR = (ROWS_P + 4 * RPI - 1) / (4 * RPI);            // row iterations per wave
                if (preload || (!rb && mode == 0 && (!ACT || relu_out < 2))) {
                    // nothing to read from memory: a store loop without a vmcnt wait (conv_mfma.hip)
                    const float lo = relu_out == 1 ? 0.f : -__builtin_inff();
                    const bool do_div = mode == 2;
#pragma unroll 4
                    for (int it = 0; it < NR; ++it) {
                        const int r0 = wid * RPI + it * 4 * RPI;
                        const int rl = r0 + (LPR >= 64 ? 0 : lane / LPR);
                        const int co = co_blk0 + ps * ROWS_P + rl;
                        if (rl >= ROWS_P || ps * ROWS_P + rl >= CO_BLK || co >= Cout) continue;
                        const float bsv = epb[ps * ROWS_P + rl], scv = epb[CO_BLK + ps * ROWS_P + rl];
#pragma unroll
                        for (int cg = 0; cg < CPL; ++cg) {
                            const int col = ((LPR >= 64 ? lane : lane % LPR) + 64 * cg) * 4;
                            const int q = q0 + col;
                            if (q >= n_out) continue;
                            const float4 a4 = *reinterpret_cast<const float4*>(ep + rl * NT_BLK + col);
                            float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = fmaxf((v[e] + bsv) * scv, lo);
                                if (do_div) x = x / div;
                                v[e] = x;
                            }
                            float* yp = yb + (int64_t)co * p.y_cs + q;
                            if (q + 3 < n_out) {
                                *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (q + e < n_out) yp[e] = v[e];
                            }
                        }
                    }
                    continue;
                }
                if constexpr (!preload) {
#pragma unroll 2
                for (int it = 0; it < NR; ++it) {
                    const int r0 = wid * RPI + it * 4 * RPI;
                    const int rl = r0 + (LPR >= 64 ? 0 : lane / LPR);
                    const int co = co_blk0 + ps * ROWS_P + rl;
                    if (rl >= ROWS_P || ps * ROWS_P + rl >= CO_BLK || co >= Cout) continue;   // NPASS * ROWS_P may exceed the tile
                    const float bsv = epb[ps * ROWS_P + rl], scv = epb[CO_BLK + ps * ROWS_P + rl];
#pragma unroll
                    for (int cg = 0; cg < CPL; ++cg) {
                        const int col = ((LPR >= 64 ? lane : lane % LPR) + 64 * cg) * 4;
                        const int q = q0 + col;
                        if (q >= n_out) continue;
                        const float4 a4 = *reinterpret_cast<const float4*>(ep + rl * NT_BLK + col);
                        float v[4] = {a4.x, a4.y, a4.z, a4.w};
                        float* yp = yb + (int64_t)co * p.y_cs + q;
                        const float* rp = rb ? rb + (int64_t)co * p.r_cs + q : nullptr;
                        const bool full = q + 3 < n_out;
                        float rr4[4] = {0.f, 0.f, 0.f, 0.f}, pp4[4] = {0.f, 0.f, 0.f, 0.f};
                        if (full) {
                            if (rp) { const float4 t = *reinterpret_cast<const float4*>(rp); rr4[0] = t.x; rr4[1] = t.y; rr4[2] = t.z; rr4[3] = t.w; }
                            if (mode != 0) { const float4 t = *reinterpret_cast<const float4*>(yp); pp4[0] = t.x; pp4[1] = t.y; pp4[2] = t.z; pp4[3] = t.w; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (q + e < n_out) {
                                    if (rp) rr4[e] = rp[e];
                                    if (mode != 0) pp4[e] = yp[e];
                                }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = v[e] + bsv;
                            if constexpr (ACT) { if (relu_out == 2) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
                            x = x * scv + rr4[e];
                            if (relu_out == 1) x = fmaxf(x, 0.f);
                            if constexpr (ACT) { if (relu_out == 3) x = tanhf(x); }
                            if (mode == 1) x = pp4[e] + x;
                            else if (mode == 2) x = (pp4[e] + x) / div;
                            v[e] = x;
                        }
                        if (full) {
                            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (q + e < n_out) yp[e] = v[e];
                        }
                    }
                }
                }   // !preload
            }
            return;
        }
    } else {
    // epilogue: bias, residual, activation, accumulate modes.  Per half tile all loads
    // (bias, residual, previous y) are issued first and only then consumed.
    if (!wave_active) return;
    const int co_w0 = co_blk0 + wm * MT * 32;
    float* __restrict__ yb = p.y + (int64_t)b * p.y_bs + phase;
    const float* __restrict__ rb = p.res ? p.res + (int64_t)b * p.r_bs + phase : nullptr;
    const float* __restrict__ bias = p.bias;
    const float* __restrict__ scale = p.scale;
    const int mode = p.mode, relu_out = p.relu_out, Cout = p.Cout;
    const int y_cs = p.y_cs, y_ts = p.y_ts, r_cs = p.r_cs;
    const float div = p.div;
    if (NPL == 1 && p.y_packed) {
        // lane (kk, l31) of a 32x32 tile holds rows 8*oo + 4*kk + {0,1,2,3} of column l31: the bf16 pairs
        // (rows +0,+2) and (+1,+3) are dword kk of the entries (octet oo, kk' = 0) and (octet oo, kk' = 1)
        unsigned* __restrict__ yp = reinterpret_cast<unsigned*>(p.y) + (int64_t)b * (Cout / 2) * y_cs;
        const float ps = p.pack_slope;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int q = q0 + qw0 + j * 32 + l31;
#pragma unroll
                for (int oo = 0; oo < 4; ++oo) {
                    const int co0 = co_w0 + i * 32 + 8 * oo + 4 * kk;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int coc = min(co0 + e, Cout - 1);
                        float t = acc[i][j][4 * oo + e] + (bias ? bias[coc] : 0.f);
                        if (scale) t *= scale[coc];
                        if (relu_out == 1) t = fmaxf(t, 0.f);
                        v[e] = t > 0.f ? t : t * ps;
                    }
                    if (q < n_out && co0 < Cout) {
                        const int64_t e0 = ((int64_t)((co0 >> 3) * 2) * y_cs + q) * 2 + kk;
                        yp[e0] = bf16_rne(v[0]) | (bf16_rne(v[2]) << 16);
                        yp[e0 + 2 * (int64_t)y_cs] = bf16_rne(v[1]) | (bf16_rne(v[3]) << 16);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int q = q0 + qw0 + j * 32 + l31;
            const bool q_ok = q < n_out;
            const int qc = q_ok ? q : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float bv[8], rv[8], pv[8], sv[8];
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = h * 8 + r8;
                    const int co = co_w0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    const int coc = min(co, Cout - 1);
                    bv[r8] = bias ? bias[coc] : 0.f;
                    sv[r8] = scale ? scale[coc] : 1.f;
                    rv[r8] = rb ? rb[(int64_t)coc * r_cs + (int64_t)qc * y_ts] : 0.f;
                    pv[r8] = mode != 0 ? yb[(int64_t)coc * y_cs + (int64_t)qc * y_ts] : 0.f;
                }
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = h * 8 + r8;
                    const int co = co_w0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    float v = acc[i][j][r] + bv[r8];
                    if constexpr (ACT) { if (relu_out == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }  // nn.GELU()
                    v = v * sv[r8] + rv[r8];
                    if (relu_out == 1) v = fmaxf(v, 0.f);
                    if constexpr (ACT) { if (relu_out == 3) v = tanhf(v); }
                    if (mode == 1) v = pv[r8] + v;
                    else if (mode == 2) v = (pv[r8] + v) / div;
                    if (q_ok && co < Cout) yb[(int64_t)co * y_cs + (int64_t)q * y_ts] = v;
                }
            }
        }
    }
    }   // EPI == 2
}


template <int K, int MT, int NTL, int WM, int WN, int NPL, int EPI, bool ACT>
static int32_t launch_act_bf16(const ConvParams& p, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> lds_done{0};          // per instantiation: devices already opted in (common.hpp: lds_opt_in)
    TTS_CHECK_HIP(lds_opt_in((const void*)conv1d_mfma_bf16<K, MT, NTL, WM, WN, NPL, EPI, ACT>, (int)lds, lds_done));
    hipLaunchKernelGGL((conv1d_mfma_bf16<K, MT, NTL, WM, WN, NPL, EPI, ACT>), grid, dim3(256), lds, stream, p);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int K, int MT, int NTL, int WM, int WN, int NPL>
static int32_t launch_cfg_bf16(const ConvParams& p_in, hipStream_t stream) {
    ConvParams p = p_in;
    constexpr int CO_BLK = WM * MT * 32, NT_BLK = WN * NTL * 32;
    using G = GeoB<K, NT_BLK, CO_BLK, NPL>;
    TTS_REQUIRE(p.Cin % G::KC == 0, "conv: Cin=%d must be a multiple of %d for K=%d", p.Cin, G::KC, K);
    TTS_REQUIRE(p.w_bf16 != nullptr, "conv: bf16 weights were not packed for this layer");
    TTS_REQUIRE(!(p.x_packed || p.y_packed) || (NPL == 1 && p.n_phase == 1 && p.y_ts == 1),
                "conv: packed bf16 activations need the plain bf16 mode and a non-transposed conv");
    TTS_REQUIRE(!p.y_packed || (p.Cout % 8 == 0 && p.mode == 0 && p.res == nullptr), "conv: y_packed needs Cout %% 8 == 0, mode 0, no residual");
    constexpr bool HAS_ACT = K == 1 || K == 5;      // GELU: Vocos pwconv1 (k1); tanh: Tacotron2 postnet (k5)
    TTS_REQUIRE(p.relu_out < 2 || HAS_ACT, "conv: GELU / tanh epilogues are built for kernel sizes 1 and 5 only (K=%d)", K);
    const size_t lds = (size_t)G::NSTAGE * G::BUF4 * sizeof(uint2);
    dim3 grid((p.Nout + NT_BLK - 1) / NT_BLK, (p.CoutP / CO_BLK) * p.n_phase, p.batch);
    p.tile_major = tile_major_order(p, grid.x) ? 1 : 0;
    if (p.tile_major) std::swap(grid.x, grid.z);
    const bool vec_ok = !p.y_packed && p.y_ts == 1 && p.n_phase == 1 && (p.y_cs & 3) == 0 && (p.y_bs & 3) == 0 &&
                        ((uintptr_t)p.y & 15) == 0 &&
                        (!p.res || ((p.r_cs & 3) == 0 && (p.r_bs & 3) == 0 && ((uintptr_t)p.res & 15) == 0));
    const bool act = HAS_ACT && p.relu_out >= 2;
    const bool pre_ok = p.res != nullptr && p.scale == nullptr && p.relu_out < 2 &&
                        (int64_t)p.Cout * std::max(p.r_cs, p.y_cs) * 4 < ((int64_t)1 << 31);
    if (vec_ok) {
        if (act) return launch_act_bf16<K, MT, NTL, WM, WN, NPL, 0, HAS_ACT>(p, grid, lds, stream);
        if (pre_ok) return launch_act_bf16<K, MT, NTL, WM, WN, NPL, 3, false>(p, grid, lds, stream);
        return launch_act_bf16<K, MT, NTL, WM, WN, NPL, 0, false>(p, grid, lds, stream);
    }
    if (act) return launch_act_bf16<K, MT, NTL, WM, WN, NPL, 2, HAS_ACT>(p, grid, lds, stream);
    return launch_act_bf16<K, MT, NTL, WM, WN, NPL, 2, false>(p, grid, lds, stream);
}

template <int K, int NPL>
static int32_t launch_k_bf16(const ConvParams& p, hipStream_t stream) {
    auto blocks = [&](int co_blk, int nt_blk) -> int64_t {
        return (int64_t)((p.Nout + nt_blk - 1) / nt_blk) * (p.CoutP / co_blk) * p.n_phase * p.batch;
    };
    const int64_t want = 768;
    const bool tiny = p.Nout <= 96;
    if (K == 11 && NPL == 2) {   // the 2x2-tile split-bf16 k=11 instantiations spill registers: 1x2 tiles
        if (!tiny) return launch_cfg_bf16<K, 1, 2, 1, 4, NPL>(p, stream);
        return launch_cfg_bf16<K, 1, 1, 1, 4, NPL>(p, stream);
    }
    if (p.CoutP % 128 == 0) {
        if (!tiny && blocks(128, 128) >= want) return launch_cfg_bf16<K, 2, 2, 2, 2, NPL>(p, stream);
        return launch_cfg_bf16<K, 1, 1, 2, 2, NPL>(p, stream);
    }
    if (p.CoutP % 64 == 0) {
        if (!tiny && blocks(64, 256) >= want) return launch_cfg_bf16<K, 2, 2, 1, 4, NPL>(p, stream);
        return launch_cfg_bf16<K, 1, 1, 2, 2, NPL>(p, stream);
    }
    if (!tiny && blocks(32, 256) >= want) return launch_cfg_bf16<K, 1, 2, 1, 4, NPL>(p, stream);
    return launch_cfg_bf16<K, 1, 1, 1, 4, NPL>(p, stream);
}

template <int NPL>
static int32_t launch_conv_bf16(const ConvParams& p, hipStream_t stream) {
    switch (p.K) {
        case 1: return launch_k_bf16<1, NPL>(p, stream);
        case 2: return launch_k_bf16<2, NPL>(p, stream);
        case 3: return launch_k_bf16<3, NPL>(p, stream);
        case 5: return launch_k_bf16<5, NPL>(p, stream);
        case 7: return launch_k_bf16<7, NPL>(p, stream);
        case 11: return launch_k_bf16<11, NPL>(p, stream);
        default:
            set_error("conv: kernel size %d not instantiated (1,2,3,5,7,11)", p.K);
            return TTSAMD_EINVAL;
    }
}

int32_t launch_conv_bf16_any(const ConvParams& p, hipStream_t stream) {
    return p.precision == 2 ? launch_conv_bf16<2>(p, stream) : launch_conv_bf16<1>(p, stream);
}

}  // namespace ttsamd
