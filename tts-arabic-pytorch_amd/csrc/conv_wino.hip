// Winograd F(2,3) Conv1d(k = 3, dilation 1, "same" padding) on the fp32 matrix cores -- the one lever that REMOVES matrix work from
// the exact-fp32 path instead of re-arranging it: two neighbouring outputs from four multiplies instead of six,
//     m0 = (x0 - x2) g0,   m1 = (x1 + x2) (g0 + g1 + g2) / 2,   m2 = (x2 - x1) (g0 - g1 + g2) / 2,   m3 = (x1 - x3) g2
//     y[q] = m0 + m1 + m2,   y[q + 1] = m1 - m2 - m3            (x_s = x[q - 1 + s]; transform entries 0, +-1, +-1/2 only)
// i.e. FOUR pointwise GEMMs M_i = U_i V_i over the input channels instead of a 3-tap conv: 2/3 of the MFMAs of conv1d_mfma_f32<3>
// for the same outputs.  Serves FastPitch's PositionwiseConvFF convs (384 -> 1536 -> 384, k3: 97 % of a layer's FLOPs,
// models/fastpitch/fastpitch/transformer.py:59-65,72-90) and the dilation-1 k = 3 convs of HiFi-GAN's C = 256 stage
// (vocoder/hifigan/models.py:30-44).  Numerics: products and sums in fp32 as before; the transforms add one rounding per operand
// (measured 0.5-1.6x the direct conv's error against float64, profiles/r4/NOTES.md) -- results differ from conv1d_mfma_f32 in the
// last bits, the oracle tolerances (mel 1e-3, wave 1e-4) are met with the same margin.  TTSAMD_WINO=0 keeps the direct kernel.
//
// The anatomy of conv_mfma.hip: block = 4 waves (WM x WN), operands in LDS as float4 = four consecutive channel pairs so that one
// ds_read_b128 feeds four MFMA k-steps, a ring of 3 stages with one barrier per 8-channel chunk, the global loads of chunk c + 2
// and their LDS writes issued one by one inside the gaps between MFMAs (sched_barrier-pinned), row epilogue through the dead ring.
// What differs:
//   * a chunk's X region holds the four TRANSFORMED planes V_i [kk][i][pair] (8 x NPAIR float4), written by the staging jobs: every
//     thread loads 2 channels x 4 positions and writes its two float4 components of the four planes (8-byte LDS writes);
//   * the weights are the four transformed filters U_i, packed on the host like a 4-tap conv ([Cin/8][4][2][CoutP][4]);
//   * operand group g = plane g feeds its OWN accumulators acc[g] (4 x MT x 16 registers for a 32 MT x 64 output wave tile);
//   * the epilogue forms y[2j] = M0 + M1 + M2, y[2j + 1] = M1 - M2 - M3 and writes the pair as one float2 into the row buffer; the
//     residual preload (EPI 3) puts res[2j] into M0 and -res[2j + 1] into M3.
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include <vector>

#include "conv_mfma_common.hpp"

namespace ttsamd {

template <int MT, int WM, int WN>
struct WinoGeo {
    static constexpr int CO_BLK = WM * MT * 32;
    static constexpr int NPAIR = WN * 32;                     // output pairs per block (one 32-pair tile per wave)
    static constexpr int NT_BLK = 2 * NPAIR;                  // outputs per block
    static constexpr int X4 = 2 * 4 * NPAIR;                  // X float4s per stage: [kk][plane][pair]
    static constexpr int W4 = 4 * 2 * CO_BLK;                 // W float4s per stage: [plane][kk][co]
    static constexpr int NW = (W4 + 255) / 256;               // ... per thread
    static constexpr int BUF4 = X4 + W4;
    static constexpr int NSTAGE = 3;
    static constexpr int NGRP = 4;                            // operand groups per chunk = planes
    static constexpr int XTHR = 4 * NPAIR;                    // threads with an X item: (half, kk, pair)
    static_assert(XTHR <= 256, "one X item per thread");
    static_assert(3 * BUF4 * 16 <= 80 * 1024, "two blocks per CU");
};

// EPI 0: row epilogue (bias / residual / ReLU / accumulate modes);  3: the same with the residual (and the previous y) preloaded
template <int MT, int WM, int WN, int EPI>
__global__ __launch_bounds__(256, TTS_MINWAVES) void conv1d_wino_f32(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    using G = WinoGeo<MT, WM, WN>;
    constexpr int CO_BLK = G::CO_BLK, NPAIR = G::NPAIR, NT_BLK = G::NT_BLK, NW = G::NW, NGRP = G::NGRP, NSTAGE = G::NSTAGE;
    int b = blockIdx.z;
    int q0 = blockIdx.x * NT_BLK;
    if (p.compact) {   // dead blocks last (live_tile, common.hpp)
        int tile = 0;
        if (!live_tile(p.lens_out, p.len_out_mul, p.Nout, NT_BLK, p.batch, blockIdx.z * gridDim.x + blockIdx.x, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * NT_BLK;
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int co_blk0 = blockIdx.y * CO_BLK;
    int n_out = p.Nout;
    if (p.lens_out) n_out = min(n_out, (int)p.lens_out[b] * p.len_out_mul);
    if (q0 >= n_out) return;
    int in_len = p.Lin;
    if (p.lens_in) in_len = min(in_len, (int)p.lens_in[b] * p.len_in_mul);

    const int x_cs = p.x_cs, CoutP = p.CoutP;
    const int n_chunks = p.Cin / 8;
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    const float4* __restrict__ wp4 = reinterpret_cast<const float4*>(p.w_wino) + co_blk0;      // [octet][plane][kk][CoutP][4]
    const float in_slope = p.in_slope;

    float ep_bias = 0.f;
    if (tid < CO_BLK && p.bias) ep_bias = p.bias[min(co_blk0 + tid, p.Cout - 1)];

    const int kk = lane >> 5, l31 = lane & 31;
    constexpr bool preload = EPI == 3;
    f32x16 acc[4][MT];

    // residual preload: y[2j] = M0 + M1 + M2 takes res[2j] through M0, y[2j + 1] = M1 - M2 - M3 takes res[2j + 1] through -M3 (and, in
    // the accumulate modes, the previous y the same way).  Two 4-byte buffer loads per row and lane: an 8-byte load of the pair is
    // miscompiled by hipcc on ROCm 7.2 (llvm.amdgcn.raw.buffer.load.v2i32 whose two dwords feed different instructions is narrowed to
    // two loads of the SAME address; __builtin_amdgcn_raw_buffer_load_b64 emits one 4-byte load) -- tests/test_gpu_wino.py caught both.
#define TTS_INIT_ACC()                                                                                           \
    {                                                                                                            \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                            \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                       \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[g][i][r] = 0.f;                               \
        if (preload) {                                                                                           \
            const int wm_s = __builtin_amdgcn_readfirstlane(wm);                                                 \
            const int row0 = co_blk0 + wm_s * MT * 32;                                                           \
            const int q = q0 + 2 * (wn * 32 + l31);                                                              \
            /* q is even and the rows are float4-aligned, so q + 1 never leaves the row; a pair cut by the utterance end loads one value */ \
            /* that is never stored */                                                                           \
            const int voff = (q < n_out ? q : 0) * 4;                                                            \
            {                                                                                                    \
                const int r_cs = p.r_cs;                                                                         \
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res + (int64_t)b * p.r_bs), 0, \
                                                                  p.Cout * r_cs * 4, 0x00020000);                \
                const int vk = 4 * kk * r_cs * 4;                                                                \
                _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                   \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        const int so = (row0 + i * 32 + (r & 3) + 8 * (r >> 2)) * r_cs * 4;                      \
                        acc[0][i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + vk, so, 0));      \
                        acc[3][i][r] = -__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + vk + 4, so, 0)); \
                    }                                                                                            \
            }                                                                                                    \
            if (p.mode != 0) {                                                                                   \
                const int y_cs_ = p.y_cs;                                                                        \
                const auto ys = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * p.y_bs, 0, p.Cout * y_cs_ * 4, 0x00020000); \
                const int vk = 4 * kk * y_cs_ * 4;                                                               \
                _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                 \
                    f32x16 t0, t1;                                                                               \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        const int so = (row0 + i * 32 + (r & 3) + 8 * (r >> 2)) * y_cs_ * 4;                     \
                        t0[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, voff + vk, so, 0));     \
                        t1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, voff + vk + 4, so, 0)); \
                    }                                                                                            \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        acc[0][i][r] = t0[r] + acc[0][i][r];                                                     \
                        acc[3][i][r] = acc[3][i][r] - t1[r];                                                     \
                    }                                                                                            \
                }                                                                                                \
            }                                                                                                    \
        }                                                                                                        \
    }

    // ---- staging.  X item of thread (h, kks, pc): channels 8 c + 2 (2 h + pp) + kks, pp = 0 / 1, at the four positions
    // q0 + 2 pc - 1 + s, s = 0..3 -> components 2 h, 2 h + 1 of the float4 of each plane at [kks][plane][pc]
    const bool x_thr = tid < G::XTHR;
    const int sh = (tid / (2 * NPAIR)) & 1, skk = (tid / NPAIR) & 1, spc = tid % NPAIR;
    float sx[8], sw[4 * NW];
    int pos_c[4];
    bool pos_ok[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        const int pos = q0 + 2 * spc - 1 + s_;
        pos_ok[s_] = pos >= 0 && pos < in_len;
        pos_c[s_] = min(max(pos, 0), max(in_len - 1, 0));
    }
    const int ch_off = (4 * sh + skk) * x_cs;                 // channel 2 (2 h) + kks; pp adds 2 x_cs

    constexpr int NLJ = 8 + NW;                // load jobs per chunk
    constexpr int NWJ = 4 + NW;                // write jobs per chunk
    constexpr int NM = 4 * MT;                 // MFMAs per operand group
    constexpr int NF = MT + 1;                 // ds_read_b128 per operand fetch
    constexpr int GL = (NGRP - 1) * NM;        // gaps carrying load jobs
    static_assert(NF <= NM, "operand group layout");
#define TTS_LOAD_JOB_(J, SX, SW, XC, WC)                                                             \
    {                                                                                                \
        if ((J) < 8) {                                                                               \
            SX[(J)] = XC[ch_off + 2 * ((J) >> 2) * x_cs + pos_c[(J) & 3]];                           \
        } else {                                                                                     \
            const int i_ = (J)-8;                                                                    \
            const int e = min(tid + 256 * i_, G::W4 - 1);                                            \
            const float4 t4 = WC[(int64_t)(e / CO_BLK) * CoutP + (e % CO_BLK)];                      \
            SW[4 * i_] = t4.x; SW[4 * i_ + 1] = t4.y; SW[4 * i_ + 2] = t4.z; SW[4 * i_ + 3] = t4.w;  \
        }                                                                                            \
    }
#define TTS_XA(SX, PP, S) (pos_ok[S] ? ((SX)[4 * (PP) + (S)] > 0.f ? (SX)[4 * (PP) + (S)] : (SX)[4 * (PP) + (S)] * in_slope) : 0.f)
#define TTS_WRITE_JOB_(J, SB, SX, SW)                                                                \
    {                                                                                                \
        if ((J) < 4) {                                                                               \
            if (x_thr) {                                                                             \
                float2 v2;                                                                           \
                if ((J) == 0) { v2.x = TTS_XA(SX, 0, 0) - TTS_XA(SX, 0, 2); v2.y = TTS_XA(SX, 1, 0) - TTS_XA(SX, 1, 2); } \
                else if ((J) == 1) { v2.x = TTS_XA(SX, 0, 1) + TTS_XA(SX, 0, 2); v2.y = TTS_XA(SX, 1, 1) + TTS_XA(SX, 1, 2); } \
                else if ((J) == 2) { v2.x = TTS_XA(SX, 0, 2) - TTS_XA(SX, 0, 1); v2.y = TTS_XA(SX, 1, 2) - TTS_XA(SX, 1, 1); } \
                else { v2.x = TTS_XA(SX, 0, 1) - TTS_XA(SX, 0, 3); v2.y = TTS_XA(SX, 1, 1) - TTS_XA(SX, 1, 3); } \
                reinterpret_cast<float2*>((SB) + (skk * 4 + (J)) * NPAIR + spc)[sh] = v2;            \
            }                                                                                        \
        } else {                                                                                     \
            const int i_ = (J)-4;                                                                    \
            const int e = tid + 256 * i_;                                                            \
            if (e < G::W4)                                                                           \
                (SB)[G::X4 + e] = make_float4(SW[4 * i_], SW[4 * i_ + 1], SW[4 * i_ + 2], SW[4 * i_ + 3]); \
        }                                                                                            \
    }
#define TTS_LOAD_JOB(J) TTS_LOAD_JOB_(J, sx, sw, xc, wc)
#define TTS_WRITE_JOB(J, SB) TTS_WRITE_JOB_(J, SB, sx, sw)
    // part PART (< NF) of the operand fetch of group (= plane) GRP of stage STG into register slot SLOT
#define TTS_FETCH_PART(SLOT, STG, GRP, PART)                                                         \
    {                                                                                                \
        if ((PART) < MT) {                                                                           \
            const float4 t4 = sA[(STG)*G::BUF4 + (GRP)*2 * CO_BLK + (PART)*32];                      \
            a[SLOT][(PART) % MT][0] = t4.x; a[SLOT][(PART) % MT][1] = t4.y;                          \
            a[SLOT][(PART) % MT][2] = t4.z; a[SLOT][(PART) % MT][3] = t4.w;                          \
        } else {                                                                                     \
            const float4 t4 = sB[(STG)*G::BUF4 + (GRP)*NPAIR];                                       \
            bq[SLOT][0] = t4.x; bq[SLOT][1] = t4.y; bq[SLOT][2] = t4.z; bq[SLOT][3] = t4.w;          \
        }                                                                                            \
    }

    const float4* sB = smem4 + kk * 4 * NPAIR + wn * 32 + l31;
    const float4* sA = smem4 + G::X4 + kk * CO_BLK + wm * MT * 32 + l31;
    float a[2][MT][4], bq[2][4];

    // prologue: fill two stages (both chunks' loads before the first LDS write: one memory round trip), fetch the first operands
    if (n_chunks >= 2) {
        float sxb[8], swb[4 * NW + 1];
        const float* __restrict__ xc0 = xb;
        const float4* __restrict__ wc0 = wp4;
        const float* __restrict__ xc1 = xb + (int64_t)8 * x_cs;
        const float4* __restrict__ wc1 = wp4 + (int64_t)4 * 2 * CoutP;
#pragma unroll
        for (int J = 0; J < NLJ; ++J) TTS_LOAD_JOB_(J, sx, sw, xc0, wc0)
#pragma unroll
        for (int J = 0; J < NLJ; ++J) TTS_LOAD_JOB_(J, sxb, swb, xc1, wc1)
        TTS_INIT_ACC()
#pragma unroll
        for (int J = 0; J < NWJ; ++J) TTS_WRITE_JOB_(J, smem4, sx, sw)
#pragma unroll
        for (int J = 0; J < NWJ; ++J) TTS_WRITE_JOB_(J, smem4 + G::BUF4, sxb, swb)
    } else {
        const float* __restrict__ xc = xb;
        const float4* __restrict__ wc = wp4;
#pragma unroll
        for (int J = 0; J < NLJ; ++J) TTS_LOAD_JOB(J)
        TTS_INIT_ACC()
#pragma unroll
        for (int J = 0; J < NWJ; ++J) TTS_WRITE_JOB(J, smem4)
    }
    __syncthreads();
#pragma unroll
    for (int P = 0; P < NF; ++P) TTS_FETCH_PART(0, 0, 0, P)

    int stage = 0;  // c % NSTAGE
    for (int c = 0; c < n_chunks; ++c) {
        const int cl = min(c + NSTAGE - 1, n_chunks - 1);       // tail: the last chunk is re-staged into a dead stage (branch-free body)
        const float* __restrict__ xc = xb + (int64_t)cl * 8 * x_cs;
        const float4* __restrict__ wc = wp4 + (int64_t)cl * 4 * 2 * CoutP;
        const int stage_next = (stage + 1 == NSTAGE) ? 0 : stage + 1;
        const int stage_fill = (stage == 0) ? NSTAGE - 1 : stage - 1;
        float4* sbf = smem4 + stage_fill * G::BUF4;
        const int sn = (c + 1 < n_chunks) ? stage_next : stage;
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int pq = m / MT, i = m % MT;
                acc[g][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][pq], bq[cur][pq], acc[g][i], 0, 0, 0);
                // ---- gap work ----
                if (m < NF) {
                    if (g + 1 < NGRP) TTS_FETCH_PART(nxt, stage, g + 1, m)
                    else TTS_FETCH_PART(nxt, sn, 0, m)
                }
                if (g + 1 < NGRP) {
                    const int t = g * NM + m;
#pragma unroll
                    for (int J = 0; J < NLJ; ++J)
                        if (J >= t * NLJ / GL && J < (t + 1) * NLJ / GL) TTS_LOAD_JOB(J)
                } else {
#pragma unroll
                    for (int J = 0; J < NWJ; ++J)
                        if (J >= m * NWJ / NM && J < (m + 1) * NWJ / NM) TTS_WRITE_JOB(J, sbf)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // NGRP = 4 is even: the operands prefetched by the last group sit in slot 0, where the next chunk starts
        stage = stage_next;
    }
#undef TTS_INIT_ACC
#undef TTS_LOAD_JOB
#undef TTS_WRITE_JOB
#undef TTS_LOAD_JOB_
#undef TTS_WRITE_JOB_
#undef TTS_FETCH_PART
#undef TTS_XA

    // ---- epilogue: output transform, then the row epilogue of conv_mfma.hip (bias, residual, ReLU, accumulate modes)
    constexpr int LDS_F = NSTAGE * G::BUF4 * 4;                         // floats of LDS this block owns
    static_assert(LDS_F >= CO_BLK * NT_BLK + CO_BLK, "the tile goes through the dead ring in one pass");
    constexpr int LPR = NT_BLK / 4;                                     // lanes per row (one float4 each)
    static_assert(64 % LPR == 0 || LPR % 64 == 0, "epilogue rows");
    float* ep = reinterpret_cast<float*>(smem4);
    float* epb = ep + LDS_F - CO_BLK;                                   // [CO_BLK] bias
    float* __restrict__ yb = p.y + (int64_t)b * p.y_bs;
    const float* __restrict__ rb = (p.res && !preload) ? p.res + (int64_t)b * p.r_bs : nullptr;
    const int mode = p.mode, relu_out = p.relu_out, Cout = p.Cout;
    const float div = p.div;
    __syncthreads();                                                    // ring stages are dead
    if (tid < CO_BLK) epb[tid] = ep_bias;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            float2 y2;
            y2.x = acc[0][i][r] + acc[1][i][r] + acc[2][i][r];
            y2.y = acc[1][i][r] - acc[2][i][r] - acc[3][i][r];
            *reinterpret_cast<float2*>(ep + row * NT_BLK + 2 * (wn * 32 + l31)) = y2;
        }
    __syncthreads();
    constexpr int RPI = LPR >= 64 ? 1 : 64 / LPR;                       // rows per wave instruction
    constexpr int CPL = LPR >= 64 ? LPR / 64 : 1;                       // float4 column groups per lane
    constexpr int NR = (CO_BLK + 4 * RPI - 1) / (4 * RPI);              // row iterations per wave
    if (preload || (!rb && mode == 0)) {
        // nothing to read from memory: a loop without a single vmcnt wait (conv_mfma.hip: why)
        const float lo = relu_out == 1 ? 0.f : -__builtin_inff();
        const bool do_div = mode == 2;
#pragma unroll 4
        for (int it = 0; it < NR; ++it) {
            const int rl = wid * RPI + it * 4 * RPI + (LPR >= 64 ? 0 : lane / LPR);
            const int co = co_blk0 + rl;
            if (rl >= CO_BLK || co >= Cout) continue;
            const float bsv = epb[rl];
#pragma unroll
            for (int cg = 0; cg < CPL; ++cg) {
                const int col = ((LPR >= 64 ? lane : lane % LPR) + 64 * cg) * 4;
                const int q = q0 + col;
                if (q >= n_out) continue;
                const float4 a4 = *reinterpret_cast<const float4*>(ep + rl * NT_BLK + col);
                float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = fmaxf(v[e] + bsv, lo);
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
        return;
    }
    if constexpr (!preload) {
#pragma unroll 2
        for (int it = 0; it < NR; ++it) {
            const int rl = wid * RPI + it * 4 * RPI + (LPR >= 64 ? 0 : lane / LPR);
            const int co = co_blk0 + rl;
            if (rl >= CO_BLK || co >= Cout) continue;
            const float bsv = epb[rl];
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
                    float x = v[e] + bsv + rr4[e];
                    if (relu_out == 1) x = fmaxf(x, 0.f);
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
    }
}

template <int MT, int WM, int WN, int EPI>
static int32_t launch_wino_epi(const ConvParams& q, dim3 grid, hipStream_t stream) {
    using G = WinoGeo<MT, WM, WN>;
    constexpr size_t lds = (size_t)G::NSTAGE * G::BUF4 * sizeof(float4);
    static std::atomic<uint64_t> lds_done{0};
    const auto kern = conv1d_wino_f32<MT, WM, WN, EPI>;
    TTS_CHECK_HIP(lds_opt_in((const void*)kern, (int)lds, lds_done));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// Routing of an exact-fp32 conv launch: 0 = the direct kernel, 1 = conv1d_wino_f32 (this file: k = 3, dilation 1, 128-row tiles),
// 2 = conv1d_wino2_f32 (conv_wino2.hip: k = 3 / 7 / 11, dilation 1 / 3 / 5, 128- or 64-row tiles).  Wanted when the Winograd weights
// are there, the padding is 'same', rows are float4-aligned (row epilogue, residual preload) and the launch has at least 192 blocks.
// The two switches are read ONCE per launch (A/B runs and the parity tests of all three kernels flip them between calls):
//   TTSAMD_WINO=0        everything on the direct kernel
//   TTSAMD_WINO2=<mask>  what the decomposition kernel takes: bit 0 / 1 / 2 = k 3 / 7 / 11, bit 3 = their dilated (3 / 5) convs, bit 4
//                        = Cout = 64.  Default 31 (same-box A/B of the bench step, tools/ab_env.sh: 74.2 ms with none, 73.1 with k = 3,
//                        68.5 with k = 7 + 11, 67.5 with the three, 65.6 with dilations, 64.6-65.1 with Cout = 64)
int wino_route(const ConvParams& p) {
    if (p.precision != 0) return 0;
    const char* e = opt_str(OPT_WINO);
    if (e && e[0] == '0') return 0;
    if (p.K == 1) {
        // k = 1 (Vocos' pointwise convs and head, FastPitch's qkv / o_net projections) on the F(4,3) kernel's skeleton with nothing to transform
        // (conv_wino4.hip, Wino4Geo::WSHARE): the direct engine's packed weights as they are; TTSAMD_WINO4 bit 4.  Same tile as the F(4,3) launches.
        const char* e4 = opt_str(OPT_WINO4);
        const int mask4 = e4 ? atoi(e4) : 31;
        const bool ok1 = (mask4 & 16) && p.w != nullptr && p.dil == 1 && p.pad == 0 && p.n_phase == 1 && p.y_ts == 1 && p.CoutP % 64 == 0 &&
                         p.Cin % 32 == 0 && p.in_slope >= 0.f && p.in_slope <= 1.f && !p.x_packed && !p.y_packed &&
                         (p.y_cs & 3) == 0 && (p.y_bs & 3) == 0 && ((uintptr_t)p.y & 15) == 0 &&
                         (!p.res || ((p.r_cs & 3) == 0 && (p.r_bs & 3) == 0 && ((uintptr_t)p.res & 15) == 0)) &&
                         (int64_t)p.Cout * std::max(std::max(p.r_cs, p.y_cs), 1) * 4 < ((int64_t)1 << 31);
        if (!ok1) return 0;
        const int64_t blocks1 = (int64_t)((p.Nout + 255) / 256) * (p.CoutP / 64) * p.batch * wino4_ksplit(p);
        // one full round of the chip's 512 block slots or more: Vocos' GEMMs (512 ... 1536 blocks); FastPitch's qkv / o_net at batch 32 (192 / 384
        // blocks) stay on the direct kernel's smaller tiles -- same-box A/B of the step: 52.43 ms with them here, 52.33 without
        return (blocks1 >= 512 && p.Nout >= 256) ? 3 : 0;
    }
    if ((p.w_wino == nullptr && p.w_wino4 == nullptr) || (p.K != 3 && p.K != 7 && p.K != 11)) return 0;
    if (p.w_wino4 != nullptr) {
        // F(4,3) decomposition (conv_wino4.hip): TTSAMD_WINO4=<mask>, bit 0 / 1 / 2 = k 3 / 7 / 11, bit 3 = their dilated convs; default 15.
        // (k = 3 -- 6 instead of 8 products per quad -- did not pay with the kernel's first staging path: 414 / 330 vs 392 / 315 us on
        // FastPitch's conv-FF pair; with the aligned 16-byte window loads it does: same-box A/B of the step 55.46 (mask 14) vs 54.68 ms.)
        // 64 rows x 64 quads per block, float4-aligned rows, at least 192 blocks (below: the F(2,3) / direct routing that follows)
        const char* e4 = opt_str(OPT_WINO4);
        const int mask4 = e4 ? atoi(e4) : 31;
        const int kbit4 = p.K == 3 ? 1 : (p.K == 7 ? 2 : 4);
        const bool ok4 = (mask4 & kbit4) && (p.dil == 1 || (mask4 & 8)) && (p.K != 3 || p.Cin % 16 == 0) &&
                         (p.dil == 1 || p.dil == 3 || p.dil == 5) && p.pad == p.dil * (p.K - 1) / 2 && p.n_phase == 1 && p.y_ts == 1 &&
                         p.CoutP % 64 == 0 && p.Cin % 8 == 0 && p.scale == nullptr && p.relu_out < 2 && p.in_slope >= 0.f && p.in_slope <= 1.f &&
                         (p.y_cs & 3) == 0 && (p.y_bs & 3) == 0 && ((uintptr_t)p.y & 15) == 0 &&
                         (!p.res || ((p.r_cs & 3) == 0 && (p.r_bs & 3) == 0 && ((uintptr_t)p.res & 15) == 0)) &&
                         (int64_t)p.Cout * std::max(std::max(p.r_cs, p.y_cs), 1) * 4 < ((int64_t)1 << 31);
        if (ok4) {
            const int bo4 = wino4_block_outputs(p.dil);
            const int64_t blocks4 = (int64_t)((p.Nout + bo4 - 1) / bo4) * (p.CoutP / 64) * p.batch * wino4_ksplit(p);
            if (blocks4 >= 192 && p.Nout >= 256) return 3;
        }
        if (p.w_wino == nullptr) return 0;
    }
    const char* e2 = opt_str(OPT_WINO2);
    const int mask = e2 ? atoi(e2) : 31;
    const int kbit = p.K == 3 ? 1 : (p.K == 7 ? 2 : 4);
    const bool rows64 = p.CoutP % 128 != 0;
    // what the decomposition kernel can take of this launch (its k = 3 chunks are 16 channels)
    const bool w2 = (mask & kbit) && (p.dil == 1 || (mask & 8)) && (!rows64 || (mask & 16)) && (p.K != 3 || p.Cin % 16 == 0);
    const bool w1 = p.K == 3 && p.dil == 1 && !rows64;
    if (!w2 && !w1) return 0;
    if ((p.dil != 1 && p.dil != 3 && p.dil != 5) || p.pad != p.dil * (p.K - 1) / 2 || p.n_phase != 1 || p.y_ts != 1) return 0;
    if (p.CoutP % 64 != 0 || p.Cin % 8 != 0 || p.scale != nullptr || p.relu_out >= 2) return 0;
    if (!(p.in_slope >= 0.f && p.in_slope <= 1.f)) return 0;         // conv_wino2.hip activates with max(x, slope x)
    const bool vec_ok = (p.y_cs & 3) == 0 && (p.y_bs & 3) == 0 && ((uintptr_t)p.y & 15) == 0 &&
                        (!p.res || ((p.r_cs & 3) == 0 && (p.r_bs & 3) == 0 && ((uintptr_t)p.res & 15) == 0));
    if (!vec_ok || (int64_t)p.Cout * std::max(std::max(p.r_cs, p.y_cs), 1) * 4 >= ((int64_t)1 << 31)) return 0;
    const int bo = wino2_block_outputs(p.CoutP, p.dil);
    const int64_t blocks = (int64_t)((p.Nout + bo - 1) / bo) * (p.CoutP / (rows64 ? 64 : 128)) * p.batch;
    // under ~3/4 of a block per CU the direct kernel's 64 x 64 tiles with split K fill the chip better (batch 1: 3.68 ms per step with
    // the limit at 200, 3.74 at 100, 3.9-4.0 at 50 / 1 and with the Winograd kernels off; batch 4: 10.0 vs 10.5-11.0: tools/ab_small.sh);
    // short sequences (FastPitch's 64-token encoder) would leave half of a 128-output tile empty
    if (blocks < 192 || p.Nout < 256) return 0;
    return w2 ? 2 : 1;
}

template <int MT, int WM, int WN>
static int32_t launch_wino_cfg(const ConvParams& p, hipStream_t stream) {
    using G = WinoGeo<MT, WM, WN>;
    dim3 grid((p.Nout + G::NT_BLK - 1) / G::NT_BLK, p.CoutP / G::CO_BLK, p.batch);
    ConvParams q = p;
    q.ksplit = 1;
    q.compact = compact_order(p.lens_out, p.batch) ? 1 : 0;
    if (p.res != nullptr) return launch_wino_epi<MT, WM, WN, 3>(q, grid, stream);
    return launch_wino_epi<MT, WM, WN, 0>(q, grid, stream);
}

int32_t launch_wino(const ConvParams& p, hipStream_t stream) {
    // 128 co x 128 outputs (64 pairs).  (A 128 co x 64 outputs tile for launches under two blocks per CU -- FastPitch's 1536 -> 384 conv
    // at batch 32 is 384 blocks -- measured slower: 75.04 vs 74.45 ms per step; everything on it: 76.67.  A 64 co x 128 outputs tile,
    // three blocks per CU: 74.61 vs 74.22-74.35; everything on it: 75.44.)
    return launch_wino_cfg<2, 2, 2>(p, stream);
}

// host: torch Conv1d weight [Cout][Cin][3] -> the four transformed filters as a 4-tap conv in the engine's packed layout
// [Cin/8][4][2][CoutP][4]:  U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2   (sums in double, one rounding)
void pack_wino_weight(const float* w, int cout, int cin, float* out) {
    std::vector<float> u((size_t)cout * cin * 4);
    for (int64_t i = 0; i < (int64_t)cout * cin; ++i) {
        const double g0 = w[3 * i], g1 = w[3 * i + 1], g2 = w[3 * i + 2];
        u[4 * i] = (float)g0;
        u[4 * i + 1] = (float)((g0 + g1 + g2) * 0.5);
        u[4 * i + 2] = (float)((g0 - g1 + g2) * 0.5);
        u[4 * i + 3] = (float)g2;
    }
    pack_conv_weight(u.data(), cout, cin, 4, out);
}

}  // namespace ttsamd
