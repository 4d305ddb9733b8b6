// Implicit-GEMM Conv1d / ConvTranspose1d / Linear on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TFLOP/s peak).
//
// Replaces every F.conv1d / F.conv_transpose1d / F.linear the reference issues on the hot
// path: HiFi-GAN conv_pre / ups / ResBlock1 convs (vocoder/hifigan/models.py:46-53,
// 111-127), FastPitch conv-FF, qkv/o_net/proj, predictor convs
// (models/fastpitch/fastpitch/transformer.py:59-65,122,148; model.py:54-57,406), Vocos' pointwise convs and the
// forward DFT of the denoiser's bias spectrum.
//
// GEMM view per utterance b:  Y[co][q] = sum_{ci,tap} W[co][ci][tap] * X[ci][q + tap*dil - pad]
//   M = co (A operand = weights), N = q (time, B operand = activations), K = (ci, tap).
// Activations are channel-first [B][C][T]: a wave's 32 N-lanes are 32 consecutive time steps.
// A block is 4 waves tiled WM x WN, each wave owning MT x NTL 32x32 accumulators.
//
// Operand feeding (what the roofline fraction hinges on):
//  * both operands live in LDS as float4 = FOUR consecutive channel pairs, so ONE
//    ds_read_b128 per operand tile feeds four MFMA k-steps (16 MFMAs per 4 LDS reads at
//    MT = NTL = 2, no address VALU: 16-bit immediate offsets).  The k-pair of an MFMA is
//    (ci = 8o+2p, 8o+2p+1), p = float4 component, lanes 32-63 take the odd channel:
//      X_lds[o][kk][col][p]      W_lds[o][tap][kk][co][p]     (kk = lane >> 5)
//    and the weights are stored in exactly that order in HBM (pack_conv_weight) so their
//    staging is a linear float4 copy;
//  * a ring of NSTAGE (3 when it fits) LDS stages: the global loads of chunk c+NSTAGE-1 are
//    issued at the top of chunk c and written to LDS inside its last operand group, one
//    barrier per chunk; operand registers ping-pong one group ahead (pinned with
//    sched_barrier — hipcc otherwise re-uses the registers and exposes the LDS latency) and
//    with 3 stages the last group of a chunk already fetches the first operands of the next
//    chunk, so the MFMA stream never drains at a chunk boundary;
//  * input activation (leaky-relu) and the utterance-edge zero padding are applied once on
//    the LDS write; bias / residual / ReLU / ResBlock sum and /3 are fused in the epilogue.
// Ragged batches: positions >= lens_in[b] read as zero at the INPUT of every layer
// (SURVEY.md §3.4-5), tiles past lens_out[b] exit early.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "conv_mfma_common.hpp"

namespace ttsamd {

// octets (8 input channels) staged per chunk


template <int K, int NT_BLK, int CO_BLK>
struct Geo {
    // octets (8 input channels) per chunk.  k = 3 on the 128 x 128 tile takes two: a one-octet chunk is 48 MFMAs per wave (1.3 us)
    // between barriers; two halve the barriers and still leave two blocks per CU (FastPitch's 384 -> 1536 conv 121 -> 128 TFLOP/s,
    // HiFi-GAN C = 128 k = 3 106 -> 108; the smaller k = 3 tiles lose 2-3 % with it: tools/conv_bench -DTTS_NOCT3=2)
    static constexpr int NOCT = (K == 3 && NT_BLK == 128 && CO_BLK == 128) ? 2 : OctsOf<K>::NOCT;
    static constexpr int KC = 8 * NOCT;                       // input channels per chunk
    static constexpr int WS = NT_BLK + (K - 1) * DMAX;        // staged columns (one float4 each)
    static constexpr int XI = 2 * NOCT * WS;                  // X float4s per stage
    static constexpr int NXI = (XI + 255) / 256;              // ... per thread
    static constexpr int W4 = NOCT * K * 2 * CO_BLK;          // W float4s per stage
    static constexpr int NW = (W4 + 255) / 256;               // ... per thread
    static constexpr int BUF4 = XI + W4;                      // float4s per stage
#ifdef TTS_FORCE_NSTAGE
    static constexpr int NSTAGE = TTS_FORCE_NSTAGE;
#else
    // three stages when they fit in half of the LDS; the 32-channel tiles (HiFi-GAN stage 4: 4 chunks per block, all
    // prologue and epilogue) do better with two stages and one more resident block (+2...5 %, tools/conv_bench), and
    // so do the 64-channel k = 7 tiles (64 x 128: 4 resident blocks instead of 2, +3.8 % on stage 1; 64 x 256: +2.2 %)
    static constexpr int NSTAGE = (CO_BLK > 32 && 3 * BUF4 * 16 <= 80 * 1024 && !(K == 7 && CO_BLK == 64 && NT_BLK >= 128)) ? 3 : 2;
#endif
    static constexpr int NGRP = NOCT * K;                     // operand groups per chunk
};

// EPI selects the epilogue that is compiled in (the launcher decides on the host, launch_cfg): with all of them in one
// kernel the code was 140 KB, of which the main loop is 2 KB -- more than the 64 KB instruction cache two CUs share.
//   0: row-major float4 epilogue, bias / scale / residual / ReLU / accumulate modes   (every ResBlock and FFT conv)
//   1: the same plus GELU and tanh                                                    (Vocos pwconv1, Tacotron2 postnet)
//   2: per-lane epilogue for everything else: polyphase upsamplers, unaligned rows, split-K partial sums
//   3: row epilogue with the residual (and, in the accumulate modes, the previous y) PRELOADED into the accumulators
//      (every c2 conv of a ResBlock, o_net and the second conv-FF conv): nothing is read after the main loop
template <int K, int MT, int NTL, int WM, int WN, int EPI>
__device__ __forceinline__ void conv_tile(const ConvParams& p, float4* smem4, const int b, const int q0, const unsigned by_) {
    constexpr int CO_BLK = WM * MT * 32;
    constexpr int NT_BLK = WN * NTL * 32;
    using G = Geo<K, NT_BLK, CO_BLK>;
    constexpr int KC = G::KC, WS = G::WS, NXI = G::NXI, NW = G::NW, NGRP = G::NGRP, NSTAGE = G::NSTAGE;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int n_co_tiles = p.CoutP / CO_BLK;
    const int tiles_y = n_co_tiles * p.n_phase;
    const int ks = by_ / tiles_y, by = by_ % tiles_y;   // ks = split-K slice (0 when ksplit == 1)
    const int phase = by / n_co_tiles;
    const int co_blk0 = (by % n_co_tiles) * CO_BLK;

    int n_out = p.Nout;
    if (p.lens_out) n_out = min(n_out, (int)p.lens_out[b] * p.len_out_mul);
#ifdef TTS_TIMING   /* tools/conv_bench: block timeline (start, prologue done, main loop done, end) per block */
    const unsigned long long t_start = wall_clock64();
    if (q0 >= n_out) {   // dead block of a ragged batch: stamp it too (pro = main = 0)
        if (p.timing && threadIdx.x == 0) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            unsigned long long* tp = p.timing + (size_t)lin * 8;
            tp[0] = t_start; tp[1] = 0; tp[2] = 0; tp[3] = wall_clock64();
        }
        return;
    }
#else
    if (q0 >= n_out) return;
#endif
#ifdef TTS_TIMING
    const unsigned long long c_start = clock64();   // s_memtime: shader-clock ticks
    unsigned long long t_pro = 0, t_main = 0, t_e1 = 0;
#endif
    int in_len = p.Lin;
    if (p.lens_in) in_len = min(in_len, (int)p.lens_in[b] * p.len_in_mul);

    const int dil = p.dil;
    int pad = p.pad;
    if (p.n_phase > 1) pad = -((phase + p.phase_p) / p.n_phase);
    const int span = (K - 1) * (dil < 0 ? -dil : dil);
    const int lo = (dil < 0 ? (K - 1) * dil : 0) - pad;  // first input position relative to q0
    const int W = NT_BLK + span;                          // staged columns actually used (<= WS)

    const int x_cs = p.x_cs, CoutP = p.CoutP;
    const int chunks_all = p.Cin / KC;
    const int c_beg = (int)((int64_t)ks * chunks_all / p.ksplit), c_end = (int)((int64_t)(ks + 1) * chunks_all / p.ksplit);
    const int n_chunks = c_end - c_beg;
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs + (int64_t)c_beg * KC * x_cs;
    // packed weights: [phase][octet][tap][kk][CoutP][4]
    const float4* __restrict__ wp4 = reinterpret_cast<const float4*>(p.w) +
                                     (int64_t)phase * (p.Cin / 8) * K * 2 * p.CoutP + co_blk0 +
                                     (int64_t)c_beg * G::NOCT * K * 2 * CoutP;
    const float in_slope = p.in_slope;

    // this block's bias / scale (one output channel per thread) for the row epilogue: loaded now, parked in LDS after the
    // main loop, so that the rolled row loop reads them with lgkmcnt waits only (a vmcnt wait would also wait for the
    // previous row's store on this ISA)
    float ep_bias = 0.f, ep_scale = 1.f;
    if (EPI != 2 && tid < CO_BLK) {
        const int co_ = min(co_blk0 + tid, p.Cout - 1);
        if (p.bias) ep_bias = p.bias[co_];
        if (p.scale) ep_scale = p.scale[co_];
    }

    const int qw0 = wn * NTL * 32;
    const bool wave_active = (q0 + qw0) < n_out;   // wave-uniform
    const int kk = lane >> 5, l31 = lane & 31;

    // Residual preload: the accumulators START from the residual tile -- plus, in the accumulate modes, the previous
    // value of y -- (y = [y_prev +] res + sum ..., then + bias [, / div] in the epilogue) instead of the epilogue reading them.  The loads go out with the first chunk's staging loads and land directly in
    // the accumulator registers in the MFMA C layout (row = 8*(r>>2) + 4*kk + (r&3), 32 consecutive columns per
    // half-wave = one 128-byte line per row), so the residual costs no extra registers and no memory round trip after
    // the main loop: the row epilogue becomes LDS transposition + bias + stores, nothing to wait for.  (Before: one
    // residual round trip per row iteration, with every vmcnt wait also draining the previous row's store; the epilogue
    // lasted as long as the main loop)
    constexpr bool preload = EPI == 3;
    f32x16 acc[MT][NTL];
    // issued between the first chunk's staging loads and its LDS writes (see the prologue), so that the operands the
    // first MFMA needs are at the head of the memory queue
    // buffer loads: ONE per-lane 32-bit offset per column tile + a scalar row offset per load (64 flat loads would keep
    // 64 x 64-bit addresses live: +44 VGPRs, one resident block less per CU).  Rows >= Cout fall outside num_records and
    // read as zero; out-of-range columns read column 0 (never stored).
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
    float sx[4 * NXI], sw[4 * NW];   // scalar arrays: float4 arrays end up in scratch
    bool st_ok[NXI], st_in[NXI];
    int st_off[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int it = tid + 256 * i;
        const int ockk = min(it / WS, 2 * G::NOCT - 1), col = it % WS;
        const int pos = q0 + lo + col;
        st_in[i] = it < G::XI;
        st_ok[i] = st_in[i] && (col < W) && (pos >= 0) && (pos < in_len);
        st_off[i] = ((ockk >> 1) * 8 + (ockk & 1)) * x_cs + min(max(pos, 0), max(in_len - 1, 0));
    }

    // One staging "job" = one memory instruction (+ its VALU).  Jobs are spread one by one
    // over the gaps between MFMAs (an MFMA occupies the matrix pipe for 64 cycles, during
    // which the wave can issue ~50 cycles of other work for free); every gap is pinned with
    // sched_barrier(0).  Load jobs of chunk c+NSTAGE-1 go into the first NGRP-1 operand
    // groups of chunk c, write jobs into its last group.
    constexpr int NLJ = 4 * NXI + NW;          // load jobs per chunk
    constexpr int NWJ = NXI + NW;              // write jobs per chunk
    constexpr int NM = 4 * MT * NTL;           // MFMAs per operand group
    constexpr int NF = MT + NTL;               // ds_read_b128 per operand fetch
    constexpr int GL = (NGRP - 1) * NM;        // gaps carrying load jobs
    static_assert(NGRP >= 2 && NF <= NM, "operand group layout");
#define TTS_LOAD_JOB_(J, SX, SW, XC, WC)                                                                           \
    {                                                                                                \
        if ((J) < 4 * NXI) {                                                                         \
            SX[(J)] = XC[st_off[(J) / 4] + 2 * ((J) % 4) * x_cs];                                    \
        } else {                                                                                     \
            const int i_ = (J)-4 * NXI;                                                              \
            const int e = min(tid + 256 * i_, G::W4 - 1);                                            \
            const float4 t4 = WC[(int64_t)(e / CO_BLK) * CoutP + (e % CO_BLK)];                      \
            SW[4 * i_] = t4.x; SW[4 * i_ + 1] = t4.y; SW[4 * i_ + 2] = t4.z; SW[4 * i_ + 3] = t4.w;  \
        }                                                                                            \
    }
#define TTS_LRELU(v) ((v) > 0.f ? (v) : (v)*in_slope)
#define TTS_WRITE_JOB_(J, SB, SX, SW)                                                                         \
    {                                                                                                \
        if ((J) < NXI) {                                                                             \
            const int i_ = (J);                                                                      \
            if (st_in[i_]) {                                                                         \
                const float v0 = SX[4 * i_], v1 = SX[4 * i_ + 1], v2 = SX[4 * i_ + 2], v3 = SX[4 * i_ + 3]; \
                (SB)[tid + 256 * i_] =                                                               \
                    st_ok[i_] ? make_float4(TTS_LRELU(v0), TTS_LRELU(v1), TTS_LRELU(v2), TTS_LRELU(v3)) \
                              : make_float4(0.f, 0.f, 0.f, 0.f);                                     \
            }                                                                                        \
        } else {                                                                                     \
            const int i_ = (J)-NXI;                                                                  \
            const int e = tid + 256 * i_;                                                            \
            if (e < G::W4)                                                                           \
                (SB)[G::XI + e] = make_float4(SW[4 * i_], SW[4 * i_ + 1], SW[4 * i_ + 2], SW[4 * i_ + 3]); \
        }                                                                                            \
    }
#define TTS_LOAD_JOB(J) TTS_LOAD_JOB_(J, sx, sw, xc, wc)
#define TTS_WRITE_JOB(J, SB) TTS_WRITE_JOB_(J, SB, sx, sw)
    // part PART (< NF) of the operand fetch of group GRP of stage STG into register slot SLOT
#define TTS_FETCH_PART(SLOT, STG, GRP, PART)                                                         \
    {                                                                                                \
        if ((PART) < MT) {                                                                           \
            const float4 t4 = sA[(STG)*G::BUF4 + (GRP)*2 * CO_BLK + (PART)*32];                      \
            a[SLOT][(PART) % MT][0] = t4.x; a[SLOT][(PART) % MT][1] = t4.y;                          \
            a[SLOT][(PART) % MT][2] = t4.z; a[SLOT][(PART) % MT][3] = t4.w;                          \
        } else {                                                                                     \
            const int j_ = ((PART)-MT) % NTL;                                                        \
            const float4 t4 = sB[(STG)*G::BUF4 + ((GRP) / K) * 2 * WS + ((GRP) % K) * dil + j_ * 32]; \
            bq[SLOT][j_][0] = t4.x; bq[SLOT][j_][1] = t4.y; bq[SLOT][j_][2] = t4.z; bq[SLOT][j_][3] = t4.w; \
        }                                                                                            \
    }

    const float4* sB = smem4 + kk * WS + (qw0 + l31 - pad - lo);
    const float4* sA = smem4 + G::XI + kk * CO_BLK + wm * MT * 32 + l31;
    float a[2][MT][4], bq[2][NTL][4];

    // prologue: fill NSTAGE-1 stages (bulk), fetch the first operands.  With three stages the loads of both chunks
    // are issued before the first LDS write (second register set, dead after the prologue): one memory round trip
    // instead of two at the start of every block.
    if (NSTAGE == 3 && n_chunks >= 2) {
        float sxb[4 * NXI], swb[4 * NW + 1];
        const float* __restrict__ xc0 = xb;
        const float4* __restrict__ wc0 = wp4;
        const float* __restrict__ xc1 = xb + (int64_t)KC * x_cs;
        const float4* __restrict__ wc1 = wp4 + (int64_t)G::NOCT * K * 2 * CoutP;
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
        // one stage to fill: two-stage ring, or a single chunk (n_chunks >= 1 always)
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

#ifdef TTS_TIMING
    t_pro = wall_clock64();
#endif
    int stage = 0;  // c % NSTAGE
    for (int c = 0; c < n_chunks; ++c) {
        // chunk to stage during this chunk (clamped: at the tail the last chunk is re-staged
        // into a dead stage, which keeps the loop body branch-free)
        const int cl = min(c + NSTAGE - 1, n_chunks - 1);
        const float* __restrict__ xc = xb + (int64_t)cl * KC * x_cs;
        const float4* __restrict__ wc = wp4 + (int64_t)cl * G::NOCT * K * 2 * CoutP;
        const int stage_next = (stage + 1 == NSTAGE) ? 0 : stage + 1;           // chunk c+1
        const int stage_fill = (stage == 0) ? NSTAGE - 1 : stage - 1;           // chunk c+NSTAGE-1
        float4* sbf = smem4 + stage_fill * G::BUF4;
        const int sn = (c + 1 < n_chunks) ? stage_next : stage;
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int pq = m / (MT * NTL), i = (m / NTL) % MT, j = m % NTL;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][pq], bq[cur][j][pq], acc[i][j], 0, 0, 0);
                // ---- gap work ----
                if (m < NF) {
                    if (g + 1 < NGRP) TTS_FETCH_PART(nxt, stage, g + 1, m)
                    else if (NSTAGE == 3) TTS_FETCH_PART(nxt, sn, 0, m)
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
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[0][i][e] = a[1][i][e];
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) bq[0][j][e] = bq[1][j][e];
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
#undef TTS_LOAD_JOB_
#undef TTS_WRITE_JOB_
#undef TTS_FETCH_PART
#undef TTS_LRELU

#ifdef TTS_TIMING
    t_main = wall_clock64();
#endif
    // ---- epilogue: bias, residual, activation, accumulate modes ------------------------------------------
    const int co_w0 = co_blk0 + wm * MT * 32;
    if constexpr (EPI != 2) {
        {

            constexpr int LDS_F = NSTAGE * G::BUF4 * 4;                         // floats of LDS this block owns
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
                            if (NPASS == 1 || (row >= ps * ROWS_P && row < (ps + 1) * ROWS_P))   // NPASS > 1: a branch per element
                                ep[(row - ps * ROWS_P) * NT_BLK + qw0 + j * 32 + l31] = acc[i][j][r];
                        }
                __syncthreads();
#ifdef TTS_TIMING
                if (ps == 0) t_e1 = wall_clock64();
#endif
                constexpr int RPI = LPR >= 64 ? 1 : 64 / LPR;                     // rows per wave instruction
                constexpr int CPL = LPR >= 64 ? LPR / 64 : 1;                     // float4 columns groups per lane
                constexpr int NR = (ROWS_P + 4 * RPI - 1) / (4 * RPI);            // row iterations per wave
                if (preload || (!rb && mode == 0 && relu_out < 2)) {
                    // nothing to read from memory (no residual, or it came in through the accumulators): a loop without
                    // a single vmcnt wait -- on this ISA a vmcnt wait also drains the wave's previous STORE, which made
                    // every row iteration of the generic loop below one full memory round trip (~1.7 us)
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
                            if constexpr (EPI == 1) { if (relu_out == 2) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
                            x = x * scv + rr4[e];
                            if (relu_out == 1) x = fmaxf(x, 0.f);
                            if constexpr (EPI == 1) { if (relu_out == 3) x = tanhf(x); }
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
#ifdef TTS_TIMING
            if (p.timing && threadIdx.x == 0) {
                const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
                unsigned hwid, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                unsigned long long* tp = p.timing + (size_t)lin * 8;
                tp[0] = t_start; tp[1] = t_pro; tp[2] = t_main; tp[3] = wall_clock64(); tp[4] = hwid; tp[5] = xcc; tp[6] = t_e1; tp[7] = clock64() - c_start;
            }
#endif
            return;
        }
    } else {
    if (!wave_active) return;
    if (p.ksplit > 1) {   // raw partial sums; bias / activation / residual happen in splitk_reduce_kernel
        float* __restrict__ pb = p.splitk_ws + ((int64_t)ks * p.batch + b) * p.Cout * p.Nout;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int q = q0 + qw0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co_w0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (q < n_out && co < p.Cout) pb[(int64_t)co * p.Nout + q] = acc[i][j][r];
                }
            }
        return;
    }
    float* __restrict__ yb = p.y + (int64_t)b * p.y_bs + phase;
    const float* __restrict__ rb = p.res ? p.res + (int64_t)b * p.r_bs + phase : nullptr;
    const float* __restrict__ bias = p.bias;
    const float* __restrict__ scale = p.scale;
    const int mode = p.mode, relu_out = p.relu_out, Cout = p.Cout;
    const int y_cs = p.y_cs, y_ts = p.y_ts, r_cs = p.r_cs;
    const float div = p.div;
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
                    if constexpr (K == 1 || K == 5) { if (relu_out == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }   // nn.GELU()
                    v = v * sv[r8] + rv[r8];
                    if (relu_out == 1) v = fmaxf(v, 0.f);
                    if constexpr (K == 1 || K == 5) { if (relu_out == 3) v = tanhf(v); }
                    if (mode == 1) v = pv[r8] + v;
                    else if (mode == 2) v = (pv[r8] + v) / div;
                    if (q_ok && co < Cout) yb[(int64_t)co * y_cs + (int64_t)q * y_ts] = v;
                }
            }
        }
    }
    }   // EPI == 2
}

template <int K, int MT, int NTL, int WM, int WN, int EPI>
__global__ __launch_bounds__(256, TTS_MINWAVES) void conv1d_mfma_f32(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    constexpr int NT_BLK = WN * NTL * 32;
    // ragged batches: tile-major block order (x = utterance slot, z = time tile), see ConvParams::tile_major
    unsigned bx_ = blockIdx.x, by_ = blockIdx.y, bz_ = blockIdx.z;
    if (p.xcd_w) {
        // weight locality: workgroups go to the 8 XCDs round-robin by linear id; every XCD gets its own class of co-tiles, so
        // that the weight slice its 4 MB L2 has to hold next to the streaming activations is 1/g of the layer (C=256 k=11:
        // 2.9 MB of weights were re-fetched ~36x per XCD and launch; stand-alone +4 % there, +2.4 % for C=128 k=11, production
        // +0.2 % because three concurrent launches share the L2; TTSAMD_XCD_W=0 disables)
        // g = gcd(n_co_tiles, 8) classes of co-tiles; XCD x serves class x % g = the n_co_tiles / g co-tiles {x % g + g*k}
        const unsigned lin = bx_ + gridDim.x * (by_ + gridDim.y * bz_), xcd = lin & 7u, slot = lin >> 3;
        const unsigned nct = gridDim.y, g = (unsigned)p.xcd_w & 0xffu, per = nct / g;
        unsigned rest;
        if (p.xcd_w & 0x100) {
            // input locality on top: an XCD owns WHOLE time tiles of its class -- the per co-tiles of one time tile are consecutive
            // slots of one XCD, run side by side on its CUs and read the tile's input window through one L2 (with per = 1 co-tile
            // per class every window crosses the fabric nct times: C = 256 on 64-row tiles 4x, C = 128 k = 11 2x -- measured
            // 2.6-4.1x and 1.36-1.56x the algorithmic bytes, profiles/r4/traffic.json).  g is the launcher's choice: the fewest
            // classes whose weight slice still sits in the 4 MB L2 next to the streaming windows.
            rest = (slot / per) * (8u / g) + xcd / g;
            by_ = xcd % g + g * (slot % per);
        } else {
            const unsigned idx2 = slot * (8u / g) + xcd / g;
            rest = idx2 / per;
            by_ = xcd % g + g * (idx2 % per);
        }
        bx_ = rest % gridDim.x; bz_ = rest / gridDim.x;
    }
    int b = p.tile_major ? (int)((bx_ + bz_) % (unsigned)p.batch) : (int)bz_;
    int q0 = (p.tile_major ? bz_ : bx_) * NT_BLK;
    if (p.compact) {   // dead blocks last (live_tile, conv_mfma_common.hpp)
        int tile = 0;
        if (!live_tile(p.lens_out, p.len_out_mul, p.Nout, NT_BLK, p.batch, bz_ * gridDim.x + bx_, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * NT_BLK;
    }
    conv_tile<K, MT, NTL, WM, WN, EPI>(p, smem4, b, q0, by_);
}

// Second half of a split-K conv: y = epilogue(sum_ks partial[ks]) with exactly the epilogue of the main kernel.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams p) {
    const int b = blockIdx.z, co = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    int n_out = p.Nout;
    if (p.lens_out) n_out = min(n_out, (int)p.lens_out[b] * p.len_out_mul);
    if (q >= n_out) return;
    const int64_t per = (int64_t)p.batch * p.Cout * p.Nout;
    const float* __restrict__ pp = p.splitk_ws + ((int64_t)b * p.Cout + co) * p.Nout + q;
    // eight slices' loads in flight at a time (a plain `v += pp[ks * per]` loop waits for every load before the next one: up to
    // ksplit serial memory round trips); the sum still runs in slice order
    float v = 0.f;
    for (int k0 = 0; k0 < p.ksplit; k0 += 8) {
        float tq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tq[j] = pp[(int64_t)min(k0 + j, p.ksplit - 1) * per];
#pragma unroll
        for (int j = 0; j < 8; ++j) v = (k0 + j < p.ksplit) ? v + tq[j] : v;
    }
    v += p.bias ? p.bias[co] : 0.f;
    if (p.relu_out == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    v = v * (p.scale ? p.scale[co] : 1.f) + (p.res ? p.res[(int64_t)b * p.r_bs + (int64_t)co * p.r_cs + q] : 0.f);
    if (p.relu_out == 1) v = fmaxf(v, 0.f);
    else if (p.relu_out == 3) v = tanhf(v);
    float* yp = p.y + (int64_t)b * p.y_bs + (int64_t)co * p.y_cs + q;
    if (p.mode == 1) v = *yp + v;
    else if (p.mode == 2) v = (*yp + v) / p.div;
    *yp = v;
}

int32_t launch_splitk_reduce(const ConvParams& q, hipStream_t stream) {
    dim3 rg((q.Nout + 255) / 256, q.Cout, q.batch);
    hipLaunchKernelGGL(splitk_reduce_kernel, rg, dim3(256), 0, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int K, int MT, int NTL, int WM, int WN, int EPI>
static int32_t launch_epi(const ConvParams& q, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> lds_done{0};          // per instantiation: devices already opted in (common.hpp: lds_opt_in)
    const auto kern = conv1d_mfma_f32<K, MT, NTL, WM, WN, EPI>;
    TTS_CHECK_HIP(lds_opt_in((const void*)kern, (int)lds, lds_done));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// Block order experiment for ragged batches (TTSAMD_TILE_MAJOR=1; default off).  With the time tile in blockIdx.x every
// utterance ends in a run of dead blocks (tiles past its length).  On a stand-alone ragged launch whose utterances are
// 16 % shorter than the padded length on average, the in-order workgroup dispatcher leaves slots empty behind those runs
// (1.59 resident blocks per CU instead of 1.90, tools/conv_slots.py) and tile-major order (x = utterance slot, z = time
// tile, utterance = (slot + tile) % batch so that the XCD <-> utterance assignment rotates) recovers +4-5 % there.  On the
// bench workload (10 % shorter on average) it measures -0.5 % (84.2 vs 83.8 ms per step: neighbouring blocks no longer
// share their halo columns in L2), so the default stays time-major.
// Dead blocks last (live_tile, conv_mfma_common.hpp): default for every ragged launch; TTSAMD_COMPACT=0 restores the plain
// (time tile, co tile, utterance) grid.
bool compact_order(const void* lens, int batch) {
    if (lens == nullptr || batch <= 1) return false;
    const char* e = exp_env("TTSAMD_COMPACT");                  // read per call (a dozen ns next to a launch): tests and A/B runs flip it
    return !(e && e[0] == '0');
}

bool tile_major_order(const ConvParams& p, unsigned n_tiles) {
    static const int force = [] { const char* e = exp_env("TTSAMD_TILE_MAJOR"); return e ? atoi(e) : 0; }();
    return force != 0 && p.lens_out != nullptr && p.batch > 1 && n_tiles <= 65535;
}

// Weights one XCD's L2 is asked to keep when co-tiles share an XCD for the input window's sake.  Stand-alone, FETCH_SIZE per launch
// (profiles/r4/xcd_probe.txt): C = 256 k = 3 / 7 / 11 519 -> 330 / 727 -> 342 / 755 -> 492 MB, C = 128 k = 11 1452 -> 1020 MB -- at the same
// time per launch to 0.3 % (so the fabric bytes were never what these launches wait for), step 76.5 ms either way (tools/ab_xcd.sh).
// Kept on: 20 GB less fabric traffic per step for nothing.
constexpr int kXcdWeightKB = 3000;

template <int K, int MT, int NTL, int WM, int WN>
static int32_t launch_cfg(const ConvParams& p, hipStream_t stream) {
    constexpr int CO_BLK = WM * MT * 32, NT_BLK = WN * NTL * 32;
    using G = Geo<K, NT_BLK, CO_BLK>;
    TTS_REQUIRE(p.Cin % G::KC == 0, "conv: Cin=%d must be a multiple of %d for K=%d", p.Cin, G::KC, K);
    const size_t lds = (size_t)G::NSTAGE * G::BUF4 * sizeof(float4);
    dim3 grid((p.Nout + NT_BLK - 1) / NT_BLK, (p.CoutP / CO_BLK) * p.n_phase, p.batch);
    ConvParams q = p;
    q.ksplit = 1;
    q.tile_major = tile_major_order(p, grid.x) ? 1 : 0;
    if (q.tile_major) std::swap(grid.x, grid.z);
    q.compact = (!q.tile_major && compact_order(p.lens_out, p.batch)) ? 1 : 0;
    {
        const char* xe = opt_str(OPT_XCD_W);                    // read per call, like the other schedule switches
        const bool xw = !(xe && xe[0] == '0');
        const unsigned nct = grid.y;
        const unsigned g = (nct % 8 == 0) ? 8 : (nct % 4 == 0 ? 4 : (nct % 2 == 0 ? 2 : 1));
        const bool can = xw && !q.tile_major && p.n_phase == 1 && ((int64_t)grid.x * grid.y * grid.z) % 8 == 0;
        q.xcd_w = (can && g > 1) ? (int)g : 0;
        // an XCD owns whole time tiles: the co-tiles of its class are consecutive slots and share the input window through its L2.  As few
        // classes (g2 <= g) as keep a class's weights under TTSAMD_XCD_WMAX_KB (0 = the plain map above); with more co-tiles than classes
        // (FastPitch's 384 -> 1536 conv: 12 co-tiles in 4 classes; 1536 -> 384: 3 in 1) the same g already saves the re-reads
        const char* we = opt_str(OPT_XCD_WMAX_KB);
        const int64_t wmax = (we ? (int64_t)atoi(we) : (int64_t)kXcdWeightKB) * 1024;
        if (can && wmax > 0 && nct > 1) {
            const int64_t wbytes = (int64_t)p.CoutP * p.Cin * K * 4;
            unsigned g2 = g;
            while (g2 > 1 && wbytes / (g2 / 2) <= wmax) g2 /= 2;
            if (nct / g2 > 1 && ((int64_t)grid.x * grid.z * g2) % 8 == 0)
                q.xcd_w = (int)g2 | 0x100 | (q.xcd_w << 16);   // (bits 16..: the plain choice, for a split-K launch)
        }
    }
    const int64_t nblk = (int64_t)grid.x * grid.y * grid.z, per = (int64_t)p.batch * p.Cout * p.Nout;
    const int n_chunks = p.Cin / G::KC;
    static const int sk_blocks = [] { const char* e = exp_env("TTSAMD_SPLITK_BLOCKS"); return e ? atoi(e) : 320; }();
    static const int sk_target = [] { const char* e = exp_env("TTSAMD_SPLITK_TARGET"); return e ? atoi(e) : 640; }();
    static const int sk_chunks = [] { const char* e = exp_env("TTSAMD_SPLITK_MIN_CHUNKS"); return e ? atoi(e) : 8; }();
    if (p.splitk_ws && p.n_phase == 1 && p.y_ts == 1 && nblk < sk_blocks && n_chunks >= sk_chunks) {
        int64_t ks = std::min<int64_t>((sk_target + nblk - 1) / nblk, n_chunks / 4);
        ks = std::min<int64_t>(ks, p.splitk_floats / per);
        if (ks >= 2) q.ksplit = (int)ks;
        if (ks >= 2 && (q.xcd_w & 0x100)) q.xcd_w >>= 16;
    }
    if (q.xcd_w & 0x100) q.xcd_w &= 0x1ff;
    grid.y *= q.ksplit;
    // epilogue kind (see the kernel): the float4 row epilogue needs 16-byte aligned rows of y (and of the residual)
    const bool vec_ok = q.ksplit == 1 && p.y_ts == 1 && p.n_phase == 1 && (p.y_cs & 3) == 0 && (p.y_bs & 3) == 0 &&
                        ((uintptr_t)p.y & 15) == 0 &&
                        (!p.res || ((p.r_cs & 3) == 0 && (p.r_bs & 3) == 0 && ((uintptr_t)p.res & 15) == 0));
#ifdef TTS_NO_VEC_EPILOGUE
    const int epi = 2;
#else
    // GELU (Vocos pwconv1, k = 1) and tanh (Tacotron2 postnet, k = 5) are compiled into those kernel sizes only
    TTS_REQUIRE(p.relu_out < 2 || K == 1 || K == 5, "conv: GELU / tanh epilogues are built for kernel sizes 1 and 5 only (K=%d)", K);
    // the residual rides in through the accumulators (EPI 3) unless a per-channel scale sits between conv and residual
    // (Vocos gamma) or the kernel was built without it
#ifdef TTS_NO_PRELOAD
    const bool pre_ok = false;
#else
    const bool pre_ok = p.res != nullptr && p.scale == nullptr && p.relu_out < 2 &&
                        (int64_t)p.Cout * std::max(p.r_cs, p.y_cs) * 4 < ((int64_t)1 << 31);
#endif
    const int epi = !vec_ok ? 2 : (p.relu_out >= 2 ? 1 : (pre_ok ? 3 : 0));
#endif
    int32_t rc;
    if (epi == 3) rc = launch_epi<K, MT, NTL, WM, WN, 3>(q, grid, lds, stream);
    else if (epi == 0) rc = launch_epi<K, MT, NTL, WM, WN, 0>(q, grid, lds, stream);
    else if (epi == 1) rc = launch_epi<K, MT, NTL, WM, WN, (K == 1 || K == 5) ? 1 : 2>(q, grid, lds, stream);
    else rc = launch_epi<K, MT, NTL, WM, WN, 2>(q, grid, lds, stream);
    if (rc != 0) return rc;
    if (q.ksplit > 1) return launch_splitk_reduce(q, stream);
    return 0;
}

// Tile choice: the largest tile that still gives the chip >= 3 blocks per CU (768); small
// problems (batch 1, encoder-side S = 64) fall back to smaller tiles to fill the 256 CUs.
template <int K>
static int32_t launch_k(const ConvParams& p, hipStream_t stream) {
    auto blocks = [&](int co_blk, int nt_blk) -> int64_t {
        return (int64_t)((p.Nout + nt_blk - 1) / nt_blk) * (p.CoutP / co_blk) * p.n_phase * p.batch;
    };
    static const int64_t want_env = [] { const char* e = exp_env("TTSAMD_WANT_BLOCKS"); return e ? (int64_t)atoi(e) : (int64_t)-1; }();
    // blocks a launch should have: 3 per CU -- unless the whole problem is about one round of the smallest tiles (batch 1: 914 tiles
    // of 64 x 64 for a stage-2 conv): then one block per CU of a LARGE tile (64 x 256: 230 blocks at 0.85 of the matrix peak) beats
    // four of the small one (0.62): batch 1 5.04 -> 4.86 ms per call, measured with TTSAMD_WANT_BLOCKS=200 / 768
    const int64_t want = want_env > 0 ? want_env : (blocks(64, 64) <= 1024 ? (int64_t)200 : (int64_t)768);
    const bool tiny = p.Nout <= 96;
#ifdef TTS_FORCE_CFG   /* tile autotuning with tools/conv_bench.hip */
    switch (TTS_FORCE_CFG) {
        case 0: if (p.CoutP % 128 == 0) return launch_cfg<K, 2, 2, 2, 2>(p, stream); break;
        case 1: if (p.CoutP % 128 == 0) return launch_cfg<K, 1, 2, 4, 1>(p, stream); break;
        case 2: if (p.CoutP % 64 == 0) return launch_cfg<K, 2, 2, 1, 4>(p, stream); break;
        case 3: if (p.CoutP % 64 == 0) return launch_cfg<K, 1, 1, 2, 2>(p, stream); break;
        case 4: return launch_cfg<K, 1, 2, 1, 4>(p, stream);
        case 5: return launch_cfg<K, 1, 1, 1, 4>(p, stream);
        case 6: if (p.CoutP % 64 == 0) return launch_cfg<K, 1, 2, 2, 2>(p, stream); break;   // 64 co x 128 t
        case 7: if (p.CoutP % 128 == 0) return launch_cfg<K, 2, 1, 2, 2>(p, stream); break;  // 128 co x 64 t, 2x1 tiles per wave
    }
#endif
    // (measured and not kept, round 4: a 96 co x 128 t tile for FastPitch's second conv-FF conv -- 1536 -> 384, exactly two blocks per CU
    // instead of 2.6-2.8 of the 128 x 64 tile -- runs the step in 77.83 vs 77.82 ms: tools/ab_tile96.sh)
    if (p.CoutP % 128 == 0) {
        // deep-K layers on short sequences (HiFi-GAN stage 1: C = 256, 8 positions per frame) have few, long blocks;
        // a launch is then 2-4 rounds of blocks and its tail costs 15-19 %: finer tiles pay there
        const bool few_long = p.Cin >= 256 && p.CoutP <= p.Cin && !tiny && blocks(128, 128) < 4 * want;
        if (few_long && K == 3 && blocks(128, 64) >= want) return launch_cfg<K, 1, 2, 4, 1>(p, stream);  // 128 co x 64 t
        if (few_long && K >= 7 && blocks(64, 128) >= want) return launch_cfg<K, 1, 2, 2, 2>(p, stream);  //  64 co x 128 t
        if (K < 11 && !tiny && blocks(128, 128) >= want && (K != 3 || p.Cin % 16 == 0))                 // (k = 3: 16-channel chunks)
            return launch_cfg<K, 2, 2, 2, 2>(p, stream);                                                // 128 co x 128 t
        if (K == 11 && !tiny && blocks(64, 256) >= want) return launch_cfg<K, 2, 2, 1, 4>(p, stream);   //  64 co x 256 t
        // deep K, a few hundred tiles (FastPitch's second conv-FF conv, 1536 -> 384, at batch 7..13): 128 x 64 tiles with K split
        // (launch_cfg: < 320 tiles -> 2..4 slices) instead of twice as many 64 x 64 tiles that each walk all 96 chunks
        if (p.splitk_ws && p.Cin >= 1024 && !tiny && blocks(128, 64) >= 160 && blocks(128, 64) < 320 &&
            (int64_t)2 * p.batch * p.Cout * p.Nout <= p.splitk_floats) {
            const char* e = opt_str(OPT_DEEP_SPLITK);                    // read per launch of this (rare) shape: the tests flip it
            if (!(e && e[0] == '0')) return launch_cfg<K, 1, 2, 4, 1>(p, stream);
        }
        if (blocks(128, 64) >= want || (tiny && blocks(64, 64) < 2 * want)) return launch_cfg<K, 1, 2, 4, 1>(p, stream);   // 128 co x 64 t
        return launch_cfg<K, 1, 1, 2, 2>(p, stream);                                                    //  64 co x  64 t
    }
    if (p.CoutP % 64 == 0) {
        if (!tiny && blocks(64, 256) >= want) return launch_cfg<K, 2, 2, 1, 4>(p, stream);              //  64 co x 256 t
        return launch_cfg<K, 1, 1, 2, 2>(p, stream);                                                    //  64 co x  64 t
    }
    if (!tiny && blocks(32, 256) >= want) return launch_cfg<K, 1, 2, 1, 4>(p, stream);                  //  32 co x 256 t
    return launch_cfg<K, 1, 1, 1, 4>(p, stream);                                                        //  32 co x 128 t
}

// TTSAMD_CONV_LOG=<file>: one CSV line per conv launch, in launch order (= rocprofv3 dispatch order), so that
// profiles/traffic_from_pmc.py can put the algorithmic HBM bytes of each launch next to the measured ones.
void conv_log(const char* kind, int K, int cin, int cout, int nout, int batch, int has_res, int mode, int len_mul, int ragged,
              int n_phase) {
    static FILE* f = [] { const char* e = getenv("TTSAMD_CONV_LOG"); return e ? fopen(e, "a") : (FILE*)nullptr; }();
    if (!f) return;
    fprintf(f, "%s,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d\n", kind, K, cin, cout, nout, batch, has_res, mode, len_mul, ragged, n_phase);
    fflush(f);
}

int32_t launch_conv(const ConvParams& p, hipStream_t stream) {
    TTS_REQUIRE(p.CoutP % 32 == 0 && p.CoutP >= p.Cout, "conv: bad CoutP=%d", p.CoutP);
    TTS_REQUIRE(p.n_phase >= 1 && p.batch >= 1, "conv: bad n_phase/batch");
    TTS_REQUIRE(p.dil >= -DMAX && p.dil <= DMAX && p.dil != 0, "conv: dilation %d outside [-%d,%d]", p.dil, DMAX, DMAX);
    if (p.Nout <= 0) return 0;
    const int wino = wino_route(p);
    conv_log(p.precision == 0 ? (wino == 3 ? "f32_wino4" : (wino ? "f32_wino" : "f32")) : "bf16", p.K, p.Cin, p.Cout, p.Nout, p.batch, p.res != nullptr, p.mode, p.len_out_mul,
             p.lens_out != nullptr, p.n_phase);
    if (p.precision != 0) return launch_conv_bf16_any(p, stream);
    TTS_REQUIRE(!p.x_packed && !p.y_packed, "conv: packed bf16 activations exist only in the bf16 mode");
#ifdef TTS_WITH_DIRECT   /* tools/conv_bench.hip only: A/B against tools/conv_direct_f32.hip (round 3, no gain: DESIGN.md §4) */
    {
        const char* de = exp_env("TTSAMD_DIRECT");
        if (de && de[0] == '1' && direct_supported(p)) return launch_direct(p, stream);   // opt in: the tool's default is the product kernel
    }
#endif
#ifdef TTS_ONLY_K   /* kernel experiments: compile one kernel size only (tools/conv_bench, 10 s instead of 90 s) */
    if (p.K == TTS_ONLY_K) return launch_k<TTS_ONLY_K>(p, stream);
    set_error("conv: built with TTS_ONLY_K=%d", TTS_ONLY_K);
    return TTSAMD_EINVAL;
#else
    if (wino) return wino == 3 ? launch_wino4(p, stream) : (wino == 2 ? launch_wino2(p, stream) : launch_wino(p, stream));
    switch (p.K) {
        case 1: return launch_k<1>(p, stream);
        case 2: return launch_k<2>(p, stream);
        case 3: return launch_k<3>(p, stream);
        case 5: return launch_k<5>(p, stream);
        case 7: return launch_k<7>(p, stream);
        case 11: return launch_k<11>(p, stream);
        default:
            set_error("conv: kernel size %d not instantiated (1,2,3,5,7,11)", p.K);
            return TTSAMD_EINVAL;
    }
#endif
}

// torch Conv1d weight [Cout][Cin][K] -> [Cin/8][K][2][CoutP][4]:
//   out[(((o*K + t)*2 + kk)*CoutP + co)*4 + p] = w[co][8*o + 2*p + kk][t]
void pack_conv_weight(const float* w, int cout, int cin, int k, float* out) {
    const int cp = cout_padded(cout);
    for (int o = 0; o < cin / 8; ++o)
        for (int t = 0; t < k; ++t)
            for (int kk = 0; kk < 2; ++kk) {
                float* dst = out + (((int64_t)o * k + t) * 2 + kk) * cp * 4;
                for (int co = 0; co < cp; ++co)
                    for (int pq = 0; pq < 4; ++pq)
                        dst[co * 4 + pq] = co < cout ? w[((int64_t)co * cin + (8 * o + 2 * pq + kk)) * k + t] : 0.f;
            }
}

static inline uint16_t host_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float host_bf16_to_f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// fp32 packed weights (any number of phases) -> two bf16 planes in the same element order:
// out[0 .. n) = hi = bf16(w), out[n .. 2n) = lo = bf16(w - hi)
void split_packed_bf16(const float* packed, int64_t n, uint16_t* out) {
    for (int64_t i = 0; i < n; ++i) {
        const uint16_t h = host_bf16_rne(packed[i]);
        out[i] = h;
        out[n + i] = host_bf16_rne(packed[i] - host_bf16_to_f(h));
    }
}

// ConvTranspose1d(stride u, kernel Kt = 2u, padding p): y[co][q*u+rho] =
//   sum_ci W[ci][co][ka] x[ci][q+delta] + W[ci][co][ka+u] x[ci][q+delta-1],
//   ka = (rho+p) % u, delta = (rho+p) / u     (vocoder/hifigan/models.py:96-99)
// -> u polyphase 2-tap filters in the layout above: [u][Cin/8][2 taps][2][CoutP][4]
void pack_convt_weight(const float* w, int cin, int cout, int kt, int u, int p, float* out) {
    const int cp = cout_padded(cout);
    for (int rho = 0; rho < u; ++rho) {
        const int ka = (rho + p) % u;
        for (int o = 0; o < cin / 8; ++o)
            for (int t = 0; t < 2; ++t) {
                const int kidx = ka + t * u;
                for (int kk = 0; kk < 2; ++kk) {
                    float* dst = out + ((((int64_t)rho * (cin / 8) + o) * 2 + t) * 2 + kk) * cp * 4;
                    for (int co = 0; co < cp; ++co)
                        for (int pq = 0; pq < 4; ++pq)
                            dst[co * 4 + pq] = (co < cout && kidx < kt)
                                                   ? w[((int64_t)(8 * o + 2 * pq + kk) * cout + co) * kt + kidx]
                                                   : 0.f;
                }
            }
    }
}

}  // namespace ttsamd
