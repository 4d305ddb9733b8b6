// Winograd F(2,3) DECOMPOSITION kernel of the exact-fp32 engine: a k = 3 / 7 / 11 Conv1d is the sum of NS = k / 3 three-tap
// sub-filters (each one F(2,3): 4 products per output pair instead of 6) and NL = k - 3 NS single taps (2 products per pair):
// 4 instead of 6 products per pair at k = 3, 10 instead of 14 at k = 7 (0.71), 16 instead of 22 at k = 11 (0.73) -- with the
// transforms of F(2,3) only (entries 0, +-1, +-1/2: fp32-safe; a direct F(2,7) / F(2,11) is not).  The output transform is linear
// and the same for every sub-filter, so ALL groups of a conv accumulate into the same four planes
//     P_i += U_{s,i} V_{s,i}   (sub-filter s, i = 0..3),      P_0 += g_t x[q + (t - pad) d],  P_3 += (-g_t) x[q + d + (t - pad) d]   (single tap t)
//     y[q] = P_0 + P_1 + P_2,   y[q + d] = P_1 - P_2 - P_3          (dilation d: the output pair is (q, q + d))
// Serves every ResBlock conv of HiFi-GAN's stages with 128 / 256 channels and the un-fused ones with 64 (vocoder/hifigan/models.py:
// 30-53) and FastPitch's conv-FF convs (transformer.py:59-65); routing: wino_route (conv_wino.hip).  tests/test_wino_decomposition_cpu.py
// states the arithmetic in numpy, tests/test_gpu_wino.py checks the kernel against float64.
//
// Anatomy (what differs from the round's first Winograd kernel, conv_wino.hip):
//   * the WEIGHTS DO NOT GO THROUGH LDS (NG x 2 x 128 float4 per octet would be 40-64 KB per stage): a wave reads its own A fragments
//     -- one 16-byte load per lane and (group, 32-row tile), 1 KB contiguous per wave instruction, the packed layout
//     [Cin/8][NG][2][CoutP][4] is the operand order -- from L2 into a register queue PF groups ahead (as resblock_fused2.hip);
//   * LDS = the transformed / shifted input planes only: [octet][kk][group][pair] float4, 16-40 KB per stage;
//   * staging without clamps or masks: a wave instruction reads one channel row, the row is the base of a raw buffer descriptor and the
//     range check returns the halo zeros; leaky-relu once per loaded value pair, planes as packed adds -- 0.8 VALU per MFMA (the first
//     version, activation and mask inside every plane expression: 3.1, and no faster than the direct kernel);
//   * 128 rows x 64 pairs per block (2 x 2 waves), or 64 rows x 128 pairs (1 x 4) for Cout = 64 with k = 11 in two PHASES of 8 groups.
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <vector>

#include "conv_mfma_common.hpp"
#include "bfo.hpp"

namespace ttsamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int K, int NOCT_, int NSTAGE_, int WM_, int NPH_>
struct Wino2Geo {
    static constexpr int MT = 2, WM = WM_, WN = 4 / WM_;      // 2 x 2 waves: 128 rows x 64 pairs;  1 x 4: 64 rows x 128 pairs (Cout = 64)
    static constexpr int NS = K / 3, NL = K - 3 * NS, NG = 4 * NS + 2 * NL;   // sub-filters, single taps, operand groups per octet
    static constexpr int NOCT = NOCT_, NSTAGE = NSTAGE_;
    // PHASES: an octet's NG groups in NPH steps of NGP groups, each with its own LDS stage (k = 11 at 128 pairs: 2 x 8 groups, a stage
    // of all 16 would be 64 KB); a phase stages only the positions its groups touch
    static constexpr int NPH = NPH_, NGP = NG / NPH_;
    static constexpr int CO_BLK = WM * MT * 32;
    static constexpr int NPAIR = WN * 32;                     // output pairs per block
    static constexpr int NHALF = NPAIR / 64;                  // 64-pair halves a staging thread covers
    static constexpr int NT_BLK = 2 * NPAIR;                  // outputs per block (dilation 1)
    static constexpr int BUF4 = NOCT * 2 * NGP * NPAIR;       // float4s per stage: [octet][kk][group of the phase][pair]
    static constexpr int NGC = NOCT * NGP;                    // operand groups per step
    static constexpr int PF = 2;                              // weight groups in flight ahead of the one being multiplied
    static constexpr int NM = 4 * MT;                         // MFMAs per operand group
    static constexpr int NGAP = NGC * NM;                     // gaps (one per MFMA) per step
    static constexpr int DA = 2 * NM;                         // a value is activated 16 MFMAs (~1000 cycles) after its load was issued
    // first / last input position (0..K) of the groups [g_lo, g_hi] a phase holds
    __host__ __device__ static constexpr int gpos(int g, bool last) {
        return g < 4 * NS ? 3 * (g / 4) + (last ? 3 : 0) : 3 * NS + (g - 4 * NS) / 2 + (last ? 1 : 0);
    }
    __host__ __device__ static constexpr int glo(int ph) { return ph * NGP; }
    __host__ __device__ static constexpr int mlo(int ph) { return gpos(ph * NGP, false); }
    __host__ __device__ static constexpr int npos(int ph) { return gpos(ph * NGP + NGP - 1, true) - mlo(ph) + 1; }
    static constexpr int NPOSP = npos(0) > npos(NPH - 1) ? npos(0) : npos(NPH - 1);
    __host__ __device__ static constexpr int nlj(int ph) { return NOCT * 2 * NHALF * npos(ph); }      // load jobs (one value each) of a phase
    static constexpr int NWJ = NGC * NHALF;                   // write jobs (one plane of one half each)
    static constexpr int LPG = 3 * (NOCT * 2 * NHALF * NPOSP) > NGAP ? 2 : 1;                          // loads per gap
    __host__ __device__ static constexpr int tw0(int ph) { return (nlj(ph) + LPG - 1) / LPG + DA; }    // first gap with every value activated
    __host__ __device__ static constexpr int ws(int ph) { return (NGAP - tw0(ph)) / NWJ; }             // one plane write every ws gaps after it
    static_assert(NG % NPH == 0 && (NPH == 1 || (NOCT == 1 && (NGP % 4 == 0))), "phases split at sub-filter boundaries");
    static_assert(NGC % PF == 0, "queue slots line up across steps");
    static_assert(ws(0) >= 1 && ws(NPH - 1) >= 1, "one write job per gap at most");
    static_assert((size_t)NSTAGE * BUF4 * 16 <= 80 * 1024, "two blocks per CU");
    static_assert(NPH == 1 || NSTAGE == 2, "phases alternate between two stages");
};

// plane (accumulator) a group feeds: (s, i) -> i;  single tap l: its y[2j] half -> 0, its y[2j + 1] half -> 3
template <int K>
__device__ __host__ constexpr int wino2_plane(int g) {
    return g < 4 * (K / 3) ? (g & 3) : (((g - 4 * (K / 3)) & 1) ? 3 : 0);
}

// outputs per tile at dilation d (see the kernel): the largest multiple of 2 d and of 4 in 2 * npair
__device__ __host__ constexpr int wino2_tile(int d, int npair) { return (2 * npair / ((d & 1) ? 4 * d : 2 * d)) * ((d & 1) ? 4 * d : 2 * d); }

template <int K, int NOCT_, int NSTAGE_, int WM_, int NPH_, int EPI>
__global__ __launch_bounds__(256, TTS_MINWAVES) void conv1d_wino2_f32(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    using G = Wino2Geo<K, NOCT_, NSTAGE_, WM_, NPH_>;
    constexpr int MT = G::MT, WN = G::WN, NS = G::NS, NG = G::NG, NOCT = G::NOCT, NSTAGE = G::NSTAGE;
    constexpr int CO_BLK = G::CO_BLK, NPAIR = G::NPAIR, NT_BLK = G::NT_BLK, NGC = G::NGC, PF = G::PF;
    constexpr int NHALF = G::NHALF, NGP = G::NGP, NPH = G::NPH, NPOSP = G::NPOSP, NM = G::NM, DA = G::DA, LPG = G::LPG, NWJ = G::NWJ;
    // DILATION d: the conv over x[q + (t - pad) d] is the dilation-1 conv of every d-th sample, so the output pair of F(2,3) is
    // (q, q + d): pair pc of a tile sits at column (pc / d) 2 d + pc % d.  d = 1: every pair slot = 128 (256) outputs per tile; d = 3 /
    // 5: 120 (252 / 240) outputs (a multiple of 2 d and of 4: tiles stay float4-aligned), the last pair slots of the MFMA tile idle.
    const int dil = p.dil;
    const int nt_eff = wino2_tile(dil, NPAIR), npair_eff = nt_eff / 2;
    int b = blockIdx.z;
    int q0 = blockIdx.x * nt_eff;
    if (p.compact) {   // dead blocks last (live_tile, common.hpp)
        int tile = 0;
        if (!live_tile(p.lens_out, p.len_out_mul, p.Nout, nt_eff, p.batch, blockIdx.z * gridDim.x + blockIdx.x, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * nt_eff;
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
    const int n_chunks = p.Cin / (8 * NOCT);
    const int n_groups = (p.Cin / 8) * NG;                    // operand groups of the whole conv
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    const float in_slope = p.in_slope;
    const int pad = (K - 1) / 2;

    float ep_bias = 0.f;
    if (tid < CO_BLK && p.bias) ep_bias = p.bias[min(co_blk0 + tid, p.Cout - 1)];

    const int kk = lane >> 5, l31 = lane & 31;
    const int pe = wn * 32 + l31;                                           // the pair this lane holds in the accumulators
    const int col_e = dil == 1 ? 2 * pe : (pe / dil) * 2 * dil + pe % dil;  // ... its first column in the tile (the second: + dil)
    constexpr bool preload = EPI == 3;
    f32x16 acc[4][MT];

    // ---- weight queue: group gf = octet * NG + g of the conv sits at w_wino + gf * 2 CoutP float4 (the scalar offset of the buffer
    // load, advanced by one group per refill and clamped at the conv's last group: an L1 hit, unused), this lane's fragment of row
    // tile mt at + kk * CoutP + co_blk0 + 32 (wm MT + mt) + l31.  The first PF groups go out before anything else.
    const bfo_i4 wrs = bfo_rsrc(p.w_wino, (unsigned)n_groups * 2u * (unsigned)CoutP * 16u);
    const int wv = (kk * CoutP + co_blk0 + wm * MT * 32 + l31) * 16;
    const int wstep = 2 * CoutP * 16, wlast = (n_groups - 1) * wstep;
    int wso = 0;                                              // scalar offset of the next group to fetch
    f32x4 aq[PF][MT];
#pragma unroll
    for (int g = 0; g < PF; ++g) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aq[g][mt] = __builtin_bit_cast(f32x4, bfo_ld16(wrs, wv + 32 * 16 * mt, wso, 0));
        wso = min(wso + wstep, wlast);
    }

#define TTS_INIT_ACC()                                                                                           \
    {                                                                                                            \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                            \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                       \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[g][i][r] = 0.f;                               \
        if (preload) {                                                                                           \
            const int wm_s = __builtin_amdgcn_readfirstlane(wm);                                                 \
            const int row0 = co_blk0 + wm_s * MT * 32;                                                           \
            const int q = q0 + col_e;                                                                            \
            /* a pair cut by the utterance end (or an idle pair slot) loads values that are never stored; past the tensor the range */ \
            /* check returns zeros */                                                                            \
            const int voff = (q < n_out ? q : 0) * 4;                                                            \
            const int vd = 4 * dil;                                                                              \
            {                                                                                                    \
                const int r_cs = p.r_cs;                                                                         \
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res + (int64_t)b * p.r_bs), 0, \
                                                                  p.Cout * r_cs * 4, 0x00020000);                \
                const int vk = 4 * kk * r_cs * 4;                                                                \
                _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                   \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        const int so = (row0 + i * 32 + (r & 3) + 8 * (r >> 2)) * r_cs * 4;                      \
                        acc[0][i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + vk, so, 0));      \
                        acc[3][i][r] = -__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + vk + vd, so, 0)); \
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
                        t1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, voff + vk + vd, so, 0)); \
                    }                                                                                            \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        acc[0][i][r] = t0[r] + acc[0][i][r];                                                     \
                        acc[3][i][r] = acc[3][i][r] - t1[r];                                                     \
                    }                                                                                            \
                }                                                                                                \
            }                                                                                                    \
        }                                                                                                        \
    }


    // ---- staging.  X item of thread (h, kks, pc), octet ol and 64-pair half hf: channels 8 (c NOCT + ol) + 2 (2 h + pp) + kks, pp = 0 / 1,
    // at the positions q0 + col(pc + 64 hf) + (m - pad) d, m = the phase's window -> components 2 h, 2 h + 1 of the float4 of each
    // plane of the phase at [ol][kks][g][pc + 64 hf].
    // (h, kks) = the wave index: a wave instruction reads ONE channel row, so the row is the base of a raw buffer descriptor (two
    // scalar adds per row and chunk) of in_len * 4 bytes and the load's range check returns the zeros of the halo: left of
    // position 0 (the negative offset wraps) and right of the utterance.  No clamps, no masks.
    const int sh = __builtin_amdgcn_readfirstlane((tid >> 7) & 1), skk = __builtin_amdgcn_readfirstlane((tid >> 6) & 1);
    f32x2 sx[NOCT][NHALF][NPOSP];                            // [pp] = the two channels of the item, packed: one v_pk_add_f32 per plane
    int xv0[NHALF];
#pragma unroll
    for (int hf = 0; hf < NHALF; ++hf) {
        const int spe = min(lane + 64 * hf, npair_eff - 1);    // idle pair slots (d > 1) repeat the last pair: never stored
        xv0[hf] = (q0 + (dil == 1 ? 2 * spe : (spe / dil) * 2 * dil + spe % dil) - pad * dil) * 4;
    }
    const int xvd = 4 * dil;
    const int ch_off = (4 * sh + skk) * x_cs;                 // channel 2 (2 h) + kks; pp adds 2 rows, the octet 8

    // load job J of phase PH -> (octet, half, position, pp): the two channels of a value pair in consecutive jobs
#define TTS_JOB_IDX(PH, J)                                                                           \
        const int np_ = G::npos(PH);                                                                 \
        const int ol_ = (J) / (2 * NHALF * np_), hf_ = ((J) / (2 * np_)) % NHALF, mi_ = ((J) / 2) % np_, pp_ = (J) % 2;
#define TTS_LOAD_JOB(PH, J, XSO)                                                                     \
    {                                                                                                \
        TTS_JOB_IDX(PH, J)                                                                           \
        const bfo_i4 xrs_ = bfo_rsrc(xb + ((XSO) + ch_off + (8 * ol_ + 2 * pp_) * x_cs), (unsigned)in_len * 4u);     \
        sx[ol_][hf_][mi_][pp_] = bfo_ld4f(xrs_, xv0[hf_] + xvd * (G::mlo(PH) + mi_), 0, 0);         \
    }
    // leaky-relu on load, once per value pair, after its second load (slopes in [0, 1]: max(x, slope x); slope 1 = the identity, exactly)
#define TTS_ACT_JOB(PH, J)                                                                           \
    if ((J) & 1) {                                                                                   \
        TTS_JOB_IDX(PH, J)                                                                           \
        (void)pp_;                                                                                   \
        const f32x2 w_ = sx[ol_][hf_][mi_] * in_slope;                                               \
        sx[ol_][hf_][mi_].x = fmaxf(sx[ol_][hf_][mi_].x, w_.x);                                      \
        sx[ol_][hf_][mi_].y = fmaxf(sx[ol_][hf_][mi_].y, w_.y);                                      \
    }
    // plane g of octet ol from the staged values: (s, i) -> the F(2,3) input transform of positions 3 s + {0..3}; single tap l ->
    // the value at position 3 NS + l (for y[2j]) or 3 NS + l + 1 (for y[2j + 1]).  Job J -> (octet, group of the phase, half)
#define TTS_SX(M) sx[ol_][hf_][(M) - G::mlo(PH_)]
#define TTS_WRITE_JOB(PH, J, WR)                                                                     \
    {                                                                                                \
        const int PH_ = (PH);                                                                        \
        const int hf_ = (J) % NHALF, ol_ = ((J) / NHALF) / NGP, gl_ = ((J) / NHALF) % NGP, g_ = G::glo(PH_) + gl_;   \
        f32x2 v2;                                                                                    \
        if (g_ < 4 * NS) {                                                                           \
            const int s3 = 3 * (g_ / 4), i_ = g_ % 4;                                                \
            if (i_ == 0) v2 = TTS_SX(s3) - TTS_SX(s3 + 2);                                           \
            else if (i_ == 1) v2 = TTS_SX(s3 + 1) + TTS_SX(s3 + 2);                                  \
            else if (i_ == 2) v2 = TTS_SX(s3 + 2) - TTS_SX(s3 + 1);                                  \
            else v2 = TTS_SX(s3 + 1) - TTS_SX(s3 + 3);                                               \
        } else {                                                                                     \
            v2 = TTS_SX(3 * NS + (g_ - 4 * NS) / 2 + ((g_ - 4 * NS) & 1));                           \
        }                                                                                            \
        (WR)[2 * ((ol_ * 2 * NGP + gl_) * NPAIR + 64 * hf_)] = v2;                                   \
    }

    const float4* sB = smem4 + kk * NGP * NPAIR + wn * 32 + l31;      // + stage * BUF4 + (ol * 2 NGP + g) * NPAIR
    f32x2* sW = reinterpret_cast<f32x2*>(smem4 + skk * NGP * NPAIR + lane) + sh;    // + 2 (stage * BUF4 + (ol * 2 NGP + g) * NPAIR + 64 hf)
    float4 bq[2];

    // prologue: fill NSTAGE - 1 stages (steps 0 .. NSTAGE - 2: phase st % NPH of chunk st / NPH), fetch the first B operand
    {
#pragma unroll
        for (int st = 0; st < NSTAGE - 1; ++st) {
            const int ph = st % NPH;
            const int xso = min(st / NPH, n_chunks - 1) * 8 * NOCT * x_cs;
#pragma unroll
            for (int J = 0; J < G::nlj(ph); ++J) TTS_LOAD_JOB(ph, J, xso)
            if (st == 0) TTS_INIT_ACC()
#pragma unroll
            for (int J = 0; J < G::nlj(ph); ++J) TTS_ACT_JOB(ph, J)
            f32x2* wr = sW + 2 * st * G::BUF4;
#pragma unroll
            for (int J = 0; J < NWJ; ++J) TTS_WRITE_JOB(ph, J, wr)
        }
    }
    __syncthreads();
    bq[0] = sB[0];

    int stage = 0;  // step % NSTAGE
    for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            // the step being staged: NSTAGE - 1 ahead (tail: the last chunk is re-staged into a dead stage -- branch-free body)
            const int tph = (ph + NSTAGE - 1) % NPH;
            const int xso = min(c + (ph + NSTAGE - 1) / NPH, n_chunks - 1) * 8 * NOCT * x_cs;
            const int stage_next = (stage + 1 == NSTAGE) ? 0 : stage + 1;
            const int stage_fill = (stage == 0) ? NSTAGE - 1 : stage - 1;
            f32x2* wr = sW + 2 * stage_fill * G::BUF4;
            const float4* rd = sB + stage * G::BUF4;
            const int sn = (c + 1 < n_chunks || ph + 1 < NPH) ? stage_next : stage;
#pragma unroll
            for (int g = 0; g < NGC; ++g) {
                const int cur = g & 1, nxt = cur ^ 1;
                const int plane = wino2_plane<K>(G::glo(ph) + g % NGP);
                f32x4 a4[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a4[mt] = aq[g % PF][mt];
                // refill the queue slot with the group PF ahead
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) aq[g % PF][mt] = __builtin_bit_cast(f32x4, bfo_ld16(wrs, wv + 32 * 16 * mt, wso, 0));
                wso = min(wso + wstep, wlast);
                // B operand of the next group: same stage, or (three stages) the first group of the next step's stage
                if (g + 1 < NGC) bq[nxt] = rd[(((g + 1) / NGP) * 2 * NGP + (g + 1) % NGP) * NPAIR];
                else if (NSTAGE >= 3) bq[nxt] = sB[sn * G::BUF4];
                __builtin_amdgcn_sched_barrier(0);
                const float bv[4] = {bq[cur].x, bq[cur].y, bq[cur].z, bq[cur].w};
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int pq = m / MT, i = m % MT;
                    acc[plane][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][pq], bv[pq], acc[plane][i], 0, 0, 0);
                    // ---- gap work for the step NSTAGE - 1 ahead: LPG loads per gap first, each value activated DA gaps later, then
                    // the plane writes into the stage the previous step has left
                    const int t = g * NM + m;
#pragma unroll
                    for (int u = 0; u < LPG; ++u)
                        if (t * LPG + u < G::nlj(tph)) TTS_LOAD_JOB(tph, t * LPG + u, xso)
#pragma unroll
                    for (int u = 0; u < LPG; ++u)
                        if (t >= DA && (t - DA) * LPG + u < G::nlj(tph)) TTS_ACT_JOB(tph, (t - DA) * LPG + u)
                    if (t >= G::tw0(tph) && (t - G::tw0(tph)) % G::ws(tph) == 0 && (t - G::tw0(tph)) / G::ws(tph) < NWJ)
                        TTS_WRITE_JOB(tph, (t - G::tw0(tph)) / G::ws(tph), wr)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            if (NSTAGE < 3) bq[0] = sB[sn * G::BUF4];            // two stages: the next step's stage has only just been written
            stage = stage_next;
        }
    }
#undef TTS_INIT_ACC
#undef TTS_LOAD_JOB
#undef TTS_ACT_JOB
#undef TTS_JOB_IDX
#undef TTS_SX
#undef TTS_WRITE_JOB

    // ---- epilogue: output transform, then the row epilogue of conv_mfma.hip (bias, residual, ReLU, accumulate modes)
    constexpr int LDS_F = CO_BLK * NT_BLK + CO_BLK;                       // the launcher allocates max(ring, this)                         // floats of LDS this block owns
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
            if (pe < npair_eff) {
                ep[row * NT_BLK + col_e] = y2.x;
                ep[row * NT_BLK + col_e + dil] = y2.y;
            }
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
                if (q >= n_out || col >= nt_eff) continue;
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
                if (q >= n_out || col >= nt_eff) continue;
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


template <int K, int NOCT, int NSTAGE, int WM, int NPH, int EPI>
static int32_t launch_wino2_epi(const ConvParams& q, dim3 grid, hipStream_t stream) {
    using G = Wino2Geo<K, NOCT, NSTAGE, WM, NPH>;
    constexpr size_t ring = (size_t)G::NSTAGE * G::BUF4 * sizeof(float4);
    constexpr size_t epi = ((size_t)G::CO_BLK * G::NT_BLK + G::CO_BLK) * sizeof(float);
    constexpr size_t lds = ring > epi ? ring : epi;
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    static std::atomic<uint64_t> lds_done{0};
    const auto kern = conv1d_wino2_f32<K, NOCT, NSTAGE, WM, NPH, EPI>;
    TTS_CHECK_HIP(lds_opt_in((const void*)kern, (int)lds, lds_done));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int K, int NOCT, int NSTAGE, int WM, int NPH>
static int32_t launch_wino2_cfg(const ConvParams& p, hipStream_t stream) {
    using G = Wino2Geo<K, NOCT, NSTAGE, WM, NPH>;
    const int nt = wino2_tile(p.dil, G::NPAIR);
    dim3 grid((p.Nout + nt - 1) / nt, p.CoutP / G::CO_BLK, p.batch);
    ConvParams q = p;
    q.ksplit = 1;
    q.compact = compact_order(p.lens_out, p.batch) ? 1 : 0;
    if (p.res != nullptr) return launch_wino2_epi<K, NOCT, NSTAGE, WM, NPH, 3>(q, grid, stream);
    return launch_wino2_epi<K, NOCT, NSTAGE, WM, NPH, 0>(q, grid, stream);
}

int32_t launch_wino2(const ConvParams& p, hipStream_t stream) {
    // larger chunks (k = 3: 32 channels, two stages; k = 7: 16 channels, two stages) measured equal within the noise (65.2 / 65.6 vs 65.3 ms)
    if (p.CoutP % 128 == 0) {      // 128 rows x 64 pairs per block
        if (p.K == 3) return launch_wino2_cfg<3, 2, 3, 2, 1>(p, stream);      // 16-channel chunks: 64 MFMAs per wave between barriers, 48 KB ring
        if (p.K == 7) return launch_wino2_cfg<7, 1, 3, 2, 1>(p, stream);      // 80 MFMAs, 60 KB ring
        if (p.K == 11) return launch_wino2_cfg<11, 1, 2, 2, 1>(p, stream);    // 128 MFMAs, 64 KB ring (two stages)
    } else {                       // Cout = 64: 64 rows x 128 pairs
        if (p.K == 3) return launch_wino2_cfg<3, 2, 2, 1, 1>(p, stream);      // 64 MFMAs, 64 KB ring
        if (p.K == 7) return launch_wino2_cfg<7, 1, 2, 1, 1>(p, stream);      // 80 MFMAs, 80 KB ring
        if (p.K == 11) return launch_wino2_cfg<11, 1, 2, 1, 2>(p, stream);    // two phases of 8 groups: 64 MFMAs each, 64 KB ring
    }
    set_error("wino2: kernel size %d not built (3, 7, 11)", p.K);
    return TTSAMD_EINVAL;
}

// outputs per block of the kernel launch_wino2 picks (the routing's block count)
int wino2_block_outputs(int coutp, int dil) { return wino2_tile(dil, coutp % 128 == 0 ? 64 : 128); }

int wino2_groups(int k) { return 4 * (k / 3) + 2 * (k - 3 * (k / 3)); }

// host: torch Conv1d weight [Cout][Cin][K] -> the NG group filters as an NG-tap conv in the engine's packed layout [Cin/8][NG][2][CoutP][4]
void pack_wino2_weight(const float* w, int cout, int cin, int k, float* out) {
    const int ns = k / 3, nl = k - 3 * ns, ng = 4 * ns + 2 * nl;
    std::vector<float> u((size_t)cout * cin * ng);
    for (int64_t i = 0; i < (int64_t)cout * cin; ++i) {
        const float* g = w + i * k;
        float* o = u.data() + i * ng;
        for (int s = 0; s < ns; ++s) {
            const double g0 = g[3 * s], g1 = g[3 * s + 1], g2 = g[3 * s + 2];
            o[4 * s] = (float)g0;
            o[4 * s + 1] = (float)((g0 + g1 + g2) * 0.5);
            o[4 * s + 2] = (float)((g0 - g1 + g2) * 0.5);
            o[4 * s + 3] = (float)g2;
        }
        for (int l = 0; l < nl; ++l) {
            o[4 * ns + 2 * l] = g[3 * ns + l];
            o[4 * ns + 2 * l + 1] = -g[3 * ns + l];
        }
    }
    pack_conv_weight(u.data(), cout, cin, ng, out);
}

}  // namespace ttsamd
