// Second generation of the fused c1 -> c2 ResBlock1 pair in exact fp32 (vocoder/hifigan/models.py:46-53):
//     y = x + conv1d(lrelu(conv1d(lrelu(x), w1, dil d) + b1), w2, dil 1) + b2
// for C = 32 / 64 / 128 at k = 3 / 7 / 11, on v_mfma_f32_32x32x2_f32.
//
// What bounds resblock_pair (resblock_fused.hip) at k = 3 is not the matrix pipe but the block's own rhythm: one barrier per
// 8-channel weight chunk (24-48 MFMAs = 0.6-1.3 us), 2 C / 8 of them per block, a prologue that queues the first operands
// behind 64-128 residual loads, and a weight ring that costs a quarter of the LDS (one resident block less).  Here:
//   * WEIGHTS DO NOT GO THROUGH LDS.  A wave owns all C output rows of its columns and reads only its own A fragments -- one
//     16-byte load per lane and (octet, tap, 32-row tile), 1 KB contiguous per wave instruction, the packed layout
//     [Cin/8][K][2][C][4] is exactly the operand order -- from L2 into a register queue that runs two (octet, tap) groups
//     ahead (an fp32 group is 8-32 MFMAs = 0.5-2 k cycles, longer than an L2 round trip).  No barrier inside a conv.
//   * THE WINDOW IS STAGED RAW, once, by the whole block (leaky-relu is applied on the operand path: two VALU operations per
//     value next to 64-cycle MFMAs).  So the residual comes straight out of the LDS window after phase A, bit-exact, and
//     nothing but the window and the first weights stands in front of the first MFMA.
//   * Barriers per block: 1 after staging, 2 around the intermediate's LDS write, 2 in the epilogue's transposition --
//     5 instead of 19 (C = 64, k = 3).
//   * LDS = the window only (4 C (NB + (k - 1) d) bytes, sized by the ACTUAL dilation): C = 64 keeps two blocks per CU at
//     every k, C = 128 fits at all (one block of 256 columns, or two of 128 columns at k = 3 / 7).
// Tile: 4 waves side by side along time, each C rows x (32 NTW) columns; NB = 128 NTW columns per block of which
// TS = (NB - (k - 1)) & ~3 are stored (the (k - 1) / 2-column halo of the intermediate is recomputed by the neighbour).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "conv_mfma_common.hpp"

namespace ttsamd {

struct FusedPair2Params {
    const float* x;        // [B][C][L] input = residual
    float* y;              // [B][C][L]; must not alias x (other blocks read x's halo)
    const float4* w1;      // packed [C/8 oct][K][2][C][4]
    const float4* w2;
    const float4* w1w;     // WM = 2 kernels: conv 1 as Winograd groups too
    const float4* w2w;     // WM >= 1 kernels: conv 2 as Winograd F(2,3) groups, packed [C/8 oct][NGW][2][C][4] (conv_wino2.hip: pack_wino2_weight)
    const float* b1;
    const float* b2;
    const int64_t* lens;   // valid length = lens[b] * len_mul (nullptr -> L)
    int32_t len_mul, L, dil, batch;
    int32_t mode;          // 0: y = v   1: y = y + v   2: y = (y + v) / div
    float div, slope;
    int32_t compact;       // ragged batch: blocks take the lin-th LIVE tile (common.hpp: live_tile)
    int32_t exp;           // -DTTS_F2_EXP builds only (timing experiments, results wrong): TTSAMD_F2_EXP bit 0 no window loads,
                           // 1 no phase A, 2 no phase B, 3 no output stores
};

template <int K, int C, int NTW>
struct Fused2Geo {
    static constexpr int NOCT = C / 8, MT = C / 32;
    static constexpr int H = (K - 1) / 2;
    static constexpr int NB = 128 * NTW;                       // MFMA columns per block
    static constexpr int TS = (NB - 2 * H) & ~3;               // outputs per block
    static constexpr int TSTR = NB + K - 1;                    // columns of the intermediate incl. the over-read of dead MFMA columns
    static constexpr int NG = NOCT * K;                        // (octet, tap) operand groups per conv
    static constexpr int NS = K / 3, NL = K - 3 * NS, NGW = 4 * NS + 2 * NL;   // Winograd phase B: sub-filters, single taps, groups per octet
    static constexpr int TSH = TSTR / 2;                       // ... its intermediate: even / odd columns apart, TSH entries each (TSTR is even)
    static constexpr int PF = 2;                               // weight groups in flight ahead of the one being multiplied
    // octets per unrolled body of the group loop: the whole conv when it is at most ~640 MFMAs per wave, else as many octets as
    // fit (always an even number of groups per body, so the two queue slots line up across iterations; NOCT % UO == 0)
    static constexpr int MPO = 4 * MT * NTW * K;               // MFMAs per octet and wave
    static constexpr int UO = NOCT * MPO <= 640 ? NOCT : (4 * MPO <= 640 ? 4 : 2);
    static_assert(NOCT % UO == 0 && (UO * K) % PF == 0, "unrolled body of the group loop");
    // resident blocks per CU the register budget is declared for (the LDS window decides at run time whether they fit)
    static constexpr int WAVES = (C == 128 && NTW == 2) ? 1 : (C == 32 ? 4 : 2);
    // Winograd phase A at dilation d: output pairs (n, n + d), 128 pair slots of which d * (128 / d) tile the columns in groups of 2 d
    // -> 256 / 252 / 250 intermediate columns at d = 1 / 3 / 5, outputs per block = those minus the halo, a multiple of 4
    __host__ __device__ static constexpr int npa(int d) { return d * (128 / d); }
    __host__ __device__ static constexpr int ts_wa(int d) { return (2 * npa(d) - 2 * H) & ~3; }
    static size_t lds_bytes(int dil) { return (size_t)2 * NOCT * (NB + (K - 1) * dil) * sizeof(float4); }
};

// leaky_relu(v, slope) for 0 < slope <= 1 as max(v, v * slope): one multiply and ONE v_med3_f32(v, v * slope, +inf) -- fmaxf adds a
// canonicalising v_max(v, v) per operand on this target, and an inline-asm v_max is opaque to hipcc's hazard padding in front of the
// MFMA that reads it.  A NaN input stays a NaN either way.
__device__ __forceinline__ float lrelu_max(const float v, const float slope) {
    return __builtin_amdgcn_fmed3f(v, v * slope, __builtin_inff());
}

// One conv of the pair on one wave: acc[mt][j] += sum over the NG = (C/8) K groups g = (octet o, tap t) of
//   A = this lane's fragments of w[o][t] (queue `aq`, filled PF groups ahead from `wl`; the last PF slots are refilled with
//       the first groups of `wnext`, the weights of the conv that follows),
//   B = LDS entries (o, kk, column + 32 j + t bdil), row stride bstr; ACT: leaky-relu applied here (the window is staged raw).
// `sb` already points at (octet 0, this lane's kk, this lane's column).
template <int K, int C, int NTW, bool ACT>
__device__ __forceinline__ void conv_phase2(f32x16 (&acc)[C / 32][NTW], float4 (&aq)[2][C / 32], const float4* sb, const int bstr,
                                            const int bdil, const float4* __restrict__ wl, const float4* __restrict__ wnext,
                                            const float slope) {
    using G = Fused2Geo<K, C, NTW>;
    constexpr int NOCT = G::NOCT, MT = G::MT, NG = G::NG, PF = G::PF, UO = G::UO, GB = UO * K;
    float4 bq[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) bq[j] = sb[32 * j];
#pragma unroll 1
    for (int ob = 0; ob < NOCT; ob += UO) {
        const float4* sbo = sb + ob * 2 * bstr;
        const int g0 = ob * K;
#pragma unroll
        for (int gl = 0; gl < GB; ++gl) {
            float4 a4[MT], b4[NTW];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a4[mt] = aq[gl % PF][mt];
#pragma unroll
            for (int j = 0; j < NTW; ++j) b4[j] = bq[j];
            {   // refill the queue slot: group g + PF of this conv, or the first groups of the next one (scalar select)
                const int gn = g0 + gl + PF;
                const float4* __restrict__ src = gn < NG ? wl + (int64_t)gn * 2 * C : wnext + (int64_t)(gn - NG) * 2 * C;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) aq[gl % PF][mt] = src[32 * mt];
            }
            {   // B operands of the next group
                const int o1 = (gl + 1) / K, t1 = (gl + 1) % K;                 // relative to ob
                const float4* nb = (gl + 1 < GB) ? sbo + o1 * 2 * bstr + t1 * bdil
                                                 : sb + min(ob + UO, NOCT - 1) * 2 * bstr;   // (past the last group: unused)
#pragma unroll
                for (int j = 0; j < NTW; ++j) bq[j] = nb[32 * j];
            }
            // the loads above stay above the MFMAs of this group (hipcc otherwise sinks each one to right in front of its use and
            // waits for it with the matrix pipe idle): the B operands are then one group (>= 512 cycles) ahead, the weights PF groups
            __builtin_amdgcn_sched_barrier(0);
            float bv[NTW][4];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                bv[j][0] = b4[j].x; bv[j][1] = b4[j].y; bv[j][2] = b4[j].z; bv[j][3] = b4[j].w;
                if (ACT) {
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) bv[j][pq] = lrelu_max(bv[j][pq], slope);
                }
            }
#pragma unroll
            for (int pq = 0; pq < 4; ++pq)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float av = pq == 0 ? a4[mt].x : (pq == 1 ? a4[mt].y : (pq == 2 ? a4[mt].z : a4[mt].w));
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j][pq], acc[mt][j], 0, 0, 0);
                }
        }
    }
}

// Phase B as Winograd F(2,3) (conv_wino2.hip's decomposition: k / 3 three-tap sub-filters + k % 3 single taps over the same four
// planes): the pair of outputs (2 jw, 2 jw + 1) of this lane from the intermediate stored as EVEN / ODD column arrays -- column
// 2 jw + c is entry jw + c / 2 of array c & 1, so every operand read is contiguous over the lanes (stride-2 reads of an interleaved
// row would be 2-way bank conflicts).  Per sub-filter: four float4 reads (x0..x3, one bundle ahead), V = x0 - x2, x1 + x2, x2 - x1,
// x1 - x3 as packed adds, then its four groups of 4 MT MFMAs: 0.25 / MT reads and 0.5 / MT VALU per MFMA, 4/6, 10/14, 16/22 of the
// direct phase's MFMAs.  `sb` = row (octet 0, this lane's kk), entry jw of the even array; rows are 2 TSH entries apart.
// EO = false (phase A on the activated window): column c of the pair's sub-sequence is entry c * cstep (= the dilation) of a row of
// `rstride` entries; `wnext`: the conv that follows (its first PF groups refill the queue at the tail), nullptr = none.
// MTW: 32-row tiles this wave owns (C / 32, or half of them in the 8-wave C = 128 kernel: `wl` then points at its first row).
template <int K, int C, bool EO, int MTW>
__device__ __forceinline__ void conv_phase2_wino(f32x16 (&acc)[4][MTW], float4 (&aq)[2][MTW], const float4* sb,
                                                 const float4* __restrict__ wl, const int rstride, const int cstep,
                                                 const float4* __restrict__ wnext) {
    using G = Fused2Geo<K, C, 2>;
    constexpr int NOCT = G::NOCT, MT = MTW, PF = G::PF, NS = G::NS, NL = G::NL, NGW = G::NGW, TSH = G::TSH;
    constexpr int NBU = NS + NL;                               // operand bundles per octet (a sub-filter: 4 entries, a single tap: 2)
    constexpr int NGT = NOCT * NGW;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    // entry of intermediate column 2 jw + c
#define TTS_ENT(SBO, CC) (SBO)[EO ? ((CC) & 1) * TSH + ((CC) >> 1) : (CC) * cstep]
    float4 bq[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bq[0][m] = TTS_ENT(sb, m);     // bundle 0 of octet 0 = sub-filter 0 (every k has one)
#pragma unroll 1
    for (int o = 0; o < NOCT; ++o) {
        const float4* sbo = sb + o * 2 * rstride;              // (2 rows per octet)
        const float4* sbn = sb + min(o + 1, NOCT - 1) * 2 * rstride;
#pragma unroll
        for (int u = 0; u < NBU; ++u) {
            const int cur = u & 1, nxt = cur ^ 1;              // (NBU odd would flip the slots between octets: both k = 7 and 11 have NBU = 3 / 5 ... handled by `cur` below)
            // next bundle's reads: the next sub-filter / tap of this octet, or sub-filter 0 of the next octet
            {
                const int un = u + 1;
                if (un < NS) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) bq[nxt][m] = TTS_ENT(sbo, 3 * un + m);
                } else if (un < NBU) {
                    bq[nxt][0] = TTS_ENT(sbo, 3 * NS + (un - NS));
                    bq[nxt][1] = TTS_ENT(sbo, 3 * NS + (un - NS) + 1);
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) bq[nxt][m] = TTS_ENT(sbn, m);
                }
            }
            f32x4v x[4], v[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) x[m] = f32x4v{bq[cur][m].x, bq[cur][m].y, bq[cur][m].z, bq[cur][m].w};
            const bool sub = u < NS;
            if (sub) { v[0] = x[0] - x[2]; v[1] = x[1] + x[2]; v[2] = x[2] - x[1]; v[3] = x[1] - x[3]; }
            else { v[0] = x[0]; v[1] = x[1]; }
            const int ng = sub ? 4 : 2;                        // groups of this bundle
            const int g0 = sub ? 4 * u : 4 * NS + 2 * (u - NS);
#pragma unroll
            for (int i = 0; i < ng; ++i) {
                const int g = g0 + i;                          // group of the octet; its plane: (s, i) -> i, a tap's halves -> 0 and 3
                const int plane = sub ? i : (i ? 3 : 0);
                const int slot = g % PF;                       // NGW is even: the slots line up across octets
                float4 a4[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a4[mt] = aq[slot][mt];
                {
                    const int gn = o * NGW + g + PF;                       // (past the last group of the last conv: an L1 hit, unused)
                    const float4* __restrict__ src = gn < NGT ? wl + (int64_t)gn * 2 * C
                                                              : (wnext ? wnext + (int64_t)(gn - NGT) * 2 * C : wl + (int64_t)(NGT - 1) * 2 * C);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) aq[slot][mt] = src[32 * mt];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pq = 0; pq < 4; ++pq)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float av = pq == 0 ? a4[mt].x : (pq == 1 ? a4[mt].y : (pq == 2 ? a4[mt].z : a4[mt].w));
                        acc[plane][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, v[i][pq], acc[plane][mt], 0, 0, 0);
                    }
            }
        }
        if (NBU & 1) {   // an odd number of bundles per octet: the prefetched first bundle of the next octet sits in slot 1 -> move it
#pragma unroll
            for (int m = 0; m < 4; ++m) bq[0][m] = bq[1][m];
        }
    }
#undef TTS_ENT
}

// WM = 0: both convs direct; 1: phase B (c2, dilation 1) as Winograd F(2,3); 2: phase A too -- then the window is staged ACTIVATED (one
// leaky-relu per value instead of one per use) and the residual comes from global memory in the row epilogue (an L2 hit).
#ifndef TTS_F2_WA32_WAVES
#define TTS_F2_WA32_WAVES 3     /* resident blocks per CU the C = 32 kernels with both phases on Winograd are compiled for: 170 registers, no spills (4: 128 registers, 40-56 bytes of scratch at k = 7 / 11; 61.80 vs 61.94 ms per step) */
#endif
// C = 128 with both phases on Winograd: EIGHT waves per block (two row halves x four column groups) -- four planes x 128 rows do not
// fit one wave's registers; one block per CU (the 136 KB window).
template <int C, int WM>
constexpr int f2_row_halves() { return (WM == 2 && C == 128) ? 2 : 1; }

template <int K, int C, int NTW, int WM>
__global__ __launch_bounds__((256 * f2_row_halves<C, WM>()), ((WM == 2 && C == 32) ? TTS_F2_WA32_WAVES : Fused2Geo<K, C, NTW>::WAVES))
void resblock_pair2(const FusedPair2Params p) {
    using G = Fused2Geo<K, C, NTW>;
    constexpr int NOCT = G::NOCT, MT = G::MT, H = G::H, NB = G::NB, TSTR = G::TSTR, PF = G::PF, TSH = G::TSH;
    constexpr bool WB = WM >= 1, WA = WM == 2;
    constexpr int RH = f2_row_halves<C, WM>(), MTW = MT / RH, NTH = 256 * RH;       // row halves, row tiles per wave, threads per block
    static_assert(!WB || NTW == 2, "Winograd phase B: a wave's 64 columns are its 32 output pairs");
    const int TS = WA ? G::ts_wa(p.dil) : G::TS;               // outputs per block
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int kk = lane >> 5, l31 = lane & 31;
    const int cg = wid & 3, mt0 = RH == 2 ? MTW * (wid >> 2) : 0;      // column group; first 32-row tile of this wave
    int b = blockIdx.z;
    int q0 = blockIdx.x * TS;
    if (p.compact) {   // ragged batch: dead blocks last (common.hpp: live_tile)
        int tile = 0;
        if (!live_tile(p.lens, p.len_mul, p.L, TS, p.batch, blockIdx.z * gridDim.x + blockIdx.x, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * TS;
    }
    int len = p.L;
    if (p.lens) len = min(len, (int)p.lens[b] * p.len_mul);
    if (q0 >= len) return;
#ifdef TTS_F2_EXP
    if (p.exp & 16) {      // desynchronise the co-resident blocks of the first round: the second block of every CU starts (exp >> 8) us late
        const unsigned lin0 = blockIdx.z * gridDim.x + blockIdx.x;
        if (lin0 >= 256 && lin0 < 512) {
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < (unsigned long long)(p.exp >> 8) * 100) __builtin_amdgcn_s_sleep(8);
        }
    }
#endif
    const int dil = p.dil, L = p.L;
    const int pad1 = H * dil;
    const int W1 = NB + (K - 1) * dil;                         // staged columns = row stride of the window
    const int x0 = q0 - H - pad1;                              // position of staged column 0
    const float slope = p.slope;
    const float* __restrict__ xb = p.x + (int64_t)b * C * L;
    float4* Xs = smem4;                                        // [o][kk][W1]: channels 8o + kk + {0,2,4,6} at position x0 + col, RAW

    // ---- weight queue: group g = (octet, tap) of conv 1 sits at w1 + g * 2C, this lane's fragment of row tile mt at
    // + kk * C + 32 mt + l31.  The first PF groups go out before anything else.
    const float4* __restrict__ wl1 = (WA ? p.w1w : p.w1) + kk * C + l31 + 32 * mt0;
    const float4* __restrict__ wl2 = (WB ? p.w2w : p.w2) + kk * C + l31 + 32 * mt0;
    float4 aq[PF][MTW];
#pragma unroll
    for (int g = 0; g < PF; ++g)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) aq[g][mt] = wl1[g * 2 * C + 32 * mt];

    // ---- the window: entry e = (o, kk_e, col), four scalar loads (consecutive lanes = consecutive positions), zero outside
    // the utterance, written raw.  ALL of a thread's loads go out before its first LDS write (nothing else is live yet: up to
    // 4 NE registers): one memory round trip for the whole window instead of one per batch of entries.
    {
        constexpr int NE = (2 * NOCT * (NB + (K - 1) * DMAX) + NTH - 1) / NTH;     // entries per thread at the widest dilation
        const int n_ent = 2 * NOCT * W1;
        float v[NE][4];
        int eo[NE];
        bool ok[NE];
        int okk = 0, col = tid;
        while (col >= W1) { col -= W1; ++okk; }
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int pos = x0 + col;
            const int okc = min(okk, 2 * NOCT - 1);
            ok[u] = pos >= 0 && pos < len;
            eo[u] = tid + NTH * u < n_ent ? okc * W1 + col : -1;
            const float* src = xb + (int64_t)((okc >> 1) * 8 + (okc & 1)) * L + min(max(pos, 0), max(len - 1, 0));
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) v[u][pc] = src[(int64_t)2 * pc * L];
            col += NTH;
            while (col >= W1) { col -= W1; ++okk; }
        }
#pragma unroll
        for (int u = 0; u < NE; ++u)
            if (eo[u] >= 0) {
                if (WA) {
#pragma unroll
                    for (int pc = 0; pc < 4; ++pc) v[u][pc] = lrelu_max(v[u][pc], slope);
                }
                Xs[eo[u]] = ok[u] ? make_float4(v[u][0], v[u][1], v[u][2], v[u][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(smem4);               // [C][NB]: the epilogue's row buffer (the intermediate is dead by then)
    if constexpr (WA) {
        // ---- BOTH phases as Winograd F(2,3).  Phase A: pair slot jw of the block -> intermediate columns (na, na + dil), na =
        // (jw / d) 2 d + jw % d; slots past d (128 / d) idle.  Planes start from b1 -> P0, -b1 -> P3.
        const int jw = cg * 32 + l31;
        const int npa = G::npa(dil);
        const int pa = min(jw, npa - 1);
        const int na = dil == 1 ? 2 * pa : (pa / dil) * 2 * dil + pa % dil;
        {
            f32x16 accA[4][MTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = p.b1[32 * (mt0 + mt) + (r & 3) + 8 * (r >> 2) + 4 * kk];
                    accA[0][mt][r] = bv; accA[1][mt][r] = 0.f; accA[2][mt][r] = 0.f; accA[3][mt][r] = -bv;
                }
            conv_phase2_wino<K, C, false, MTW>(accA, aq, Xs + kk * W1 + na, wl1, W1, dil, wl2);
            __syncthreads();                                   // every wave is done with the window
            // intermediate -> LDS: lrelu(y), zero outside the utterance, even / odd columns apart (phase B's layout)
            if (jw < npa) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int col = na + hh * dil;
                    const int pos = q0 - H + col;
                    const bool live = pos >= 0 && pos < len;
                    const int ent = (col & 1) * TSH + (col >> 1);
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                        for (int oc = 0; oc < 4; ++oc)
#pragma unroll
                            for (int k2 = 0; k2 < 2; ++k2) {
                                const int r = 4 * oc + k2;
                                float v0 = hh ? accA[1][mt][r] - accA[2][mt][r] - accA[3][mt][r] : accA[0][mt][r] + accA[1][mt][r] + accA[2][mt][r];
                                float v1 = hh ? accA[1][mt][r + 2] - accA[2][mt][r + 2] - accA[3][mt][r + 2]
                                              : accA[0][mt][r + 2] + accA[1][mt][r + 2] + accA[2][mt][r + 2];
                                v0 = lrelu_max(v0, slope);
                                v1 = lrelu_max(v1, slope);
                                const float2 w2v = live ? make_float2(v0, v1) : make_float2(0.f, 0.f);
                                *reinterpret_cast<float2*>(reinterpret_cast<float*>(Xs + ((4 * (mt0 + mt) + oc) * 2 + k2) * 2 * TSH + ent) + 2 * kk) = w2v;
                            }
                }
            }
            __syncthreads();
        }
        // phase B planes: zero, + the running ResBlock sum (y[2 jw] -> P0, -y[2 jw + 1] -> P3); the residual joins in the row epilogue
        f32x16 accw[4][MTW];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) accw[g][mt][r] = 0.f;
        if (p.mode != 0) {
            const int n = 2 * jw, q = q0 + n;
            const int voff = ((n < TS && q < len) ? q : 0) * 4 + 4 * kk * L * 4;
            const auto ys = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * C * L, 0, C * L * 4, 0x00020000);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int so = (32 * (mt0 + mt) + (r & 3) + 8 * (r >> 2)) * L * 4;
                    accw[0][mt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, voff, so, 0));
                    accw[3][mt][r] = -__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, voff + 4, so, 0));
                }
        }
        conv_phase2_wino<K, C, true, MTW>(accw, aq, Xs + kk * 2 * TSH + jw, wl2, 2 * TSH, 0, nullptr);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float2 y2;
                y2.x = accw[0][mt][r] + accw[1][mt][r] + accw[2][mt][r];
                y2.y = accw[1][mt][r] - accw[2][mt][r] - accw[3][mt][r];
                *reinterpret_cast<float2*>(ep + (32 * (mt0 + mt) + (r & 3) + 8 * (r >> 2) + 4 * kk) * NB + 2 * jw) = y2;
            }
    } else {
    const int colw = wid * 32 * NTW + l31;                     // this lane's MFMA column (j = 0), + 32 j
    // phase A accumulators start from b1 (row = channel 32mt + (r&3) + 8(r>>2) + 4kk)
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = p.b1[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kk];
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[mt][j][r] = bv;
        }

#ifdef TTS_F2_EXP
    if (!(p.exp & 2))
#endif
    conv_phase2<K, C, NTW, true>(acc, aq, Xs + kk * W1 + colw, W1, dil, wl1, wl2, slope);

    if constexpr (WB) {
        // ---- Winograd phase B: four planes per row tile for this lane's output pair (2 jw, 2 jw + 1); the residual (raw x out of
        // the window, exact) and the running ResBlock sum enter as x[2 jw] -> P0, -x[2 jw + 1] -> P3
        const int jw = wid * 32 + l31;
        f32x16 accw[4][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int colx = 2 * jw + H + pad1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { accw[1][mt][r] = 0.f; accw[2][mt][r] = 0.f; }
#pragma unroll
            for (int oc = 0; oc < 4; ++oc)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const float* src = reinterpret_cast<const float*>(Xs + ((4 * mt + oc) * 2 + k2) * W1 + colx) + 2 * kk;
                    const float2 r0 = *reinterpret_cast<const float2*>(src);
                    const float2 r1 = *reinterpret_cast<const float2*>(src + 4);
                    accw[0][mt][4 * oc + k2] = r0.x;
                    accw[0][mt][4 * oc + k2 + 2] = r0.y;
                    accw[3][mt][4 * oc + k2] = -r1.x;
                    accw[3][mt][4 * oc + k2 + 2] = -r1.y;
                }
        }
        if (p.mode != 0) {
            const int n = 2 * jw, q = q0 + n;
            const int voff = ((n < TS && q < len) ? q : 0) * 4 + 4 * kk * L * 4;        // (q even, rows float4-aligned: q + 1 stays in the row)
            const auto ys = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * C * L, 0, C * L * 4, 0x00020000);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 t0, t1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int so = (32 * mt + (r & 3) + 8 * (r >> 2)) * L * 4;
                    t0[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, voff, so, 0));
                    t1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ys, voff + 4, so, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    accw[0][mt][r] = t0[r] + accw[0][mt][r];
                    accw[3][mt][r] = accw[3][mt][r] - t1[r];
                }
            }
        }
        __syncthreads();                                       // every wave is done with the window

        // ---- intermediate -> LDS: lrelu(acc) (b1 is in), zero outside the utterance, even / odd columns apart: column c of row
        // (octet, k2) is entry (c & 1) TSH + (c >> 1) of the row's 2 TSH
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int col = wid * 32 * NTW + l31 + 32 * j;
            const int pos = q0 - H + col;
            const bool live = pos >= 0 && pos < len;
            const int ent = (col & 1) * TSH + (col >> 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int oc = 0; oc < 4; ++oc)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int r = 4 * oc + k2;
                        float v0 = acc[mt][j][r], v1 = acc[mt][j][r + 2];
                        v0 = lrelu_max(v0, slope);
                        v1 = lrelu_max(v1, slope);
                        const float2 w2v = live ? make_float2(v0, v1) : make_float2(0.f, 0.f);
                        *reinterpret_cast<float2*>(reinterpret_cast<float*>(Xs + ((4 * mt + oc) * 2 + k2) * 2 * TSH + ent) + 2 * kk) = w2v;
                    }
        }
        __syncthreads();

        conv_phase2_wino<K, C, true, MT>(accw, aq, Xs + kk * 2 * TSH + jw, wl2, 2 * TSH, 0, nullptr);

        // ---- output transform into the row buffer: y[2 jw] = P0 + P1 + P2, y[2 jw + 1] = P1 - P2 - P3
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float2 y2;
                y2.x = accw[0][mt][r] + accw[1][mt][r] + accw[2][mt][r];
                y2.y = accw[1][mt][r] - accw[2][mt][r] - accw[3][mt][r];
                *reinterpret_cast<float2*>(ep + (32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kk) * NB + 2 * jw) = y2;
            }
    } else {
    // ---- residual out of the window (raw x, exact): register r of tile (mt, j) is channel 32mt + (r&3) + 8(r>>2) + 4kk at
    // position q0 + colw + 32j = window column colw + 32j + H + pad1; registers (r, r+2), r&3 in {0,1}, are components
    // (2kk, 2kk+1) of entry (4mt + (r>>2), r&1, col): one 8-byte read for the two.
    f32x16 acc2[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int colx = colw + 32 * j + H + pad1;
#pragma unroll
            for (int oc = 0; oc < 4; ++oc)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const float2 rv = *reinterpret_cast<const float2*>(
                        reinterpret_cast<const float*>(Xs + ((4 * mt + oc) * 2 + k2) * W1 + colx) + 2 * kk);
                    acc2[mt][j][4 * oc + k2] = rv.x;
                    acc2[mt][j][4 * oc + k2 + 2] = rv.y;
                }
        }
    if (p.mode != 0) {      // + the running ResBlock sum (buffer loads: one per-lane offset per column tile, scalar row offsets)
        int voff[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = colw + 32 * j;
            const int q = q0 + n;
            voff[j] = ((n < TS && q < len) ? q : 0) * 4 + 4 * kk * L * 4;
        }
        const auto ys = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * C * L, 0, C * L * 4, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    t[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        ys, voff[j], (32 * mt + (r & 3) + 8 * (r >> 2)) * L * 4, 0));
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][j][r] = t[r] + acc2[mt][j][r];
            }
    }
    __syncthreads();                                           // every wave is done with the window

    // ---- intermediate -> LDS: lrelu(acc) (b1 is in), zero outside the utterance (c2 pads at the true edge), B-operand layout
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int col = colw + 32 * j;
        const int pos = q0 - H + col;
        const bool live = pos >= 0 && pos < len;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int oc = 0; oc < 4; ++oc)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int r = 4 * oc + k2;
                    float v0 = acc[mt][j][r], v1 = acc[mt][j][r + 2];
                    v0 = lrelu_max(v0, slope);
                    v1 = lrelu_max(v1, slope);
                    const float2 w2v = live ? make_float2(v0, v1) : make_float2(0.f, 0.f);
                    *reinterpret_cast<float2*>(reinterpret_cast<float*>(Xs + ((4 * mt + oc) * 2 + k2) * TSTR + col) + 2 * kk) = w2v;
                }
    }
    __syncthreads();

#ifdef TTS_F2_EXP
    if (!(p.exp & 4))
#endif
    conv_phase2<K, C, NTW, false>(acc2, aq, Xs + kk * TSTR + colw, TSTR, 1, wl2, wl2, slope);

    // ---- epilogue: + b2 [, / div], transposed through LDS (the intermediate is dead after the barrier), float4 row stores
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kk) * NB + colw + 32 * j] = acc2[mt][j][r];
    }
    }
    __syncthreads();
    const bool do_div = p.mode == 2;
    const float div = p.div;
    float* __restrict__ yb = p.y + (int64_t)b * C * L;
    constexpr int LPR = NB / 4;                                 // lanes per row (64 or 32)
    constexpr int RPW = 64 / LPR;                               // rows per wave instruction
    const int n = (lane % LPR) * 4;
    const int rsub = lane / LPR;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
#pragma unroll 4
    for (int it = 0; it < C / (4 * RH * RPW); ++it) {
        const int ch = (wid_u + 4 * RH * it) * RPW + rsub;
        const float bs = p.b2[ch];
        const int q = q0 + n;
        if (n >= TS || q >= len) continue;
        const float4 v4 = *reinterpret_cast<const float4*>(ep + ch * NB + n);
        float vv[4] = {v4.x + bs, v4.y + bs, v4.z + bs, v4.w + bs};
        if (WA) {      // the residual: raw x from memory (the window was staged activated); q is a multiple of 4, rows are float4-aligned
            const float* xp = xb + (int64_t)ch * L + q;
            if (q + 3 < len) {
                const float4 x4 = *reinterpret_cast<const float4*>(xp);
                vv[0] += x4.x; vv[1] += x4.y; vv[2] += x4.z; vv[3] += x4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q + e < len) vv[e] += xp[e];
            }
        }
        if (do_div) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = vv[e] / div;
        }
        float* yp = yb + (int64_t)ch * L + q;
#ifdef TTS_F2_EXP
        if ((p.exp & 8) && vv[0] != 12345.678f) continue;
#endif
        if (q + 3 < len) {
            *reinterpret_cast<float4*>(yp) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q + e < len) yp[e] = vv[e];
        }
    }
}


template <int K, int C, int NTW, int WM>
static int32_t launch_fused2_k(const FusedPair2Params& p, hipStream_t stream) {
    using G = Fused2Geo<K, C, NTW>;
    static std::atomic<uint64_t> lds_done{0};          // per instantiation: devices already opted in (common.hpp: lds_opt_in)
    TTS_CHECK_HIP(lds_opt_in((const void*)resblock_pair2<K, C, NTW, WM>, (int)G::lds_bytes(DMAX), lds_done));
    const int ts = WM == 2 ? G::ts_wa(p.dil) : G::TS;
    dim3 grid((p.L + ts - 1) / ts, 1, p.batch);
    hipLaunchKernelGGL((resblock_pair2<K, C, NTW, WM>), grid, dim3(256 * f2_row_halves<C, WM>()), G::lds_bytes(p.dil), stream, p);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// geometry the second-generation kernel is built for; ntw = 2: 256-column blocks, ntw = 1: 128-column blocks
bool fused_pair2_supported(int32_t channels, int32_t k, int32_t dil, int32_t L, const float* x, const float* y, int32_t ntw) {
    if (!(channels == 32 || channels == 64 || channels == 128) || !(k == 3 || k == 7 || k == 11) || !(ntw == 1 || ntw == 2)) return false;
    if (dil < 1 || dil > DMAX || (L & 3) != 0 || ((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0 || x == y) return false;
    if ((int64_t)channels * L * 4 >= ((int64_t)1 << 31)) return false;
    // the window must fit the 160 KB of one CU
    const size_t lds = (size_t)2 * (channels / 8) * (128 * ntw + (k - 1) * dil) * 16;
    return lds <= 160 * 1024;
}

int32_t launch_fused_pair2(int32_t channels, const float* x, float* y, const float* w1, const float* b1, const float* w2,
                           const float* b2, int32_t k, int32_t dil, const int64_t* lens, int32_t len_mul, int32_t L, int32_t batch,
                           int32_t mode, float div, float slope, int32_t ntw, hipStream_t stream, const float* w2_wino,
                           const float* w1_wino) {
    TTS_REQUIRE(fused_pair2_supported(channels, k, dil, L, x, y, ntw),
                "fused ResBlock pair (direct weights): unsupported geometry (C=%d, k=%d, dil=%d, L=%d, ntw=%d)", channels, k, dil, L, ntw);
    TTS_REQUIRE(slope > 0.f && slope <= 1.f, "fused ResBlock pair: leaky-relu slope %g outside (0, 1]", (double)slope);
    // w2_wino (conv 2 as Winograd groups, pack_wino2_weight): phase B on F(2,3) -- 256-column blocks of C = 32 / 64 only
    // ... + w1_wino: phase A too (dilations 1 / 3 / 5: 256 / 252 / 250 intermediate columns per block)
    const bool wa = w2_wino != nullptr && w1_wino != nullptr && ntw == 2 && (dil == 1 || dil == 3 || dil == 5);   // C = 128: 8 waves
    const bool wb = w2_wino != nullptr && ntw == 2 && (channels <= 64 || wa);
    conv_log(wa ? "fused_pair2ww" : wb ? "fused_pair2w" : (ntw == 2 ? "fused_pair2" : "fused_pair2n"), k, channels, channels, L, batch, 1, mode, len_mul, lens != nullptr, 1);
    FusedPair2Params p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.y = y;
    p.w1 = reinterpret_cast<const float4*>(w1); p.w2 = reinterpret_cast<const float4*>(w2);
    p.w2w = reinterpret_cast<const float4*>(w2_wino);
    p.w1w = reinterpret_cast<const float4*>(w1_wino);
    p.b1 = b1; p.b2 = b2; p.lens = lens; p.len_mul = len_mul; p.L = L; p.dil = dil; p.batch = batch;
    p.mode = mode; p.div = div; p.slope = slope;
    p.compact = compact_order(lens, batch) ? 1 : 0;
#ifdef TTS_F2_EXP
    if (const char* e = exp_env("TTSAMD_F2_EXP")) p.exp = atoi(e);
#endif
#define TTS_F2W(KK, CC) if (wa && k == KK && channels == CC) return launch_fused2_k<KK, CC, 2, 2>(p, stream); \
                        if (wb && k == KK && channels == CC) return launch_fused2_k<KK, CC, 2, 1>(p, stream);
    TTS_F2W(3, 32) TTS_F2W(7, 32) TTS_F2W(11, 32) TTS_F2W(3, 64) TTS_F2W(7, 64) TTS_F2W(11, 64)
#undef TTS_F2W
    if (wa && channels == 128) {
        if (k == 3) return launch_fused2_k<3, 128, 2, 2>(p, stream);
        if (k == 7) return launch_fused2_k<7, 128, 2, 2>(p, stream);
        if (k == 11) return launch_fused2_k<11, 128, 2, 2>(p, stream);
    }
#define TTS_F2(KK, CC, NN) if (k == KK && channels == CC && ntw == NN) return launch_fused2_k<KK, CC, NN, 0>(p, stream);
    TTS_F2(3, 32, 2) TTS_F2(7, 32, 2) TTS_F2(11, 32, 2)
    TTS_F2(3, 64, 2) TTS_F2(7, 64, 2) TTS_F2(11, 64, 2)
    TTS_F2(3, 128, 2) TTS_F2(7, 128, 2) TTS_F2(11, 128, 2)
    TTS_F2(3, 32, 1) TTS_F2(7, 32, 1) TTS_F2(11, 32, 1)
    TTS_F2(3, 64, 1) TTS_F2(7, 64, 1) TTS_F2(11, 64, 1)
    TTS_F2(3, 128, 1) TTS_F2(7, 128, 1) TTS_F2(11, 128, 1)
#undef TTS_F2
    set_error("fused ResBlock pair (direct weights): no instantiation for C=%d, k=%d, ntw=%d", channels, k, ntw);
    return TTSAMD_EINVAL;
}

}  // namespace ttsamd
