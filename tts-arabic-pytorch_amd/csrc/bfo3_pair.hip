// One launch for a whole c1 -> c2 pair of a HiFi-GAN ResBlock1 in the split-bf16 mode of the octet engine (bfo3.hpp):
//     y = x + conv1d(lrelu(conv1d(lrelu(x), w1, dil d) + b1), w2, dil 1) + b2          (vocoder/hifigan/models.py:46-53)
// for C = 32 / 64 / 128 at k = 3 / 7 / 11: fp32-class results (x = hi + lo, three bf16 MFMAs per product) with the
// anatomy of bfo_pair.hip -- the whole input window in LDS, weights from L2 into a register ring, the c1 -> c2
// intermediate never leaves the CU -- and two things the plain bf16 pair does not have:
//   * the window is TWO planes (hi, lo) of octet entries, so LDS holds 2 x C x (NCOLS + (K - 1) d) x 2 bytes; the three
//     products re-read the hi plane instead of staging it twice (two ds_read_b128 per three MFMAs);
//   * HBM accesses are 16 bytes per lane (the 32-byte x3 entry = the C-layout halves of lanes kk = 0 / 1 side by side).
// Block = NW waves, each owning a 32-row x (NT x 32)-column slab; WM = C / 32 row slabs x WN = NW / WM column slabs.
// C <= 64: NW = 4, two blocks per CU (<= 80 KB of LDS each): one block's staging / exchange / stores run under the other's
// MFMAs.  C = 128: NW = 8 (4 x 2 waves, 256 columns), one block per CU (136-154 KB; DM = the largest dilation served).
//   stage    2 NE 16-byte pieces (piece u = half kk = u & 1 of entry u >> 1), fully coalesced; halves -> the two planes;
//   phase A  T = conv(window, w1) on NCOLS = TS + K - 1 positions, accumulators start from b1 (bfo3_mma);
//   T -> LDS lrelu(T), zero outside the utterance, split into hi / lo, over the dead window;
//   phase B  Y = conv(T, w2) on the TS outputs; accumulators start from b2 + the residual read back from the LDS window;
//   epilogue [+ running ResBlock sum] [/ n_kernels], leaky-relu of the CONSUMER, split, 16-byte stores.
#include <cstdlib>
#include <cstring>

#include "bfo3.hpp"

namespace ttsamd {

template <int K, int C, int NT_, int NW_, int DM_>
struct Bfo3PairGeo {
    static constexpr int NO = C / 8, NH = C / 16;
    static constexpr int NW = NW_, NTHR = 64 * NW_;
    static constexpr int WM = C / 32, WN = NW_ / WM;        // waves over rows / over columns
    static constexpr int NT = NT_;
    static constexpr int NCOLS = WN * NT * 32;              // columns of phase A
    static constexpr int H = (K - 1) / 2;
    static constexpr int TS = NCOLS - (K - 1);              // outputs per block
    static constexpr int WS = NCOLS + (K - 1) * DM_;        // LDS entries per octet row
    static constexpr int NE = NO * WS;                      // entries of ONE plane
    static constexpr int NXI = (2 * NE + NTHR - 1) / NTHR;  // 16-byte pieces per thread
    static constexpr int PH = K <= 3 ? 2 : 1;               // 16-channel groups the A ring runs ahead
    static constexpr size_t LDS = (size_t)NE * 32;
    static constexpr int BPC = LDS <= 80 * 1024 ? 2 : 1;    // blocks per CU (256-column blocks, three per CU, measured slower at C = 32: 350 vs 338 us at k = 7)
    static_assert(NW_ % WM == 0 && WN >= 1, "waves must tile the rows");
    static_assert(LDS <= 160 * 1024, "window does not fit a CU");
};

template <int K, int C, int NT_, int NW_, int DM_>
__global__ __launch_bounds__(64 * NW_, (Bfo3PairGeo<K, C, NT_, NW_, DM_>::BPC * NW_) / 4)
void bfo3_resblock_pair(const BfoPairParams p) {
    using G = Bfo3PairGeo<K, C, NT_, NW_, DM_>;
    constexpr int NO = G::NO, NH = G::NH, WN = G::WN, NT = G::NT, H = G::H, TS = G::TS, WS = G::WS, NXI = G::NXI, NE = G::NE;
    extern __shared__ __attribute__((aligned(16))) uint4 Xs[];          // [hi plane NE][lo plane NE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5, l31 = lane & 31;
    const int wm = wid / WN, wn = wid % WN;
    int b = blockIdx.z;
    int q0 = blockIdx.x * TS;
    if (p.compact) {   // ragged batch: dead blocks last (common.hpp: live_tile)
        int tile = 0;
        if (!live_tile(p.lens, p.len_mul, p.L, TS, p.batch, blockIdx.z * gridDim.x + blockIdx.x, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * TS;
    }
    const int L = p.L;
    int len = L;
    if (p.lens) len = min(len, (int)p.lens[b] * p.len_mul);
    if (q0 >= len) return;
    const int dil = p.dil;
    const int W1 = G::NCOLS + (K - 1) * dil;                // staged columns actually used
    const int x0 = q0 - H - (K - 1) * dil / 2;              // position of staged column 0
    const bfo_i4 xrs = bfo_rsrc((const char*)p.x + (int64_t)b * NO * L * 32, (unsigned)NO * L * 32);

    // ---- stage the window: all loads first, then the LDS writes (8 bytes to each plane)
    {
        bfo_i4 xv[NXI];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int u = tid + G::NTHR * i;
            const int e = u >> 1, hk = u & 1;
            const int o = e / WS, col = e - o * WS;
            const int pos = x0 + col;
            const bool ok = e < NE && col < W1 && pos >= 0 && pos < len;
            xv[i] = bfo_ld16(xrs, ok ? ((o * L + pos) * 2 + hk) * 16 : BFO_OOB, 0, 0);
        }
        char* const xb = reinterpret_cast<char*>(Xs);
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int u = tid + G::NTHR * i;
            if (u < 2 * NE) {
                bfo_i2 h2, l2;
                h2.x = xv[i].x; h2.y = xv[i].y; l2.x = xv[i].z; l2.y = xv[i].w;
                *reinterpret_cast<bfo_i2*>(xb + u * 8) = h2;
                *reinterpret_cast<bfo_i2*>(xb + NE * 16 + u * 8) = l2;
            }
        }
    }
    __syncthreads();

    const int wv = (kk * C + 32 * wm + l31) * 32;           // this lane's A pair inside a (h, tap) step
    const int cw = wn * (NT * 32) + l31;                    // this lane's column in tile 0
    const uint4* sB = Xs + kk * WS + cw;

    // ---- phase A
    bfo_f16 acc[NT];
    {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = p.b1[32 * wm + 8 * (r >> 2) + 4 * kk + (r & 3)];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = bv[r];
    }
    bfo3_mma<K, G::PH, NT>(acc, bfo_rsrc(p.w1, (unsigned)NH * K * 2 * C * 32), wv, 2 * C * 32, sB, NE, NH, 2 * WS, dil);

    // residual (= the activated input at the output positions) read back from the LDS window before the intermediate
    // overwrites it: the hi and lo halves of (octet 4 wm + g, column n + h + pad), 8 bytes each
    int vo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = cw + 32 * j, q = q0 + n;
        vo[j] = (n < TS && q < len) ? (q * 2 + kk) * 16 : BFO_OOB;
    }
    bfo_i2 rh[NT][4], rl[NT][4];
    {
        const int rc0 = cw + H + (K - 1) * dil / 2;
        const char* const xb = reinterpret_cast<const char*>(Xs);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int eo = (((4 * wm + g) * WS + rc0 + 32 * j) * 2 + kk) * 8;
                rh[j][g] = *reinterpret_cast<const bfo_i2*>(xb + eo);
                rl[j][g] = *reinterpret_cast<const bfo_i2*>(xb + NE * 16 + eo);
            }
    }
    __syncthreads();                                        // every wave is done with the window
    {
        const float ms = p.mid_slope;
        char* const xb = reinterpret_cast<char*>(Xs);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = cw + 32 * j;
            const int pos = q0 - H + col;
            const int live = (pos >= 0 && pos < len) ? -1 : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bfo_i4 w = bfo3_act4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], ms, live);
                const int eo = (((4 * wm + g) * WS + col) * 2 + kk) * 8;
                bfo_i2 h2, l2;
                h2.x = w.x; h2.y = w.y; l2.x = w.z; l2.y = w.w;
                *reinterpret_cast<bfo_i2*>(xb + eo) = h2;
                *reinterpret_cast<bfo_i2*>(xb + NE * 16 + eo) = l2;
            }
        }
    }
    __syncthreads();

    // ---- phase B: accumulators start from b2 + x (x = a >= 0 ? a : a / in_slope)
    {
        const float inv = 1.f / p.in_slope;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = p.b2[32 * wm + 8 * (r >> 2) + 4 * kk + (r & 3)];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a[4];
                bfo3_join4(rh[j][g].x, rh[j][g].y, rl[j][g].x, rl[j][g].y, a);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = bv[4 * g + e] + bfo_unrelu(a[e], inv);
            }
    }
    bfo3_mma<K, G::PH, NT>(acc, bfo_rsrc(p.w2, (unsigned)NH * K * 2 * C * 32), wv, 2 * C * 32, sB, NE, NH, 2 * WS, 1);

    // ---- epilogue
    const bfo_i4 yrs = bfo_rsrc((char*)p.y + (int64_t)b * NO * L * 32, (unsigned)NO * L * 32);
    const float os = p.out_slope;
    if (p.mode != 0) {
        const bfo_i4 srs = bfo_rsrc((const char*)p.sum_in + (int64_t)b * NO * L * 32, (unsigned)NO * L * 32);
        const float sc = p.mode == 2 ? 1.f / p.div : 1.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bfo_i4 sv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) sv[g] = bfo_ld16(srs, vo[j], (4 * wm + g) * L * 32, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s[4];
                bfo3_join4(sv[g].x, sv[g].y, sv[g].z, sv[g].w, s);
                bfo3_st16(bfo3_act4((acc[j][4 * g] + s[0]) * sc, (acc[j][4 * g + 1] + s[1]) * sc, (acc[j][4 * g + 2] + s[2]) * sc,
                                   (acc[j][4 * g + 3] + s[3]) * sc, os, -1),
                         yrs, vo[j], (4 * wm + g) * L * 32);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bfo3_st16(bfo3_act4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], os, -1), yrs, vo[j],
                         (4 * wm + g) * L * 32);
    }
}

template <int K, int C, int NT, int NW, int DM>
static int32_t bfo3_launch_pair_cfg(const BfoPairParams& p, hipStream_t stream) {
    using G = Bfo3PairGeo<K, C, NT, NW, DM>;
    static std::atomic<uint64_t> lds_done{0};
    TTS_CHECK_HIP(lds_opt_in((const void*)bfo3_resblock_pair<K, C, NT, NW, DM>, (int)G::LDS, lds_done));
    dim3 grid((p.L + G::TS - 1) / G::TS, 1, p.batch);
    BfoPairParams q = p;
    q.compact = compact_order(p.lens, p.batch) ? 1 : 0;
    hipLaunchKernelGGL((bfo3_resblock_pair<K, C, NT, NW, DM>), grid, dim3(G::NTHR), G::LDS, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

bool bfo3_pair_supported(int32_t channels, int32_t k, int32_t dil, int32_t L) {
    return (channels == 32 || channels == 64 || channels == 128) && (k == 3 || k == 7 || k == 11) && dil >= 1 && dil <= BFO_DMAX &&
           (int64_t)channels * L * 4 < ((int64_t)1 << 31);
}

int32_t bfo3_launch_pair(int32_t channels, int32_t k, const BfoPairParams& p, hipStream_t stream) {
    TTS_REQUIRE(bfo3_pair_supported(channels, k, p.dil, p.L), "split-bf16 ResBlock pair: unsupported geometry (C=%d, k=%d, dil=%d, L=%d)",
                channels, k, p.dil, p.L);
    TTS_REQUIRE(p.x != p.y, "split-bf16 ResBlock pair: x and y must differ (halo reads)");
    TTS_REQUIRE(p.mode == 0 || p.sum_in != nullptr, "split-bf16 ResBlock pair: mode %d needs sum_in", p.mode);
    conv_log("bfo3_pair", k, channels, channels, p.L, p.batch, 1, p.mode, p.len_mul, p.lens != nullptr, 1);
    // C <= 64: two blocks of 4 waves x (32 rows x 128 columns) per CU.  C = 128: ONE block of 8 waves per CU (WM = 4 row slabs x
    // WN = 2 column slabs, 256 columns: half the halo of a 128-column block, 136-154 KB of LDS)
#define BFO3_CASE(KK, CC) if (k == KK && channels == CC) return bfo3_launch_pair_cfg<KK, CC, 4, 4, BFO_DMAX>(p, stream);
    BFO3_CASE(3, 32) BFO3_CASE(7, 32) BFO3_CASE(11, 32)
    BFO3_CASE(3, 64) BFO3_CASE(7, 64) BFO3_CASE(11, 64)
#undef BFO3_CASE
    // (measured on the bench step, tools/ab_x3_w8.sh: 26.05-26.08 ms against 26.18 with two 4-wave blocks per CU where they fit)
#define BFO3_CASE8(KK) if (k == KK && channels == 128) return bfo3_launch_pair_cfg<KK, 128, 4, 8, BFO_DMAX>(p, stream);
    BFO3_CASE8(3) BFO3_CASE8(7) BFO3_CASE8(11)
#undef BFO3_CASE8
    return TTSAMD_EINVAL;
}

}  // namespace ttsamd
