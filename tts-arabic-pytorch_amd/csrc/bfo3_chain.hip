// One launch for a WHOLE k = 3 ResBlock1 of HiFi-GAN in the split-bf16 mode of the octet engine (bfo3.hpp): the three
// c1 -> c2 pairs
//     x <- x + conv1d(lrelu(conv1d(lrelu(x), w1_m, dil d_m) + b1_m), w2_m, dil 1) + b2_m,   m = 0, 1, 2   (d = 1, 3, 5)
// (vocoder/hifigan/models.py:46-53) chained through LDS.  At k = 3 a split-bf16 pair moves 8 bytes per element for 36 C
// MFMA-FLOP: the C = 32 / 64 launches run at 3-4.2 TB/s with the matrix pipe 30-48 % busy (profiles/r5): three pairs per
// launch stage the window once and store once.
//
// The anatomy of bfo3_pair.hip plus what bfo_chain.hip adds to the bf16 pair: the OUTPUT of pair m (activated for its consumer
// and split into hi + lo exactly where the un-fused launch splits it for HBM, so the results are bit-identical to three pair
// launches) is written over the window as the input of pair m + 1.  Every phase computes all NCOLS columns of the tile; the
// columns that depend on data outside the staged window are garbage and never reach a stored output: the stored outputs are the
// TS = NCOLS - 2 * sum (d_m + 1) = NCOLS - 24 positions [q0, q0 + TS).
#include <cstdlib>
#include <cstring>

#include "bfo3.hpp"

namespace ttsamd {

template <int C, int NT_, int NW_>
struct Bfo3ChainGeo {
    static constexpr int K = 3, NP = 3, H = 1;
    static constexpr int NO = C / 8, NH = C / 16;
    static constexpr int NW = NW_, NTHR = 64 * NW_;
    static constexpr int WM = C / 32, WN = NW_ / WM;
    static constexpr int NT = NT_;
    static constexpr int NCOLS = WN * NT * 32;
    static constexpr int WS = NCOLS + (K - 1) * BFO_DMAX;   // LDS entries per octet row (as the pair kernel)
    static constexpr int NE = NO * WS;                      // entries of ONE plane
    static constexpr int NXI = (2 * NE + NTHR - 1) / NTHR;
    static constexpr int PH = 2;                            // as the k = 3 pair kernel: same accumulation order, same bits
    static constexpr size_t LDS = (size_t)NE * 32;
    static constexpr int BPC = LDS <= 80 * 1024 ? 2 : 1;
    static_assert(NW_ % WM == 0 && WN >= 1 && LDS <= 160 * 1024, "geometry");
};

template <int C, int NT_, int NW_>
__global__ __launch_bounds__(64 * NW_, (Bfo3ChainGeo<C, NT_, NW_>::BPC * NW_) / 4)
void bfo3_resblock_chain(const BfoChainParams p) {
    using G = Bfo3ChainGeo<C, NT_, NW_>;
    constexpr int K = G::K, H = G::H, NO = G::NO, NH = G::NH, WN = G::WN, NT = G::NT, WS = G::WS, NXI = G::NXI, NE = G::NE;
    extern __shared__ __attribute__((aligned(16))) uint4 Xs[];          // [hi plane NE][lo plane NE]
    char* const xb = reinterpret_cast<char*>(Xs);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5, l31 = lane & 31;
    const int wm = wid / WN, wn = wid % WN;
    const int HT = H * (p.dil[0] + p.dil[1] + p.dil[2] + 3);   // columns lost on each side over the three pairs
    const int TS = G::NCOLS - 2 * HT;
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
    const bfo_i4 xrs = bfo_rsrc((const char*)p.x + (int64_t)b * NO * L * 32, (unsigned)NO * L * 32);

    // ---- stage the window of pair 0: column c = position xw + c
    int xw = q0 - HT;                                        // position of LDS column 0 for the current pair's INPUT
    {
        const int W1 = G::NCOLS + (K - 1) * p.dil[0];
        bfo_i4 xv[NXI];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int u = tid + G::NTHR * i;
            const int e = u >> 1, hk = u & 1;
            const int o = e / WS, col = e - o * WS;
            const int pos = xw + col;
            const bool ok = e < NE && col < W1 && pos >= 0 && pos < len;
            xv[i] = bfo_ld16(xrs, ok ? ((o * L + pos) * 2 + hk) * 16 : BFO_OOB, 0, 0);
        }
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

    const int wv = (kk * C + 32 * wm + l31) * 32;
    const int cw = wn * (NT * 32) + l31;
    const uint4* sB = Xs + kk * WS + cw;
    const unsigned wbytes = (unsigned)NH * K * 2 * C * 32;
    const float inv_in = 1.f / p.in_slope;
    // acc -> the two planes at column `col` (activated with `slope`, zero where pos is outside the utterance)
    const auto to_lds = [&](const bfo_f16 (&a)[NT], const float slope, const int pos0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = cw + 32 * j;
            const int pos = pos0 + col;
            const int live = (pos >= 0 && pos < len) ? -1 : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bfo_i4 w = bfo3_act4(a[j][4 * g], a[j][4 * g + 1], a[j][4 * g + 2], a[j][4 * g + 3], slope, live);
                const int eo = (((4 * wm + g) * WS + col) * 2 + kk) * 8;
                bfo_i2 h2, l2;
                h2.x = w.x; h2.y = w.y; l2.x = w.z; l2.y = w.w;
                *reinterpret_cast<bfo_i2*>(xb + eo) = h2;
                *reinterpret_cast<bfo_i2*>(xb + NE * 16 + eo) = l2;
            }
        }
    };

    bfo_f16 acc[NT];
#pragma unroll 1
    for (int m = 0; m < G::NP; ++m) {
        const int dil = p.dil[m];
        // ---- phase A: T column c = position xw + H dil + c
        {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = p.b1[m][32 * wm + 8 * (r >> 2) + 4 * kk + (r & 3)];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = bv[r];
        }
        bfo3_mma<K, G::PH, NT>(acc, bfo_rsrc(p.w1[m], wbytes), wv, 2 * C * 32, sB, NE, NH, 2 * WS, dil);

        // residual = the pair's (activated) input at its output positions: output column n reads window column n + H (dil + 1)
        bfo_i2 rh[NT][4], rl[NT][4];
        {
            const int rc0 = cw + H * (dil + 1);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int eo = (((4 * wm + g) * WS + rc0 + 32 * j) * 2 + kk) * 8;
                    rh[j][g] = *reinterpret_cast<const bfo_i2*>(xb + eo);
                    rl[j][g] = *reinterpret_cast<const bfo_i2*>(xb + NE * 16 + eo);
                }
        }
        __syncthreads();                                    // every wave is done with the window
        to_lds(acc, p.mid_slope, xw + H * dil);
        __syncthreads();

        // ---- phase B: output column n = position xw + H (dil + 1) + n; accumulators start from b2 + x
        {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = p.b2[m][32 * wm + 8 * (r >> 2) + 4 * kk + (r & 3)];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a[4];
                    bfo3_join4(rh[j][g].x, rh[j][g].y, rl[j][g].x, rl[j][g].y, a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = bv[4 * g + e] + bfo_unrelu(a[e], inv_in);
                }
        }
        bfo3_mma<K, G::PH, NT>(acc, bfo_rsrc(p.w2[m], wbytes), wv, 2 * C * 32, sB, NE, NH, 2 * WS, 1);
        xw += H * (dil + 1);
        if (m + 1 == G::NP) break;

        // ---- the pair's output, activated for the next pair (in_slope) and split as the un-fused launch splits it for HBM,
        // becomes the next window: column n = position xw + n, zero outside the utterance (the next conv pads at the TRUE edge)
        __syncthreads();                                    // every wave is done with the intermediate
        to_lds(acc, p.in_slope, xw);
        __syncthreads();
    }

    // ---- epilogue of the last pair (xw == q0 now)
    int vo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = cw + 32 * j, q = q0 + n;
        vo[j] = (n < TS && q < len) ? (q * 2 + kk) * 16 : BFO_OOB;
    }
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

template <int C, int NT, int NW>
static int32_t bfo3_launch_chain_cfg(const BfoChainParams& p, hipStream_t stream) {
    using G = Bfo3ChainGeo<C, NT, NW>;
    static std::atomic<uint64_t> lds_done{0};
    TTS_CHECK_HIP(lds_opt_in((const void*)bfo3_resblock_chain<C, NT, NW>, (int)G::LDS, lds_done));
    const int TS = G::NCOLS - 2 * G::H * (p.dil[0] + p.dil[1] + p.dil[2] + 3);
    dim3 grid((p.L + TS - 1) / TS, 1, p.batch);
    BfoChainParams q = p;
    q.compact = compact_order(p.lens, p.batch) ? 1 : 0;
    hipLaunchKernelGGL((bfo3_resblock_chain<C, NT, NW>), grid, dim3(G::NTHR), G::LDS, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

bool bfo3_chain_supported(int32_t channels, int32_t k, const int32_t* dil, int32_t n_pairs, int32_t L) {
    if (n_pairs != 3 || k != 3 || !(channels == 32 || channels == 64 || channels == 128)) return false;
    for (int m = 0; m < 3; ++m)
        if (dil[m] < 1 || dil[m] > BFO_DMAX) return false;
    return (int64_t)channels * L * 4 < ((int64_t)1 << 31);
}

int32_t bfo3_launch_chain(int32_t channels, const BfoChainParams& p, hipStream_t stream) {
    TTS_REQUIRE(bfo3_chain_supported(channels, p.k, p.dil, 3, p.L), "split-bf16 ResBlock chain: unsupported geometry (C=%d, k=%d, L=%d)",
                channels, p.k, p.L);
    TTS_REQUIRE(p.x != p.y, "split-bf16 ResBlock chain: x and y must differ (halo reads)");
    TTS_REQUIRE(p.mode == 0 || p.sum_in != nullptr, "split-bf16 ResBlock chain: mode %d needs sum_in", p.mode);
    conv_log("bfo3_chain", p.k, channels, channels, p.L, p.batch, 1, p.mode, p.len_mul, p.lens != nullptr, 3);
    if (channels == 128) return bfo3_launch_chain_cfg<128, 4, 8>(p, stream);
    if (channels == 64) return bfo3_launch_chain_cfg<64, 4, 4>(p, stream);
    return bfo3_launch_chain_cfg<32, 4, 4>(p, stream);
}

}  // namespace ttsamd
