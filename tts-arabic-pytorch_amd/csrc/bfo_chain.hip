// One launch for a WHOLE k = 3 ResBlock1 of HiFi-GAN in the bf16 octet engine (bfo.hpp): the three c1 -> c2 pairs
//     x <- x + conv1d(lrelu(conv1d(lrelu(x), w1_m, dil d_m) + b1_m), w2_m, dil 1) + b2_m,   m = 0, 1, 2   (d = 1, 3, 5)
// (vocoder/hifigan/models.py:46-53) chained through LDS.  bfo_resblock_pair spends half of a k = 3 block's life outside its MFMAs
// (window staging 5 us, epilogue 4 us of 24) and the C = 32 / 64 stages are at 0.45-0.6 of the HBM roof:
// three pairs per launch stage the window once, store once and move a third of the bytes.
//
// Same anatomy as bfo_pair.hip (4 waves, 32 rows x NT 32-column tiles per wave, weights streamed from L2 into a register ring,
// the c1 -> c2 intermediate over the dead window), plus: the OUTPUT of pair m (activated for its consumer, rounded to bf16
// exactly where the un-fused launch rounds it for HBM: the results are bit-identical) is written over the window as the input
// of pair m + 1.  Every phase computes all NCOLS columns of the tile; the columns that depend on data outside the staged window
// are garbage and never reach a stored output: column n of pair m sits at position x0 + sum_{i<m} (d_i + 1) + ... (see below),
// the stored outputs are the TS = NCOLS - 2 * sum (d_m + 1) = NCOLS - 24 positions [q0, q0 + TS).  The halo costs 10 % more MFMAs than
// three pair launches (232 of 256 columns useful instead of 254), on launches that keep the matrix pipe 50 % busy.
#include <cstdlib>
#include <cstring>

#include "bfo.hpp"

namespace ttsamd {

// the k = 7 ResBlock of the C = 32 / 64 stages as one launch: by default only when the padded batch holds at most this many columns
// (same-box A/B, bf16 one-stream step: batch 1 2.382 -> 2.349 ms, batch 8 4.674 -> 4.623, batch 32 11.66 -> 11.74 -- at k = 7 the
// chain's halo is 72 of 256 / 512 columns and its LDS leaves one block per CU, so on a full chip the saved round trips do not pay
// for the recomputed columns; what is left is the launch count, which shows where the launches are short).  profiles/r4/ab_chain7.txt
constexpr int64_t kBfoChain7MaxColumns = 1536 * 1024;

template <int C, int NT_, int K_>
struct BfoChainGeo {
    static constexpr int K = K_, NP = 3, H = (K_ - 1) / 2;
    static constexpr int NO = C / 8, NH = C / 16;
    static constexpr int WM = C / 32, WN = 4 / WM;
    static constexpr int NT = NT_;
    static constexpr int NCOLS = WN * NT * 32;
    static constexpr int WS = NCOLS + (K - 1) * BFO_DMAX;   // LDS entries per octet row (as the pair kernel)
    static constexpr int NE = NO * WS;
    static constexpr int NXI = (NE + 255) / 256;
    static constexpr int PH = K_ <= 3 ? (C <= 64 ? 4 : 2) : 1;   // 16-channel groups the weight ring runs ahead (as the pair kernels)
    static constexpr size_t LDS = (size_t)NE * 16;
};

template <int C, int NT_, int K_>
__global__ __launch_bounds__(256, (NT_ <= 4 ? 3 : 2)) void bfo_resblock_chain(const BfoChainParams p) {
    using G = BfoChainGeo<C, NT_, K_>;
    constexpr int K = K_, H = G::H, NO = G::NO, NH = G::NH, WN = G::WN, NT = G::NT, WS = G::WS, NXI = G::NXI;
    extern __shared__ __attribute__((aligned(16))) uint4 Xs[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5, l31 = lane & 31;
    const int wm = wid / WN, wn = wid % WN;
    const int HT = H * (p.dil[0] + p.dil[1] + p.dil[2] + 3);   // columns lost on each side over the three pairs: (K-1)/2 (d + 1) per pair
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
    const bfo_i4 xrs = bfo_rsrc((const char*)p.x + (int64_t)b * NO * L * 16, (unsigned)NO * L * 16);

    // ---- stage the window of pair 0: column c = position x0 + c, all loads first, then the LDS writes
    int xw = q0 - HT;                                        // position of LDS column 0 for the current pair's INPUT
    {
        const int W1 = G::NCOLS + (K - 1) * p.dil[0];
        bfo_i4 xv[NXI];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int e = tid + 256 * i;
            const int o = e / WS, col = e - o * WS;
            const int pos = xw + col;
            const bool ok = e < G::NE && col < W1 && pos >= 0 && pos < len;
            xv[i] = bfo_ld16(xrs, ok ? (o * L + pos) * 16 : BFO_OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int e = tid + 256 * i;
            if (e < G::NE) Xs[e] = __builtin_bit_cast(uint4, xv[i]);
        }
    }
    __syncthreads();

    const int wv = (kk * C + 32 * wm + l31) * 16;           // this lane's A fragment inside a (h, tap) step
    const int cw = wn * (NT * 32) + l31;                    // this lane's column in tile 0
    const uint4* sB = Xs + kk * WS + cw;
    const unsigned wbytes = (unsigned)NH * K * 2 * C * 16;
    const float inv_in = 1.f / p.in_slope;

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
        bfo_mma<K, G::PH, NT>(acc, bfo_rsrc(p.w1[m], wbytes), wv, 2 * C * 16, sB, NH, 2 * WS, dil);

        // residual = the pair's (activated) input at its output positions: output column n reads window column n + H (dil + 1)
        bfo_i2 rv[NT][4];
        {
            const int rc0 = cw + H * (dil + 1);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    rv[j][g] = *reinterpret_cast<const bfo_i2*>(reinterpret_cast<const char*>(Xs + (4 * wm + g) * WS + rc0 + 32 * j) + 8 * kk);
        }
        __syncthreads();                                    // every wave is done with the window
        {
            const float ms = p.mid_slope;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = cw + 32 * j;
                const int pos = xw + H * dil + col;
                const int live = (pos >= 0 && pos < len) ? -1 : 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bfo_i2 w = bfo_act4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], ms, live);
                    *reinterpret_cast<bfo_i2*>(reinterpret_cast<char*>(Xs + (4 * wm + g) * WS + col) + 8 * kk) = w;
                }
            }
        }
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
                    const float a0 = bfo_lo(rv[j][g].x), a1 = bfo_hi(rv[j][g].x), a2 = bfo_lo(rv[j][g].y), a3 = bfo_hi(rv[j][g].y);
                    acc[j][4 * g] = bv[4 * g] + bfo_unrelu(a0, inv_in);
                    acc[j][4 * g + 1] = bv[4 * g + 1] + bfo_unrelu(a1, inv_in);
                    acc[j][4 * g + 2] = bv[4 * g + 2] + bfo_unrelu(a2, inv_in);
                    acc[j][4 * g + 3] = bv[4 * g + 3] + bfo_unrelu(a3, inv_in);
                }
        }
        bfo_mma<K, G::PH, NT>(acc, bfo_rsrc(p.w2[m], wbytes), wv, 2 * C * 16, sB, NH, 2 * WS, 1);
        xw += H * (dil + 1);
        if (m + 1 == G::NP) break;

        // ---- the pair's output, activated for the next pair (in_slope) and rounded as the un-fused launch rounds it for HBM,
        // becomes the next window: column n = position xw + n, zero outside the utterance (the next conv pads at the TRUE edge)
        __syncthreads();                                    // every wave is done with the intermediate
        {
            const float is = p.in_slope;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = cw + 32 * j;
                const int pos = xw + col;
                const int live = (pos >= 0 && pos < len) ? -1 : 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bfo_i2 w = bfo_act4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], is, live);
                    *reinterpret_cast<bfo_i2*>(reinterpret_cast<char*>(Xs + (4 * wm + g) * WS + col) + 8 * kk) = w;
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue of the last pair (xw == q0 now): [+ running ResBlock sum] [/ n_kernels], consumer's leaky-relu, 8-byte stores
    int vo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = cw + 32 * j, q = q0 + n;
        vo[j] = (n < TS && q < len) ? q * 16 + 8 * kk : BFO_OOB;
    }
    const bfo_i4 yrs = bfo_rsrc((char*)p.y + (int64_t)b * NO * L * 16, (unsigned)NO * L * 16);
    const float os = p.out_slope;
    if (p.mode != 0) {
        const bfo_i4 srs = bfo_rsrc((const char*)p.sum_in + (int64_t)b * NO * L * 16, (unsigned)NO * L * 16);
        const float sc = p.mode == 2 ? 1.f / p.div : 1.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bfo_i2 sv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) sv[g] = bfo_ld8(srs, vo[j], (4 * wm + g) * L * 16, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v0 = (acc[j][4 * g] + bfo_lo(sv[g].x)) * sc, v1 = (acc[j][4 * g + 1] + bfo_hi(sv[g].x)) * sc;
                const float v2 = (acc[j][4 * g + 2] + bfo_lo(sv[g].y)) * sc, v3 = (acc[j][4 * g + 3] + bfo_hi(sv[g].y)) * sc;
                bfo_st8(bfo_act4(v0, v1, v2, v3, os, -1), yrs, vo[j], (4 * wm + g) * L * 16, 0);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bfo_st8(bfo_act4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], os, -1), yrs, vo[j],
                        (4 * wm + g) * L * 16, 0);
    }
}

template <int C, int NT, int K>
static int32_t bfo_launch_chain_nt(const BfoChainParams& p, hipStream_t stream) {
    using G = BfoChainGeo<C, NT, K>;
    static std::atomic<uint64_t> lds_done{0};          // per instantiation: devices already opted in (common.hpp: lds_opt_in)
    TTS_CHECK_HIP(lds_opt_in((const void*)bfo_resblock_chain<C, NT, K>, (int)G::LDS, lds_done));
    const int TS = G::NCOLS - 2 * G::H * (p.dil[0] + p.dil[1] + p.dil[2] + 3);
    dim3 grid((p.L + TS - 1) / TS, 1, p.batch);
    BfoChainParams q = p;
    q.compact = compact_order(p.lens, p.batch) ? 1 : 0;
    hipLaunchKernelGGL((bfo_resblock_chain<C, NT, K>), grid, dim3(256), G::LDS, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// A ResBlock (three pairs) in one launch: k = 3 at C = 32 / 64 / 128, k = 7 at C = 32 / 64 (there the pairs keep the matrix pipe 34-54 %
// busy and sit at 0.4 of the HBM roof: the halo's extra MFMAs -- 72 of 256 / 512 columns -- are affordable, the two saved tensor round trips
// are not free; at k = 11 and at C = 128 k = 7 the pairs run at the power-managed matrix roof and the halo would only cost).
// TTSAMD_BFO_CHAIN=0 keeps the three pair launches, TTSAMD_BFO_CHAIN7=0 / 1 forces the k = 7 choice (k = 3 measured: batch 32 11.63 -> 11.32 ms per step,
// batch 8 4.38 -> 4.21, batch 1 2.09 -> 1.97).
// what the kernel can run (the C-ABI entry ttsamd_bfo_resblock_chain accepts exactly this) ...
bool bfo_chain_supported(int32_t channels, int32_t k, const int32_t* dil, int32_t n_pairs, int32_t L, int32_t batch) {
    (void)batch;
    if (n_pairs != 3) return false;
    if (k == 7) {
        if (!(channels == 32 || channels == 64)) return false;
    } else if (k != 3 || !(channels == 32 || channels == 64 || channels == 128)) {
        return false;
    }
    for (int m = 0; m < 3; ++m)
        if (dil[m] < 1 || dil[m] > BFO_DMAX) return false;
    if ((int64_t)channels * L * 2 >= ((int64_t)1 << 31)) return false;
    const int ncols = channels == 32 ? 512 : 256;
    return ncols - (k - 1) * (dil[0] + dil[1] + dil[2] + 3) >= ncols / 2;
}

// ... and what the generator routes to it (the switches are read per call, like the other schedule switches: tests and A/B runs flip them)
bool bfo_chain_wanted(int32_t channels, int32_t k, const int32_t* dil, int32_t n_pairs, int32_t L, int32_t batch) {
    const char* ce = opt_str(OPT_BFO_CHAIN);
    if (ce && ce[0] == '0') return false;
    if (k == 7) {
        const char* c7 = opt_str(OPT_BFO_CHAIN7);           // 0 / 1 force it; default: small batches only
        const bool on7 = c7 ? c7[0] != '0' : (int64_t)batch * L <= kBfoChain7MaxColumns;
        if (!on7) return false;
    }
    return bfo_chain_supported(channels, k, dil, n_pairs, L, batch);
}

int32_t bfo_launch_chain(int32_t channels, const BfoChainParams& p, hipStream_t stream) {
    TTS_REQUIRE(p.k == 3 || p.k == 7, "bf16 ResBlock chain: kernel size %d (3 or 7)", p.k);
    TTS_REQUIRE(bfo_chain_supported(channels, p.k, p.dil, 3, p.L, p.batch), "bf16 ResBlock chain: unsupported geometry (C=%d, k=%d, L=%d)", channels,
                p.k, p.L);
    TTS_REQUIRE(p.x != p.y, "bf16 ResBlock chain: x and y must differ (halo reads)");
    TTS_REQUIRE(p.mode == 0 || p.sum_in != nullptr, "bf16 ResBlock chain: mode %d needs sum_in", p.mode);
    conv_log("bfo_chain", p.k, channels, channels, p.L, p.batch, 1, p.mode, p.len_mul, p.lens != nullptr, 3);
    if (p.k == 7) {
        if (channels == 64) return bfo_launch_chain_nt<64, 4, 7>(p, stream);
        return bfo_launch_chain_nt<32, 4, 7>(p, stream);
    }
    if (channels == 128) return bfo_launch_chain_nt<128, 8, 3>(p, stream);
    if (channels == 64) return bfo_launch_chain_nt<64, 4, 3>(p, stream);
    return bfo_launch_chain_nt<32, 4, 3>(p, stream);
}

}  // namespace ttsamd
