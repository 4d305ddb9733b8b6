// One launch for a whole c1 -> c2 pair of a HiFi-GAN ResBlock1 in the bf16 octet engine (bfo.hpp):
//     y = x + conv1d(lrelu(conv1d(lrelu(x), w1, dil d) + b1), w2, dil 1) + b2          (vocoder/hifigan/models.py:46-53)
// for C = 32 / 64 / 128 at k = 3 / 7 / 11, on v_mfma_f32_32x32x16_bf16, two tensor passes through HBM per pair (read
// lrelu(x) once with its halo, write once; the residual re-read is an L2 hit on the columns the block has just staged).
//
// Block = 4 waves, each owning a 32-row x 256-column slab (8 accumulators): C = 128 -> 4 row slabs x 256 columns,
// C = 64 -> 2 x 512, C = 32 -> 1 x 1024.
//   stage    the whole input window (C channels x (NCOLS + (K-1) d) positions, 66-78 KB) as 16-byte entries: plain copies
//            (the tensor is stored activated), zero outside the utterance;
//   phase A  T = conv(window, w1) for NCOLS = TS + K - 1 positions [q0 - h, q0 - h + NCOLS), accumulators start from b1;
//            the weights never touch LDS: every wave streams the A fragments of its own 32 rows from L2 into a register
//            ring (bfo_mma), so a conv has no barrier inside;
//   T -> LDS lrelu(T), zero outside the utterance (c2 pads at the TRUE edge), bf16, over the dead window (8-byte writes
//            straight from the C layout);
//   phase B  Y = conv(T, w2) for the TS outputs [q0, q0 + TS); accumulators start from b2 + the residual, whose loads are
//            read back from the LDS window before the intermediate overwrites it;
//   epilogue [+ running ResBlock sum] [/ n_kernels], leaky-relu of the CONSUMER, bf16, 8-byte stores from the C layout
//            (512 contiguous bytes per wave instruction).
// LDS <= 78 KB -> two blocks per CU: one block's loads / exchange / stores run under the other's MFMAs.
#include <cstdlib>
#include <cstring>

#include "bfo.hpp"

namespace ttsamd {

// 32-column tiles per wave: 8 (256 columns), or 4 -- half the window, 3-4 blocks per CU -- for k = 3 at C <= 64 (152 -> 137 us at
// C = 64, 118 -> 111 at C = 32 with the deeper weight ring below; no change at C = 128: tools/bfo_pair_bench) and for small
// batches, where 256-column tiles leave CUs without a block (batch 1: 115 blocks at C = 128)
template <int K, int C>
constexpr int bfo_pair_default_nt() { return (K == 3 && C <= 64) ? 4 : 8; }

template <int K, int C, int NT_>
struct BfoPairGeo {
    static constexpr int NO = C / 8, NH = C / 16;
    static constexpr int WM = C / 32, WN = 4 / WM;          // waves over rows / over columns
    static constexpr int NT = NT_;
    static constexpr int NCOLS = WN * NT * 32;              // columns of phase A
    static constexpr int H = (K - 1) / 2;
    static constexpr int TS = NCOLS - (K - 1);              // outputs per block
    static constexpr int WS = NCOLS + (K - 1) * BFO_DMAX;   // LDS entries per octet row
    static constexpr int NE = NO * WS;                      // entries of the window
    static constexpr int NXI = (NE + 255) / 256;            // ... per thread
    static constexpr int PH = K <= 3 ? (C <= 64 ? 4 : 2) : 1;   // 16-channel groups the A ring runs ahead
    static constexpr size_t LDS = (size_t)NE * 16;
};

template <int K, int C, int NT_>
__global__ __launch_bounds__(256, (NT_ <= 4 ? 3 : 2)) void bfo_resblock_pair(const BfoPairParams p) {
    using G = BfoPairGeo<K, C, NT_>;
    constexpr int NO = G::NO, NH = G::NH, WN = G::WN, NT = G::NT, H = G::H, TS = G::TS, WS = G::WS, NXI = G::NXI;
    extern __shared__ __attribute__((aligned(16))) uint4 Xs[];
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
#ifdef BFO_TIMING
    const unsigned long long wc0 = wall_clock64();
    unsigned long long tst[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) tst[i] = 0;
    tst[0] = clock64();
#define BFO_STAMP(i) tst[i] = clock64();
#else
#define BFO_STAMP(i)
#endif
    const int dil = p.dil;
    const int W1 = G::NCOLS + (K - 1) * dil;                // staged columns actually used
    const int x0 = q0 - H - (K - 1) * dil / 2;              // position of staged column 0
    const bfo_i4 xrs = bfo_rsrc((const char*)p.x + (int64_t)b * NO * L * 16, (unsigned)NO * L * 16);

    // ---- stage the window: all loads first, then the LDS writes
    {
        bfo_i4 xv[NXI];
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int e = tid + 256 * i;
            const int o = e / WS, col = e - o * WS;
            const int pos = x0 + col;
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
    BFO_STAMP(1)

    const int wv = (kk * C + 32 * wm + l31) * 16;           // this lane's A fragment inside a (h, tap) step
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
    bfo_mma<K, G::PH, NT>(acc, bfo_rsrc(p.w1, (unsigned)NH * K * 2 * C * 16), wv, 2 * C * 16, sB, NH, 2 * WS, dil);

    BFO_STAMP(2)
    // residual (= the activated input at the output positions): 8 bytes per (tile, octet) in the C layout
    int vo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = cw + 32 * j, q = q0 + n;
        vo[j] = (n < TS && q < len) ? q * 16 + 8 * kk : BFO_OOB;
    }
    // ... read back from the LDS window (column n + h + pad of the staged tile) before the intermediate overwrites it: the block's
    // vector-memory pipe already carries the window, the weight stream and the stores (a CU sustains ~11 B per cycle on that path)
    bfo_i2 rv[NT][4];
    {
        const int rc0 = cw + H + (K - 1) * dil / 2;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                rv[j][g] = *reinterpret_cast<const bfo_i2*>(reinterpret_cast<const char*>(Xs + (4 * wm + g) * WS + rc0 + 32 * j) + 8 * kk);
    }

    BFO_STAMP(6)
    __syncthreads();                                        // every wave is done with the window
    BFO_STAMP(7)
    {
        const float ms = p.mid_slope;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = cw + 32 * j;
            const int pos = q0 - H + col;
            const int live = (pos >= 0 && pos < len) ? -1 : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bfo_i2 w = bfo_act4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3], ms, live);
                *reinterpret_cast<bfo_i2*>(reinterpret_cast<char*>(Xs + (4 * wm + g) * WS + col) + 8 * kk) = w;
            }
        }
    }
    BFO_STAMP(8)
    __syncthreads();
    BFO_STAMP(3)

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
                const float a0 = bfo_lo(rv[j][g].x), a1 = bfo_hi(rv[j][g].x), a2 = bfo_lo(rv[j][g].y), a3 = bfo_hi(rv[j][g].y);
                acc[j][4 * g] = bv[4 * g] + bfo_unrelu(a0, inv);
                acc[j][4 * g + 1] = bv[4 * g + 1] + bfo_unrelu(a1, inv);
                acc[j][4 * g + 2] = bv[4 * g + 2] + bfo_unrelu(a2, inv);
                acc[j][4 * g + 3] = bv[4 * g + 3] + bfo_unrelu(a3, inv);
            }
    }
    BFO_STAMP(9)
    bfo_mma<K, G::PH, NT>(acc, bfo_rsrc(p.w2, (unsigned)NH * K * 2 * C * 16), wv, 2 * C * 16, sB, NH, 2 * WS, 1);

    BFO_STAMP(4)
    // ---- epilogue
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
#ifdef BFO_TIMING
    tst[5] = clock64();
    if (p.timing && tid == 0) {
        unsigned long long* tp = p.timing + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16;
#pragma unroll
        for (int i = 0; i < 12; ++i) tp[i] = tst[i];
        tp[12] = wall_clock64();
        tp[13] = wc0;
    }
#endif
#undef BFO_STAMP
}

template <int K, int C, int NT>
static int32_t bfo_launch_pair_nt(const BfoPairParams& p, hipStream_t stream) {
    using G = BfoPairGeo<K, C, NT>;
    static std::atomic<uint64_t> lds_done{0};          // per instantiation: devices already opted in (common.hpp: lds_opt_in)
    TTS_CHECK_HIP(lds_opt_in((const void*)bfo_resblock_pair<K, C, NT>, (int)G::LDS, lds_done));
    dim3 grid((p.L + G::TS - 1) / G::TS, 1, p.batch);
    BfoPairParams q = p;
    q.compact = compact_order(p.lens, p.batch) ? 1 : 0;
    hipLaunchKernelGGL((bfo_resblock_pair<K, C, NT>), grid, dim3(256), G::LDS, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int K, int C>
static int32_t bfo_launch_pair_k(const BfoPairParams& p, hipStream_t stream) {
    constexpr int NT0 = bfo_pair_default_nt<K, C>();
    if constexpr (NT0 == 8) {
        // fewer than 1.5 blocks per CU with 256-column tiles: halve them (TTSAMD_BFO_SMALL_TILES=0/1 forces either)
        using G8 = BfoPairGeo<K, C, 8>;
        const int64_t blocks8 = (int64_t)((p.L + G8::TS - 1) / G8::TS) * p.batch;
        const char* e = opt_str(OPT_BFO_SMALL_TILES);
        const bool small = e ? e[0] == '1' : blocks8 < 384;
        if (small) return bfo_launch_pair_nt<K, C, 4>(p, stream);
    }
    return bfo_launch_pair_nt<K, C, NT0>(p, stream);
}

bool bfo_pair_supported(int32_t channels, int32_t k, int32_t dil, int32_t L) {
    return (channels == 32 || channels == 64 || channels == 128) && (k == 3 || k == 7 || k == 11) && dil >= 1 && dil <= BFO_DMAX &&
           (int64_t)channels * L * 2 < ((int64_t)1 << 31);
}

int32_t bfo_launch_pair(int32_t channels, int32_t k, const BfoPairParams& p, hipStream_t stream) {
    TTS_REQUIRE(bfo_pair_supported(channels, k, p.dil, p.L), "bf16 ResBlock pair: unsupported geometry (C=%d, k=%d, dil=%d, L=%d)",
                channels, k, p.dil, p.L);
    TTS_REQUIRE(p.x != p.y, "bf16 ResBlock pair: x and y must differ (halo reads)");
    TTS_REQUIRE(p.mode == 0 || p.sum_in != nullptr, "bf16 ResBlock pair: mode %d needs sum_in", p.mode);
    conv_log("bfo_pair", k, channels, channels, p.L, p.batch, 1, p.mode, p.len_mul, p.lens != nullptr, 1);
#define BFO_CASE(KK, CC) if (k == KK && channels == CC) return bfo_launch_pair_k<KK, CC>(p, stream);
    BFO_CASE(3, 32) BFO_CASE(7, 32) BFO_CASE(11, 32)
    BFO_CASE(3, 64) BFO_CASE(7, 64) BFO_CASE(11, 64)
    BFO_CASE(3, 128) BFO_CASE(7, 128) BFO_CASE(11, 128)
#undef BFO_CASE
    return TTSAMD_EINVAL;
}

}  // namespace ttsamd
