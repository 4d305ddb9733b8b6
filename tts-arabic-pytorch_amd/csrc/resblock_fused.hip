// One launch for a whole c1 -> c2 pair of a HiFi-GAN ResBlock1 at C = 32 (the 256x-upsampled stage; C = 64 at k = 3), exact fp32 MFMA:
//     y = x + conv1d(lrelu(conv1d(lrelu(x), w1, dil d) + b1), w2, dil 1) + b2          (vocoder/hifigan/models.py:46-53)
// with the intermediate activation kept in LDS.  The unfused pair moves 5 tensor passes through HBM (c1: read x, write t;
// c2: read t, read x, write y = 2.35 GB at batch 32) and at C = 32 that, not the matrix pipe, is the bound (k = 3: 3.6-4.5
// TB/s); fused it is read x + write y (+ the residual re-read, an L2 hit: the block has just staged the
// same columns).  Price: each block recomputes the (K-1)/2-column halo of the intermediate on both sides (1.6 / 3.1 / 4.7 %
// more MFMA work at k = 3 / 7 / 11).
//
// (C = 64, round 2: the same kernel with two 32-row MFMA tiles per wave, 8 input octets, 16 weight chunks; used where the halo is
// cheap and the unfused pair is far from the matrix pipe -- k = 3, 92 TF unfused.)
// Block = 4 waves x (C channels x 64 columns) = 256 MFMA columns:
//   phase A  T[32][256] = conv(lrelu(x)) for positions [q0-h, q0-h+256), h = (K-1)/2.  The whole input window
//            (32 channels x (256 + (K-1)d) positions) is staged once, as float4 = four channel pairs per LDS entry
//            (same operand layout as conv_mfma.hip); the two weight sets stream through a two-stage ring, one 8-channel
//            octet per chunk, eight chunks (4 of w1, 4 of w2) in one sequence;
//   T -> LDS  lrelu(T + b1), zero outside the utterance (c2 pads at the true edge), in the B-operand layout, over the
//            dead input window;
//   phase B  Y = conv(T, w2) for the TS = (256 - 2h) & ~3 outputs [q0, q0+TS); the accumulators start from the residual
//            (+ the running ResBlock sum in the accumulate modes), as in conv_mfma.hip's EPI 3;
//   epilogue + b2 [, / n_kernels], transposed through LDS, float4 row stores.
#include <cstring>

#include "conv_mfma_common.hpp"

namespace ttsamd {

struct FusedPairParams {
    const float* x;        // [B][C][L] input = residual
    float* y;              // [B][C][L]; must not alias x (other blocks read x's halo)
    const float4* w1;      // packed [C/8 oct][K][2][C][4]
    const float4* w2;
    const float* b1;
    const float* b2;
    const int64_t* lens;   // valid length = lens[b] * len_mul (nullptr -> L)
    int32_t len_mul, L, dil, batch;
    int32_t mode;          // 0: y = v   1: y = y + v   2: y = (y + v) / div
    float div, slope;
    int32_t compact;       // set by the launcher: ragged batch, blocks take the lin-th LIVE tile (common.hpp: live_tile)
};

template <int K, int C>
struct FusedGeo {
    static constexpr int NOCT = C / 8, MT = C / 32;            // input octets = weight chunks per conv; 32-row MFMA tiles per wave
    static constexpr int H = (K - 1) / 2;
    static constexpr int TS = (256 - 2 * H) & ~3;              // outputs per block
    static constexpr int W1S = 256 + (K - 1) * DMAX;           // staged input columns (one float4 each), dilation <= DMAX
    static constexpr int TSTR = 256 + K - 1;                   // columns of the intermediate incl. the over-read of dead MFMA columns
    static constexpr int XT4 = 2 * NOCT * W1S;                 // float4s of the input window (>= 2 NOCT TSTR and >= C x 256 floats)
    static constexpr int WCH4 = K * 2 * C;                     // float4s of one weight chunk: [K][2][C]
    static constexpr int NWL = (WCH4 + 255) / 256;             // weight float4 loads per thread and chunk
    static constexpr int LDS4 = XT4 + 2 * WCH4;
    // prefetch distance in chunks (weights: register sets in flight; input octets): a k = 3 chunk is 24 MFMAs = 0.64 us,
    // shorter than one L2 / HBM round trip
#ifdef TTS_FUSED_PF64
    static constexpr int PFW = C > 32 ? TTS_FUSED_PF64 : (K <= 3 ? 3 : (K <= 7 ? 2 : 1));
#else
    static constexpr int PFW = C > 32 ? (K <= 3 ? 2 : 1) : (K <= 3 ? 3 : (K <= 7 ? 2 : 1));
#endif
    static constexpr int PFX = PFW;
    static_assert(C * 64 <= XT4, "the epilogue's [C][256] transposition buffer must fit in the input window");
};

template <int K, int C>
__global__ __launch_bounds__(256, 2) void resblock_pair(const FusedPairParams p) {
    using G = FusedGeo<K, C>;
    constexpr int NOCT = G::NOCT, MT = G::MT, NCH = 2 * NOCT;
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    constexpr int H = G::H, TS = G::TS, W1S = G::W1S, TSTR = G::TSTR;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int kk = lane >> 5, l31 = lane & 31;
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
    const int dil = p.dil;
    const int pad1 = (K - 1) * dil / 2;
    const int W1 = 256 + (K - 1) * dil;                        // staged columns actually used
    const int x0 = q0 - H - pad1;                              // position of staged column 0
    const float slope = p.slope;
    const int L = p.L;
    const float* __restrict__ xb = p.x + (int64_t)b * C * L;

    float4* Xs = smem4;                                        // [o][kk][W1S]
    float4* Wr = smem4 + G::XT4;                               // 2 stages x [K][2][C]

    // ---- weight chunk 0 (first octet of w1) and the input window: loads first, then the residual preload, then LDS ---
    constexpr int PFW = G::PFW, PFX = G::PFX;
    float w0reg[4 * G::NWL], wreg[PFW][4 * G::NWL];            // chunk 0; chunks in flight (set = chunk % PFW); scalars: float4 arrays go to scratch
#define TTS_W_LOAD(DST, CH)                                                                             \
    {                                                                                                   \
        const float4* __restrict__ wsrc = ((CH) < NOCT ? p.w1 : p.w2) + (int64_t)((CH) % NOCT) * G::WCH4; \
        _Pragma("unroll") for (int i = 0; i < G::NWL; ++i) {                                          \
            const float4 t4 = wsrc[min(tid + 256 * i, G::WCH4 - 1)];                                    \
            DST[4 * i] = t4.x; DST[4 * i + 1] = t4.y; DST[4 * i + 2] = t4.z; DST[4 * i + 3] = t4.w;     \
        }                                                                                               \
    }
    TTS_W_LOAD(w0reg, 0)
#pragma unroll
    for (int cc = 1; cc <= PFW; ++cc) TTS_W_LOAD(wreg[cc % PFW], cc)

    // accumulators of phase B start from the residual (+ the running sum): buffer loads, a scalar row offset per load
    f32x16 acc2[MT][2];
    {
        const int wid_s = __builtin_amdgcn_readfirstlane(wid);
        int voff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wid_s * 64 + j * 32 + l31;
            const int q = q0 + n;
            voff[j] = ((n < TS && q < len) ? q : 0) * 4 + 4 * kk * L * 4;
        }
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, C * L * 4, 0x00020000);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc2[mt][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rs, voff[j], (32 * mt + (r & 3) + 8 * (r >> 2)) * L * 4, 0));
        if (p.mode != 0) {
            const auto ys = __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)b * C * L, 0, C * L * 4, 0x00020000);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 t;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        t[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            ys, voff[j], (32 * mt + (r & 3) + 8 * (r >> 2)) * L * 4, 0));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[mt][j][r] = t[r] + acc2[mt][j][r];
                }
        }
    }

    // input window: entry (o, kk_e, col) holds channels 8o + kk_e + {0,2,4,6} at position x0 + col.  Octet 0 is staged
    // here; octets 1-3 are loaded during the MFMAs of the chunk before they are needed (as the weight chunks are), so the
    // first MFMA waits for a quarter of the window only.
    constexpr int NXO = (2 * W1S + 255) / 256;                 // entries per thread and octet
    float xv[NOCT][NXO][4];                                    // per octet (static indices: the chunk loop is unrolled)
    bool x_ok[NXO], x_in[NXO];
    int x_off[NXO];
#pragma unroll
    for (int i = 0; i < NXO; ++i) {
        const int e = tid + 256 * i;
        const int kke = min(e / W1S, 1), col = e % W1S;
        const int pos = x0 + col;
        x_in[i] = e < 2 * W1S;
        x_ok[i] = x_in[i] && col < W1 && pos >= 0 && pos < len;
        x_off[i] = kke * L + min(max(pos, 0), max(len - 1, 0));
    }
#define TTS_X_LOAD(O)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < NXO; ++i) {                                                   \
        const float* src = xb + (int64_t)(O) * 8 * L + x_off[i];                                        \
        _Pragma("unroll") for (int pc_ = 0; pc_ < 4; ++pc_) xv[O][i][pc_] = src[(int64_t)2 * pc_ * L];  \
    }
#define TTS_X_WRITE(O)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NXO; ++i) {                                                   \
        if (x_in[i]) {                                                                                  \
            float4 o4;                                                                                  \
            o4.x = x_ok[i] ? (xv[O][i][0] > 0.f ? xv[O][i][0] : xv[O][i][0] * slope) : 0.f;                      \
            o4.y = x_ok[i] ? (xv[O][i][1] > 0.f ? xv[O][i][1] : xv[O][i][1] * slope) : 0.f;                      \
            o4.z = x_ok[i] ? (xv[O][i][2] > 0.f ? xv[O][i][2] : xv[O][i][2] * slope) : 0.f;                      \
            o4.w = x_ok[i] ? (xv[O][i][3] > 0.f ? xv[O][i][3] : xv[O][i][3] * slope) : 0.f;                      \
            Xs[(O) * 2 * W1S + tid + 256 * i] = o4;                                                     \
        }                                                                                               \
    }
    TTS_X_LOAD(0)
#pragma unroll
    for (int oo = 1; oo <= PFX && oo < NOCT; ++oo) TTS_X_LOAD(oo)
    TTS_X_WRITE(0)
#pragma unroll
    for (int i = 0; i < G::NWL; ++i)
        if (tid + 256 * i < G::WCH4) Wr[tid + 256 * i] = make_float4(w0reg[4 * i], w0reg[4 * i + 1], w0reg[4 * i + 2], w0reg[4 * i + 3]);
    __syncthreads();

    // ---- main loop: 2 NOCT weight chunks; the first NOCT = phase A on the input window, the rest = phase B on the intermediate
    // phase A accumulators start from b1 (row = channel 32mt + (r&3) + 8(r>>2) + 4kk): no bias registers to keep alive
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][0][r] = acc[mt][1][r] = p.b1[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kk];

    const int colw = wid * 64 + l31;                           // this lane's MFMA column (j = 0), + 32 for j = 1
    const float4* sA = Wr + kk * C + l31;                      // + stage * WCH4 + tap * 2C + 32 mt
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int o = c % NOCT;
        const bool phase_b = c >= NOCT;
        // chunk c+1 goes from registers to LDS at the end of this chunk; its register set is free again only then, so the
        // loads of chunk c+1+PFW are issued after that write (below); input octet c+1+PFX likewise

        const float4* sAc = sA + (c & 1) * G::WCH4;
        const int bstr = phase_b ? TSTR : W1S;
        const int bdil = phase_b ? 1 : dil;
        const float4* sB = Xs + (o * 2 + kk) * bstr + colw;
        float4 a4[MT], b4[2] = {sB[0], sB[32]};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a4[mt] = sAc[32 * mt];
#pragma unroll
        for (int t = 0; t < K; ++t) {
            float4 an[MT], bn[2] = {b4[0], b4[1]};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) an[mt] = a4[mt];
            if (t + 1 < K) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) an[mt] = sAc[(t + 1) * 2 * C + 32 * mt];
                bn[0] = sB[(t + 1) * bdil];
                bn[1] = sB[(t + 1) * bdil + 32];
            }
            const float b0[4] = {b4[0].x, b4[0].y, b4[0].z, b4[0].w};
            const float b1[4] = {b4[1].x, b4[1].y, b4[1].z, b4[1].w};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float av[4] = {a4[mt].x, a4[mt].y, a4[mt].z, a4[mt].w};
                if (!phase_b) {
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) {
                        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pq], b0[pq], acc[mt][0], 0, 0, 0);
                        acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pq], b1[pq], acc[mt][1], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) {
                        acc2[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pq], b0[pq], acc2[mt][0], 0, 0, 0);
                        acc2[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pq], b1[pq], acc2[mt][1], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a4[mt] = an[mt];
            b4[0] = bn[0]; b4[1] = bn[1];
        }
        // ring: the other stage was last read in chunk c-1, which every wave left at the previous barrier
        if (c + 1 < NCH) {
            float4* wfill = Wr + ((c + 1) & 1) * G::WCH4;
#pragma unroll
            for (int i = 0; i < G::NWL; ++i)
                if (tid + 256 * i < G::WCH4)
                    wfill[tid + 256 * i] = make_float4(wreg[(c + 1) % PFW][4 * i], wreg[(c + 1) % PFW][4 * i + 1],
                                                       wreg[(c + 1) % PFW][4 * i + 2], wreg[(c + 1) % PFW][4 * i + 3]);
            if (c + 1 + PFW < NCH) TTS_W_LOAD(wreg[(c + 1 + PFW) % PFW], c + 1 + PFW)
        }
        if (c + 1 < NOCT) {
            TTS_X_WRITE(c + 1)
            if (c + 1 + PFX < NOCT) TTS_X_LOAD(c + 1 + PFX)
        }
        __syncthreads();
        if (c == NOCT - 1) {
            // ---- intermediate -> LDS (over the dead input window; every wave passed the barrier above) -----------
            // acc[mt][j][r] = channel 32mt + (r&3) + 8(r>>2) + 4kk at column colw + 32j; entry (o, kk', col) component p holds
            // channel 8o + 2p + kk': registers (r, r+2) with r&3 in {0,1} are components (2kk, 2kk+1) of entry (4mt + (r>>2), r&1, col)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
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
                            v0 = v0 > 0.f ? v0 : v0 * slope;
                            v1 = v1 > 0.f ? v1 : v1 * slope;
                            float2 w2v = live ? make_float2(v0, v1) : make_float2(0.f, 0.f);
                            *reinterpret_cast<float2*>(reinterpret_cast<float*>(Xs + ((4 * mt + oc) * 2 + k2) * TSTR + col) + 2 * kk) = w2v;
                        }
            }
            __syncthreads();
        }
    }

#undef TTS_X_LOAD
#undef TTS_X_WRITE
#undef TTS_W_LOAD
    // ---- epilogue: + b2 [, / div], transposed through LDS (the intermediate is dead after the last barrier) ------------
    float* ep = reinterpret_cast<float*>(smem4);               // [C][256]
    const bool do_div = p.mode == 2;
    const float div = p.div;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kk) * 256 + colw + 32 * j] = acc2[mt][j][r];
    __syncthreads();
    float* __restrict__ yb = p.y + (int64_t)b * C * L;
    const int n = lane * 4;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);     // wave-uniform row -> the bias comes in through the scalar cache
#pragma unroll 4
    for (int it = 0; it < C / 4; ++it) {
        const int ch = wid_u + 4 * it;
        const float bs = p.b2[ch];
        const int q = q0 + n;
        if (n >= TS || q >= len) continue;
        const float4 v4 = *reinterpret_cast<const float4*>(ep + ch * 256 + n);
        float vv[4] = {v4.x + bs, v4.y + bs, v4.z + bs, v4.w + bs};
        if (do_div) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = vv[e] / div;
        }
        float* yp = yb + (int64_t)ch * L + q;
        if (q + 3 < len) {
            *reinterpret_cast<float4*>(yp) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q + e < len) yp[e] = vv[e];
        }
    }
}

template <int K, int C>
static int32_t launch_fused_k(const FusedPairParams& p, hipStream_t stream) {
    using G = FusedGeo<K, C>;
    static std::atomic<uint64_t> lds_done{0};          // per instantiation: devices already opted in (common.hpp: lds_opt_in)
    const size_t lds = (size_t)G::LDS4 * sizeof(float4);
    TTS_CHECK_HIP(lds_opt_in((const void*)resblock_pair<K, C>, (int)lds, lds_done));
    dim3 grid((p.L + G::TS - 1) / G::TS, 1, p.batch);
    hipLaunchKernelGGL((resblock_pair<K, C>), grid, dim3(256), lds, stream, p);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// true if the fused kernel covers this pair (C = 32: k = 3 / 7 / 11; C = 64: k = 3, the one that wins there; fp32, aligned rows)
bool fused_pair_supported(int32_t channels, int32_t k, int32_t dil, int32_t L, const float* x, const float* y) {
    const char* e64 = exp_env("TTSAMD_FUSED_PAIR_C64");         // read per call: the tests and A/B runs flip it
    const bool c64 = !(e64 && e64[0] == '0');
    const bool geo = (channels == 32 && (k == 3 || k == 7 || k == 11)) || (channels == 64 && k == 3 && c64);
    return geo && dil >= 1 && dil <= DMAX && (L & 3) == 0 &&
           (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && x != y && (int64_t)channels * L * 4 < ((int64_t)1 << 31);
}

int32_t launch_fused_pair(int32_t channels, const float* x, float* y, const float* w1, const float* b1, const float* w2,
                          const float* b2, int32_t k, int32_t dil, const int64_t* lens, int32_t len_mul, int32_t L, int32_t batch,
                          int32_t mode, float div, float slope, hipStream_t stream) {
    TTS_REQUIRE(fused_pair_supported(channels, k, dil, L, x, y), "fused ResBlock pair: unsupported geometry (C=%d, k=%d, dil=%d, L=%d)",
                channels, k, dil, L);
    conv_log("fused_pair", k, channels, channels, L, batch, 1, mode, len_mul, lens != nullptr, 1);
    FusedPairParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.y = y;
    p.w1 = reinterpret_cast<const float4*>(w1); p.w2 = reinterpret_cast<const float4*>(w2);
    p.b1 = b1; p.b2 = b2; p.lens = lens; p.len_mul = len_mul; p.L = L; p.dil = dil; p.batch = batch;
    p.mode = mode; p.div = div; p.slope = slope;
    p.compact = compact_order(lens, batch) ? 1 : 0;
    if (channels == 64) return launch_fused_k<3, 64>(p, stream);
    switch (k) {
        case 3: return launch_fused_k<3, 32>(p, stream);
        case 7: return launch_fused_k<7, 32>(p, stream);
        default: return launch_fused_k<11, 32>(p, stream);
    }
}

}  // namespace ttsamd
