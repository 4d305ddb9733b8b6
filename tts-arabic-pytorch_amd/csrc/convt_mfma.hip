// ConvTranspose1d(stride u, kernel 2u, padding u/2) + leaky-relu on load: the HiFi-GAN upsamplers
// (vocoder/hifigan/models.py:96-99, 114-115), exact fp32 MFMA, with ALL u output phases of a tile in one wave.
//
//   y[co][q*u + rho] = b[co] + sum_ci W[ci][co][ka] x[ci][q + d] + W[ci][co][ka + u] x[ci][q + d - 1],
//   ka = (rho + p) % u, d = (rho + p) / u in {0, 1}          (polyphase form; weights packed by pack_convt_weight)
//
// The generic engine (conv_mfma.hip) runs the u phases as separate blocks (grid.y = phase x co-tile), so a block owns one
// phase and can only store with stride u: one dword per lane at stride 4u bytes.  Measured with the fabric counters
// (profiles/r2/traffic.json): WRITE_SIZE counts 8x the bytes for that pattern at u = 8 (every 4-byte store is its own
// partial-line write): 5.5 GB moved for a 0.59 GB launch -- the u = 8 upsamplers were bound by write requests, not by the
// matrix pipe.  Here a wave keeps u accumulators per (co-tile, q-tile), one per phase: the phases share the staged input
// (two column shifts, d = 0 / 1) and differ only in the A operand, and in the C layout lane q then owns the u consecutive
// outputs y[co][q*u .. q*u+u) of every row: u = 8 -> two float4 stores per row, u = 2 -> one float2, full lines either way.
#include <cstring>

#include <cstdlib>
#include "conv_mfma_common.hpp"

namespace ttsamd {

template <int U, int MT, int NTL, int WM>
struct ConvtGeo {
    static constexpr int WN = 4 / WM;
    static constexpr int CO_BLK = WM * MT * 32;
    static constexpr int NT_BLK = WN * NTL * 32;               // input positions per block
    static constexpr int XS = NT_BLK + 2;                      // staged columns: positions q0-1 .. q0+NT_BLK
    static constexpr int X4 = 2 * XS;                          // float4s: [kk][XS]
    static constexpr int W4 = U * 4 * CO_BLK;                  // float4s: [rho][tap][kk][CO_BLK]
    static constexpr int STG4 = X4 + W4;
    static constexpr int NW = (W4 + 255) / 256;
    static constexpr int NX = (X4 + 255) / 256;
};

template <int U, int MT, int NTL, int WM>
__global__ __launch_bounds__(256, 2) void convt_mfma_f32(const ConvParams p) {
    using G = ConvtGeo<U, MT, NTL, WM>;
    constexpr int WN = G::WN, CO_BLK = G::CO_BLK, NT_BLK = G::NT_BLK, XS = G::XS, NW = G::NW, NX = G::NX;
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int kk = lane >> 5, l31 = lane & 31;
    int b = blockIdx.z;
    const int co_blk0 = blockIdx.y * CO_BLK;
    int q0 = blockIdx.x * NT_BLK;
    if (p.compact) {   // ragged batch: dead blocks last (common.hpp: live_tile)
        int tile = 0;
        if (!live_tile(p.lens_out, p.len_out_mul, p.Nout, NT_BLK, p.batch, blockIdx.z * gridDim.x + blockIdx.x, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * NT_BLK;
    }
    int n_out = p.Nout;                                        // input positions that produce output
    if (p.lens_out) n_out = min(n_out, (int)p.lens_out[b] * p.len_out_mul);
    if (q0 >= n_out) return;
    int in_len = p.Lin;
    if (p.lens_in) in_len = min(in_len, (int)p.lens_in[b] * p.len_in_mul);
    const int x_cs = p.x_cs, CoutP = p.CoutP, n_oct = p.Cin / 8;
    const float slope = p.in_slope;
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    const float4* __restrict__ wp4 = reinterpret_cast<const float4*>(p.w) + co_blk0;

    // ---- staging descriptors ------------------------------------------------------------------------------------
    // X entry e = kk_e * XS + col: channels 8o + kk_e + {0,2,4,6} at position q0 - 1 + col
    bool x_ok[NX], x_in[NX];
    int x_off[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tid + 256 * i;
        const int kke = min(e / XS, 1), col = e % XS;
        const int pos = q0 - 1 + col;
        x_in[i] = e < G::X4;
        x_ok[i] = x_in[i] && pos >= 0 && pos < in_len;
        x_off[i] = kke * x_cs + min(max(pos, 0), max(in_len - 1, 0));
    }
    // W float4 f = ((rho * 2 + tap) * 2 + kk_w) * CO_BLK + co  <-  wp4[(((rho * n_oct + o) * 2 + tap) * 2 + kk_w) * CoutP + co]
    int w_off[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int f = min(tid + 256 * i, G::W4 - 1);
        const int co = f % CO_BLK, r = f / CO_BLK;             // r = (rho*2 + tap)*2 + kk_w
        const int rho = r >> 2, tk = r & 3;
        w_off[i] = ((rho * n_oct * 4) + tk) * CoutP + co;      // + o * 4 * CoutP per octet
    }
    float xv[NX][4], wv[NW][4];
#define TTS_CT_LOAD(O)                                                                                  \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NX; ++i) {                                                \
            const float* src = xb + (int64_t)(O) * 8 * x_cs + x_off[i];                                 \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) xv[i][c_] = src[(int64_t)2 * c_ * x_cs];   \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                                \
            const float4 t4 = wp4[(int64_t)(O) * 4 * CoutP + w_off[i]];                                 \
            wv[i][0] = t4.x; wv[i][1] = t4.y; wv[i][2] = t4.z; wv[i][3] = t4.w;                         \
        }                                                                                               \
    }
#define TTS_CT_LRELU(v) ((v) > 0.f ? (v) : (v)*slope)
#define TTS_CT_WRITE(SB)                                                                                \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NX; ++i)                                                  \
            if (x_in[i])                                                                                \
                (SB)[tid + 256 * i] = x_ok[i] ? make_float4(TTS_CT_LRELU(xv[i][0]), TTS_CT_LRELU(xv[i][1]), \
                                                            TTS_CT_LRELU(xv[i][2]), TTS_CT_LRELU(xv[i][3])) \
                                              : make_float4(0.f, 0.f, 0.f, 0.f);                        \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                  \
            if (tid + 256 * i < G::W4) (SB)[G::X4 + tid + 256 * i] = make_float4(wv[i][0], wv[i][1], wv[i][2], wv[i][3]); \
    }

    TTS_CT_LOAD(0)

    // accumulators start from the bias: acc[rho][i][j][r] = b[co], co = co_w + 32i + (r&3) + 8(r>>2) + 4kk
    f32x16 acc[U][MT][NTL];
    {
        const int co_w = co_blk0 + wm * MT * 32;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = min(co_w + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk, p.Cout - 1);
                const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
                for (int rho = 0; rho < U; ++rho)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) acc[rho][i][j][r] = bv;
            }
    }
    TTS_CT_WRITE(smem4)
    __syncthreads();

    // B operand of column tile j, shift s in {-1, 0, +1}: X[kk][wn*NTL*32 + 32j + l31 + 1 + s]
    const int colw = wn * NTL * 32 + l31 + 1;
    for (int o = 0; o < n_oct; ++o) {
        const float4* st = smem4 + (o & 1) * G::STG4;
        float4* fill = smem4 + ((o + 1) & 1) * G::STG4;
        if (o + 1 < n_oct) TTS_CT_LOAD(o + 1)
        const float4* sX = st + kk * XS + colw;
        const float4* sW = st + G::X4 + kk * CO_BLK + wm * MT * 32 + l31;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // phases with d = 0 read x[q - t], d = 1 read x[q + 1 - t]
            float4 bq[2][NTL];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int j = 0; j < NTL; ++j) bq[d][j] = sX[32 * j + d - t];
#pragma unroll
            for (int rho = 0; rho < U; ++rho) {
                float4 a4[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a4[i] = sW[((rho * 2 + t) * 2) * CO_BLK + 32 * i];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        const float4 b4 = bq[(2 * rho >= U) ? 1 : 0][j];   // d = (rho + u/2) / u: padding u/2 (convt_supported)
                        const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                        for (int pq = 0; pq < 4; ++pq)
                            acc[rho][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[pq], bv[pq], acc[rho][i][j], 0, 0, 0);
                    }
                }
            }
        }
        if (o + 1 < n_oct) TTS_CT_WRITE(fill)
        __syncthreads();
    }
#undef TTS_CT_LOAD
#undef TTS_CT_WRITE
#undef TTS_CT_LRELU

    // ---- epilogue: lane (kk, l31) owns y[co][q*U .. q*U+U) for its 16 rows per co-tile: contiguous across the lanes ----
    float* __restrict__ yb = p.y + (int64_t)b * p.y_bs;
    const int y_cs = p.y_cs, Cout = p.Cout;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int q = q0 + wn * NTL * 32 + 32 * j + l31;
            if (q >= n_out) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co_blk0 + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (co >= Cout) continue;
                float* yp = yb + (int64_t)co * y_cs + (int64_t)q * U;
                if constexpr (U % 4 == 0) {
#pragma unroll
                    for (int g = 0; g < U / 4; ++g)
                        *reinterpret_cast<float4*>(yp + 4 * g) = make_float4(acc[4 * g][i][j][r], acc[4 * g + 1][i][j][r],
                                                                             acc[4 * g + 2][i][j][r], acc[4 * g + 3][i][j][r]);
                } else {
                    static_assert(U == 2, "phase count");
                    *reinterpret_cast<float2*>(yp) = make_float2(acc[0][i][j][r], acc[1][i][j][r]);
                }
            }
        }
}

template <int U, int MT, int NTL, int WM>
static int32_t launch_convt_cfg(const ConvParams& p, hipStream_t stream) {
    using G = ConvtGeo<U, MT, NTL, WM>;
    static std::atomic<uint64_t> lds_done{0};          // per instantiation: devices already opted in (common.hpp: lds_opt_in)
    const size_t lds = (size_t)2 * G::STG4 * sizeof(float4);
    TTS_CHECK_HIP(lds_opt_in((const void*)convt_mfma_f32<U, MT, NTL, WM>, (int)lds, lds_done));
    dim3 grid((p.Nout + G::NT_BLK - 1) / G::NT_BLK, p.CoutP / G::CO_BLK, p.batch);
    ConvParams q = p;
    q.compact = compact_order(p.lens_out, p.batch) ? 1 : 0;
    hipLaunchKernelGGL((convt_mfma_f32<U, MT, NTL, WM>), grid, dim3(256), lds, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// The polyphase launch of conv_mfma.hip (ConvParams with n_phase = u, K = 2, dil = -1, y_ts = u) as ONE kernel that owns
// all phases; returns false if this geometry is not covered (the caller then uses the generic engine).
bool convt_supported(const ConvParams& p) {
    const bool u_ok = (p.n_phase == 8 || p.n_phase == 2) && p.phase_p * 2 == p.n_phase;
    // a grid that leaves most CUs without a block (batch 1: 32 blocks for the first upsampler) does
    // better on the polyphase launch of the generic engine, which has one block per (phase, co tile) and splits K
    static const int min_blocks = [] { const char* e = exp_env("TTSAMD_CONVT_MIN_BLOCKS"); return e ? atoi(e) : 100; }();
    if (u_ok && p.n_phase == 8 && (int64_t)((p.Nout + 63) / 64) * (p.CoutP / 64) * p.batch < min_blocks) return false;
    return p.precision == 0 && u_ok && p.K == 2 && p.dil == -1 && p.y_ts == p.n_phase && p.res == nullptr && p.mode == 0 &&
           p.scale == nullptr && p.relu_out == 0 && p.Cin % 8 == 0 && !p.x_packed && !p.y_packed &&
           (p.y_cs % 4) == 0 && (p.y_bs % 4) == 0 && (((uintptr_t)p.y) & 15) == 0 &&
           (p.n_phase == 8 ? p.CoutP % 64 == 0 : p.CoutP % 32 == 0);
}

int32_t launch_convt(const ConvParams& p, hipStream_t stream) {
    TTS_REQUIRE(convt_supported(p), "convt: unsupported geometry");
    if (p.Nout <= 0) return 0;
    conv_log("convt", 2, p.Cin, p.Cout, p.Nout, p.batch, 0, 0, p.len_out_mul, p.lens_out != nullptr, p.n_phase);
    if (p.n_phase == 8) return launch_convt_cfg<8, 1, 1, 2>(p, stream);                 // 64 co x 64 q (x 8 phases)
    if (p.CoutP % 64 == 0) return launch_convt_cfg<2, 1, 2, 2>(p, stream);              // 64 co x 128 q (x 2)
    return launch_convt_cfg<2, 1, 4, 1>(p, stream);                                     // 32 co x 512 q (x 2)
}

}  // namespace ttsamd
