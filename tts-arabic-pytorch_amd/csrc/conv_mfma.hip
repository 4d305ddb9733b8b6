// Implicit-GEMM Conv1d / ConvTranspose1d / Linear on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TFLOP/s peak).
//
// Replaces every F.conv1d / F.conv_transpose1d / F.linear the reference issues on the hot
// path: HiFi-GAN conv_pre / ups / ResBlock1 convs (vocoder/hifigan/models.py:46-53,
// 111-127), FastPitch conv-FF, qkv/o_net/proj, predictor convs
// (models/fastpitch/fastpitch/transformer.py:59-65,122,148; model.py:54-57,406).
//
// GEMM view per utterance b:  Y[co][q] = sum_{ci,tap} Wp[ci][tap][co] * X[ci][q + tap*dil - pad]
//   M = co (weights, A operand, read straight from L2 — all blocks share them),
//   N = q  (time, B operand, staged once per 16-channel chunk in LDS with the input
//           activation (leaky-relu) and the utterance-edge zero padding applied on load),
//   K = (ci, tap) walked two input channels per MFMA.
// Activations are channel-first [B][C][T] so a wave's 32 N-lanes read 32 consecutive time
// steps (coalesced HBM, conflict-free LDS).  A block is 4 waves tiled WM x WN, each wave
// owning MT x NTL 32x32 accumulators.  Ragged batches: positions >= lens_in[b] read as
// zero at the INPUT of every layer, tiles past lens_out[b] exit early.
#include "common.hpp"

namespace ttsamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DMAX = 5;  // largest |dilation| the LDS row is sized for

// Input channels staged per LDS chunk: KC*K ~ 48 (ci,tap) rows so that every chunk carries
// the same ~6k cycles of MFMA work per wave between two barriers.
#ifndef TTS_KC3
#define TTS_KC3 8
#endif
#ifndef TTS_KC7
#define TTS_KC7 4
#endif
#ifndef TTS_KC11
#define TTS_KC11 4
#endif
#ifndef TTS_MINWAVES
#define TTS_MINWAVES 3
#endif
template <int K> struct ChunkOf { static constexpr int KC = K >= 11 ? TTS_KC11 : (K >= 7 ? TTS_KC7 : (K >= 3 ? TTS_KC3 : 16)); };

template <int K, int NT_BLK, int CO_BLK>
struct Geo {
    static constexpr int KC = ChunkOf<K>::KC;
    static constexpr int WS = ((NT_BLK + (K - 1) * DMAX + 3) / 4) * 4;  // X row stride (floats)
    static constexpr int NJ = (NT_BLK + (K - 1) * DMAX + 63) / 64;      // 64-wide column steps
    static constexpr int XROWS = (KC + 3) / 4;                          // X rows per wave
    static constexpr int NX = XROWS * NJ;                               // staged X floats / thread
    static constexpr int W4 = KC * K * CO_BLK / 4;                      // float4s of the W chunk
    static constexpr int NW = (W4 + 255) / 256;                         // staged W float4s / thread
    static constexpr int X_FLOATS = KC * WS;
    static constexpr int W_FLOATS = KC * K * CO_BLK;
    static constexpr int BUF_FLOATS = X_FLOATS + W_FLOATS;              // one pipeline stage
};

template <int K, int MT, int NTL, int WM, int WN>
__global__ __launch_bounds__(256, TTS_MINWAVES) void conv1d_mfma_f32(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CO_BLK = WM * MT * 32;
    constexpr int NT_BLK = WN * NTL * 32;
    using G = Geo<K, NT_BLK, CO_BLK>;
    constexpr int KC = G::KC, WS = G::WS, NJ = G::NJ, NX = G::NX, NW = G::NW, XROWS = G::XROWS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int b = blockIdx.z;
    const int n_co_tiles = p.CoutP / CO_BLK;
    const int phase = blockIdx.y / n_co_tiles;
    const int co_blk0 = (blockIdx.y % n_co_tiles) * CO_BLK;
    const int q0 = blockIdx.x * NT_BLK;

    int n_out = p.Nout;
    if (p.lens_out) n_out = min(n_out, (int)p.lens_out[b] * p.len_out_mul);
    if (q0 >= n_out) return;
    int in_len = p.Lin;
    if (p.lens_in) in_len = min(in_len, (int)p.lens_in[b] * p.len_in_mul);

    const int dil = p.dil;
    int pad = p.pad;
    if (p.n_phase > 1) pad = -((phase + p.phase_p) / p.n_phase);
    const int span = (K - 1) * (dil < 0 ? -dil : dil);
    const int lo = (dil < 0 ? (K - 1) * dil : 0) - pad;  // first input position relative to q0
    const int W = NT_BLK + span;                          // staged row length (<= WS)

    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    const float* __restrict__ wp = p.w + (int64_t)phase * p.Cin * K * p.CoutP + co_blk0;
    const float in_slope = p.in_slope;
    const int n_chunks = p.Cin / KC;
    const int x_cs = p.x_cs, CoutP = p.CoutP;   // locals: the lambdas below must not capture `p`

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int qw0 = wn * NTL * 32;
    const bool wave_active = (q0 + qw0) < n_out;   // wave-uniform
    const int kk = lane >> 5, l31 = lane & 31;
    const int soff = qw0 + l31 - pad - lo;         // >= 0

    // ---- staging registers: X rows (wave w owns rows w, w+4, ..; lanes walk the row) and
    // the W chunk (a linear float4 copy).  All loads are unconditional (clamped address +
    // select) so they issue back to back and are only waited for at the end of the chunk.
    float sx[NX];
    float sw[4 * NW];
    bool st_ok[NJ];
    int st_pos[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = lane + 64 * j;
        const int pos = q0 + lo + col;
        st_ok[j] = (col < W) && (pos >= 0) && (pos < in_len);
        st_pos[j] = min(max(pos, 0), max(in_len - 1, 0));
    }
#define TTS_STAGE_LOAD(CH)                                                                          \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < XROWS; ++i) {                                         \
            const int r = min(wid + 4 * i, KC - 1);                                                 \
            const float* __restrict__ xc = xb + (int64_t)((CH)*KC + r) * x_cs;                      \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) sx[i * NJ + j] = xc[st_pos[j]];          \
        }                                                                                           \
        const float* __restrict__ wc = wp + (int64_t)(CH)*KC * K * CoutP;                           \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                            \
            const int e = min(tid + 256 * i, G::W4 - 1);                                            \
            const int row = e / (CO_BLK / 4), c4 = e % (CO_BLK / 4);                                \
            const float4 t4 = *reinterpret_cast<const float4*>(wc + (int64_t)row * CoutP + 4 * c4); \
            sw[4 * i + 0] = t4.x; sw[4 * i + 1] = t4.y; sw[4 * i + 2] = t4.z; sw[4 * i + 3] = t4.w; \
        }                                                                                           \
    }
#define TTS_STAGE_WRITE(BUF)                                                                        \
    {                                                                                               \
        float* sb = smem + (BUF)*G::BUF_FLOATS;                                                     \
        _Pragma("unroll") for (int i = 0; i < XROWS; ++i) {                                         \
            if (wid + 4 * i < KC) {                                                                 \
                _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                    \
                    const int col = lane + 64 * j;                                                  \
                    if (col < WS) {                                                                 \
                        float v = sx[i * NJ + j];                                                   \
                        v = v > 0.f ? v : v * in_slope;                                             \
                        sb[(wid + 4 * i) * WS + col] = st_ok[j] ? v : 0.f;                          \
                    }                                                                               \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
        float4* swp = reinterpret_cast<float4*>(sb + G::X_FLOATS);                                  \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) {                                            \
            const int e = tid + 256 * i;                                                            \
            if (e < G::W4) swp[e] = make_float4(sw[4 * i], sw[4 * i + 1], sw[4 * i + 2], sw[4 * i + 3]); \
        }                                                                                           \
    }

    // 3-stage LDS ring, one barrier per chunk, MFMA stream continuous across chunk boundaries:
    //   top of chunk c : issue the global loads of chunk c+2 (registers)
    //   steps 0..N-1   : MFMAs of chunk c from stage c%3; the operands of step s+1 are fetched
    //                    from LDS before the MFMAs of step s are issued (register ping-pong);
    //                    the LAST step fetches step 0 of chunk c+1 (stage (c+1)%3, made visible
    //                    by the previous barrier), so no wave ever drains at a chunk boundary
    //                    (waves of a SIMD run in lockstep: occupancy cannot hide such a bubble)
    //   before the last step: write chunk c+2 into stage (c+2)%3 (last read in chunk c-1)
    //   barrier        : covered by the 4 MFMAs just issued
    // Waves whose whole time range lies past n_out still run the MFMAs (results discarded):
    // keeping the MFMA block unconditional lets the accumulators stay in registers across chunks.
    constexpr int NSTEP = (KC / 2) * K;
    const float* sx0 = smem + kk * WS + soff;
    const float* sw0 = smem + G::X_FLOATS + kk * K * CO_BLK + wm * MT * 32 + l31;
    float a[2][MT], bq[2][NTL];

    TTS_STAGE_LOAD(0)
    TTS_STAGE_WRITE(0)
    if (n_chunks > 1) {
        TTS_STAGE_LOAD(1)
        TTS_STAGE_WRITE(1)
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) a[0][i] = sw0[i * 32];
#pragma unroll
    for (int j = 0; j < NTL; ++j) bq[0][j] = sx0[j * 32];

    int stage = 0;                                   // c % 3
    for (int c = 0; c < n_chunks; ++c) {
        const bool more2 = c + 2 < n_chunks;
#ifndef TTS_EXP_NOLOAD
        if (more2) TTS_STAGE_LOAD(c + 2)
#endif
        const float* sx_ = sx0 + stage * G::BUF_FLOATS;
        const float* sw_ = sw0 + stage * G::BUF_FLOATS;
        const int stage1 = stage == 2 ? 0 : stage + 1;   // (c+1) % 3
        const int stage2 = stage == 0 ? 2 : stage - 1;   // (c+2) % 3
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st + 1 < NSTEP) {
                const int pr = (st + 1) / K, tap = (st + 1) % K;
#if defined(TTS_EXP_B64)   /* timing-only experiment: wrong addresses */
                if constexpr (MT == 2 && NTL == 2) {
                    const float2 ta = *reinterpret_cast<const float2*>(sw_ + ((2 * pr) * K + tap) * CO_BLK + l31);
                    a[nxt][0] = ta.x; a[nxt][1] = ta.y;
#if TTS_EXP_B64 >= 2
                    const float2 tb = *reinterpret_cast<const float2*>(sx_ + (2 * pr) * WS + 2 * tap + l31);
                    bq[nxt][0] = tb.x; bq[nxt][1] = tb.y;
#else
                    for (int j = 0; j < NTL; ++j) bq[nxt][j] = sx_[(2 * pr) * WS + tap * dil + j * 32];
#endif
                } else
#endif
                {
#pragma unroll
                for (int i = 0; i < MT; ++i) a[nxt][i] = sw_[((2 * pr) * K + tap) * CO_BLK + i * 32];
#pragma unroll
                for (int j = 0; j < NTL; ++j) bq[nxt][j] = sx_[(2 * pr) * WS + tap * dil + j * 32];
                }
            } else {
#ifndef TTS_EXP_NOWRITE
                if (more2) TTS_STAGE_WRITE(stage2)
#endif
                // step 0 of the next chunk (clamped re-read of a valid stage on the last chunk)
                const int sn = (c + 1 < n_chunks) ? stage1 : stage;
#pragma unroll
                for (int i = 0; i < MT; ++i) a[nxt][i] = sw0[sn * G::BUF_FLOATS + i * 32];
#pragma unroll
                for (int j = 0; j < NTL; ++j) bq[nxt][j] = sx0[sn * G::BUF_FLOATS + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this step's MFMAs
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], bq[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NSTEP & 1) {                               // odd step count: operands of the next chunk sit in slot 1
#pragma unroll
            for (int i = 0; i < MT; ++i) a[0][i] = a[1][i];
#pragma unroll
            for (int j = 0; j < NTL; ++j) bq[0][j] = bq[1][j];
        }
#ifndef TTS_EXP_NOBARRIER
        __syncthreads();
#endif
        stage = stage1;
    }
#undef TTS_STAGE_LOAD
#undef TTS_STAGE_WRITE

    // epilogue: bias, residual, activation, accumulate modes.  Per 32x32 tile all loads
    // (bias, residual, previous y) are issued first and only then consumed, so a tile costs
    // one memory round trip instead of sixteen.
    if (!wave_active) return;
    const int co_w0 = co_blk0 + wm * MT * 32;
    float* __restrict__ yb = p.y + (int64_t)b * p.y_bs + phase;
    const float* __restrict__ rb = p.res ? p.res + (int64_t)b * p.r_bs + phase : nullptr;
    const float* __restrict__ bias = p.bias;
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
                float bv[8], rv[8], pv[8];
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = h * 8 + r8;
                    const int co = co_w0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    const int coc = min(co, Cout - 1);
                    bv[r8] = bias ? bias[coc] : 0.f;
                    rv[r8] = rb ? rb[(int64_t)coc * r_cs + (int64_t)qc * y_ts] : 0.f;
                    pv[r8] = mode != 0 ? yb[(int64_t)coc * y_cs + (int64_t)qc * y_ts] : 0.f;
                }
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = h * 8 + r8;
                    const int co = co_w0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    float v = acc[i][j][r] + bv[r8];
                    v += rv[r8];
                    if (relu_out) v = fmaxf(v, 0.f);
                    if (mode == 1) v = pv[r8] + v;
                    else if (mode == 2) v = (pv[r8] + v) / div;
                    if (q_ok && co < Cout) yb[(int64_t)co * y_cs + (int64_t)q * y_ts] = v;
                }
            }
        }
    }
}

template <int K, int MT, int NTL, int WM, int WN>
static int32_t launch_cfg(const ConvParams& p, hipStream_t stream) {
    constexpr int CO_BLK = WM * MT * 32, NT_BLK = WN * NTL * 32;
    using G = Geo<K, NT_BLK, CO_BLK>;
    TTS_REQUIRE(p.Cin % G::KC == 0, "conv: Cin=%d must be a multiple of %d for K=%d", p.Cin, G::KC, K);
    const size_t lds = (size_t)3 * G::BUF_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        TTS_CHECK_HIP(hipFuncSetAttribute((const void*)conv1d_mfma_f32<K, MT, NTL, WM, WN>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((p.Nout + NT_BLK - 1) / NT_BLK, (p.CoutP / CO_BLK) * p.n_phase, p.batch);
    hipLaunchKernelGGL((conv1d_mfma_f32<K, MT, NTL, WM, WN>), grid, dim3(256), lds, stream, p);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int K>
static int32_t launch_k(const ConvParams& p, hipStream_t stream) {
    const bool is_long = p.Nout > 96;
    if (p.CoutP % 128 == 0) {
        if (is_long) return launch_cfg<K, 2, 2, 2, 2>(p, stream);   // 128 co x 128 t
        return launch_cfg<K, 1, 2, 4, 1>(p, stream);                // 128 co x  64 t
    }
    if (p.CoutP % 64 == 0) {
        if (is_long) return launch_cfg<K, 2, 2, 1, 4>(p, stream);   //  64 co x 256 t
        return launch_cfg<K, 1, 1, 2, 2>(p, stream);                //  64 co x  64 t
    }
    if (is_long) return launch_cfg<K, 1, 2, 1, 4>(p, stream);       //  32 co x 256 t
    return launch_cfg<K, 1, 1, 1, 4>(p, stream);                    //  32 co x 128 t
}

int32_t launch_conv(const ConvParams& p, hipStream_t stream) {
    TTS_REQUIRE(p.CoutP % 32 == 0 && p.CoutP >= p.Cout, "conv: bad CoutP=%d", p.CoutP);
    TTS_REQUIRE(p.n_phase >= 1 && p.batch >= 1, "conv: bad n_phase/batch");
    TTS_REQUIRE(p.dil >= -DMAX && p.dil <= DMAX && p.dil != 0, "conv: dilation %d outside [-%d,%d]", p.dil, DMAX, DMAX);
    if (p.Nout <= 0) return 0;
    switch (p.K) {
        case 1: return launch_k<1>(p, stream);
        case 2: return launch_k<2>(p, stream);
        case 3: return launch_k<3>(p, stream);
        case 7: return launch_k<7>(p, stream);
        case 11: return launch_k<11>(p, stream);
        default:
            set_error("conv: kernel size %d not instantiated (1,2,3,7,11)", p.K);
            return TTSAMD_EINVAL;
    }
}

void pack_conv_weight(const float* w, int cout, int cin, int k, float* out) {
    const int cp = cout_padded(cout);
    for (int ci = 0; ci < cin; ++ci)
        for (int t = 0; t < k; ++t) {
            float* o = out + ((int64_t)ci * k + t) * cp;
            for (int co = 0; co < cp; ++co)
                o[co] = co < cout ? w[((int64_t)co * cin + ci) * k + t] : 0.f;
        }
}

// ConvTranspose1d(stride u, kernel Kt = 2u, padding p): y[co][q*u+rho] =
//   sum_ci W[ci][co][ka] x[ci][q+delta] + W[ci][co][ka+u] x[ci][q+delta-1],
//   ka = (rho+p) % u, delta = (rho+p) / u     (vocoder/hifigan/models.py:96-99).
void pack_convt_weight(const float* w, int cin, int cout, int kt, int u, int p, float* out) {
    const int cp = cout_padded(cout);
    for (int rho = 0; rho < u; ++rho) {
        const int ka = (rho + p) % u;
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 2; ++t) {
                const int kidx = ka + t * u;
                float* o = out + (((int64_t)rho * cin + ci) * 2 + t) * cp;
                for (int co = 0; co < cp; ++co)
                    o[co] = (co < cout && kidx < kt) ? w[((int64_t)ci * cout + co) * kt + kidx] : 0.f;
            }
    }
}

}  // namespace ttsamd
