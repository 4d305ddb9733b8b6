// HBM-/latency-bound kernels around the MFMA conv engine.  All activations are channel-first
// [B][C][T] so that consecutive lanes touch consecutive time steps (coalesced, 64-wide waves).
#include <cstdint>

#include <cstdlib>

#include "kernels.hpp"

namespace ttsamd {

// ------------------------------------------------------------------------------------
// HiFi-GAN tail (vocoder/hifigan/models.py:123-125): leaky_relu(x, 0.01) -> Conv1d(C->1,k7,p3)
// -> tanh.  Each thread produces 4 consecutive samples from a 10-wide register window.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, int64_t x_bs, int x_cs,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        const int64_t* __restrict__ lens, int len_mul, int C, int L,
                                                        float slope, float* __restrict__ wave, int64_t wave_bs) {
    const int b = blockIdx.y;
    int n = L;
    if (lens) n = min(n, (int)lens[b] * len_mul);
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t0 >= n) return;
    const float* xb = x + (int64_t)b * x_bs;
    float acc[4];
    const float b0 = bias ? bias[0] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = 0.f;
    // window = positions t0-4 .. t0+7 as three aligned float4 loads (t0 % 4 == 0); rows are 16-byte aligned when
    // the row stride is a multiple of 4 (always: L = 256 * frames), otherwise element loads
    const bool vec = (x_cs % 4 == 0) && (x_bs % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && t0 + 8 <= L;
    for (int c = 0; c < C; ++c) {
        const float* xr = xb + (int64_t)c * x_cs;
        float win[12];
        if (vec) {
            const float4 m = *reinterpret_cast<const float4*>(xr + t0);
            const float4 l = t0 >= 4 ? *reinterpret_cast<const float4*>(xr + t0 - 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 r = *reinterpret_cast<const float4*>(xr + t0 + 4);
            win[0] = l.x; win[1] = l.y; win[2] = l.z; win[3] = l.w;
            win[4] = m.x; win[5] = m.y; win[6] = m.z; win[7] = m.w;
            win[8] = r.x; win[9] = r.y; win[10] = r.z; win[11] = r.w;
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int pos = t0 - 4 + i;
                win[i] = (pos >= 0 && pos < L) ? xr[pos] : 0.f;
            }
        }
#pragma unroll
        for (int i = 1; i < 11; ++i) {
            const int pos = t0 - 4 + i;
            const float v = win[i];
            win[i] = (pos >= 0 && pos < n) ? (v > 0.f ? v : v * slope) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float wk = w[c * 7 + k];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(wk, win[i + k + 1], acc[i]);
        }
    }
    float* o = wave + (int64_t)b * wave_bs;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (t0 + i < n) o[t0 + i] = tanhf(acc[i] + b0);
}

int32_t launch_conv_post(const float* x, int64_t x_bs, int32_t x_cs, const float* w, const float* bias,
                         const int64_t* lens, int32_t len_mul, int32_t B, int32_t C, int32_t L, float in_slope,
                         float* wave, int64_t wave_bs, hipStream_t s) {
    if (L <= 0 || B <= 0) return 0;
    dim3 grid((L + 1023) / 1024, B);
    hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), 0, s, x, x_bs, x_cs, w, bias, lens, len_mul, C, L,
                       in_slope, wave, wave_bs);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Depthwise Conv1d k=7, p=3 (vocoder/vocos/modules.py:31): HBM-bound, one thread per (c, t).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv7_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      const int64_t* __restrict__ lens, int C, int S,
                                                      float* __restrict__ y) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= S) return;
    int n = S;
    if (lens) n = min(n, (int)lens[b]);
    const float* xr = x + ((int64_t)b * C + c) * S;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int pos = t + k - 3;
        const float v = (pos >= 0 && pos < n) ? xr[pos] : 0.f;
        acc = fmaf(w[c * 7 + k], v, acc);
    }
    y[((int64_t)b * C + c) * S + t] = acc + bias[c];
}

int32_t launch_dwconv7(const float* x, const float* w, const float* bias, const int64_t* lens, int32_t B, int32_t C,
                       int32_t S, float* y, hipStream_t s) {
    if (S <= 0 || B <= 0) return 0;
    dim3 grid((S + 255) / 256, C, B);
    hipLaunchKernelGGL(dwconv7_kernel, grid, dim3(256), 0, s, x, w, bias, lens, C, S, y);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// LayerNorm over channels of a channel-first tensor.  Block = 64 time steps x 4 channel
// groups; two-pass mean / variance (biased, eps 1e-5) like torch.nn.LayerNorm.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_cf_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const int64_t* __restrict__ lens, int apply_mask, int C,
                                                           int S, float eps) {
    __shared__ float red[4][64];
    const int b = blockIdx.y;
    const int tl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    const bool ok = t < S;
    const float* xb = x + (int64_t)b * C * S;
    float* yb = y + (int64_t)b * C * S;
    const int cpg = (C + 3) / 4;
    const int c_lo = g * cpg, c_hi = min(C, c_lo + cpg);
    float sum = 0.f;
    if (ok)
        for (int c = c_lo; c < c_hi; ++c) sum += xb[(int64_t)c * S + t];
    red[g][tl] = sum;
    __syncthreads();
    const float mean = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / (float)C;
    __syncthreads();
    float sq = 0.f;
    if (ok)
        for (int c = c_lo; c < c_hi; ++c) {
            const float d = xb[(int64_t)c * S + t] - mean;
            sq = fmaf(d, d, sq);
        }
    red[g][tl] = sq;
    __syncthreads();
    const float var = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (!ok) return;
    float m = 1.f;
    if (apply_mask && lens && t >= (int)lens[b]) m = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
        const float v = (xb[(int64_t)c * S + t] - mean) * rstd * gamma[c] + beta[c];
        yb[(int64_t)c * S + t] = v * m;
    }
}

// Single-read variant for C <= 512: block = 32 positions x 8 channel groups, the thread's <= 64 values stay
// in registers between the mean, variance and normalise steps (1 read + 1 write of the tensor, 128 B
// segments per wave-load); 4x more blocks than the kernel above, which matters at batch 1.
constexpr int LN_TP = 32, LN_G = 8, LN_MAXV = 64;
// NV_ = C / 8 as a compile-time constant (48: d_model 384, 32: the predictors' 256; C % 8 == 0, every thread owns exactly NV_
// channels), or 0 = run-time bound: then each of the 64 predicated loads sits behind its own exec-masked branch and vmcnt(0)
// (64 serial memory round trips: 12 us per launch at batch 1 against 5-6)
template <int NV_>
__global__ __launch_bounds__(256) void layernorm_cf_reg_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const int64_t* __restrict__ lens, int apply_mask,
                                                               int C, int S, float eps) {
    __shared__ float red[LN_G][LN_TP];
    const int b = blockIdx.y;
    const int tl = threadIdx.x & (LN_TP - 1), g = threadIdx.x / LN_TP;
    const int t = blockIdx.x * LN_TP + tl;
    const bool ok = t < S;
    const float* xb = x + (int64_t)b * C * S + (ok ? t : 0);
    constexpr int NV = NV_ ? NV_ : LN_MAXV;
    float v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = g + LN_G * i;
        v[i] = (NV_ || c < C) ? xb[(int64_t)c * S] : 0.f;
        sum += v[i];
    }
    red[g][tl] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < LN_G; ++k) tot += red[k][tl];
    const float mean = tot / (float)C;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float d = (NV_ || g + LN_G * i < C) ? v[i] - mean : 0.f;
        sq = fmaf(d, d, sq);
    }
    red[g][tl] = sq;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int k = 0; k < LN_G; ++k) tot += red[k][tl];
    const float rstd = 1.0f / sqrtf(tot / (float)C + eps);
    if (!ok) return;
    float m = 1.f;
    if (apply_mask && lens && t >= (int)lens[b]) m = 0.f;
    float* yb = y + (int64_t)b * C * S + t;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = g + LN_G * i;
        if (NV_ || c < C) yb[(int64_t)c * S] = ((v[i] - mean) * rstd * gamma[c] + beta[c]) * m;
    }
}

int32_t launch_layernorm_cf(const float* x, float* y, const float* gamma, const float* beta, const int64_t* lens,
                            int32_t apply_mask, int32_t B, int32_t C, int32_t S, hipStream_t s, float eps) {
    if (S <= 0 || B <= 0) return 0;
    if (C <= LN_G * LN_MAXV) {
        dim3 grid((S + LN_TP - 1) / LN_TP, B);
        if (C == 384) hipLaunchKernelGGL(layernorm_cf_reg_kernel<48>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, apply_mask, C, S, eps);
        else if (C == 256) hipLaunchKernelGGL(layernorm_cf_reg_kernel<32>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, apply_mask, C, S, eps);
        else if (C == 512) hipLaunchKernelGGL(layernorm_cf_reg_kernel<64>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, apply_mask, C, S, eps);
        else hipLaunchKernelGGL(layernorm_cf_reg_kernel<0>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, apply_mask, C, S, eps);
    } else {
        dim3 grid((S + 63) / 64, B);
        hipLaunchKernelGGL(layernorm_cf_kernel, grid, dim3(256), 0, s, x, y, gamma, beta, lens, apply_mask, C, S, eps);
    }
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Encoder embedding (transformer.py:212-219): word_emb gather + sinusoid table * mask + speaker.
// pos_table is channel-first [C][pos_stride].
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids,
                                                    const float* __restrict__ word_emb,
                                                    const float* __restrict__ pos, int pos_stride,
                                                    const float* __restrict__ spk, int pad_idx, int n_symbols, int L, int C,
                                                    float* __restrict__ x, int64_t* __restrict__ lens) {
    const int b = blockIdx.y;
    const int tl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    const int64_t* idb = ids + (int64_t)b * L;
    if (blockIdx.x == 0 && blockIdx.z == 0 && g == 0) {
        int cnt = 0;
        for (int i = tl; i < L; i += 64) cnt += (idb[i] != pad_idx) ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (tl == 0) lens[b] = cnt;
    }
    if (t >= L) return;
    const int64_t id = idb[t];
    const float m = (id != pad_idx) ? 1.f : 0.f;
    // out-of-range ids never leave the table (the host-side wrappers raise IndexError as nn.Embedding does)
    const float* er = word_emb + min(max(id, (int64_t)0), (int64_t)n_symbols - 1) * C;
    float* xb = x + (int64_t)b * C * L;
    for (int c = g + 4 * blockIdx.z; c < C; c += 4 * gridDim.z) {
        float v = er[c] + pos[(int64_t)c * pos_stride + t] * m;
        if (spk) v += spk[c];
        xb[(int64_t)c * L + t] = v;
    }
}

int32_t launch_embed(const int64_t* ids, const float* word_emb, const float* pos_table, int32_t pos_stride,
                     const float* spk, int32_t pad_idx, int32_t n_symbols, int32_t B, int32_t L, int32_t C, float* x, int64_t* lens, hipStream_t s) {
    dim3 grid((L + 63) / 64, B, 8);     // z: channel slices (the kernel is latency-bound at one block per utterance)
    hipLaunchKernelGGL(embed_kernel, grid, dim3(256), 0, s, ids, word_emb, pos_table, pos_stride, spk, pad_idx, n_symbols, L,
                       C, x, lens);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Single-head attention, fp32 VALU: 0.1 % of the path's FLOPs.
// Block = QT = 16*RA queries of one utterance; key/value tiles of 64 streamed through LDS, the next
// tile's 32 values per thread are already in flight (registers) while the current one is consumed.
// Thread (ti = tid/16, tj = tid%16) owns score rows i0=RA*ti.. and score cols / out dims 4*tj..
// Every 64-key tile yields a tile-local softmax (m_t, l_t, o_t) that is merged into the running (m, l, o)
// in tile order (att_merge): the result depends neither on RA (64 queries per block when that gives every
// CU a block, 16 otherwise) nor on WHO merges -- at small batches (batch 1: 29 blocks of 16 queries, each
// walking 8 key tiles in sequence: 34 us for 0.06 GFLOP) every (query block, key tile) is its own block
// (attention_tile_kernel) and attention_merge_kernel replays the same merges in the same order: the same
// bits as the one-kernel path (tests/test_gpu_parity.py), 34 -> 11 us per layer at batch 1.
// ------------------------------------------------------------------------------------
constexpr int ATT_D = 64;
constexpr int ATT_PS = 68;          // floats per (query, tile) partial: m, l, 2 pad, o[64]

// running (m, l, o[4]) <- merged with a tile's (mt, lt, ot[4]); explicit roundings: both kernels must agree bit for bit
__device__ __forceinline__ void att_merge(float& m, float& l, float (&o)[4], const float mt, const float lt, const float (&ot)[4]) {
    const float m_new = fmaxf(m, mt);
    const float ea = expf(m - m_new), eb = expf(mt - m_new);
    l = __fmaf_rn(l, ea, __fmul_rn(lt, eb));
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = __fmaf_rn(o[c], ea, __fmul_rn(ot[c], eb));
    m = m_new;
}

// one 64-key tile (already in Ks / Vs) against the block's queries: tile-local max, sum and weighted values
template <int RA, int QT>
__device__ __forceinline__ void att_tile(const float (&Qs)[ATT_D][QT + 4], const float (&Ks)[ATT_D][64 + 4], const float (&Vs)[ATT_D][64 + 1],
                                         float (&Ps)[QT][64 + 4], const int i0, const int j0, const int kt, const int len, const float scale,
                                         float (&mt)[RA], float (&lt)[RA], float (&ot)[RA][4]) {
    float sc[RA][4];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) sc[a][c] = 0.f;
#pragma unroll 8
    for (int d = 0; d < ATT_D; ++d) {
        float qa[RA];
        if constexpr (RA == 4) {
            const float4 qv = *reinterpret_cast<const float4*>(&Qs[d][i0]);
            qa[0] = qv.x; qa[1] = qv.y; qa[2] = qv.z; qa[3] = qv.w;
        } else {
#pragma unroll
            for (int a = 0; a < RA; ++a) qa[a] = Qs[d][i0 + a];
        }
        const float4 kv = *reinterpret_cast<const float4*>(&Ks[d][j0]);
        const float ka[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) sc[a][c] = fmaf(qa[a], ka[c], sc[a][c]);
    }
#pragma unroll
    for (int a = 0; a < RA; ++a) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sc[a][c] = (kt + j0 + c < len) ? sc[a][c] * scale : -INFINITY;
            mx = fmaxf(mx, sc[a][c]);
        }
        for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float rs = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float pv = expf(sc[a][c] - mx);
            sc[a][c] = pv;
            rs += pv;
        }
        for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off);
        mt[a] = mx;
        lt[a] = rs;
        *reinterpret_cast<float4*>(&Ps[i0 + a][j0]) = make_float4(sc[a][0], sc[a][1], sc[a][2], sc[a][3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) ot[a][c] = 0.f;
    }
    __syncthreads();
    // O_t[i0+a][d0+c] = sum_j P[i0+a][j] * V[d0+c][j], d0 = j0
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        float pa[RA], va[4];
#pragma unroll
        for (int a = 0; a < RA; ++a) pa[a] = Ps[i0 + a][j];
#pragma unroll
        for (int c = 0; c < 4; ++c) va[c] = Vs[j0 + c][j];
#pragma unroll
        for (int a = 0; a < RA; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) ot[a][c] = fmaf(pa[a], va[c], ot[a][c]);
    }
}

template <int RA>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv,
                                                        const int64_t* __restrict__ lens, int S, float scale,
                                                        float* __restrict__ out) {
    constexpr int QT = 16 * RA;
    __shared__ float Qs[ATT_D][QT + 4];
    __shared__ float Ks[ATT_D][64 + 4];
    __shared__ float Vs[ATT_D][64 + 1];
    __shared__ float Ps[QT][64 + 4];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int i0 = ti * RA, j0 = tj * 4;
    const int qbase = blockIdx.x * QT;
    int len = S;
    if (lens) len = min(S, (int)lens[b]);
    const float* qb = qkv + (int64_t)b * 3 * ATT_D * S;
    const float* kb = qb + (int64_t)ATT_D * S;
    const float* vb = kb + (int64_t)ATT_D * S;
    if (qbase >= len) {
        // a block of queries past the utterance's end (ragged batch: the longest of config 1's lines is 1.55x the mean of its group):
        // their rows are masked by the LayerNorm behind o_net (a k = 1 conv: no neighbour reads them), so zeros instead of a walk over
        // all key tiles -- finite, because that LayerNorm masks by multiplication
        float* ob = out + (int64_t)b * ATT_D * S;
        for (int e = tid; e < ATT_D * QT; e += 256) {
            const int d = e / QT, i = qbase + e % QT;
            if (i < S) ob[(int64_t)d * S + i] = 0.f;
        }
        return;
    }

    for (int e = tid; e < ATT_D * QT; e += 256) {
        const int d = e / QT, i = e % QT;
        Qs[d][i] = (qbase + i < S) ? qb[(int64_t)d * S + qbase + i] : 0.f;
    }
    float m_run[RA], l_run[RA], o[RA][4];
#pragma unroll
    for (int a = 0; a < RA; ++a) {
        m_run[a] = -INFINITY;
        l_run[a] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[a][c] = 0.f;
    }
    // element e = tid + 256*r of a [64 d][64 j] tile: d = 4*r + tid/64, j = tid%64
    const int lj = tid & 63, ld = tid >> 6;
    float kreg[16], vreg[16];
    auto fetch = [&](int kt) {
        const bool ok = kt + lj < len;
        const int64_t off = (int64_t)ld * S + kt + lj;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            kreg[r] = ok ? kb[off + (int64_t)(4 * r) * S] : 0.f;
            vreg[r] = ok ? vb[off + (int64_t)(4 * r) * S] : 0.f;
        }
    };
    if (len > 0) fetch(0);
    for (int kt = 0; kt < len; kt += 64) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Ks[4 * r + ld][lj] = kreg[r];
            Vs[4 * r + ld][lj] = vreg[r];
        }
        __syncthreads();
        if (kt + 64 < len) fetch(kt + 64);
        float mt[RA], lt[RA], ot[RA][4];
        att_tile<RA, QT>(Qs, Ks, Vs, Ps, i0, j0, kt, len, scale, mt, lt, ot);
#pragma unroll
        for (int a = 0; a < RA; ++a) att_merge(m_run[a], l_run[a], o[a], mt[a], lt[a], ot[a]);
    }
    float* ob = out + (int64_t)b * ATT_D * S;
#pragma unroll
    for (int a = 0; a < RA; ++a) {
        const int i = qbase + i0 + a;
        if (i >= S) continue;
        const float inv = 1.0f / l_run[a];
#pragma unroll
        for (int c = 0; c < 4; ++c) ob[(int64_t)(j0 + c) * S + i] = o[a][c] * inv;
    }
}

// Small batches: block (x, y, z) = 16 queries of utterance y against key tile z; its (m_t, l_t, o_t) go to
// part[((y * gridDim.x + x) * gridDim.z + z) * 16 + query][ATT_PS]
__global__ __launch_bounds__(256) void attention_tile_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ lens, int S,
                                                             float scale, float* __restrict__ part) {
    constexpr int QT = 16;
    __shared__ float Qs[ATT_D][QT + 4];
    __shared__ float Ks[ATT_D][64 + 4];
    __shared__ float Vs[ATT_D][64 + 1];
    __shared__ float Ps[QT][64 + 4];
    const int b = blockIdx.y, kt = blockIdx.z * 64;
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int j0 = tj * 4;
    const int qbase = blockIdx.x * QT;
    int len = S;
    if (lens) len = min(S, (int)lens[b]);
    if (kt >= len) return;
    const float* qb = qkv + (int64_t)b * 3 * ATT_D * S;
    const float* kb = qb + (int64_t)ATT_D * S;
    const float* vb = kb + (int64_t)ATT_D * S;
    const int lj = tid & 63, ld = tid >> 6;
    {
        const bool ok = kt + lj < len;
        const int64_t off = (int64_t)ld * S + kt + lj;
        float kreg[16], vreg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            kreg[r] = ok ? kb[off + (int64_t)(4 * r) * S] : 0.f;
            vreg[r] = ok ? vb[off + (int64_t)(4 * r) * S] : 0.f;
        }
        for (int e = tid; e < ATT_D * QT; e += 256) {
            const int d = e / QT, i = e % QT;
            Qs[d][i] = (qbase + i < S) ? qb[(int64_t)d * S + qbase + i] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Ks[4 * r + ld][lj] = kreg[r];
            Vs[4 * r + ld][lj] = vreg[r];
        }
    }
    __syncthreads();
    float mt[1], lt[1], ot[1][4];
    att_tile<1, QT>(Qs, Ks, Vs, Ps, ti, j0, kt, len, scale, mt, lt, ot);
    float* pp = part + ((((int64_t)b * gridDim.x + blockIdx.x) * gridDim.z + blockIdx.z) * 16 + ti) * ATT_PS;
    if (tj == 0) { pp[0] = mt[0]; pp[1] = lt[0]; }
    *reinterpret_cast<float4*>(pp + 4 + j0) = make_float4(ot[0][0], ot[0][1], ot[0][2], ot[0][3]);
}

// ... and the merges of attention_kernel's tile loop, in its order, for 16 queries per block
__global__ __launch_bounds__(256) void attention_merge_kernel(const float* __restrict__ part, const int64_t* __restrict__ lens, int S,
                                                              int n_tiles_max, float* __restrict__ out) {
    const int b = blockIdx.y, tid = threadIdx.x, ti = tid >> 4, j0 = (tid & 15) * 4;
    int len = S;
    if (lens) len = min(S, (int)lens[b]);
    float m = -INFINITY, l = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
    const float* pp = part + ((((int64_t)b * gridDim.x + blockIdx.x) * n_tiles_max) * 16 + ti) * ATT_PS;
    for (int kt = 0; kt < len; kt += 64, pp += 16 * ATT_PS) {
        const float4 ov = *reinterpret_cast<const float4*>(pp + 4 + j0);
        const float ot[4] = {ov.x, ov.y, ov.z, ov.w};
        att_merge(m, l, o, pp[0], pp[1], ot);
    }
    const int i = blockIdx.x * 16 + ti;
    if (i >= S) return;
    const float inv = 1.0f / l;
    float* ob = out + (int64_t)b * ATT_D * S;
#pragma unroll
    for (int c = 0; c < 4; ++c) ob[(int64_t)(j0 + c) * S + i] = o[c] * inv;
}

int32_t launch_attention(const float* qkv, const int64_t* lens, int32_t B, int32_t D, int32_t S, float scale,
                         float* out, hipStream_t s, float* ws, int64_t ws_floats) {
    TTS_REQUIRE(D == ATT_D, "attention: d_head=%d, only %d is built", D, ATT_D);
    if (S <= 0 || B <= 0) return 0;
    if (default_precision() == 1) {                              // config 3: bf16 MFMA attention (TTSAMD_BF16_ATTN=0: the fp32 kernel)
        const char* e = opt_str(OPT_BF16_ATTN);
        if (!(e && e[0] == '0')) return launch_attention_bf16(qkv, lens, B, D, S, scale, out, s);
    }
    // 64 queries per block once that gives every CU a block (batch 32 x 450 frames: 78.14 -> 77.95 ms per step), 16 otherwise
    // (batch 1 / 8: 4.60 / 21.94 ms against 4.80 / 22.14); TTSAMD_ATT_RA=1/2/4 forces the tile
    const char* rae = opt_str(OPT_ATT_RA);
    const int ra = rae ? atoi(rae) : ((int64_t)((S + 63) / 64) * B >= 256 ? 4 : 1);
    // fewer than half a block per CU and several key tiles: one block per (16 queries, key tile) + the merge launch (same bits);
    // TTSAMD_ATT_SPLIT=0/1 forces either
    const int qb = (S + 15) / 16, nt = (S + 63) / 64;
    const char* spe = opt_str(OPT_ATT_SPLIT);
    const bool fits = ws != nullptr && (int64_t)B * qb * nt * 16 * ATT_PS <= ws_floats;
    const bool split = fits && nt >= 2 && (spe ? spe[0] == '1' : (ra == 1 && (int64_t)qb * B < 128));
    if (split) {
        hipLaunchKernelGGL(attention_tile_kernel, dim3(qb, B, nt), dim3(256), 0, s, qkv, lens, S, scale, ws);
        TTS_CHECK_HIP(hipGetLastError());
        hipLaunchKernelGGL(attention_merge_kernel, dim3(qb, B), dim3(256), 0, s, (const float*)ws, lens, S, nt, out);
    } else if (ra == 4) {
        dim3 grid((S + 63) / 64, B);
        hipLaunchKernelGGL(attention_kernel<4>, grid, dim3(256), 0, s, qkv, lens, S, scale, out);
    } else if (ra == 2) {
        dim3 grid((S + 31) / 32, B);
        hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(256), 0, s, qkv, lens, S, scale, out);
    } else {
        dim3 grid((S + 15) / 16, B);
        hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(256), 0, s, qkv, lens, S, scale, out);
    }
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Predictor head (model.py:132) + duration transform (model.py:368) / pitch_trf (networks.py:38-42)
// ------------------------------------------------------------------------------------
// Block = 16 positions x 16 channel groups (group g sums c = g, g+16, ...), LDS reduction in group order.
__global__ __launch_bounds__(256) void pred_fc_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      const int64_t* __restrict__ lens, int C, int S,
                                                      float* __restrict__ out, float* __restrict__ out2,
                                                      float max_dur, float mul, float add) {
    __shared__ float part[16][17];
    const int b = blockIdx.y;
    const int tl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + tl;
    const float* xb = x + (int64_t)b * C * S + (t < S ? t : 0);
    float acc = 0.f;
    for (int c = g; c < C; c += 16) acc = fmaf(w[c], xb[(int64_t)c * S], acc);
    part[g][tl] = acc;
    __syncthreads();
    if (g != 0 || t >= S) return;
    acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += part[k][tl];
    acc += bias[0];
    if (lens && t >= (int)lens[b]) acc = 0.f;   // "* enc_out_mask"
    if (out2) out2[(int64_t)b * S + t] = fminf(fmaxf(expf(acc) - 1.0f, 0.f), max_dur);
    out[(int64_t)b * S + t] = mul * acc + add;
}

int32_t launch_pred_fc(const float* x, const float* w, const float* bias, const int64_t* lens, int32_t B,
                       int32_t C, int32_t S, float* out, float* out2, float max_dur, float mul, float add,
                       hipStream_t s) {
    dim3 grid((S + 15) / 16, B);
    hipLaunchKernelGGL(pred_fc_kernel, grid, dim3(256), 0, s, x, w, bias, lens, C, S, out, out2, max_dur, mul,
                       add);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// enc[b][c][t] += bias[c] + sum_k w[c][k] * src[b][t + k - K/2]    (zero padded)
__global__ __launch_bounds__(256) void scalar_emb_add_kernel(float* __restrict__ enc,
                                                             const float* __restrict__ src,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int C, int S, int K) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= S) return;
    const float* sb = src + (int64_t)b * S;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int pos = t + k - (K - 1) / 2;
        const float v = (pos >= 0 && pos < S) ? sb[pos] : 0.f;
        acc = fmaf(w[c * K + k], v, acc);
    }
    acc += bias[c];
    enc[((int64_t)b * C + c) * S + t] += acc;
}

int32_t launch_scalar_emb_add(float* enc, const float* src, const float* w, const float* bias, int32_t B,
                              int32_t C, int32_t S, int32_t K, hipStream_t s) {
    dim3 grid((S + 255) / 256, C, B);
    hipLaunchKernelGGL(scalar_emb_add_kernel, grid, dim3(256), 0, s, enc, src, w, bias, C, S, K);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Length regulator, integer half (model.py:72-76).  One wave64 per utterance.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void durations_to_reps_kernel(const float* __restrict__ dur, float pace, int L,
                                                               int64_t* __restrict__ reps,
                                                               int64_t* __restrict__ dec_lens) {
    const int b = blockIdx.x, lane = threadIdx.x;
    long long tot = 0;
    for (int i = lane; i < L; i += 64) {
        const float r = dur[(int64_t)b * L + i] / pace + 0.5f;   // fp32, as durations.float()/pace + 0.5
        const long long n = max((long long)r, 0ll);              // .long(): truncation toward zero; negative dur_tgt -> 0 repeats
                                                                 // (the reference's cumsum/one-hot would misbehave; the scan needs >= 0)
        reps[(int64_t)b * L + i] = n;
        tot += n;
    }
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0) dec_lens[b] = tot;
}

int32_t launch_durations_to_reps(const float* dur, float pace, int32_t B, int32_t L, int64_t* reps,
                                 int64_t* dec_lens, hipStream_t s) {
    hipLaunchKernelGGL(durations_to_reps_kernel, dim3(B), dim3(64), 0, s, dur, pace, L, reps, dec_lens);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Length regulator, gather half (model.py:77-85): wave64 inclusive scan of the repeat counts
// (64 tokens per step, carry across steps), binary search of each frame in the prefix sums,
// then a coalesced row gather.  Equivalent to the reference's one-hot matmul bit for bit
// (each output is 1.0*x + 0.0*...).
// ------------------------------------------------------------------------------------
constexpr int REG_MAX_L = 4096;
__global__ __launch_bounds__(256) void regulate_gather_kernel(const float* __restrict__ enc,
                                                              const int64_t* __restrict__ reps,
                                                              const float* __restrict__ pos, int pos_stride, int L,
                                                              int C, int T, float* __restrict__ out,
                                                              int32_t* __restrict__ idx) {
    __shared__ int cs[REG_MAX_L + 1];   // exclusive prefix sums, cs[L] = dec_len
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < 64) {
        int carry = 0;
        if (tid == 0) cs[0] = 0;
        for (int base = 0; base < L; base += 64) {
            const int i = base + tid;
            int v = (i < L) ? (int)reps[(int64_t)b * L + i] : 0;
            for (int o = 1; o < 64; o <<= 1) {                 // wave64 inclusive scan
                const int n = __shfl_up(v, o);
                if (tid >= o) v += n;
            }
            if (i < L) cs[i + 1] = carry + v;
            carry += __shfl(v, 63);
        }
    }
    __syncthreads();
    const int dec_len = cs[L];
    const int t = blockIdx.x * 256 + tid;
    if (t >= T) return;
    int j = -1;
    if (t < dec_len) {
        int lo = 0, hi = L;                // largest j with cs[j] <= t  (cs non-decreasing)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cs[mid] <= t) lo = mid; else hi = mid;
        }
        j = lo;
    }
    if (idx && blockIdx.z == 0) idx[(int64_t)b * T + t] = j;
    const float* eb = enc + (int64_t)b * C * L;
    float* ob = out + (int64_t)b * C * T;
    for (int c = blockIdx.z; c < C; c += gridDim.z) {
        float v = 0.f;
        if (j >= 0) {
            v = eb[(int64_t)c * L + j];
            if (pos) v += pos[(int64_t)c * pos_stride + t];
        }
        ob[(int64_t)c * T + t] = v;
    }
}

int32_t launch_regulate_gather(const float* enc, const int64_t* reps, const float* pos_table, int32_t pos_stride,
                               int32_t B, int32_t L, int32_t C, int32_t T, float* out, int32_t* idx, hipStream_t s) {
    TTS_REQUIRE(L <= REG_MAX_L, "length_regulate: n_tokens=%d exceeds %d", L, REG_MAX_L);
    if (T <= 0 || B <= 0) return 0;
    dim3 grid((T + 255) / 256, B, 16);  // z: channel slices; every slice redoes the (cheap) scan and search
    hipLaunchKernelGGL(regulate_gather_kernel, grid, dim3(256), 0, s, enc, reps, pos_table, pos_stride, L, C,
                       T, out, idx);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos,
                                                      int pos_stride, const int64_t* __restrict__ lens, int C,
                                                      int S) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= S) return;
    if (lens && t >= (int)lens[b]) return;
    x[((int64_t)b * C + c) * S + t] += pos[(int64_t)c * pos_stride + t];
}

int32_t launch_add_pos(float* x, const float* pos_table, int32_t pos_stride, const int64_t* lens, int32_t B, int32_t C, int32_t S,
                       hipStream_t s) {
    dim3 grid((S + 255) / 256, C, B);
    hipLaunchKernelGGL(add_pos_kernel, grid, dim3(256), 0, s, x, pos_table, pos_stride, lens, C, S);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// out[b] = min(lens[b] + 1, W): how far FastPitch's first conv-FF conv has to compute a row (fastpitch.hip: run_fft).  W = the width of the
// reference's padded batch: S (the encoder: the caller's ids tensor IS that batch) or, with clamp_at_max (the decoder: its rows may be wider
// than the longest utterance -- padded to 16 bytes by the caller), the longest row: a frame past it does not exist in the reference's
// arithmetic -- it is zero padding, not a hidden activation
__global__ void lens_plus1_kernel(const int64_t* __restrict__ lens, int S, int B, int clamp_at_max, int64_t* __restrict__ out) {
    __shared__ long long red[256];
    long long mx = clamp_at_max ? 0 : (long long)S;
    for (int b = threadIdx.x; b < B; b += 256) mx = max(mx, (long long)lens[b]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const long long cap = min((long long)S, red[0]);
    for (int b = threadIdx.x; b < B; b += 256) out[b] = min((long long)lens[b] + 1, cap);
}

int32_t launch_lens_plus1(const int64_t* lens, int32_t S, int32_t B, int32_t clamp_at_max, int64_t* out, hipStream_t s) {
    hipLaunchKernelGGL(lens_plus1_kernel, dim3(1), dim3(256), 0, s, lens, S, B, clamp_at_max, out);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ttsamd
