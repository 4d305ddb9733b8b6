// 1-head self attention of the FastPitch FFT blocks on the bf16 matrix cores (config 3):
//     out[b][d][i] = sum_j softmax_j(q_i . k_j * scale | j < lens[b]) v_j[d]      (models/fastpitch/fastpitch/transformer.py:131-141)
// q, k, v = rows [0,64), [64,128), [128,192) of the channel-first fp32 qkv tensor [B][192][S]; out [B][64][S] fp32.
//
// Block = one 32-query tile; its 4 waves split the KEYS (flash-decoding style) and meet once at the end, so a decoder layer at
// batch 32 x 496 frames is 512 blocks and nothing is staged through LDS:
//   S^T[key][query] = K^T Q   v_mfma_f32_32x32x16_bf16, A = 8 consecutive d of one key (8 coalesced 4-byte loads, rounded to
//                             bf16), B = the wave's Q tile, loaded once (16 VGPRs);
//   softmax                   in the C layout: a lane holds 16 of the 32 keys of ITS query, the other 16 sit in lane ^ 32 (one
//                             shuffle for the max, one for the sum); running max / sum per query = per lane;
//   O[d][query] += V P        A = 8 consecutive keys of one d (32 contiguous bytes), B = P: lane (query, kk) needs keys
//                             16 j + 8 kk + 0..7 -- four of them are its own accumulator registers, four are its partner's:
//                             one select + one shuffle per value;
//   combine                   (m, l, O) of the four key ranges through LDS, out = sum_w e^(m_w - M) O_w / sum_w e^(m_w - M) l_w.
#include "bfo.hpp"
#include "kernels.hpp"

namespace ttsamd {

constexpr int ATB_D = 64;
constexpr float ATB_NEG = -1.0e30f;      // "minus infinity" that survives exp() and subtraction without NaN

__global__ __launch_bounds__(256) void attention_bf16_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ lens, int S,
                                                             float scale, float* __restrict__ out, uint4* __restrict__ out_o) {
    __shared__ float Ml[4][2][32];                    // [wave][m | l][query]
    __shared__ float Os[4][ATB_D][33];                // [wave][d][query]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, q0 = blockIdx.x * 32;
    int len = S;
    if (lens) len = min(S, (int)lens[b]);
    const bfo_i4 rs = bfo_rsrc(qkv + (int64_t)b * 3 * ATB_D * S, (unsigned)3 * ATB_D * S * 4);
    const int rowb = S * 4;                           // bytes per channel row

    // Q tile as the B operand of the first GEMM: lane (query, kk), k-step h: d = 16 h + 8 kk + e
    bfo_i4 qf[4];
    {
        const int qv = (q0 + l31 < S) ? (q0 + l31) * 4 : BFO_OOB;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bfo_ld4f(rs, qv, (16 * h + 8 * kk + e) * rowb, 0) * scale;
            qf[h].x = bfo_pk(v[0], v[1]); qf[h].y = bfo_pk(v[2], v[3]); qf[h].z = bfo_pk(v[4], v[5]); qf[h].w = bfo_pk(v[6], v[7]);
        }
    }
    bfo_f16 o0, o1;                                   // O rows d = 0..31 / 32..63 of this wave's key range
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = ATB_NEG, l_run = 0.f;

    const int n_tiles = (len + 31) / 32;
    const int t_beg = wid * n_tiles / 4, t_end = (wid + 1) * n_tiles / 4;
    for (int t = t_beg; t < t_end; ++t) {
        const int k0 = t * 32;
        // ---- S^T tile: rows = keys k0 .. k0 + 31, columns = queries
        bfo_f16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        {
            const int kv = (k0 + l31) * 4;            // keys past S read 0 through the buffer bounds (masked below anyway)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = bfo_ld4f(rs, kv, (ATB_D + 16 * h + 8 * kk + e) * rowb, 0);
                bfo_i4 a;
                a.x = bfo_pk(v[0], v[1]); a.y = bfo_pk(v[2], v[3]); a.z = bfo_pk(v[4], v[5]); a.w = bfo_pk(v[6], v[7]);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfo_h8, a), __builtin_bit_cast(bfo_h8, qf[h]), sc, 0, 0, 0);
            }
        }
        // V operands of the second GEMM (issued early: they do not depend on the softmax)
        bfo_i4 vf[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int so = (2 * ATB_D + 32 * rt) * rowb + (k0 + 16 * j) * 4;              // scalar part of the address
                const int vo = l31 * rowb + 8 * kk * 4;                                       // row d = 32 rt + l31, keys + 8 kk
                const bfo_i4 lo = bfo_ld16(rs, vo, so, 0), hi = bfo_ld16(rs, vo, so + 16, 0);
                vf[rt][j].x = bfo_pk(__int_as_float(lo.x), __int_as_float(lo.y)); vf[rt][j].y = bfo_pk(__int_as_float(lo.z), __int_as_float(lo.w));
                vf[rt][j].z = bfo_pk(__int_as_float(hi.x), __int_as_float(hi.y)); vf[rt][j].w = bfo_pk(__int_as_float(hi.z), __int_as_float(hi.w));
            }
        // ---- online softmax over the keys (rows); register r = key k0 + 8 (r >> 2) + 4 kk + (r & 3)
        float mx = ATB_NEG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + 8 * (r >> 2) + 4 * kk + (r & 3);
            sc[r] = key < len ? sc[r] : ATB_NEG;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __expf(sc[r] - m_new);             // masked keys: exp(-1e30 - m) = 0
            ps += sc[r];
        }
        ps += __shfl_xor(ps, 32);
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        // ---- P as the B operand: lane (query, kk), k-step j needs keys 16 j + 8 kk + 0..7 = row group g = 2 j + kk:
        // four values are its own registers 4 g + e (keys 8 g + 4 kk + e), four are the partner's
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float own[4], got[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float mine_lo = sc[4 * (2 * j) + e], mine_hi = sc[4 * (2 * j + 1) + e];
                own[e] = kk ? mine_hi : mine_lo;      // the group this lane consumes
                got[e] = __shfl_xor(kk ? mine_lo : mine_hi, 32);
            }
            // lane kk = 0: keys 8 g + 0..3 are its own, 8 g + 4..7 the partner's; lane kk = 1: the other way round
            bfo_i4 pf;
            pf.x = kk ? bfo_pk(got[0], got[1]) : bfo_pk(own[0], own[1]);
            pf.y = kk ? bfo_pk(got[2], got[3]) : bfo_pk(own[2], own[3]);
            pf.z = kk ? bfo_pk(own[0], own[1]) : bfo_pk(got[0], got[1]);
            pf.w = kk ? bfo_pk(own[2], own[3]) : bfo_pk(got[2], got[3]);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfo_h8, vf[0][j]), __builtin_bit_cast(bfo_h8, pf), o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfo_h8, vf[1][j]), __builtin_bit_cast(bfo_h8, pf), o1, 0, 0, 0);
        }
    }
    // ---- combine the four key ranges
    if (kk == 0) { Ml[wid][0][l31] = m_run; Ml[wid][1][l31] = l_run; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = 8 * (r >> 2) + 4 * kk + (r & 3);
        Os[wid][d][l31] = o0[r];
        Os[wid][32 + d][l31] = o1[r];
    }
    __syncthreads();
    if (out_o) {
        // octet bf16 output [B][8][S][8] for the o_net conv on the octet engine: thread = (octet, query) = one 16-byte entry
        const int o = tid >> 5, q = tid & 31;
        const float m0 = Ml[0][0][q], m1 = Ml[1][0][q], m2 = Ml[2][0][q], m3 = Ml[3][0][q];
        const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float e0 = __expf(m0 - M), e1 = __expf(m1 - M), e2 = __expf(m2 - M), e3 = __expf(m3 - M);
        const float inv = 1.f / (e0 * Ml[0][1][q] + e1 * Ml[1][1][q] + e2 * Ml[2][1][q] + e3 * Ml[3][1][q]);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = 8 * o + e;
            v[e] = (e0 * Os[0][d][q] + e1 * Os[1][d][q] + e2 * Os[2][d][q] + e3 * Os[3][d][q]) * inv;
        }
        uint4 w;
        w.x = (unsigned)bfo_pk(v[0], v[1]); w.y = (unsigned)bfo_pk(v[2], v[3]); w.z = (unsigned)bfo_pk(v[4], v[5]); w.w = (unsigned)bfo_pk(v[6], v[7]);
        if (q0 + q < S) out_o[((int64_t)b * (ATB_D / 8) + o) * S + q0 + q] = w;
        return;
    }
    float* ob = out + (int64_t)b * ATB_D * S;
#pragma unroll
    for (int i = 0; i < ATB_D * 32 / 256; ++i) {
        const int idx = tid + 256 * i;
        const int d = idx >> 5, q = idx & 31;
        const float m0 = Ml[0][0][q], m1 = Ml[1][0][q], m2 = Ml[2][0][q], m3 = Ml[3][0][q];
        const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        const float e0 = __expf(m0 - M), e1 = __expf(m1 - M), e2 = __expf(m2 - M), e3 = __expf(m3 - M);
        const float den = e0 * Ml[0][1][q] + e1 * Ml[1][1][q] + e2 * Ml[2][1][q] + e3 * Ml[3][1][q];
        const float num = e0 * Os[0][d][q] + e1 * Os[1][d][q] + e2 * Os[2][d][q] + e3 * Os[3][d][q];
        if (q0 + q < S) ob[(int64_t)d * S + q0 + q] = num / den;
    }
}

int32_t launch_attention_bf16(const float* qkv, const int64_t* lens, int32_t B, int32_t D, int32_t S, float scale, float* out,
                              hipStream_t s, void* out_octet) {
    TTS_REQUIRE(D == ATB_D, "attention (bf16): d_head=%d, only %d is built", D, ATB_D);
    TTS_REQUIRE((int64_t)3 * ATB_D * S * 4 < ((int64_t)1 << 31), "attention (bf16): sequence too long for 32-bit offsets");
    if (S <= 0 || B <= 0) return 0;
    dim3 grid((S + 31) / 32, B);
    hipLaunchKernelGGL(attention_bf16_kernel, grid, dim3(256), 0, s, qkv, lens, S, scale, out, (uint4*)out_octet);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ttsamd
