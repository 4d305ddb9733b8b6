// Split-bf16 ("x3") mode of the octet engine (bfo.hpp): fp32-class results on v_mfma_f32_32x32x16_bf16.
//
// Every activation and weight is the sum of two bf16 numbers, v = hi + lo (hi = bf16(v), lo = bf16(v - hi): 16 mantissa
// bits), and a product is three MFMAs, Wh xh + Wh xl + Wl xh, accumulated in fp32 (the dropped Wl xl term is 2^-16 of the
// product).  Reference ops and tolerance: vocoder/hifigan/models.py:46-53,96-99,111-127; BASELINE.json north_star (mel 1e-3,
// wave 1e-4 max-abs) -- plain bf16 (bfo.hpp) is 30x outside that, this mode 10-20x inside.
//
// Activation layout in HBM ("x3 tensor"):  [B][C/8][L][2 (kk)][hi 4 bf16 | lo 4 bf16]  -- 32 bytes per (octet, position).
// The 16-byte half kk holds the channels 8o + 4kk + {0..3}: exactly what lane (column, kk) of the 32x32 MFMA C layout
// owns in registers 4g..4g+3 of octet g, so an epilogue store and a residual load are ONE 16-byte access per lane and
// 1 KB contiguous per wave instruction (the plain bf16 engine moves 8 bytes per lane).  In LDS the window is kept as TWO
// planes of plain octet entries (hi plane, lo plane: [C/8][WS][8 bf16] each), the B operand of the MFMA: staging writes the
// 8-byte halves of a 16-byte piece to the two planes, an operand fetch stays one ds_read_b128 per plane -- two reads per
// three MFMAs.  Like the bf16 engine the tensors are stored PRE-ACTIVATED for their consumer (bfo.hpp).
// Weights: [Cin/16][K][2 (kk)][CoutP][hi 8 bf16 | lo 8 bf16] = the two A operands of a lane side by side (32 bytes), streamed
// from L2 into a register ring; no LDS, no barrier inside a conv.
#pragma once
#include "bfo.hpp"

namespace ttsamd {

// kernel-level launchers (bfo3_pair.hip, bfo3_conv.hip); the parameter blocks are those of the bf16 engine with x / y / res /
// sum_in pointing at x3 tensors and w at x3 weights
bool bfo3_pair_supported(int32_t channels, int32_t k, int32_t dil, int32_t L);
int32_t bfo3_launch_pair(int32_t channels, int32_t k, const BfoPairParams& p, hipStream_t s);
// a whole k = 3 ResBlock (three pairs) in one launch (bfo3_chain.hip); bit-identical to three bfo3_launch_pair calls
bool bfo3_chain_supported(int32_t channels, int32_t k, const int32_t* dil, int32_t n_pairs, int32_t L);
int32_t bfo3_launch_chain(int32_t channels, const BfoChainParams& p, hipStream_t s);
int32_t bfo3_launch_conv(const BfoConvParams& p, hipStream_t s);
int32_t bfo3_launch_convt(const BfoConvParams& p, hipStream_t s);
// fp32 channel-first [B][C][L] <-> x3 tensor (leaky-relu with `slope` on the way in, its inverse on the way out)
int32_t bfo3_launch_pack(const float* x, int32_t B, int32_t C, int32_t L, float slope, void* out, hipStream_t s);
int32_t bfo3_launch_unpack(const void* in, int32_t B, int32_t C, int32_t L, float slope, float* out, hipStream_t s);
// LayerNorm over the channels of a channel-first fp32 tensor (in place or not) + its x3 copy for the conv that follows (C = 256 / 384 / 512)
int32_t launch_layernorm_cf_x3(const float* x, float* y, void* y_x3, const float* gamma, const float* beta, const int64_t* lens,
                               int32_t apply_mask, int32_t B, int32_t C, int32_t S, hipStream_t s, float eps = 1e-5f);
// HiFi-GAN tail on an x3 tensor activated with slope 0.01: wave = tanh(conv7(a) + b)   (models.py:123-125)
int32_t bfo3_launch_conv_post(const void* x, const float* w, const float* bias, const int64_t* lens, int32_t len_mul, int32_t B,
                              int32_t C, int32_t L, float* wave, int64_t wave_bs, hipStream_t s);
// host: torch Conv1d weight [Cout][Cin][K] -> [Cin/16][K][2][CoutP][hi 8 | lo 8] bf16 (uint16 elements)
int64_t bfo3_packed_conv_elems(int cout, int cin, int k);
void bfo3_pack_conv_weight(const float* w, int cout, int cin, int k, uint16_t* out);
// host: torch ConvTranspose1d weight [Cin][Cout][2u] (stride u, padding u/2) -> u polyphase 2-tap filters in that layout
int64_t bfo3_packed_convt_elems(int cin, int cout, int u);
void bfo3_pack_convt_weight(const float* w, int cin, int cout, int u, uint16_t* out);

#ifdef __HIPCC__
// 16-byte store of a freshly computed half entry.  MEASURED on gfx950 (tools/scratch history, tests/test_gpu_bfo3.py): a
// buffer_store_dwordx4 whose data registers are rewritten by the very next VALU instruction (the epilogue loops below: the next
// (j, g) entry's v_pk_mul_f32 lands in the registers the store is still reading) stores garbage in the dwords that instruction's
// high halves write, lanes 12-15 of every 16 -- and hipcc inserts no wait state when the store takes its soffset from an SGPR
// (the "12-dword store" hazard it knows is keyed on an immediate soffset).  The store and two wait states therefore go out as
// ONE asm statement, so nothing can be scheduled between them.
__device__ __forceinline__ void bfo3_st16(bfo_i4 v, bfo_i4 rsrc, int voffset, int soffset) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
// four fp32 values -> the 16-byte half entry {hi01, hi23, lo01, lo23}; `mask` = 0 zeroes it without a branch
__device__ __forceinline__ bfo_i4 bfo3_split4(float v0, float v1, float v2, float v3, int mask) {
    bfo_i4 w;
    w.x = bfo_pk(v0, v1);
    w.y = bfo_pk(v2, v3);
    w.z = bfo_pk(v0 - bfo_lo(w.x), v1 - bfo_hi(w.x));      // the differences are exact in fp32
    w.w = bfo_pk(v2 - bfo_lo(w.y), v3 - bfo_hi(w.y));
    w.x &= mask; w.y &= mask; w.z &= mask; w.w &= mask;
    return w;
}
// ... after the activation of the consumer (max(v, v * slope), slope in [0, 1])
__device__ __forceinline__ bfo_i4 bfo3_act4(float v0, float v1, float v2, float v3, float slope, int mask) {
    const bfo_f2 a = {v0, v1}, b = {v2, v3};
    const bfo_f2 sa = a * slope, sb = b * slope;
    return bfo3_split4(bfo_vmax(v0, sa.x), bfo_vmax(v1, sa.y), bfo_vmax(v2, sb.x), bfo_vmax(v3, sb.y), mask);
}
// the four values of a half entry given as its hi words (h01, h23) and lo words (l01, l23)
__device__ __forceinline__ void bfo3_join4(int h01, int h23, int l01, int l23, float (&v)[4]) {
    v[0] = bfo_lo(h01) + bfo_lo(l01);
    v[1] = bfo_hi(h01) + bfo_hi(l01);
    v[2] = bfo_lo(h23) + bfo_lo(l23);
    v[3] = bfo_hi(h23) + bfo_hi(l23);
}

// One conv over an LDS-resident window held as a hi plane and a lo plane:
//     acc[j] += sum_{h, tap} Ah(h, tap) x Bh(h, tap, j) + Ah x Bl + Al x Bh.
//   wrs / wv  : buffer resource of the x3 weights and this lane's byte offset in it (kk * CoutP + row) * 32
//   wstep     : bytes per (h, tap) step = 2 * CoutP * 32
//   sBh       : this lane's hi-plane LDS entry for (octet kk, first column tile, tap 0); + j * 32 entries per column tile
//   lo_off    : entries from the hi plane to the lo plane
//   hstride   : LDS entries per 16-channel group = 2 * window stride;   dil: entries per tap
// Per step the wave issues three sweeps of NT MFMAs (Ah Bh, Ah Bl, Al Bh: NT MFMAs between two uses of one accumulator);
// the lo fragments are refilled for the next step behind the second sweep's MFMAs, the hi fragments behind the third's
// (>= NT MFMAs = the ds_read_b128 latency ahead of their next use), the A pair of step (h + PH, tap) behind the last MFMAs
// that read the old one.  Pinned with sched_barrier like bfo_mma.
template <int K, int PH, int NT>
__device__ __forceinline__ void bfo3_mma(bfo_f16 (&acc)[NT], const bfo_i4 wrs, const int wv, const int wstep, const uint4* sBh,
                                         const int lo_off, const int n_hexa, const int hstride, const int dil, const int h0 = 0) {
    bfo_i4 Ah[PH][K], Al[PH][K];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int so = ((h0 + min(ph, n_hexa - 1)) * K + t) * wstep;
            Ah[ph][t] = bfo_ld16(wrs, wv, so, 0);
            Al[ph][t] = bfo_ld16(wrs, wv + 16, so, 0);
        }
    uint4 Bh[NT], Bl[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        Bh[j] = sBh[j * 32];
        Bl[j] = sBh[lo_off + j * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int h = 0; h < n_hexa; h += PH) {
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
            const int hc = h + ph;
            if (hc >= n_hexa) break;
            const int hn = min(hc + PH, n_hexa - 1);                      // tail: re-load the last group (L1 hit, unused)
            const uint4* sBn = sBh + min(hc + 1, n_hexa - 1) * hstride;    // first step of the next group
            const uint4* sBc = sBh + hc * hstride;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const uint4* nx = (t + 1 < K) ? sBc + (t + 1) * dil : sBn;
                const int so = ((h0 + hn) * K + t) * wstep;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfo_h8, Ah[ph][t]),
                                                                     __builtin_bit_cast(bfo_h8, Bh[j]), acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfo_h8, Ah[ph][t]),
                                                                     __builtin_bit_cast(bfo_h8, Bl[j]), acc[j], 0, 0, 0);
                    Bl[j] = nx[lo_off + j * 32];
                    if (j == NT - 1) Ah[ph][t] = bfo_ld16(wrs, wv, so, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfo_h8, Al[ph][t]),
                                                                     __builtin_bit_cast(bfo_h8, Bh[j]), acc[j], 0, 0, 0);
                    Bh[j] = nx[j * 32];
                    if (j == NT - 1) Al[ph][t] = bfo_ld16(wrs, wv + 16, so, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}
#endif  // __HIPCC__

}  // namespace ttsamd
