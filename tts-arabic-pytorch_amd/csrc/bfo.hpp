// bf16 "octet" engine (BASELINE config 3): HiFi-GAN on v_mfma_f32_32x32x16_bf16 with bf16 activations in HBM.
//
// Activation layout in HBM *and* in LDS:  [B][C/8][L][8 bf16]  -- one 16-byte ENTRY = the 8 channels 8o..8o+7 of one
// position.  That entry is exactly
//   * the B operand of one lane of v_mfma_f32_32x32x16_bf16 (lane (col, kk) holds k = 8kk..8kk+7: the entry of octet
//     2h + kk of the 16-channel group h at its column) -> one ds_read_b128 per MFMA operand, staging = plain 16-byte copies;
//   * two lanes' worth of the MFMA C layout (lane (col, kk), registers 4g..4g+3 = channels 8g + 4kk + {0..3} of the wave's
//     32-row slab) -> residual loads and output stores are 8 bytes per lane, 512 contiguous bytes per wave instruction,
//     no LDS transposition in the epilogue.
// Tensors are stored PRE-ACTIVATED: a = leaky_relu(x, slope of the consumer), rounded once from the fp32 accumulator.  A
// conv reads its input as a plain copy (vocoder/hifigan/models.py:49,51,114 apply the leaky-relu on the consumer side);
// where the raw value is needed (the ResBlock residual x + conv(...), models.py:52) it is recovered as a >= 0 ? a : a / slope,
// which carries the same 2^-9 relative rounding as storing bf16(x) would.
// Weights: [Cin/16][K][2 (kk)][CoutP][8 bf16] = the A operand per lane (row co, k = 16h + 8kk + e), streamed from L2
// straight into registers (each wave reads only its own 32-row slab; no LDS ring, no barriers inside a conv).
#pragma once
#include "common.hpp"

namespace ttsamd {

typedef __bf16 bfo_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfo_h2 __attribute__((ext_vector_type(2)));
typedef float bfo_f16 __attribute__((ext_vector_type(16)));
typedef float bfo_f2 __attribute__((ext_vector_type(2)));
typedef int bfo_i4 __attribute__((ext_vector_type(4)));
typedef int bfo_i2 __attribute__((ext_vector_type(2)));

constexpr int BFO_DMAX = 5;            // largest dilation the LDS window is sized for
constexpr int BFO_OOB = 0x7ffffff0;    // voffset of a masked-off buffer access (reads 0, stores dropped)

struct BfoPairParams {
    const void* x;         // [B][C/8][L][8] bf16, activated with in_slope: conv input AND residual
    void* y;               // same layout; must not alias x (other blocks read x's halo)
    const void* sum_in;    // mode != 0: running ResBlock sum, RAW bf16 in the same layout (may alias y)
    const void* w1;        // packed bf16 [C/16][K][2][C][8]
    const void* w2;
    const float* b1;
    const float* b2;
    const int64_t* lens;   // valid length = lens[b] * len_mul (nullptr -> L)
    int32_t len_mul, L, dil, batch;
    int32_t mode;          // 0: v = x + conv   1: v = sum_in + x + conv   2: v = (sum_in + x + conv) / div
    float div;
    float in_slope;        // activation x was stored with (0.1)
    float mid_slope;       // leaky-relu between c1 and c2 (0.1)
    float out_slope;       // y = leaky_relu(v, out_slope); 1 = raw
    int32_t compact;              // set by the launcher: ragged batch, blocks take the lin-th LIVE tile (common.hpp: live_tile)
    unsigned long long* timing;   // tools/bfo_pair_bench -DBFO_TIMING only: [blocks][16] shader-clock stamps (nullptr otherwise)
};

// A whole k = 3 ResBlock1 (three c1 -> c2 pairs, vocoder/hifigan/models.py:46-53) in one launch: bfo_chain.hip
struct BfoChainParams {
    const void* x;         // [B][C/8][L][8] bf16, activated with in_slope: input of the first pair
    void* y;               // output of the LAST pair, same layout; must not alias x
    const void* sum_in;    // mode != 0: running ResBlock sum, RAW bf16 (may alias y)
    const void* w1[3];     // packed bf16 [C/16][k][2][C][8] per pair
    const void* w2[3];
    const float* b1[3];
    const float* b2[3];
    const int64_t* lens;
    int32_t len_mul, L, batch;
    int32_t dil[3];        // dilation of each pair's first conv
    int32_t k;             // kernel size of the six convs: 3 (C = 32 / 64 / 128) or 7 (C = 32 / 64)
    int32_t mode;          // applied to the last pair's output, as BfoPairParams::mode
    float div;
    float in_slope;        // activation x is stored with AND the one between the pairs (0.1)
    float mid_slope;       // leaky-relu between c1 and c2 (0.1)
    float out_slope;       // of the last pair's output; 1 = raw
    int32_t compact;       // set by the launcher
};

struct BfoConvParams {
    const void* x;         // [B][Cin/8][Lin][8] bf16, already activated (plain copy into LDS)
    void* y;               // [B][Cout/8][Lout][8] bf16 (Lout = Lin * up)
    const void* w;         // packed bf16 [up phases][Cin/16][K][2][CoutP][8]
    const float* bias;
    const void* res;       // residual in the y layout, activated with res_slope (nullptr = none)
    const void* sum_in;    // running sum, raw bf16 (mode != 0)
    float* y_f32;          // != nullptr: the result leaves as fp32 channel-first [B][Cout][Lout] instead (y unused): FastPitch's
    const float* res_f32;  //   second conv-FF conv writes the fp32 residual stream; res_f32 = fp32 channel-first residual or nullptr
    const int64_t* lens;
    int32_t len_mul, Lin, batch;
    int32_t out_all;       // 1: `lens` masks the INPUT only; all Lin output positions are computed and stored (FastPitch's predictors,
                           //    model.py:129-133: enc_out * mask goes in, the hidden layers are not masked)
    int32_t Cin, Cout, K, dil;
    int32_t up;            // 1: Conv1d ("same" padding); u > 1: ConvTranspose1d(stride u, kernel 2u, padding u/2)
    int32_t mode;
    float div, res_slope, out_slope;
    // split K for launches that cannot fill the chip (batch 1 / 8: FastPitch's 1536 -> 384 conv is 6 blocks at batch 1): the C-in
    // slabs are dealt to `ksplit` blocks per tile (extra grid.y factor) that write raw fp32 partial sums to
    // splitk_ws[ks][b][Cout][L]; bfo_splitk_reduce then sums them IN ORDER and applies the epilogue (deterministic, no atomics)
    float* splitk_ws;      // >= splitk_floats floats of scratch, or nullptr (never split)
    int64_t splitk_floats;
    int32_t ksplit;        // set by the launcher
    int32_t compact;       // set by the launcher: ragged batch, blocks take the lin-th LIVE tile (common.hpp: live_tile)
    // y_f32 output only: LayerNorm over the channels of the result, in place, plus its bf16 octet copy for the next conv
    // (transformer.py:88,158: conv + residual -> LayerNorm): layernorm_cf_octet_kernel follows the conv (and its split-K reduction).
    const float* ln_g;     // gamma (nullptr = no LayerNorm)
    const float* ln_b;
    void* ln_octet;        // [B][Cout/8][L][8] bf16
    const int64_t* ln_lens;   // positions >= ln_lens[b] are written as zero (nullptr = none)
};

// kernel-level launchers (bfo_pair.hip, bfo_conv.hip)
bool bfo_pair_supported(int32_t channels, int32_t k, int32_t dil, int32_t L);
int32_t bfo_launch_pair(int32_t channels, int32_t k, const BfoPairParams& p, hipStream_t s);
// a whole k = 3 (or, C <= 64, k = 7) ResBlock in one launch (bfo_chain.hip); TTSAMD_BFO_CHAIN=0: three pair launches
bool bfo_chain_supported(int32_t channels, int32_t k, const int32_t* dil, int32_t n_pairs, int32_t L, int32_t batch);
bool bfo_chain_wanted(int32_t channels, int32_t k, const int32_t* dil, int32_t n_pairs, int32_t L, int32_t batch);      // + the routing switches
int32_t bfo_launch_chain(int32_t channels, const BfoChainParams& p, hipStream_t s);
int32_t bfo_launch_conv(const BfoConvParams& p, hipStream_t s);
int32_t bfo_launch_convt(const BfoConvParams& p, hipStream_t s);
// fp32 channel-first [B][C][L] <-> octet bf16 (leaky-relu with `slope` on the way in, its inverse on the way out)
int32_t bfo_launch_pack(const float* x, int32_t B, int32_t C, int32_t L, float slope, void* out, hipStream_t s);
int32_t bfo_launch_unpack(const void* in, int32_t B, int32_t C, int32_t L, float slope, float* out, hipStream_t s);
// HiFi-GAN tail on an octet tensor activated with slope 0.01: wave = tanh(conv7(a) + b)   (models.py:123-125)
int32_t bfo_launch_conv_post(const void* x, const float* w, const float* bias, const int64_t* lens, int32_t len_mul, int32_t B,
                             int32_t C, int32_t L, float* wave, int64_t wave_bs, hipStream_t s);
// host: torch Conv1d weight [Cout][Cin][K] -> [Cin/16][K][2][CoutP][8] bf16 (Cin zero-padded to a multiple of 16)
int64_t bfo_packed_conv_elems(int cout, int cin, int k);
void bfo_pack_conv_weight(const float* w, int cout, int cin, int k, uint16_t* out);
// host: torch ConvTranspose1d weight [Cin][Cout][2u] (stride u, padding u/2) -> u polyphase 2-tap filters in that layout
int64_t bfo_packed_convt_elems(int cin, int cout, int u);
void bfo_pack_convt_weight(const float* w, int cin, int cout, int u, uint16_t* out);

#ifdef __HIPCC__
__device__ bfo_i4 bfo_ld16(bfo_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ bfo_i2 bfo_ld8(bfo_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
__device__ void bfo_st16(bfo_i4 v, bfo_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4i32");
__device__ void bfo_st8(bfo_i2 v, bfo_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2i32");
__device__ float bfo_ld4f(bfo_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void bfo_st4f(float v, bfo_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");

__device__ __forceinline__ bfo_i4 bfo_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    bfo_i4 r;
    r.x = (int)(unsigned)a;
    r.y = (int)(unsigned)(a >> 32);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}

__device__ __forceinline__ float bfo_lo(int u) { return __int_as_float(u << 16); }
__device__ __forceinline__ float bfo_hi(int u) { return __int_as_float(u & (int)0xffff0000); }
__device__ __forceinline__ int bfo_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (RNE): a -> low half
    const bfo_f2 v = {a, b};
    return __builtin_bit_cast(int, __builtin_convertvector(v, bfo_h2));
}
// leaky-relu for slopes in [0, 1] as max(v, v * slope) and its inverse (1 / slope >= 1) as min(a, a / slope): two VALU
// operations instead of multiply + compare + select.  v_max / v_min are issued through inline asm: on a value that comes out of
// an MFMA the compiler otherwise prepends a canonicalising v_max(v, v) to every one (the exchange and epilogue phases of the
// fused pair are VALU-bound next to a partner wave that is issuing MFMAs).
__device__ __forceinline__ float bfo_vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float bfo_vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float bfo_lrelu(float v, float slope) { return bfo_vmax(v, v * slope); }
__device__ __forceinline__ float bfo_unrelu(float a, float inv) { return bfo_vmin(a, a * inv); }
// four accumulator values -> activated, rounded bf16 entry half (8 bytes); `mask` = 0 zeroes it without a branch
__device__ __forceinline__ bfo_i2 bfo_act4(float v0, float v1, float v2, float v3, float slope, int mask) {
    const bfo_f2 a = {v0, v1}, b = {v2, v3};
    const bfo_f2 sa = a * slope, sb = b * slope;                       // v_pk_mul_f32
    bfo_i2 w;
    w.x = bfo_pk(bfo_vmax(v0, sa.x), bfo_vmax(v1, sa.y)) & mask;
    w.y = bfo_pk(bfo_vmax(v2, sb.x), bfo_vmax(v3, sb.y)) & mask;
    return w;
}

// One conv over an LDS-resident window: acc[j] += sum_{h, tap} A(h, tap) x B(h, tap, column tile j).
//   wrs / wv  : buffer resource of the packed weights and this lane's byte offset in it (kk * CoutP + row) * 16
//   wstep     : bytes per (h, tap) step = 2 * CoutP * 16
//   sB        : this lane's LDS entry for (octet kk, first column tile, tap 0); + j * 32 entries per column tile
//   hstride   : LDS entries per 16-channel group = 2 * window stride;   dil: entries per tap
//   h0        : first 16-channel group of this call inside the weight tensor (Cin slabs)
// The A fragments run PH groups ahead of their use in a register ring (a step is NT MFMAs = NT * 32 cycles; an L2 hit
// 500-800 under load); each B fragment is re-loaded for the next step right behind the MFMA that consumed it (NT MFMAs =
// the ds_read_b128 latency ahead).
template <int K, int PH, int NT>
__device__ __forceinline__ void bfo_mma(bfo_f16 (&acc)[NT], const bfo_i4 wrs, const int wv, const int wstep, const uint4* sB,
                                        const int n_hexa, const int hstride, const int dil, const int h0 = 0) {
    bfo_i4 A[PH][K];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
        for (int t = 0; t < K; ++t) A[ph][t] = bfo_ld16(wrs, wv, ((h0 + min(ph, n_hexa - 1)) * K + t) * wstep, 0);
    uint4 Bf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) Bf[j] = sB[j * 32];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int h = 0; h < n_hexa; h += PH) {
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
            const int hc = h + ph;
            if (hc >= n_hexa) break;                                       // n_hexa need not be a multiple of PH
            const int hn = min(hc + PH, n_hexa - 1);                      // tail: re-load the last group (L1 hit, unused)
            const uint4* sBn = sB + min(hc + 1, n_hexa - 1) * hstride;     // first step of the next group
            const uint4* sBc = sB + hc * hstride;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const uint4* nx = (t + 1 < K) ? sBc + (t + 1) * dil : sBn;
                // every MFMA is followed by the LDS read that refills its B fragment for the NEXT step (NT MFMAs = 256
                // cycles ahead of its use); the A fragment is refilled behind the last MFMA that reads it.  Pinned: the
                // scheduler otherwise sinks each read to right in front of its MFMA (lgkmcnt(0) per pair of MFMAs).
#pragma unroll
                for (int j = 0; j < NT; ++j) {
#if !defined(BFO_EXP) || (BFO_EXP & 1) == 0      /* timing builds of tools/bfo_pair_bench only (results are wrong): 1 = no MFMAs, */
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfo_h8, A[ph][t]),
                                                                     __builtin_bit_cast(bfo_h8, Bf[j]), acc[j], 0, 0, 0);
#else
                    asm volatile("" : "+v"(Bf[j].x), "+v"(Bf[j].y), "+v"(Bf[j].z), "+v"(Bf[j].w));
#endif
#if !defined(BFO_EXP) || (BFO_EXP & 6) == 0      /* 2 = every second B refill dropped, 4 = no B refills at all */
                    Bf[j] = nx[j * 32];
#elif (BFO_EXP & 2)
                    if (j & 1) Bf[j] = nx[j * 32];
#endif
                    if (j == NT - 1) A[ph][t] = bfo_ld16(wrs, wv, ((h0 + hn) * K + t) * wstep, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}
#endif  // __HIPCC__

}  // namespace ttsamd
