// Winograd F(4,3) DECOMPOSITION kernel of the exact-fp32 engine (round 6): the k = 3 / 7 / 11 'same' Conv1d launches of
// conv_wino2.hip with FOUR outputs per tuple instead of two --
//     k = 3   one three-tap sub-filter, 6 products per output quad                                  (F(2,3):  8, direct 12)
//     k = 7   two sub-filters + one single tap, 2 x 6 + 4 = 16                                      (F(2,3): 20, direct 28)
//     k = 11  three sub-filters + taps (9, 10) as a sub-filter with a zero third tap whose U5 = g2 = 0 group is never issued
//             (= F(4,2)), 3 x 6 + 5 = 23                                                            (F(2,3): 32, direct 44)
// i.e. 0.75 / 0.80 / 0.72 of the F(2,3) kernel's MFMAs.  Transforms (Lavin & Gray, points 0, +-1, +-2, inf), sub-filter on the
// positions x0..x5 of the tuple's d-decimated window:
//     V0 = 4 x0 - 5 x2 + x4          V1 = (x4 - 4 x2) + (x3 - 4 x1)     V2 = (x4 - 4 x2) - (x3 - 4 x1)
//     V3 = (x4 - x2) + 2 (x3 - x1)   V4 = (x4 - x2) - 2 (x3 - x1)       V5 = 4 x1 - 5 x3 + x5
//     U0 = g0 / 4   U1 = -(g0 + g1 + g2) / 6   U2 = -(g0 - g1 + g2) / 6   U3 = g0 / 24 + g1 / 12 + g2 / 6
//     U4 = g0 / 24 - g1 / 12 + g2 / 6   U5 = g2                                     (in double on the host, one rounding)
//     y0 = P0 + (P1 + P2) + (P3 + P4)      y1 = (P1 - P2) + 2 (P3 - P4) [+ P6]
//     y2 = (P1 + P2) + 4 (P3 + P4) [+ P7]  y3 = (P1 - P2) + 8 (P3 - P4) + P5
// The output transform is linear and shared, so every sub-filter of a conv accumulates into the same six planes; the single tap
// of k = 7 has no saving to offer (4 products per quad either way) and goes in untransformed: g x[t + j] -> P0 / P6 / P7 / P5 for
// the outputs j = 0..3 (P0 and P5 reach y0 / y3 only, P6 / P7 are two extra accumulators added to y1 / y2).  Dilation d: the
// output quad is (q, q + d, q + 2 d, q + 3 d), tuple slot pc of a tile at column (pc / d) 4 d + pc % d.
// NUMERICS: tests/test_wino_f43_numerics_cpu.py runs the whole 75-conv vocoder and FastPitch with every routed conv emulated in
// fp32 through these transforms: wave 3.2e-7 / mel 3.8e-6 max-abs against float64 (direct fp32: 2.6e-7 / 2.9e-6, F(2,3): 2.3e-7 /
// 4.2e-6; tolerance 1e-4 / 1e-3).  tests/test_gpu_wino.py checks the kernel itself against float64.
// Reference ops: vocoder/hifigan/models.py:30-53 (ResBlock1 convs), models/fastpitch/fastpitch/transformer.py:59-65 (conv-FF).
//
// Anatomy = conv_wino2.hip's (weights L2 -> register queue four groups ahead, LDS = the transformed planes only, halo zeros from the buffer
// range check, one activation per loaded value, single-instruction jobs in the gaps between MFMAs) with two differences.  The WINDOW is
// fetched as 16-byte vectors -- per lane at dilation 1, per wave through a private LDS strip at dilation 3 / 5 (Wino4Geo) -- because one
// dword per position and lane (conv_wino2's way) saturates the L1 path at this kernel's MFMA count per staged value.  And MT = 1: six or eight 32 x 32 planes per
// wave are 96 / 128 accumulator registers, so a wave owns ONE 32-row tile x 32 tuples; block = 2 x 2 waves = 64 rows x 64 tuples
// (256 outputs).  k = 11 runs its 23 groups in two PHASES (12 + 11) so that two stages fit 48 KB; its packed weights carry a 24th
// all-zero group that is fetched (queue slots line up) and never multiplied.
// k = 1 (Vocos' pwconv1 / pwconv2, FastPitch's qkv / o_net: vocoder/vocos/modules.py ConvNeXtBlock, transformer.py:96-99) runs on the same
// skeleton as a plain GEMM: nothing to transform, the tuple's four positions ARE the four single-tap planes P0 / P6 / P7 / P5, one weight
// fragment per octet (the direct engine's packed weights as they are) feeds 16 MFMAs, the epilogue also knows GELU / tanh / the per-row
// factor of the direct kernel.  Same products in the same order as conv1d_mfma_f32<1>; 251 / 217 vs 263 / 233 us on Vocos' two GEMMs
// (512 <-> 1536 rows x 15 872 frames: matrix pipe 65 / 75 % busy against 61 / 70 %), Vocos at batch 32 4.99 -> 4.67 ms.
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <vector>

#include "conv_mfma_common.hpp"
#include "bfo.hpp"

// timing experiments only (tools/w4_exp.sh builds its own libraries; results are WRONG by design): bit 0 no x loads in the main loop,
// 1 no activation, 2 no plane writes, 3 no weight refill, 4 no B reads
#ifndef TTS_W4_EXP
#define TTS_W4_EXP 0
#endif

namespace ttsamd {

typedef float w4_f32x4 __attribute__((ext_vector_type(4)));
typedef float w4_f32x2 __attribute__((ext_vector_type(2)));

template <int K, int NOCT_, int NSTAGE_, int EPI_ = 0>
struct Wino4Geo {
    static constexpr int WM = 2, WN = 2;
    static constexpr int NSF = K == 1 ? 0 : (K == 3 ? 1 : (K == 7 ? 2 : 4));     // sub-filters (k = 11: the fourth is taps 9, 10 + a zero tap)
    static constexpr int NL = (K == 7 || K == 1) ? 1 : 0;            // single taps
    static constexpr int NG = K == 1 ? 4 : (K == 3 ? 6 : (K == 7 ? 16 : 23));    // operand groups per octet that are multiplied
    static constexpr int NGQ = (NG + 1) & ~1;                        // ... in the packed weights (k = 11: + one zero group)
    // k = 1 (a plain GEMM on this kernel's skeleton: no transform, the tuple's four positions are the four single-tap planes P0 / P6 / P7 / P5):
    // the four groups of an octet multiply the SAME filter, so the queue holds one fragment per octet -- the direct engine's packed weights
    // [Cin/8][1][2][CoutP][4] as they are -- and each fragment feeds 16 MFMAs
    static constexpr bool WSHARE = K == 1;
    static constexpr int NGW = WSHARE ? 1 : NGQ;                     // weight groups per octet in memory
    static constexpr int NPL = (NL || EPI_ == 3) ? 8 : 6;            // accumulator planes (P6 / P7: the single tap of k = 7, the preloaded residual)
    static constexpr int NPOS = K == 1 ? 4 : (K == 3 ? 6 : (K == 7 ? 10 : 14));     // input positions of a tuple's window
    static constexpr int NOCT = NOCT_, NSTAGE = NSTAGE_;
    static constexpr int NPH = K == 11 ? 2 : 1;
    static constexpr int NGPM = NPH == 1 ? NG : 12;                  // group slots of a stage
    static constexpr int CO_BLK = WM * 32;
    static constexpr int NTUP = WN * 32;                             // output quads per block
    static constexpr int NT_BLK = 4 * NTUP;                          // outputs per block (dilation 1)
    static constexpr int BUF4 = NOCT * 2 * NGPM * NTUP;              // float4s per stage: [octet][kk][group slot][tuple]
    static constexpr int PF = 4;                                     // weight groups in flight ahead of the one being multiplied (16 MFMAs ~ 1000 cycles;
                                                                     // two groups -- 512 cycles, under the L2 latency -- measured 59-70 % matrix-pipe busy)
    static constexpr int NM = 4;                                     // MFMAs per operand group (the four channel pairs of the octet)
    __host__ __device__ static constexpr int glo(int ph) { return NPH == 1 ? 0 : 12 * ph; }
    __host__ __device__ static constexpr int ngp(int ph) { return NPH == 1 ? NG : (ph == 0 ? 12 : NG - 12); }     // multiplied
    __host__ __device__ static constexpr int ngq(int ph) { return NPH == 1 ? NGQ : 12; }                          // fetched
    // first / last input position (0 .. NPOS - 1) group g touches
    __host__ __device__ static constexpr int gfirst(int g) { return g < 6 * NSF ? 3 * (g / 6) : 3 * NSF + (g - 6 * NSF); }
    __host__ __device__ static constexpr int glast(int g) {
        return g < 6 * NSF ? (3 * (g / 6) + 5 < NPOS ? 3 * (g / 6) + 5 : NPOS - 1) : 3 * NSF + (g - 6 * NSF);
    }
    __host__ __device__ static constexpr int plane(int g) {
        return g < 6 * NSF ? g % 6 : ((g - 6 * NSF) == 0 ? 0 : ((g - 6 * NSF) == 1 ? 6 : ((g - 6 * NSF) == 2 ? 7 : 5)));
    }
    __host__ __device__ static constexpr int mlo(int ph) { return gfirst(glo(ph)); }
    __host__ __device__ static constexpr int mhi(int ph) {
        int m = 0;
        for (int g = glo(ph); g < glo(ph) + ngp(ph); ++g) m = glast(g) > m ? glast(g) : m;
        return m;
    }
    __host__ __device__ static constexpr int npos(int ph) { return mhi(ph) - mlo(ph) + 1; }
    static constexpr int NPOSP = npos(0) > npos(NPH - 1) ? npos(0) : npos(NPH - 1);
    __host__ __device__ static constexpr int ngap(int ph) { return NOCT * ngp(ph) * NM; }     // gaps (one per MFMA) of a step
    __host__ __device__ static constexpr int nwj(int ph) { return NOCT * ngp(ph); }           // write jobs (one plane each)
    // dilation 1 (D1 kernels): the window is fetched as ALIGNED 16-byte vectors -- vector v of a lane = positions 4 (tuple + v) .. + 3 of the
    // row (a vector is entirely left of position 0 -> zeros by the range check, which drops a WHOLE dwordx4 whose first dword is out of range
    // and checks the right edge dword by dword: tools/buf_range_probe.hip) -- instead of one dword per position: a wave instruction then
    // moves 1 KB of contiguous row instead of 64 dwords at a 16-byte stride (4x the lines through the L1 per value)
    __host__ __device__ static constexpr int fdiv4(int e) { return e >= 0 ? e / 4 : -((3 - e) / 4); }
    __host__ __device__ static constexpr int vlo(int ph) { return fdiv4(mlo(ph) - (K - 1) / 2); }
    __host__ __device__ static constexpr int nvec(int ph) { return fdiv4(mhi(ph) - (K - 1) / 2) - vlo(ph) + 1; }
    static constexpr int NVP = nvec(0) > nvec(NPH - 1) ? nvec(0) : nvec(NPH - 1);
    __host__ __device__ static constexpr int nljv(int ph) { return NOCT * 2 * nvec(ph); }     // vector load jobs
    static constexpr int DAV = (K == 3 || K == 1) ? 8 : 16;
    __host__ __device__ static constexpr int tw0v(int ph) { return DAV + 2 * nljv(ph); }       // two activation jobs per vector, one per gap
    __host__ __device__ static constexpr int wsv(int ph) { return (ngap(ph) - tw0v(ph)) / nwj(ph); }
    static_assert(wsv(0) >= 1 && wsv(NPH - 1) >= 1, "one write job per gap at most (vector loads)");
    // dilation 3 / 5, k = 7 / 11 (LP = 2 kernels): each wave fetches its two channel rows of the block's window CONTIGUOUSLY (80 aligned
    // 16-byte vectors per row = 320 positions >= 240 + 13 x 5 + 3), activates them, parks them in a private LDS strip and reads its
    // tuples' positions back at stride d (4-byte LDS reads instead of one 64-lane dword load per position through the L1: the dilated
    // launches sat at 63-76 % matrix-pipe busy against 77-82 % of the dilation-1 ones).  Same wave writes and reads: no block barrier.
    static constexpr int RAWW = 320;                                  // positions per row of the strip
    static constexpr int NROW = 2 * NOCT;                             // channel rows a wave stages per chunk
    static constexpr int RAW_BYTES = 4 * (NROW * RAWW * 4 + 16);      // four waves x (their rows + a dump slot for the idle lanes of the second load)
    static constexpr int DAT = K == 7 ? 16 : (K == 11 ? 12 : 8);      // gaps between the row loads and their activation
    __host__ __device__ static constexpr int nrj(int ph) { return NROW * ((npos(ph) + 1) / 2); }  // LDS read jobs: (row, position pair)
    __host__ __device__ static constexpr bool loads_in(int ph) { return ph == 0; }              // the rows are fetched once per chunk
    __host__ __device__ static constexpr int tr_r0(int ph) { return loads_in(ph) ? DAT + 2 * NROW : 0; }      // first read job
    __host__ __device__ static constexpr int tw0t(int ph) { return tr_r0(ph) + nrj(ph) + 1; }                 // first plane write
    __host__ __device__ static constexpr int wst(int ph) { return (ngap(ph) - tw0t(ph)) / nwj(ph); }
    static_assert(wst(0) >= 1 && wst(NPH - 1) >= 1, "one write job per gap at most (strip path)");
    static_assert((WSHARE ? NOCT : NOCT * ngq(0)) % PF == 0 && (NOCT * ngq(NPH - 1)) % PF == 0, "queue slots line up across steps");
    static_assert((size_t)NSTAGE * BUF4 * 16 <= 80 * 1024, "two blocks per CU");
    static_assert(NPH == 1 || NOCT == 1, "phases split an octet's groups");
};

// outputs per tile at dilation d: the largest multiple of 4 d in 4 * ntup
__device__ __host__ constexpr int wino4_tile(int d, int ntup) { return (4 * ntup / (4 * d)) * (4 * d); }

template <int K, int NOCT_, int NSTAGE_, int EPI, int LP>
__global__ __launch_bounds__(256, TTS_MINWAVES) void conv1d_wino4_f32(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    using G = Wino4Geo<K, NOCT_, NSTAGE_, EPI>;
    constexpr int WN = G::WN, NSF = G::NSF, NOCT = G::NOCT, NSTAGE = G::NSTAGE, NPL = G::NPL, NGPM = G::NGPM;
    constexpr int CO_BLK = G::CO_BLK, NTUP = G::NTUP, NT_BLK = G::NT_BLK, PF = G::PF;
    constexpr int NPH = G::NPH, NPOSP = G::NPOSP, NM = G::NM;
    constexpr bool D1 = LP == 1, TR = LP == 2;      // window loads: 1 aligned vectors per lane (dilation 1), 2 per-wave LDS strip (dilation 3 / 5)
    static_assert(D1 || TR, "window path");
    const int dil = D1 ? 1 : p.dil;
    const int nt_eff = wino4_tile(dil, NTUP), ntup_eff = nt_eff / 4;
    // Block -> (time tile, row block, C-in slice).  The 1-D grid is dealt to the 8 XCDs round-robin (workgroup s runs on XCD s % 8, each
    // with its own L2); slot u = s / 8 of an XCD walks (row block, slice) fastest, so ALL row blocks of a time tile run on ONE XCD one
    // after the other and share the tile's input window through that L2 (Cout = 128 / 256: 2 / 4 row blocks; measured HBM bytes of the
    // C = 128 k = 11 c2 launch with the plain (x, y, z) grid: 1.38x the algorithmic bytes, profiles/r6/traffic_plain_grid.json).
    const int tiles_y = p.CoutP / CO_BLK;
    const int per_tile = tiles_y * p.ksplit;
    const unsigned sidx = blockIdx.x, xcd = sidx & 7u, u = sidx >> 3;
    const int yk = (int)(u % (unsigned)per_tile);
    const unsigned gt = (u / (unsigned)per_tile) * 8u + xcd;            // global time-tile index (utterance-major)
    const int n_tiles_x = (p.Nout + nt_eff - 1) / nt_eff;
    int b = (int)(gt / (unsigned)n_tiles_x);
    int q0 = (int)(gt % (unsigned)n_tiles_x) * nt_eff;
    if (p.compact) {   // dead blocks last (live_tile, common.hpp)
        int tile = 0;
        if (!live_tile(p.lens_out, p.len_out_mul, p.Nout, nt_eff, p.batch, gt, b, tile)) return;
        b = __builtin_amdgcn_readfirstlane(b);
        q0 = __builtin_amdgcn_readfirstlane(tile) * nt_eff;
    } else if (b >= p.batch) {
        return;                                                        // the grid is rounded up to whole groups of 8 tiles
    }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // split K (small problems: launch_wino4_cfg): slice ks of the C-in chunks, raw partial sums to splitk_ws, epilogue in splitk_reduce_kernel
    const int ks = yk / tiles_y;
    const int co_blk0 = (yk - ks * tiles_y) * CO_BLK;
    int n_out = p.Nout;
    if (p.lens_out) n_out = min(n_out, (int)p.lens_out[b] * p.len_out_mul);
    if (q0 >= n_out) return;
    int in_len = p.Lin;
    if (p.lens_in) in_len = min(in_len, (int)p.lens_in[b] * p.len_in_mul);

    const int x_cs = p.x_cs, CoutP = p.CoutP;
    const int chunks_all = p.Cin / (8 * NOCT);
    const int c_beg = (int)((int64_t)ks * chunks_all / p.ksplit), n_chunks = (int)((int64_t)(ks + 1) * chunks_all / p.ksplit);   // [c_beg, n_chunks)
    const int n_groups = (p.Cin / 8) * G::NGW;                // groups of the whole conv in the packed weights
    const float* __restrict__ xb = p.x + (int64_t)b * p.x_bs;
    const float in_slope = p.in_slope;

    float ep_bias = 0.f;
    if (tid < CO_BLK && p.bias) ep_bias = p.bias[min(co_blk0 + tid, p.Cout - 1)];

    const int kk = lane >> 5, l31 = lane & 31;
    const int pe = wn * 32 + l31;                                           // the tuple this lane holds in the accumulators
    const int col_e = dil == 1 ? 4 * pe : (pe / dil) * 4 * dil + pe % dil;  // ... its first column in the tile (then + dil, 2 dil, 3 dil)
    f32x16 acc[NPL];
#pragma unroll
    for (int g = 0; g < NPL; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    // EPI 3 (dilation 1 only: the launcher's choice): residual [+ running ResBlock sum] of the lane's quad -- one 16-byte load per row --
    // straight into the planes that reach exactly one output each: r0 -> P0 (y0), r1 -> P6 (y1), r2 -> P7 (y2), r3 -> P5 (y3).
    // A quad cut by the utterance end loads values that are never stored; past the tensor the range check returns zeros.
    constexpr bool preload = EPI == 3;
    if constexpr (preload) {
        const int wm_s = __builtin_amdgcn_readfirstlane(wm);
        const int row0 = co_blk0 + wm_s * 32;
        const int qv = q0 + col_e;
        const int voff = (qv < n_out ? qv : 0) * 4;
        {
            const int r_cs = p.r_cs;
            const bfo_i4 rs = bfo_rsrc(p.res + (int64_t)b * p.r_bs, (unsigned)p.Cout * (unsigned)r_cs * 4u);
            const int vk = 4 * kk * r_cs * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const w4_f32x4 t = __builtin_bit_cast(w4_f32x4, bfo_ld16(rs, voff + vk, (row0 + (r & 3) + 8 * (r >> 2)) * r_cs * 4, 0));
                acc[0][r] = t.x; acc[6][r] = t.y; acc[7][r] = t.z; acc[5][r] = t.w;
            }
        }
        if (p.mode != 0) {
            const int y_cs_ = p.y_cs;
            const bfo_i4 ys = bfo_rsrc(p.y + (int64_t)b * p.y_bs, (unsigned)p.Cout * (unsigned)y_cs_ * 4u);
            const int vk = 4 * kk * y_cs_ * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const w4_f32x4 t = __builtin_bit_cast(w4_f32x4, bfo_ld16(ys, voff + vk, (row0 + (r & 3) + 8 * (r >> 2)) * y_cs_ * 4, 0));
                acc[0][r] += t.x; acc[6][r] += t.y; acc[7][r] += t.z; acc[5][r] += t.w;
            }
        }
    }

    // ---- weight queue: group gf = octet * NGQ + g sits at w_wino4 + gf * 2 CoutP float4 (the scalar offset of the buffer load, advanced by
    // one group per refill and clamped at the conv's last group: an L1 hit, unused); this lane's fragment at + kk * CoutP + row
    const bfo_i4 wrs = bfo_rsrc(K == 1 ? p.w : p.w_wino4, (unsigned)n_groups * 2u * (unsigned)CoutP * 16u);
    const int wv = (kk * CoutP + co_blk0 + wm * 32 + l31) * 16;
    const int wstep = 2 * CoutP * 16, wlast = (n_groups - 1) * wstep;
    int wso = c_beg * NOCT * G::NGW * wstep;
    w4_f32x4 aq[PF];
#pragma unroll
    for (int g = 0; g < PF; ++g) {
        aq[g] = __builtin_bit_cast(w4_f32x4, bfo_ld16(wrs, wv, wso, 0));
        wso = min(wso + wstep, wlast);
    }

    // ---- staging.  X item of thread (h, kks, pc) and octet ol: channels 8 (c NOCT + ol) + 2 (2 h + pp) + kks, pp = 0 / 1, at the positions
    // q0 + col(pc) + (m - pad) d, m = the phase's window -> components 2 h, 2 h + 1 of the float4 of each plane of the phase at
    // [ol][kks][g][pc].  (h, kks) = the wave index: a wave instruction reads ONE channel row, so the row is the base of a raw buffer
    // descriptor of in_len * 4 bytes and the load's range check returns the zeros of the halo (conv_wino2.hip).
    const int sh = __builtin_amdgcn_readfirstlane((tid >> 7) & 1), skk = __builtin_amdgcn_readfirstlane((tid >> 6) & 1);
    w4_f32x2 sx[D1 ? 1 : NOCT][D1 ? 1 : NPOSP];              // strip path: [pp] = the two channels of the item, one entry per position of the phase
    w4_f32x4 sv[D1 ? NOCT : 1][2][D1 ? G::NVP : 1];          // D1: [octet][pp][vector]
    w4_f32x2 ta[NOCT], tb[NOCT];                             // partial sums shared by two consecutive plane jobs
    const int spe = min(lane, ntup_eff - 1);                 // idle tuple slots (d > 1) repeat the last tuple: never stored
    const int xw0 = (q0 + 4 * lane) * 4;                      // D1: byte offset of the lane's vector 0 in its row
    constexpr int pad_c = (K - 1) / 2;
    const int ch_off = (4 * sh + skk) * x_cs;                 // channel 2 (2 h) + kks; pp adds 2 rows, the octet 8

    // D1: vector job J -> (octet, pp, vector)
#define TTS_VJOB_IDX(PH, J)                                                                          \
        const int nv_ = G::nvec(PH);                                                                 \
        const int ol_ = (J) / (2 * nv_), pp_ = ((J) / nv_) % 2, vi_ = (J) % nv_;
#define TTS_VLOAD_JOB(PH, J, XSO)                                                                    \
    {                                                                                                \
        TTS_VJOB_IDX(PH, J)                                                                          \
        const bfo_i4 xrs_ = bfo_rsrc(xb + ((XSO) + ch_off + (8 * ol_ + 2 * pp_) * x_cs), (unsigned)in_len * 4u);     \
        sv[ol_][pp_][vi_] = __builtin_bit_cast(w4_f32x4, bfo_ld16(xrs_, xw0 + 16 * (G::vlo(PH) + vi_), 0, 0));      \
    }
    // ... its activation in two jobs (components 0, 1 / 2, 3)
#define TTS_VACT_JOB(PH, A)                                                                          \
    {                                                                                                \
        TTS_VJOB_IDX(PH, (A) / 2)                                                                    \
        if ((A) % 2 == 0) {                                                                          \
            sv[ol_][pp_][vi_].x = bfo_lrelu(sv[ol_][pp_][vi_].x, in_slope);                          \
            sv[ol_][pp_][vi_].y = bfo_lrelu(sv[ol_][pp_][vi_].y, in_slope);                          \
        } else {                                                                                     \
            sv[ol_][pp_][vi_].z = bfo_lrelu(sv[ol_][pp_][vi_].z, in_slope);                          \
            sv[ol_][pp_][vi_].w = bfo_lrelu(sv[ol_][pp_][vi_].w, in_slope);                          \
        }                                                                                            \
    }
    // LP = 2: row job J -> (pp, vector set): lane l fetches vector l (set 0) or 64 + l (set 1: lanes 0..15) of row pp; positions A + 4 v .. + 3,
    // A = the window start rounded down to a multiple of 4 (a vector is then entirely left of position 0 or not at all)
    float* const strip = reinterpret_cast<float*>(smem4 + NSTAGE * G::BUF4) + wid * (G::NROW * G::RAWW + 4);
    const int tr_a = (q0 - pad_c * dil) & ~3;                 // position of strip column 0
    const int tr_off = q0 + (dil == 1 ? 4 * spe : (spe / dil) * 4 * dil + spe % dil) - pad_c * dil - tr_a;   // strip column of the lane's position 0
    w4_f32x4 rv[TR ? G::NROW : 1][2];                          // row = 2 octet + pp
#define TTS_TLOAD_JOB(J, XSO)                                                                        \
    {                                                                                                \
        const int pp_ = (J) / 2, vs_ = (J) % 2;                                                      \
        const bfo_i4 xrs_ = bfo_rsrc(xb + ((XSO) + ch_off + (8 * (pp_ / 2) + 2 * (pp_ % 2)) * x_cs), (unsigned)in_len * 4u); \
        const int v_ = lane + 64 * vs_;                                                              \
        rv[pp_][vs_] = __builtin_bit_cast(w4_f32x4, bfo_ld16(xrs_, (vs_ == 1 && lane >= 16) ? BFO_OOB : (tr_a + 4 * v_) * 4, 0, 0)); \
    }
    // ... activation + the 16-byte strip write (idle lanes of set 1 write the dump slot behind the rows)
#define TTS_TSTORE_JOB(J)                                                                            \
    {                                                                                                \
        const int pp_ = (J) / 2, vs_ = (J) % 2;                                                      \
        w4_f32x4 a_ = rv[pp_][vs_];                                                                  \
        a_.x = bfo_lrelu(a_.x, in_slope); a_.y = bfo_lrelu(a_.y, in_slope);                          \
        a_.z = bfo_lrelu(a_.z, in_slope); a_.w = bfo_lrelu(a_.w, in_slope);                          \
        const int v_ = lane + 64 * vs_;                                                              \
        float* dst_ = (vs_ == 1 && lane >= 16) ? strip + G::NROW * G::RAWW : strip + pp_ * G::RAWW + 4 * v_;    \
        *reinterpret_cast<w4_f32x4*>(dst_) = a_;                                                     \
    }
    // ... read job J of phase PH -> (pp, positions 2 jp, 2 jp + 1 of the phase's window) into the staged values
#define TTS_TREAD_JOB(PH, J)                                                                         \
    {                                                                                                \
        const int npr_ = (G::npos(PH) + 1) / 2;                                                      \
        const int pp_ = (J) / npr_, jp_ = (J) % npr_;                                                \
        const float* src_ = strip + pp_ * G::RAWW + tr_off + (G::mlo(PH) + 2 * jp_) * dil;           \
        sx[pp_ / 2][2 * jp_][pp_ % 2] = src_[0];                                                     \
        if (2 * jp_ + 1 < G::npos(PH)) sx[pp_ / 2][2 * jp_ + 1][pp_ % 2] = src_[dil];                \
    }
    // plane g of octet ol from the staged values.  Job J -> (octet, group of the phase)
#define TTS_SXV(M) (w4_f32x2{sv[ol_][0][G::fdiv4((M) - pad_c) - G::vlo(PH_)][((M) - pad_c) - 4 * G::fdiv4((M) - pad_c)],     \
                             sv[ol_][1][G::fdiv4((M) - pad_c) - G::vlo(PH_)][((M) - pad_c) - 4 * G::fdiv4((M) - pad_c)]})
#define TTS_SX(M) (D1 ? TTS_SXV(M) : sx[D1 ? 0 : ol_][D1 ? 0 : (M) - G::mlo(PH_)])
#define TTS_WRITE_JOB(PH, J, WR)                                                                     \
    {                                                                                                \
        const int PH_ = (PH);                                                                        \
        const int ol_ = (J) / G::ngp(PH_), gl_ = (J) % G::ngp(PH_), g_ = G::glo(PH_) + gl_;         \
        w4_f32x2 v2;                                                                                 \
        if (g_ < 6 * NSF) {                                                                          \
            const int s3 = 3 * (g_ / 6), i_ = g_ % 6;                                                \
            /* the pairs (V1, V2) and (V3, V4) share their two partial sums: kept in ta / tb from the odd job to the even one */ \
            if (i_ == 0) v2 = 4.f * TTS_SX(s3) + (TTS_SX(s3 + 4) - 5.f * TTS_SX(s3 + 2));           \
            else if (i_ == 1) {                                                                      \
                ta[ol_] = TTS_SX(s3 + 4) - 4.f * TTS_SX(s3 + 2);                                     \
                tb[ol_] = TTS_SX(s3 + 3) - 4.f * TTS_SX(s3 + 1);                                     \
                v2 = ta[ol_] + tb[ol_];                                                              \
            } else if (i_ == 2) v2 = ta[ol_] - tb[ol_];                                              \
            else if (i_ == 3) {                                                                      \
                ta[ol_] = TTS_SX(s3 + 4) - TTS_SX(s3 + 2);                                           \
                tb[ol_] = 2.f * (TTS_SX(s3 + 3) - TTS_SX(s3 + 1));                                   \
                v2 = ta[ol_] + tb[ol_];                                                              \
            } else if (i_ == 4) v2 = ta[ol_] - tb[ol_];                                              \
            else v2 = 4.f * TTS_SX(s3 + 1) + (TTS_SX(s3 + 5 < G::NPOS ? s3 + 5 : s3) - 5.f * TTS_SX(s3 + 3));           \
        } else {                                                                                     \
            v2 = TTS_SX(3 * NSF + (g_ - 6 * NSF));                                                   \
        }                                                                                            \
        (WR)[2 * ((ol_ * 2 * NGPM + gl_) * NTUP)] = v2;                                              \
    }

    const float4* sB = smem4 + kk * NGPM * NTUP + wn * 32 + l31;      // + stage * BUF4 + (ol * 2 NGPM + g) * NTUP
    w4_f32x2* sW = reinterpret_cast<w4_f32x2*>(smem4 + skk * NGPM * NTUP + lane) + sh;    // + 2 (stage * BUF4 + (ol * 2 NGPM + g) * NTUP)
    float4 bq[2];

    // prologue: fill NSTAGE - 1 stages (steps 0 .. NSTAGE - 2: phase st % NPH of chunk st / NPH), fetch the first B operand
    {
#pragma unroll
        for (int st = 0; st < NSTAGE - 1; ++st) {
            const int ph = st % NPH;
            const int xso = min(c_beg + st / NPH, n_chunks - 1) * 8 * NOCT * x_cs;
            if constexpr (D1) {
#pragma unroll
                for (int J = 0; J < G::nljv(ph); ++J) TTS_VLOAD_JOB(ph, J, xso)
#pragma unroll
                for (int J = 0; J < 2 * G::nljv(ph); ++J) TTS_VACT_JOB(ph, J)
            } else {
                if (G::loads_in(ph)) {
#pragma unroll
                    for (int J = 0; J < 2 * G::NROW; ++J) TTS_TLOAD_JOB(J, xso)
#pragma unroll
                    for (int J = 0; J < 2 * G::NROW; ++J) TTS_TSTORE_JOB(J)
                }
#pragma unroll
                for (int J = 0; J < G::nrj(ph); ++J) TTS_TREAD_JOB(ph, J)
            }
            w4_f32x2* wr = sW + 2 * st * G::BUF4;
#pragma unroll
            for (int J = 0; J < G::nwj(ph); ++J) TTS_WRITE_JOB(ph, J, wr)
        }
    }
    __syncthreads();
    bq[0] = sB[0];

    int stage = 0;  // step % NSTAGE
    for (int c = c_beg; c < n_chunks; ++c) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            // the step being staged: NSTAGE - 1 ahead (tail: the last chunk is re-staged into a dead stage -- branch-free body)
            const int tph = (ph + NSTAGE - 1) % NPH;
            const int xso = min(c + (ph + NSTAGE - 1) / NPH, n_chunks - 1) * 8 * NOCT * x_cs;
            const int stage_next = (stage + 1 == NSTAGE) ? 0 : stage + 1;
            const int stage_fill = (stage == 0) ? NSTAGE - 1 : stage - 1;
            w4_f32x2* wr = sW + 2 * stage_fill * G::BUF4;
            const float4* rd = sB + stage * G::BUF4;
            const int sn = (c + 1 < n_chunks || ph + 1 < NPH) ? stage_next : stage;
#pragma unroll
            for (int gs = 0; gs < NOCT * G::ngq(ph); ++gs) {
                const int ol = gs / G::ngq(ph), gl = gs % G::ngq(ph);
                const int qs = G::WSHARE ? ol % PF : gs % PF;
                const w4_f32x4 a4 = aq[qs];
                // refill the queue slot with the group PF ahead (k = 1: one fragment per octet, refilled behind its last group)
                if (!G::WSHARE || gl == G::ngq(ph) - 1) {
                    if (!(TTS_W4_EXP & 8)) aq[qs] = __builtin_bit_cast(w4_f32x4, bfo_ld16(wrs, wv, wso, 0));
                    wso = min(wso + wstep, wlast);
                }
                if (gl >= G::ngp(ph)) continue;                    // the zero group of k = 11: fetched, never multiplied
                const int r = ol * G::ngp(ph) + gl;                // multiplied groups of the step so far
                const int cur = r & 1, nxt = cur ^ 1;
                const int plane = G::plane(G::glo(ph) + gl);
                // B operand of the next group: same stage, or (three stages) the first group of the next step's stage
                if (TTS_W4_EXP & 16) bq[nxt] = bq[cur];
                else if (r + 1 < NOCT * G::ngp(ph)) bq[nxt] = rd[(((r + 1) / G::ngp(ph)) * 2 * NGPM + (r + 1) % G::ngp(ph)) * NTUP];
                else if (NSTAGE >= 3) bq[nxt] = sB[sn * G::BUF4];       // (an odd step leaves it in slot 1: moved to slot 0 behind the barrier)
                __builtin_amdgcn_sched_barrier(0);
                const float bv[4] = {bq[cur].x, bq[cur].y, bq[cur].z, bq[cur].w};
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    acc[plane] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m], bv[m], acc[plane], 0, 0, 0);
                    // ---- gap work for the step NSTAGE - 1 ahead: the window loads first, each value activated DAV / DAT gaps later, then
                    // the plane writes into the stage the previous step has left
                    const int t = r * NM + m;
                    if constexpr (D1) {
                        if (!(TTS_W4_EXP & 1) && t < G::nljv(tph)) TTS_VLOAD_JOB(tph, t, xso)
                        if (!(TTS_W4_EXP & 2) && t >= G::DAV && t - G::DAV < 2 * G::nljv(tph)) TTS_VACT_JOB(tph, t - G::DAV)
                        if (!(TTS_W4_EXP & 4) && t >= G::tw0v(tph) && (t - G::tw0v(tph)) % G::wsv(tph) == 0 &&
                            (t - G::tw0v(tph)) / G::wsv(tph) < G::nwj(tph))
                            TTS_WRITE_JOB(tph, (t - G::tw0v(tph)) / G::wsv(tph), wr)
                    } else {
                        if (G::loads_in(tph) && t < 2 * G::NROW) TTS_TLOAD_JOB(t, xso)
                        if (G::loads_in(tph) && t >= G::DAT && t < G::DAT + 2 * G::NROW) TTS_TSTORE_JOB(t - G::DAT)
                        if (t >= G::tr_r0(tph) && t - G::tr_r0(tph) < G::nrj(tph)) TTS_TREAD_JOB(tph, t - G::tr_r0(tph))
                        if (t >= G::tw0t(tph) && (t - G::tw0t(tph)) % G::wst(tph) == 0 && (t - G::tw0t(tph)) / G::wst(tph) < G::nwj(tph))
                            TTS_WRITE_JOB(tph, (t - G::tw0t(tph)) / G::wst(tph), wr)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            if (NSTAGE < 3) bq[0] = sB[sn * G::BUF4];            // two stages: the next step's stage has only just been written
            else if ((NOCT * G::ngp(ph)) % 2 != 0) bq[0] = bq[1];
            stage = stage_next;
        }
    }
#undef TTS_TLOAD_JOB
#undef TTS_TSTORE_JOB
#undef TTS_TREAD_JOB
#undef TTS_VLOAD_JOB
#undef TTS_VACT_JOB
#undef TTS_VJOB_IDX
#undef TTS_SXV
#undef TTS_SX
#undef TTS_WRITE_JOB

    // ---- epilogue: output transform into the dead ring, then the row epilogue of conv_mfma.hip (bias, residual, ReLU, accumulate modes)
    constexpr int LDS_F = CO_BLK * NT_BLK + CO_BLK;                     // floats of LDS this block owns (the launcher allocates max(ring, this))
    constexpr int LPR = NT_BLK / 4;                                     // lanes per row (one float4 each) = 64
    static_assert(LPR == 64, "one wave instruction per row");
    float* ep = reinterpret_cast<float*>(smem4);
    float* epb = ep + LDS_F - CO_BLK;                                   // [CO_BLK] bias
    float* __restrict__ yb = p.y + (int64_t)b * p.y_bs;
    const float* __restrict__ rb = (p.res && !preload) ? p.res + (int64_t)b * p.r_bs : nullptr;
    const int mode = p.mode, relu_out = p.relu_out, Cout = p.Cout;
    const float div = p.div;
    __syncthreads();                                                    // ring stages are dead
    if (tid < CO_BLK) epb[tid] = ep_bias;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
        const float s34 = acc[3][r] + acc[4][r], d34 = acc[3][r] - acc[4][r];
        float y0 = acc[0][r] + s12 + s34;
        float y1 = d12 + 2.f * d34;
        float y2 = s12 + 4.f * s34;
        float y3 = d12 + 8.f * d34 + acc[5][r];
        if constexpr (NSF == 0) { y0 = acc[0][r]; y1 = 0.f; y2 = 0.f; y3 = acc[5][r]; }     // k = 1: P1..P4 do not exist
        if constexpr (NPL == 8) {
            y1 += acc[6][r];
            y2 += acc[7][r];
        }
        if (pe < ntup_eff) {
            if (dil == 1) {
                *reinterpret_cast<float4*>(ep + row * NT_BLK + col_e) = make_float4(y0, y1, y2, y3);
            } else {
                ep[row * NT_BLK + col_e] = y0;
                ep[row * NT_BLK + col_e + dil] = y1;
                ep[row * NT_BLK + col_e + 2 * dil] = y2;
                ep[row * NT_BLK + col_e + 3 * dil] = y3;
            }
        }
    }
    __syncthreads();
    constexpr int NR = CO_BLK / 4;                                      // row iterations per wave
    const int col = lane * 4, q = q0 + col;
    if (q >= n_out || col >= nt_eff) return;
    const bool full = q + 3 < n_out;
    if (p.ksplit > 1) {                                                 // raw partial sums of this C-in slice
        float* __restrict__ pb = p.splitk_ws + ((int64_t)ks * p.batch + b) * Cout * p.Nout;
#pragma unroll 4
        for (int it = 0; it < NR; ++it) {
            const int rl = wid + it * 4;
            const int co = co_blk0 + rl;
            if (co >= Cout) continue;
            const float4 a4 = *reinterpret_cast<const float4*>(ep + rl * NT_BLK + col);
            float* yp = pb + (int64_t)co * p.Nout + q;
            if (full && (p.Nout & 3) == 0) {
                *reinterpret_cast<float4*>(yp) = a4;
            } else {
                const float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q + e < n_out) yp[e] = v[e];
            }
        }
        return;
    }
    // k = 1 only (Vocos' pointwise convs): GELU behind the bias, a per-row factor before the residual, tanh at the end (conv_mfma.hip's epilogue)
    const float* __restrict__ scale = K == 1 ? p.scale : nullptr;
    const bool plain_act = K != 1 || (relu_out < 2 && scale == nullptr);
    if (plain_act && (preload || (!rb && mode == 0))) {
        // nothing to read from memory: a loop without a single vmcnt wait (conv_mfma.hip: why)
        const float lo = relu_out == 1 ? 0.f : -__builtin_inff();
        const bool do_div = preload && mode == 2;
#pragma unroll 4
        for (int it = 0; it < NR; ++it) {
            const int rl = wid + it * 4;
            const int co = co_blk0 + rl;
            if (co >= Cout) continue;
            const float bsv = epb[rl];
            const float4 a4 = *reinterpret_cast<const float4*>(ep + rl * NT_BLK + col);
            float v[4] = {fmaxf(a4.x + bsv, lo), fmaxf(a4.y + bsv, lo), fmaxf(a4.z + bsv, lo), fmaxf(a4.w + bsv, lo)};
            if (do_div) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] / div;
            }
            float* yp = yb + (int64_t)co * p.y_cs + q;
            if (full) {
                *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q + e < n_out) yp[e] = v[e];
            }
        }
        return;
    }
    if constexpr (preload) return;
#pragma unroll 2
    for (int it = 0; it < NR; ++it) {
        const int rl = wid + it * 4;
        const int co = co_blk0 + rl;
        if (co >= Cout) continue;
        const float bsv = epb[rl];
        const float4 a4 = *reinterpret_cast<const float4*>(ep + rl * NT_BLK + col);
        float v[4] = {a4.x, a4.y, a4.z, a4.w};
        float* yp = yb + (int64_t)co * p.y_cs + q;
        const float* rp = rb ? rb + (int64_t)co * p.r_cs + q : nullptr;
        float rr4[4] = {0.f, 0.f, 0.f, 0.f}, pp4[4] = {0.f, 0.f, 0.f, 0.f};
        if (full) {
            if (rp) { const float4 t = *reinterpret_cast<const float4*>(rp); rr4[0] = t.x; rr4[1] = t.y; rr4[2] = t.z; rr4[3] = t.w; }
            if (mode != 0) { const float4 t = *reinterpret_cast<const float4*>(yp); pp4[0] = t.x; pp4[1] = t.y; pp4[2] = t.z; pp4[3] = t.w; }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q + e < n_out) {
                    if (rp) rr4[e] = rp[e];
                    if (mode != 0) pp4[e] = yp[e];
                }
        }
        const float scv = (K == 1 && scale) ? scale[co] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[e] + bsv;
            if constexpr (K == 1) {
                if (relu_out == 2) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));   // nn.GELU()
                x = x * scv;
            }
            x += rr4[e];
            if (relu_out == 1) x = fmaxf(x, 0.f);
            if constexpr (K == 1) { if (relu_out == 3) x = tanhf(x); }
            if (mode == 1) x = pp4[e] + x;
            else if (mode == 2) x = (pp4[e] + x) / div;
            v[e] = x;
        }
        if (full) {
            *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q + e < n_out) yp[e] = v[e];
        }
    }
}

template <int K, int NOCT, int NSTAGE, int EPI, int LP>
static int32_t launch_wino4_epi(const ConvParams& p, hipStream_t stream) {
    using G = Wino4Geo<K, NOCT, NSTAGE, EPI>;
    constexpr size_t ring = (size_t)G::NSTAGE * G::BUF4 * sizeof(float4);
    constexpr size_t epi = ((size_t)G::CO_BLK * G::NT_BLK + G::CO_BLK) * sizeof(float);
    constexpr size_t strip = LP == 2 ? (size_t)G::RAW_BYTES : 0;
    constexpr size_t lds = ring + strip > epi ? ring + strip : epi;
    static_assert(lds <= 80 * 1024, "two blocks per CU");
    static std::atomic<uint64_t> lds_done{0};
    const auto kern = conv1d_wino4_f32<K, NOCT, NSTAGE, EPI, LP>;
    TTS_CHECK_HIP(lds_opt_in((const void*)kern, (int)lds, lds_done));
    const int nt = wino4_tile(p.dil, G::NTUP);
    ConvParams q = p;
    q.ksplit = EPI == 0 ? wino4_ksplit(p) : 1;
    // 1-D grid: (time tiles of the whole batch, rounded up to groups of 8) x row blocks x C-in slices; the kernel maps it XCD-aware
    const int64_t n_tiles = (int64_t)((p.Nout + nt - 1) / nt) * p.batch;
    const int64_t n_blocks = ((n_tiles + 7) / 8) * 8 * (p.CoutP / G::CO_BLK) * q.ksplit;
    TTS_REQUIRE(n_blocks < ((int64_t)1 << 31), "wino4: %lld blocks", (long long)n_blocks);
    dim3 grid((unsigned)n_blocks, 1, 1);
    q.compact = compact_order(p.lens_out, p.batch) ? 1 : 0;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, q);
    TTS_CHECK_HIP(hipGetLastError());
    if (q.ksplit > 1) return launch_splitk_reduce(q, stream);
    return 0;
}

template <int K, int NOCT, int NSTAGE>
static int32_t launch_wino4_cfg(const ConvParams& p, hipStream_t stream) {
    // residual preload (16-byte loads of the lane's quad): dilation 1 -- every c2 conv of a ResBlock, the second conv-FF conv
    // dilation 1: the window as aligned 16-byte vectors (D1)
    if (p.dil == 1) {
        if (p.res != nullptr && wino4_ksplit(p) == 1) return launch_wino4_epi<K, NOCT, NSTAGE, 3, 1>(p, stream);
        return launch_wino4_epi<K, NOCT, NSTAGE, 0, 1>(p, stream);
    }
    // dilation 3 / 5: the window through the per-wave LDS strip (two stages: the strip takes 10 / 20 KB)
    return launch_wino4_epi<K, NOCT, 2, 0, 2>(p, stream);
}

int32_t launch_wino4(const ConvParams& p, hipStream_t stream) {
    if (p.K == 1) return launch_wino4_epi<1, 4, 2, 0, 1>(p, stream);   // 32-channel chunks: 64 MFMAs per wave between barriers, 64 KB ring
    if (p.K == 3) return launch_wino4_cfg<3, 2, 3>(p, stream);       // 16-channel chunks: 48 MFMAs per wave between barriers, 72 KB ring
    if (p.K == 7) return launch_wino4_cfg<7, 1, 2>(p, stream);       // 64 MFMAs, 64 KB ring
    if (p.K == 11) return launch_wino4_cfg<11, 1, 3>(p, stream);     // two phases of 12 / 11 groups: 48 / 44 MFMAs, 72 KB ring
    set_error("wino4: kernel size %d not built (1, 3, 7, 11)", p.K);
    return TTSAMD_EINVAL;
}

int wino4_block_outputs(int dil) { return wino4_tile(dil, 64); }

// C-in slices of a launch that cannot fill the chip by its tiles alone (batch 1: HiFi-GAN's C = 256 stage is 56 blocks): enough for
// ~224 blocks, at least eight chunks per slice, partial sums within the caller's split-K workspace.  1 = no split.
int wino4_ksplit(const ConvParams& p) {
    const int bo = wino4_block_outputs(p.dil);
    const int64_t blocks = (int64_t)((p.Nout + bo - 1) / bo) * (p.CoutP / 64) * p.batch;
    // (more slices / a higher block threshold were measured: no change at batch 1, 2.34-2.40 ms at every setting -- profiles/r6/NOTES.md)
    constexpr int sk_blocks = 192, sk_target = 224;
    if (blocks >= sk_blocks || p.splitk_ws == nullptr) return 1;
    const int n_chunks = p.Cin / (p.K == 1 ? 32 : (p.K == 3 ? 16 : 8));
    const int64_t per = (int64_t)p.batch * p.Cout * p.Nout;
    // at least 8 chunks per slice: a block of 4-5 chunks is mostly prologue and output transform (FastPitch's conv-FF at batch 1 -- 48 / 12
    // blocks x 24 / 96 chunks -- ran 30 % slower on 5 / 19 slices than on the direct kernel's split-K tiles)
    int64_t ks = std::min<int64_t>((sk_target + blocks - 1) / blocks, n_chunks / 8);
    ks = std::min<int64_t>(ks, p.splitk_floats / std::max<int64_t>(per, 1));
    return ks >= 2 ? (int)ks : 1;
}

// groups per octet in the packed weights (k = 11: 23 multiplied + one zero group)
int wino4_groups(int k) { return k == 3 ? 6 : (k == 7 ? 16 : 24); }

// the group filters of one (co, ci) filter g[0..k) in double: sub-filter s -> U0..U5 of (g[3s], g[3s+1], g[3s+2]) (taps past k are
// zero; k = 11: the fourth sub-filter's U5 = 0 takes the zero group's slot), single tap of k = 7 -> four copies (P0 / P6 / P7 / P5)
void wino4_filter_groups(const float* g, int k, float* o) {
    const int nsf = k == 3 ? 1 : (k == 7 ? 2 : 4);
    for (int s = 0; s < nsf; ++s) {
        const double g0 = g[3 * s], g1 = 3 * s + 1 < k ? g[3 * s + 1] : 0.0, g2 = 3 * s + 2 < k ? g[3 * s + 2] : 0.0;
        o[6 * s] = (float)(g0 / 4);
        o[6 * s + 1] = (float)(-(g0 + g1 + g2) / 6);
        o[6 * s + 2] = (float)(-(g0 - g1 + g2) / 6);
        o[6 * s + 3] = (float)(g0 / 24 + g1 / 12 + g2 / 6);
        o[6 * s + 4] = (float)(g0 / 24 - g1 / 12 + g2 / 6);
        o[6 * s + 5] = (float)g2;
    }
    if (k == 7)
        for (int j = 0; j < 4; ++j) o[12 + j] = g[6];
}

// host: torch Conv1d weight [Cout][Cin][K] -> the group filters as an NGQ-tap conv in the engine's packed layout [Cin/8][NGQ][2][CoutP][4]
void pack_wino4_weight(const float* w, int cout, int cin, int k, float* out) {
    const int ng = wino4_groups(k);
    std::vector<float> u((size_t)cout * cin * ng);
    for (int64_t i = 0; i < (int64_t)cout * cin; ++i) wino4_filter_groups(w + i * k, k, u.data() + i * ng);
    pack_conv_weight(u.data(), cout, cin, ng, out);
}

}  // namespace ttsamd
