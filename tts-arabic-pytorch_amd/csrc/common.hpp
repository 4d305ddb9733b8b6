// Shared host-side helpers of libttsamd (error reporting, launch checks, workspace carving).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/ttsamd.h"

namespace ttsamd {

void set_error(const char* fmt, ...);

#define TTS_CHECK_HIP(expr)                                                         \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            ttsamd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                  \
            return TTSAMD_EHIP;                                                     \
        }                                                                           \
    } while (0)

#define TTS_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            ttsamd::set_error(__VA_ARGS__); \
            return TTSAMD_EINVAL;           \
        }                                   \
    } while (0)

#define TTS_TRY(expr)               \
    do {                            \
        int32_t rc_ = (expr);       \
        if (rc_ != 0) return rc_;   \
    } while (0)

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Switches of CLOSED experiments (tile orders, split-K thresholds, block-count sweeps: profiles/r2..r4) are read only by a library
// built with `make EXTRA=-DTTS_EXPERIMENT` (tools/ A/B scripts); the product library ignores them and runs the measured defaults.
// What stays a run-time switch routes between kernels that BOTH ship and is listed in INTEGRATION.md.
inline const char* exp_env(const char* name) {
#ifdef TTS_EXPERIMENT
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device).  `done` is the instantiation's bitmask of
// devices already opted in; two host threads racing on a first launch both set the attribute (idempotent), devices >= 64 set it on
// every launch.  (Rounds 2-4 kept an unsynchronised bool[16] indexed by device & 15: a data race, and device 16 aliased device 0.)
inline hipError_t lds_opt_in(const void* fn, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return e;
}

// ---- run-time routing options (include/ttsamd.h: ttsamd_set_option / ttsamd_get_option) -----------------------------------------------
// Every switch that routes between kernels / schedules that BOTH ship.  One table (api.hip): the value is seeded ONCE from the
// environment variable TTSAMD_<NAME> when the library is first used, validated (a malformed value is an error of ttsamd_options_check and
// of every ttsamd_set_option call, never silently ignored), and changed afterwards only through ttsamd_set_option -- product code never calls
// getenv on a hot path (a forward used to make several hundred getenv calls, each racing with a concurrent setenv of the host).
// X(name, kind, lo, hi): kind 0 = decimal integer in [lo, hi], 1 = hex mask <= hi.
#define TTS_OPTIONS(X)                                                                                     \
    X(HIFIGAN_STREAMS, 0, 0, 1)   /* ResBlock branches of a HiFi-GAN stage on one / three streams (default: by size) */ \
    X(WINO, 0, 0, 1)              /* fp32 engine: 0 = direct kernels only */                                \
    X(WINO2, 0, 0, 31)            /* F(2,3) decomposition kernel: bit 0 / 1 / 2 = k 3 / 7 / 11, 3 = dilated, 4 = Cout 64 */ \
    X(WINO4, 0, 0, 31)            /* F(4,3) decomposition kernel: bit 0 / 1 / 2 = k 3 / 7 / 11, 3 = dilated, 4 = k 1 (default 31) */ \
    X(FUSED_PAIR, 0, 0, 1)        /* fp32 fused c1 -> c2 pairs: 0 = every pair as two launches */           \
    X(FUSED2, 0, 0, 1)            /* second-generation fused pair off / on */                               \
    X(FUSED2_MASK, 1, 0, 0x1ff)   /* which (C, k) pairs it takes: bit 3 ci + ki (default 00f) */            \
    X(FUSED2_MASK_N1, 1, 0, 0x1ff) /* ... with 128-column direct-arithmetic blocks (default 000) */         \
    X(FUSED2_WB, 0, 0, 2)         /* its Winograd phases: 0 none, 1 phase B, 2 both (default) */            \
    X(CONVT, 0, 0, 1)             /* all-phase transposed conv (convt_mfma.hip) off / on */                 \
    X(BFO, 0, 0, 1)               /* bf16 modes: 0 = the round-2 bf16 conv engine instead of the octet engine */ \
    X(BFO_CHAIN, 0, 0, 1)         /* whole k = 3 ResBlock in one launch (octet engines) */                  \
    X(BFO_CHAIN7, 0, 0, 1)        /* ... and k = 7 (default: small batches only) */                         \
    X(BFO_FF, 0, 0, 1)            /* FastPitch conv-FF / predictors on the octet engine */                  \
    X(BFO_SPLITK, 0, 0, 1)        /* split-K of the octet slab conv */                                      \
    X(BFO_SMALL_TILES, 0, 0, 1)   /* 128-column tiles of the bf16 pair (default: by block count) */         \
    X(BF16_ATTN, 0, 0, 1)         /* bf16 MFMA attention in the bf16 mode */                                \
    X(BF16_PACKED_T, 0, 0, 1)     /* round-2 bf16 engine: packed bf16 c1 -> c2 intermediate */              \
    X(ATT_RA, 0, 1, 4)            /* fp32 attention: 16 * RA queries per block (1, 2, 4) */                 \
    X(ATT_SPLIT, 0, 0, 1)         /* ... one block per (16 queries, key tile) + merge launch */             \
    X(XCD_W, 0, 0, 1)             /* XCD-aware block -> tile map of the direct conv kernel */               \
    X(XCD_WMAX_KB, 0, 0, 1 << 20) /* ... weight footprint per XCD that keeps co-tile classes together */    \
    X(DEEP_SPLITK, 0, 0, 1)       /* 128 x 64 tiles with split K for the deep conv-FF conv at batch 7..13 */ \
    X(TACO_PERSISTENT, 0, 0, 2)   /* Tacotron2 decoder: 0 graph replay, 1 grid-barrier kernel, 2 dataflow kernel */ \
    X(TACO_SEG, 0, 8, 1 << 20)    /* ... decoder steps per persistent launch (default 512) */
enum Opt : int {
#define TTS_OPT_ENUM(name, kind, lo, hi) OPT_##name,
    TTS_OPTIONS(TTS_OPT_ENUM)
#undef TTS_OPT_ENUM
    OPT_COUNT
};
// current value as text (what the environment variable would hold), nullptr = unset (the documented default applies)
const char* opt_str(Opt o);

// Bump allocator over the caller-provided workspace (no hidden hipMalloc on hot calls).
struct Arena {
    char* base;
    int64_t size;
    int64_t off = 0;
    bool ok = true;
    Arena(void* p, int64_t n) : base((char*)p), size(n) {}
    template <typename T>
    T* take(int64_t count) {
        int64_t bytes = align_up(count * (int64_t)sizeof(T), 256);
        if (base != nullptr && off + bytes > size) ok = false;
        T* r = base ? (T*)(base + off) : nullptr;
        off += bytes;
        return r;
    }
};

// ---- generic implicit-GEMM conv1d (conv_mfma.hip) -----------------------------------
struct ConvParams {
    const float* x;      // [B][Cin][*], element (b,c,t) at x + b*x_bs + c*x_cs + t
    int64_t x_bs;
    int32_t x_cs;
    const float* w;      // packed fp32 weights [n_phase][Cin/8][K][2][CoutP][4]
    const void* w_bf16;  // same element order as bf16: plane hi then plane lo (nullptr if not packed)
    const float* w_wino; // k = 3 only: the Winograd F(2,3) filters as a 4-tap conv [Cin/8][4][2][CoutP][4] (conv_wino.hip; nullptr = none)
    const float* w_wino4; // k = 3 / 7 / 11: the Winograd F(4,3) group filters [Cin/8][wino4_groups(k)][2][CoutP][4] (conv_wino4.hip; nullptr = none)
    int32_t precision;   // 0 fp32 MFMA, 1 bf16 MFMA, 2 split bf16 (3 MFMAs per product)
    const float* bias;   // [>=Cout] or nullptr
    float* y;            // (b,co,q) at y + b*y_bs + co*y_cs + q*y_ts + phase
    int64_t y_bs;
    int32_t y_cs, y_ts;
    const float* res;    // residual, same indexing as y (nullptr = none)
    int64_t r_bs;
    int32_t r_cs;
    const int64_t* lens_in;   // per-utterance valid input length = lens_in[b]*len_in_mul (nullptr -> Lin)
    const int64_t* lens_out;  // per-utterance number of outputs  = lens_out[b]*len_out_mul (nullptr -> Nout)
    int32_t len_in_mul, len_out_mul;
    int32_t Lin, Nout;
    int32_t Cin, Cout, CoutP, K;
    int32_t dil, pad;    // input position of tap k for output q: q + k*dil - pad
    int32_t n_phase;     // >1: transposed conv as n_phase polyphase convs (K=2, dil=-1)
    int32_t phase_p;     // transposed conv padding p: delta = (phase+p)/n_phase, pad = -delta
    float in_slope;      // leaky-relu slope applied to the input on load (1 = identity)
    const float* scale;  // per-output-channel factor applied after the bias, before the residual (nullptr = 1)
    int32_t relu_out;    // output activation: 0 none, 1 ReLU, 2 GELU (erf), 3 tanh (after the residual)
    int32_t mode;        // 0: y=v   1: y=y+v   2: y=(y+v)/div
    float div;
    int32_t batch;
    // split-K for problems that cannot fill the chip (batch 1, encoder-side S = 64): when splitk_ws is set and
    // the tile grid has < 192 blocks, the Cin range is cut into `ksplit` slices (extra grid.y factor) that write
    // raw partial sums to splitk_ws[ks][b][Cout][Nout]; a second kernel sums them and applies the epilogue.
    float* splitk_ws;         // >= splitk_floats floats of scratch, or nullptr (never split)
    int64_t splitk_floats;
    int32_t ksplit;           // set by the launcher
    // bf16 mode only (precision 1): activations that only feed the next conv can cross HBM as bf16 in the LDS entry
    // order [B][C/8][2 (kk)][L][4 bf16] (entry (o,kk,t) = channels 8o+kk+{0,2,4,6} at position t), already
    // leaky-relu'd with the consumer's slope and rounded RNE: the consumer's staging is then a plain 8-byte copy.
    int32_t y_packed;         // write y in that layout (requires Cout % 8 == 0, y_ts == 1, mode 0, no phases)
    float pack_slope;         // ... after applying leaky-relu with this slope
    int32_t x_packed;         // read x in that layout (x_cs = positions per row; in_slope is ignored)
    int32_t xcd_w;            // set by the launcher: gcd(n_co_tiles, 8) co-tile classes, one per XCD residue (weight locality; 0 = off)
    int32_t tile_major;       // set by the launcher: blockIdx.x = utterance slot, blockIdx.z = time tile (ragged batches)
    int32_t compact;          // set by the launcher (ragged batches): the (utterance, time tile) pair of a block is looked up in the
                              // utterance-major list of LIVE tiles (live_tile below), so every dead block sits at the end of the grid
    unsigned long long* timing;   // tools/conv_bench -DTTS_TIMING only: [blocks][8] clock samples (nullptr otherwise)
};
constexpr int64_t kSplitKFloats = 4 << 20;   // 16 MB covers every case the launchers pick (direct kernel: < 320 blocks, ~640 blocks wanted; conv_wino4.hip: four C-in slices of a 1 x 256 x 3584 launch = 3.7 M floats)
constexpr int64_t kSplitKFloatsFp = 8 << 20; // FastPitch: 32 MB, the deep conv-FF conv (1536 -> 384) splits K at batch 4..13 too

void conv_log(const char* kind, int K, int cin, int cout, int nout, int batch, int has_res, int mode, int len_mul, int ragged,
              int n_phase);
// block order of a launch (conv_mfma.hip)
bool tile_major_order(const ConvParams& p, unsigned n_tiles);
bool compact_order(const void* lens, int batch);
// Launches the kernel; returns 0 or a negative code.
int32_t launch_conv(const ConvParams& p, hipStream_t stream);
// ConvTranspose1d with all output phases per wave (convt_mfma.hip): takes the polyphase ConvParams of the generic engine
bool convt_supported(const ConvParams& p);
int32_t launch_convt(const ConvParams& p, hipStream_t stream);
// One launch for a c1 -> c2 pair of a C = 32 ResBlock1 with the intermediate in LDS (resblock_fused.hip); x != y.
bool fused_pair_supported(int32_t channels, int32_t k, int32_t dil, int32_t L, const float* x, const float* y);
int32_t launch_fused_pair(int32_t channels, const float* x, float* y, const float* w1, const float* b1, const float* w2,
                          const float* b2, int32_t k, int32_t dil, const int64_t* lens, int32_t len_mul, int32_t L, int32_t batch,
                          int32_t mode, float div, float slope, hipStream_t stream);
// Second generation of the fused pair (resblock_fused2.hip): weights from L2 into a register queue, raw window, 5 barriers per
// block; C = 32 / 64 / 128 at k = 3 / 7 / 11; ntw = 2: 256-column blocks, 1: 128-column blocks.
bool fused_pair2_supported(int32_t channels, int32_t k, int32_t dil, int32_t L, const float* x, const float* y, int32_t ntw);
int32_t launch_fused_pair2(int32_t channels, const float* x, float* y, const float* w1, const float* b1, const float* w2,
                           const float* b2, int32_t k, int32_t dil, const int64_t* lens, int32_t len_mul, int32_t L, int32_t batch,
                           int32_t mode, float div, float slope, int32_t ntw, hipStream_t stream, const float* w2_wino = nullptr,
                           const float* w1_wino = nullptr);
// (w2_wino: conv 2 as Winograd F(2,3) groups -- pack_wino2_weight; C = 32 / 64, ntw = 2: phase B of the pair runs on them; w1_wino:
// conv 1 likewise -> phase A too)
// Winograd F(2,3) path of the k = 3, dilation-1 convs (conv_wino.hip): routing test, launcher, host-side filter transform + packing
int wino_route(const ConvParams& p);       // 0: direct kernel, 1: conv1d_wino_f32, 2: conv1d_wino2_f32, 3: conv1d_wino4_f32 (conv_wino.hip)
int32_t launch_wino(const ConvParams& p, hipStream_t stream);
void pack_wino_weight(const float* w, int cout, int cin, float* out);   // out: cin * 4 * cout_padded(cout) floats
// ... and its generalisation to k = 7 / 11 as sums of F(2,3) sub-filters + single taps (conv_wino2.hip): NG = wino2_groups(k) operand
// groups per octet, weights [Cin/8][NG][2][CoutP][4] (k = 3: identical to pack_wino_weight)
int wino2_groups(int k);
int wino2_block_outputs(int coutp, int dil);     // outputs per block of the kernel launch_wino2 picks
int32_t launch_wino2(const ConvParams& p, hipStream_t stream);
void pack_wino2_weight(const float* w, int cout, int cin, int k, float* out);   // out: cin * wino2_groups(k) * cout_padded(cout) floats
// ... and the F(4,3) decomposition (conv_wino4.hip: 6 / 16 / 23 products per output QUAD at k = 3 / 7 / 11): 64 rows x 64 quads per block
int wino4_groups(int k);                         // groups per octet in the packed weights: 6 / 16 / 24 (k = 11: 23 + one zero group)
int wino4_block_outputs(int dil);
int wino4_ksplit(const ConvParams& p);           // C-in slices the launcher will use for this launch (1 = none)
int32_t launch_splitk_reduce(const ConvParams& q, hipStream_t stream);   // conv_mfma.hip: y = epilogue(sum of the ksplit partial tensors)
int32_t launch_wino4(const ConvParams& p, hipStream_t stream);
void wino4_filter_groups(const float* g, int k, float* o);                      // one (co, ci) filter -> its wino4_groups(k) group filters
void pack_wino4_weight(const float* w, int cout, int cin, int k, float* out);   // out: cin * wino4_groups(k) * cout_padded(cout) floats
// Host-side weight re-layout: torch Conv1d [Cout][Cin][K] -> [Cin][K][CoutP]
void pack_conv_weight(const float* w, int cout, int cin, int k, float* out);
// torch ConvTranspose1d [Cin][Cout][Kt] (Kt = 2u, stride u, padding p) -> [u][Cin][2][CoutP]
void pack_convt_weight(const float* w, int cin, int cout, int kt, int u, int p, float* out);
inline int cout_padded(int cout) { return (int)align_up(cout, 32); }
void split_packed_bf16(const float* packed, int64_t n, uint16_t* out);
// process-wide default MFMA precision of the model forwards (ttsamd_set_precision)
int32_t default_precision();

#ifdef __HIPCC__
// Ragged batches, dead blocks last.  A grid of (time tiles of the LONGEST utterance) x batch has a run of dead blocks
// (tiles past the utterance's own length) behind every utterance; a dead block still needs a free slot (LDS, four
// waves) to launch and exit, and the in-order workgroup dispatcher cannot place the live block queued behind it: with
// utterances 16 % shorter than the padded length on average a stand-alone C = 128 k = 11 launch runs at 124.5 TFLOP/s
// against 137 on a uniform batch (tools/conv_bench RAGGED=auto TTSAMD_DIRECT=0).  Instead, block number `lin` of the
// utterance-major order takes the lin-th LIVE (utterance, tile) pair and all dead blocks sit at the end of the grid
// (133.8 TFLOP/s on that launch; C = 256 k = 11 119 -> 130, C = 64 k = 11 120 -> 126): every wave loads the lengths
// (64 per pass), takes a wave prefix sum of the tile counts and finds its utterance with one ballot -- a few dozen
// cycles next to the length load the kernel does anyway.  Returns false past the last live pair.
// (A persistent variant -- as many blocks as the chip holds, tiles handed out by a device counter -- measured 1-3 %
// BELOW this on the ragged launches and 3-4 % below the plain grid on uniform ones; not kept.)
__device__ __forceinline__ bool live_tile(const int64_t* __restrict__ lens, const int mul, const int n_max, const int tile_w,
                                          const int batch, const unsigned lin, int& b, int& tile) {
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    for (int u0 = 0; u0 < batch; u0 += 64) {
        int n = 0;
        if (u0 + lane < batch) n = (max(min(n_max, (int)lens[u0 + lane] * mul), 0) + tile_w - 1) / tile_w;
        int incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        const unsigned total = (unsigned)__shfl(incl, 63, 64);
        if (lin < base + total) {
            const unsigned long long m = __ballot(base + (unsigned)incl > lin);   // first lane whose inclusive sum passes lin
            const int l = __ffsll((long long)m) - 1;
            b = u0 + l;
            tile = (int)(lin - base) - (__shfl(incl, l, 64) - __shfl(n, l, 64));
            return true;
        }
        base += total;
    }
    return false;
}
#endif

}  // namespace ttsamd
