// Shared between conv_mfma.hip (fp32) and conv_mfma_bf16.hip.
#pragma once
#include "common.hpp"

namespace ttsamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DMAX = 5;  // largest |dilation| the LDS row is sized for

#ifndef TTS_MINWAVES
#define TTS_MINWAVES 2
#endif
#ifndef TTS_NOCT3
#define TTS_NOCT3 1
#endif
// octets (8 input channels) staged per chunk
template <int K> struct OctsOf { static constexpr int NOCT = K == 1 ? 4 : (K == 2 ? 2 : (K == 3 ? TTS_NOCT3 : 1)); };

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

int32_t launch_conv_bf16_any(const ConvParams& p, hipStream_t stream);

}  // namespace ttsamd
