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

int32_t launch_conv_bf16_any(const ConvParams& p, hipStream_t stream);

}  // namespace ttsamd
