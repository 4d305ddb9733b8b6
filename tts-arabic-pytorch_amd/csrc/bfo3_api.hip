// C ABI of the split-bf16 octet engine's kernel-level entries (include/ttsamd.h "split-bf16 (x3) mode"): the parity tests and the
// layer bench drive single layers through these; the model forwards (hifigan.hip, fastpitch.hip) call the launchers directly.
#include <cstring>

#include "bfo3.hpp"
#include "kernels.hpp"

using namespace ttsamd;

extern "C" {

int32_t ttsamd_bfo3_pack(const float* x, int32_t batch, int32_t channels, int32_t len, float slope, void* out, void* stream) {
    TTS_REQUIRE(x && out && channels % 8 == 0 && slope > 0.f, "bfo3_pack: bad argument (channels %% 8 must be 0, slope > 0)");
    return bfo3_launch_pack(x, batch, channels, len, slope, out, (hipStream_t)stream);
}

int32_t ttsamd_bfo3_unpack(const void* in, int32_t batch, int32_t channels, int32_t len, float slope, float* out, void* stream) {
    TTS_REQUIRE(in && out && channels % 8 == 0 && slope > 0.f, "bfo3_unpack: bad argument (channels %% 8 must be 0, slope > 0)");
    return bfo3_launch_unpack(in, batch, channels, len, slope, out, (hipStream_t)stream);
}

int64_t ttsamd_bfo3_weight_elems(int32_t cout, int32_t cin, int32_t k, int32_t up) {
    if (cout < 1 || cin < 1 || k < 1 || up < 1) return 0;
    return up > 1 ? bfo3_packed_convt_elems(cin, cout, up) : bfo3_packed_conv_elems(cout, cin, k);
}

int32_t ttsamd_bfo3_pack_weight(const float* w, int32_t cout, int32_t cin, int32_t k, int32_t up, uint16_t* out) {
    TTS_REQUIRE(w && out && cout >= 1 && cin >= 1 && k >= 1 && up >= 1, "bfo3_pack_weight: bad argument");
    if (up > 1) {
        TTS_REQUIRE(k == 2 * up && cin % 16 == 0, "bfo3_pack_weight: transposed convs need kernel = 2 * stride and Cin %% 16 == 0");
        bfo3_pack_convt_weight(w, cin, cout, up, out);
    } else {
        bfo3_pack_conv_weight(w, cout, cin, k, out);
    }
    return 0;
}

int32_t ttsamd_bfo3_conv1d(const void* x, const void* w_packed, const float* bias, const void* res, const void* sum_in,
                           const int64_t* lens, int32_t len_mul, int32_t batch, int32_t cin, int32_t cout, int32_t k,
                           int32_t dilation, int32_t up, int32_t len_in, int32_t mode, float div, float res_slope,
                           float out_slope, void* y, float* y_f32, const float* res_f32, void* stream) {
    TTS_REQUIRE(x && w_packed && (y || y_f32) && batch >= 1, "bfo3_conv1d: null argument");
    TTS_REQUIRE(mode >= 0 && mode <= 2 && out_slope >= 0.f && (!res || res_slope > 0.f), "bfo3_conv1d: bad mode / slope");
    BfoConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.y = y; p.w = w_packed; p.bias = bias; p.res = res; p.sum_in = sum_in; p.lens = lens;
    p.len_mul = len_mul; p.Lin = len_in; p.batch = batch; p.Cin = cin; p.Cout = cout; p.K = k; p.dil = dilation; p.up = up;
    p.mode = mode; p.div = div; p.res_slope = res ? res_slope : 1.f; p.out_slope = out_slope;
    p.y_f32 = y_f32; p.res_f32 = res_f32;
    hipStream_t s = (hipStream_t)stream;
    prof_begin(s, 2.0 * cout * cin * k);
    const int32_t rc = up > 1 ? bfo3_launch_convt(p, s) : bfo3_launch_conv(p, s);
    prof_end(s);
    return rc;
}

int32_t ttsamd_bfo3_resblock_pair(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                  const void* sum_in, const int64_t* lens, int32_t len_mul, int32_t batch, int32_t channels,
                                  int32_t k, int32_t dilation, int32_t len, int32_t mode, float div, float in_slope,
                                  float mid_slope, float out_slope, void* y, void* stream) {
    TTS_REQUIRE(x && w1 && b1 && w2 && b2 && y && batch >= 1, "bfo3_resblock_pair: null argument");
    TTS_REQUIRE(mode >= 0 && mode <= 2 && in_slope > 0.f && mid_slope > 0.f && out_slope > 0.f, "bfo3_resblock_pair: bad mode / slope");
    BfoPairParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.y = y; p.sum_in = sum_in; p.w1 = w1; p.w2 = w2; p.b1 = b1; p.b2 = b2; p.lens = lens;
    p.len_mul = len_mul; p.L = len; p.dil = dilation; p.batch = batch; p.mode = mode; p.div = div;
    p.in_slope = in_slope; p.mid_slope = mid_slope; p.out_slope = out_slope;
    hipStream_t s = (hipStream_t)stream;
    prof_begin(s, 2.0 * (2.0 * channels * channels * k));
    const int32_t rc = bfo3_launch_pair(channels, k, p, s);
    prof_end(s);
    return rc;
}

int32_t ttsamd_bfo3_resblock_chain(const void* x, const void* const* w1, const float* const* b1, const void* const* w2,
                                   const float* const* b2, const int32_t* dilations, const void* sum_in, const int64_t* lens,
                                   int32_t len_mul, int32_t batch, int32_t channels, int32_t len, int32_t mode, float div,
                                   float in_slope, float mid_slope, float out_slope, void* y, void* stream) {
    TTS_REQUIRE(x && w1 && b1 && w2 && b2 && dilations && y && batch >= 1, "bfo3_resblock_chain: null argument");
    TTS_REQUIRE(mode >= 0 && mode <= 2 && in_slope > 0.f && mid_slope > 0.f && out_slope > 0.f, "bfo3_resblock_chain: bad mode / slope");
    BfoChainParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.y = y; p.sum_in = sum_in; p.lens = lens;
    for (int m = 0; m < 3; ++m) {
        TTS_REQUIRE(w1[m] && b1[m] && w2[m] && b2[m], "bfo3_resblock_chain: null weight");
        p.w1[m] = w1[m]; p.w2[m] = w2[m]; p.b1[m] = b1[m]; p.b2[m] = b2[m]; p.dil[m] = dilations[m];
    }
    p.len_mul = len_mul; p.L = len; p.batch = batch; p.mode = mode; p.div = div; p.k = 3;
    p.in_slope = in_slope; p.mid_slope = mid_slope; p.out_slope = out_slope;
    hipStream_t s = (hipStream_t)stream;
    prof_begin(s, 3 * 2.0 * (2.0 * channels * channels * 3));
    const int32_t rc = bfo3_launch_chain(channels, p, s);
    prof_end(s);
    return rc;
}

int32_t ttsamd_bfo3_conv_post(const void* x, const float* w, const float* bias, const int64_t* lens, int32_t len_mul,
                              int32_t batch, int32_t channels, int32_t len, float* wave, int64_t wave_stride, void* stream) {
    TTS_REQUIRE(x && w && wave && batch >= 1, "bfo3_conv_post: null argument");
    return bfo3_launch_conv_post(x, w, bias, lens, len_mul, batch, channels, len, wave, wave_stride, (hipStream_t)stream);
}

}  // extern "C"
