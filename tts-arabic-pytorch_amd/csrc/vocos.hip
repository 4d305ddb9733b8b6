// MelVocos('22k') on the MI355X: ConvNeXt backbone + ISTFT head with "same" padding.
// Replaces vocoder/vocos/pretrained.py:34-93 (MelVocos.__init__/make_denoising_vector/forward),
// models.py:77-89 (VocosBackbone.forward), modules.py:43-60 (ConvNeXtBlock.forward),
// heads.py:41 (ISTFTHead.out) and spectral_ops.py:33-75 (ISTFT.forward, padding="same").
// embed / pwconv1 (+GELU) / pwconv2 (x gamma, + residual) / head.out run on the MFMA conv engine; the
// ISTFT is one 1024-point FFT per frame in LDS (vocos_istft_kernel, fft1024.hpp); depthwise conv,
// LayerNorm (eps 1e-6), exp/cos/sin (+ the transposition to frame-major) and the overlap-add are HBM-bound kernels.  Ragged batches: every layer reads positions >= lens[b]
// as zero, i.e. utterance b equals MelVocos.forward(mel[b:b+1, :, :lens[b]]).
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "fft1024.hpp"

namespace ttsamd {

constexpr int V_NFFT = 1024, V_HOP = 256, V_NBIN = 513, V_SPEC_CP = 1152;

struct VConv {
    int64_t w_off = 0, b_off = -1, w16_off = 0;
    int cin = 0, cout = 0, coutp = 0, k = 0;
};
struct VBlock {
    int64_t dw_w, dw_b, ln_g, ln_b, gamma;
    VConv pw1, pw2;
};
struct Vocos {
    float* dev = nullptr;
    uint16_t* dev16 = nullptr;
    int in_ch = 80, dim = 512, inter = 1536;
    VConv embed, head;
    int64_t n0_g, n0_b, fl_g, fl_b, window, twiddle;
    std::vector<VBlock> blocks;
};

using TensorMap = std::map<std::string, const ttsamd_tensor*>;

static int64_t vnumel(const ttsamd_tensor* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}

struct VBuilder {
    const TensorMap& tm;
    std::vector<float> blob;
    std::vector<uint16_t> blob16;
    int32_t rc = 0;
    explicit VBuilder(const TensorMap& t) : tm(t) {}
    const ttsamd_tensor* get(const std::string& name, int64_t n) {
        if (rc) return nullptr;
        auto it = tm.find(name);
        if (it == tm.end() || vnumel(it->second) != n) {
            set_error("vocos: missing or mis-sized tensor '%s' (expected %lld elements)", name.c_str(), (long long)n);
            rc = TTSAMD_EINVAL;
            return nullptr;
        }
        return it->second;
    }
    int64_t raw(const std::string& name, int64_t n, int64_t pad_to = 0) {
        const ttsamd_tensor* t = get(name, n);
        if (!t) return 0;
        const int64_t off = (int64_t)blob.size();
        blob.insert(blob.end(), t->data, t->data + n);
        if (pad_to > n) blob.resize(off + pad_to, 0.f);
        blob.resize(align_up((int64_t)blob.size(), 64));
        return off;
    }
    VConv conv(const std::string& base, int cin, int cout, int k, int coutp) {
        VConv c;
        c.cin = cin; c.cout = cout; c.k = k; c.coutp = coutp;
        const ttsamd_tensor* w = get(base + ".weight", (int64_t)cin * cout * k);
        if (!w) return c;
        // pack with the padded channel count: rows >= cout are zero weights
        std::vector<float> wp((size_t)coutp * cin * k, 0.f);
        std::memcpy(wp.data(), w->data, (size_t)cout * cin * k * sizeof(float));
        c.w_off = (int64_t)blob.size();
        blob.resize(blob.size() + (size_t)cin * k * coutp);
        pack_conv_weight(wp.data(), coutp, cin, k, blob.data() + c.w_off);
        {
            const int64_t nn = (int64_t)cin * k * coutp;
            c.w16_off = (int64_t)blob16.size();
            blob16.resize(blob16.size() + 2 * nn);
            split_packed_bf16(blob.data() + c.w_off, nn, blob16.data() + c.w16_off);
        }
        blob.resize(align_up((int64_t)blob.size(), 64));
        c.b_off = raw(base + ".bias", cout, coutp);
        return c;
    }
};

int32_t vocos_create(const ttsamd_tensor* weights, int32_t n, int32_t in_ch, int32_t dim, int32_t inter,
                     int32_t n_layers, Vocos** out) {
    TTS_REQUIRE(weights && out, "vocos_create: null argument");
    TTS_REQUIRE(dim % 128 == 0 && inter % 128 == 0 && in_ch % 8 == 0 && n_layers >= 1, "vocos_create: bad dims");
    TensorMap tm;
    for (int i = 0; i < n; ++i) tm[weights[i].name] = &weights[i];
    VBuilder b(tm);
    auto* h = new Vocos();
    h->in_ch = in_ch; h->dim = dim; h->inter = inter;
    h->embed = b.conv("backbone.embed", in_ch, dim, 7, dim);
    h->n0_g = b.raw("backbone.norm.weight", dim);
    h->n0_b = b.raw("backbone.norm.bias", dim);
    for (int i = 0; i < n_layers && b.rc == 0; ++i) {
        const std::string p = "backbone.convnext." + std::to_string(i) + ".";
        VBlock bl;
        bl.dw_w = b.raw(p + "dwconv.weight", (int64_t)dim * 7);
        bl.dw_b = b.raw(p + "dwconv.bias", dim);
        bl.ln_g = b.raw(p + "norm.weight", dim);
        bl.ln_b = b.raw(p + "norm.bias", dim);
        bl.pw1 = b.conv(p + "pwconv1", dim, inter, 1, inter);
        bl.pw2 = b.conv(p + "pwconv2", inter, dim, 1, dim);
        bl.gamma = b.raw(p + "gamma", dim);
        h->blocks.push_back(bl);
    }
    h->fl_g = b.raw("backbone.final_layer_norm.weight", dim);
    h->fl_b = b.raw("backbone.final_layer_norm.bias", dim);
    // head.out: Linear(dim -> n_fft + 2), rows padded to V_SPEC_CP so the spectrum buffer is [re | im | 0]
    h->head = b.conv("head.out", dim, V_NFFT + 2, 1, V_SPEC_CP);
    int32_t rc = b.rc;
    if (rc == 0) {
        std::vector<float> wnd;
        hann_window_1024(wnd);
        h->window = (int64_t)b.blob.size();
        b.blob.insert(b.blob.end(), wnd.begin(), wnd.end());
        h->twiddle = fft1024_append_twiddles(b.blob);            // vocos_istft_kernel
        hipError_t e = hipMalloc((void**)&h->dev, b.blob.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->dev, b.blob.data(), b.blob.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&h->dev16, b.blob16.size() * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemcpy(h->dev16, b.blob16.data(), b.blob16.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("vocos_create: upload failed: %s", hipGetErrorString(e));
            rc = TTSAMD_EHIP;
        }
    }
    if (rc) {
        if (h->dev) (void)hipFree(h->dev);
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

void vocos_destroy(Vocos* h) {
    if (!h) return;
    if (h->dev) (void)hipFree(h->dev);
    if (h->dev16) (void)hipFree(h->dev16);
    delete h;
}

// ISTFT head (pretrained.py:79-90, spectral_ops.py:47-75) in two launches + the overlap-add.
// 1. S[b][t][f] = clamp(exp(O[f][t]) - dn * bias[f], 0, 100) * (cos, sin)(O[513 + f][t]): the head's channel-first output [1026][T] read along t,
//    the complex spectrum written FRAME-major (f contiguous) through a 32 x 33 LDS tile, so that
// 2. one block per frame reads its 513 bins contiguously, builds conj(X) of the Hermitian extension (X[k] = S[k], X[1024 - k] = conj S[k]; the
//    imaginary parts of DC and Nyquist dropped as irfft does), runs ONE 1024-point FFT in LDS (fft1024.hpp) and writes Re / 1024 * window as
//    Y[b][t][k].  Rounds 1-5 ran the inverse DFT as a [1152 -> 1024] GEMM on the conv engine: 300 us per B = 32 call (40x the FLOPs).
__global__ __launch_bounds__(256) void vocos_spec_t_kernel(const float* __restrict__ O, const float* __restrict__ bias, float denoise,
                                                           const int64_t* __restrict__ lens, int T, float2* __restrict__ S) {
    __shared__ float2 tile[32][33];
    const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int len = lens ? min((int)lens[b], T) : T;
    if (t0 >= len) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const float* ob = O + (int64_t)b * V_SPEC_CP * T;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + ty + 8 * r, t = t0 + tx;
        float2 v = make_float2(0.f, 0.f);
        if (f < V_NBIN && t < len) {
            const float lm = ob[(int64_t)f * T + t], ph = ob[(int64_t)(V_NBIN + f) * T + t];
            float mag = expf(lm);
            if (bias) mag -= denoise * bias[f];
            mag = fminf(fmaxf(mag, 0.f), 100.f);
            v = make_float2(mag * cosf(ph), mag * sinf(ph));
        }
        tile[ty + 8 * r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty + 8 * r, f = f0 + tx;
        if (t < len && f < V_NBIN) S[((int64_t)b * T + t) * V_NBIN + f] = tile[tx][ty + 8 * r];
    }
}

__global__ __launch_bounds__(256) void vocos_istft_kernel(const float2* __restrict__ S, const int64_t* __restrict__ lens,
                                                          const float* __restrict__ win, const float2* __restrict__ tw_g, int T,
                                                          float* __restrict__ Y) {
    __shared__ float2 buf[2][V_NFFT];
    __shared__ float2 tw[V_NFFT];
    const int b = blockIdx.y, t = blockIdx.x, i = threadIdx.x;
    if (lens && t >= (int)lens[b]) return;                       // frames the overlap-add never reads
    const float2* sb = S + ((int64_t)b * T + t) * V_NBIN;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = i + 256 * r;
        tw[k] = tw_g[k];
        float2 x = sb[k <= V_NFFT / 2 ? k : V_NFFT - k];         // conj(X)[k]: conj S[k] below Nyquist, S[1024 - k] above
        if (k <= V_NFFT / 2) x.y = -x.y;
        if (k == 0 || k == V_NFFT / 2) x.y = 0.f;
        buf[0][k] = x;
    }
    __syncthreads();
    fft1024_stockham(buf[0], buf[1], tw, i);
    float* yb = Y + ((int64_t)b * T + t) * V_NFFT;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = i + 256 * r;
        yb[k] = buf[1][k].x * (1.0f / V_NFFT) * win[k];
    }
}

// bias_vec[f] = min(exp(O[f][0]), 100)   (pretrained.py:65-69)
__global__ void vocos_bias_kernel(const float* __restrict__ O, int T, float* __restrict__ out) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f < V_NBIN) out[f] = fminf(expf(O[(int64_t)f * T]), 100.f);
}

struct VWs {
    float *x, *d, *h, *o, *y;
};
static void vcarve(const Vocos* h, Arena& a, int B, int T, VWs& w) {
    w.x = a.take<float>((int64_t)B * h->dim * T);
    w.d = a.take<float>((int64_t)B * h->dim * T);
    w.h = a.take<float>((int64_t)B * h->inter * T);
    w.o = a.take<float>((int64_t)B * V_SPEC_CP * T);
    w.y = a.take<float>((int64_t)B * V_NFFT * T);
}
int64_t vocos_workspace_bytes(const Vocos* h, int32_t B, int32_t T) {
    Arena a(nullptr, 0);
    VWs w;
    vcarve(h, a, B, T, w);
    return a.off;
}

static int32_t vconv(const Vocos* h, const VConv& c, const float* x, float* y, const float* res, const float* scale,
                     const int64_t* lens, int B, int T, int act, hipStream_t s) {
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.x_bs = (int64_t)c.cin * T; p.x_cs = T;
    p.w = h->dev + c.w_off; p.bias = h->dev + c.b_off;
    p.w_bf16 = h->dev16 + c.w16_off; p.precision = default_precision();
    p.y = y; p.y_bs = (int64_t)c.coutp * T; p.y_cs = T; p.y_ts = 1;
    p.res = res; p.r_bs = (int64_t)c.coutp * T; p.r_cs = T;
    p.scale = scale;
    p.lens_in = lens; p.lens_out = lens; p.len_in_mul = 1; p.len_out_mul = 1;
    p.Lin = T; p.Nout = T; p.Cin = c.cin; p.Cout = c.coutp; p.CoutP = c.coutp; p.K = c.k;
    p.dil = 1; p.pad = c.k / 2; p.n_phase = 1; p.in_slope = 1.f; p.relu_out = act; p.mode = 0; p.div = 1.f; p.batch = B;
    prof_begin(s, 2.0 * c.cout * c.cin * c.k);
    const int32_t rc = launch_conv(p, s);
    prof_end(s);
    return rc;
}

// backbone + head.out -> w.o [B][V_SPEC_CP][T] holding (log-magnitude | phase | 0)
static int32_t vocos_features(const Vocos* h, const float* mel, const int64_t* lens, int B, int T, const VWs& w,
                              hipStream_t s) {
    const int d = h->dim;
    TTS_TRY(vconv(h, h->embed, mel, w.x, nullptr, nullptr, lens, B, T, 0, s));
    TTS_TRY(launch_layernorm_cf(w.x, w.x, h->dev + h->n0_g, h->dev + h->n0_b, nullptr, 0, B, d, T, s, 1e-6f));
    for (const VBlock& bl : h->blocks) {
        TTS_TRY(launch_dwconv7(w.x, h->dev + bl.dw_w, h->dev + bl.dw_b, lens, B, d, T, w.d, s));
        TTS_TRY(launch_layernorm_cf(w.d, w.d, h->dev + bl.ln_g, h->dev + bl.ln_b, nullptr, 0, B, d, T, s, 1e-6f));
        TTS_TRY(vconv(h, bl.pw1, w.d, w.h, nullptr, nullptr, lens, B, T, 2, s));                   // + GELU
        TTS_TRY(vconv(h, bl.pw2, w.h, w.x, w.x, h->dev + bl.gamma, lens, B, T, 0, s));             // gamma*() + residual
    }
    TTS_TRY(launch_layernorm_cf(w.x, w.d, h->dev + h->fl_g, h->dev + h->fl_b, nullptr, 0, B, d, T, s, 1e-6f));
    return vconv(h, h->head, w.d, w.o, nullptr, nullptr, lens, B, T, 0, s);
}

int32_t vocos_bias_vec(const Vocos* h, float* out513, void* ws, int64_t ws_bytes, hipStream_t s) {
    TTS_REQUIRE(h && out513, "vocos_bias_vec: null argument");
    const int T = 88;   // pretrained.py:61
    Arena a(ws, ws_bytes);
    VWs w;
    vcarve(h, a, 1, T, w);
    float* zero_mel = a.take<float>((int64_t)h->in_ch * T);
    if (!ws || !a.ok) {
        set_error("vocos_bias_vec: workspace of %lld bytes needed, %lld given", (long long)a.off, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    TTS_CHECK_HIP(hipMemsetAsync(zero_mel, 0, (size_t)h->in_ch * T * sizeof(float), s));
    TTS_TRY(vocos_features(h, zero_mel, nullptr, 1, T, w, s));
    hipLaunchKernelGGL(vocos_bias_kernel, dim3((V_NBIN + 63) / 64), dim3(64), 0, s, w.o, T, out513);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int64_t vocos_bias_workspace_bytes(const Vocos* h) {
    Arena a(nullptr, 0);
    VWs w;
    vcarve(h, a, 1, 88, w);
    a.take<float>((int64_t)h->in_ch * 88);
    return a.off;
}

int32_t vocos_forward(const Vocos* h, const float* mel, const int64_t* lens, int32_t B, int32_t T, float denoise,
                      const float* bias_vec, float* wave, void* ws, int64_t ws_bytes, hipStream_t s) {
    TTS_REQUIRE(h && mel && wave && lens && B >= 1 && T >= 1, "vocos_forward: bad argument");
    TTS_REQUIRE(denoise == 0.f || bias_vec, "vocos_forward: denoise > 0 needs bias_vec");
    Arena a(ws, ws_bytes);
    VWs w;
    vcarve(h, a, B, T, w);
    if (!ws || !a.ok) {
        set_error("vocos_forward: workspace of %lld bytes needed, %lld given", (long long)a.off, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    TTS_TRY(vocos_features(h, mel, lens, B, T, w, s));
    // complex spectrum frame-major into the (dead) hidden buffer of the backbone: 513 float2 per frame <= inter floats
    TTS_REQUIRE(2 * V_NBIN <= h->inter, "vocos_forward: the spectrum does not fit the hidden buffer (inter %d)", h->inter);
    float2* S = reinterpret_cast<float2*>(w.h);
    hipLaunchKernelGGL(vocos_spec_t_kernel, dim3((T + 31) / 32, (V_NBIN + 31) / 32, B), dim3(256), 0, s, w.o,
                       denoise != 0.f ? bias_vec : nullptr, denoise, lens, T, S);
    hipLaunchKernelGGL(vocos_istft_kernel, dim3(T, B), dim3(256), 0, s, S, lens, h->dev + h->window,
                       reinterpret_cast<const float2*>(h->dev + h->twiddle), T, w.y);
    TTS_CHECK_HIP(hipGetLastError());
    // overlap-add with "same" trimming (pad = (n_fft - hop) / 2, n_out = hop * frames) over the frame-major time-domain frames
    return launch_overlap_add(w.y, h->dev + h->window, lens, 1, 0, (V_NFFT - V_HOP) / 2, B, T, V_HOP * T, wave,
                              (int64_t)V_HOP * T, s, /*frame_major=*/1);
}

}  // namespace ttsamd
