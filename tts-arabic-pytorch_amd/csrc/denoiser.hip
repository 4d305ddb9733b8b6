// HiFi-GAN bias denoiser (vocoder/hifigan/denoiser.py:32-72) on the GPU:
//   STFT(n_fft 1024, hop 256, hann, center/reflect)  ->  |X| - strength*bias, clamp >= 0, keep
//   phase  ->  ISTFT (window, overlap-add, divide by the window envelope).
// Both DFTs run as GEMMs on the MFMA conv engine (a 1x1 "conv" over the frame axis with the
// windowed DFT matrix as weights): 4.2 MFLOP per frame, 0.7 % of the vocoder's FLOPs.
// torchaudio itself is absent from the reference tree; semantics are those of
// torch.stft/istft(center=True, pad_mode='reflect', onesided, unnormalised).
#include <cmath>
#include <cstring>
#include <vector>

#include "kernels.hpp"

namespace ttsamd {

constexpr int NFFT = 1024, HOP = 256, NBIN = NFFT / 2 + 1;        // 513
constexpr int SPEC_CP = 1152;                                     // padded to 9 x 128 (co tiles) and % 16
struct Denoiser {
    float* dev = nullptr;
    int64_t w_fwd = 0, w_inv = 0, window = 0;
};

static const double kTwoPi = 6.283185307179586476925286766559;

static void hann(std::vector<double>& win) {
    win.resize(NFFT);
    for (int k = 0; k < NFFT; ++k) win[k] = 0.5 - 0.5 * std::cos(kTwoPi * k / NFFT);   // periodic hann
}

// irfft(n=1024, onesided) * window as a 1x1 conv: torch-layout weight [Cout = 1024 (k)][Cin = SPEC_CP]
void build_idft_packed(std::vector<float>& packed_inv, std::vector<float>& window) {
    std::vector<double> win;
    hann(win);
    std::vector<float> wi((size_t)NFFT * SPEC_CP, 0.f);
    for (int f = 0; f < NBIN; ++f) {
        const double cf = (f == 0 || f == NFFT / 2) ? 1.0 : 2.0;
        for (int k = 0; k < NFFT; ++k) {
            const double ang = kTwoPi * (double)((int64_t)f * k % NFFT) / NFFT;
            wi[(size_t)k * SPEC_CP + f] = (float)(cf * std::cos(ang) * win[k] / NFFT);
            wi[(size_t)k * SPEC_CP + NBIN + f] = (float)(-cf * std::sin(ang) * win[k] / NFFT);
        }
    }
    packed_inv.resize((size_t)SPEC_CP * NFFT);
    pack_conv_weight(wi.data(), NFFT, SPEC_CP, 1, packed_inv.data());
    window.resize(NFFT);
    for (int k = 0; k < NFFT; ++k) window[k] = (float)win[k];
}

int32_t denoiser_create(Denoiser** out) {
    TTS_REQUIRE(out, "denoiser_create: null argument");
    auto* h = new Denoiser();
    std::vector<float> blob;
    std::vector<double> win;
    hann(win);
    // forward DFT as a 1x1 conv: torch-layout weight [Cout = SPEC_CP (re | im | zero pad)][Cin = 1024]
    std::vector<float> wf((size_t)SPEC_CP * NFFT, 0.f);
    for (int f = 0; f < NBIN; ++f)
        for (int k = 0; k < NFFT; ++k) {
            const double ang = kTwoPi * (double)((int64_t)f * k % NFFT) / NFFT;
            wf[(size_t)f * NFFT + k] = (float)(win[k] * std::cos(ang));
            wf[(size_t)(NBIN + f) * NFFT + k] = (float)(-win[k] * std::sin(ang));
        }
    h->w_fwd = 0;
    blob.resize((size_t)NFFT * SPEC_CP);
    pack_conv_weight(wf.data(), SPEC_CP, NFFT, 1, blob.data());
    std::vector<float> inv, wnd;
    build_idft_packed(inv, wnd);
    h->w_inv = (int64_t)blob.size();
    blob.insert(blob.end(), inv.begin(), inv.end());
    h->window = (int64_t)blob.size();
    blob.insert(blob.end(), wnd.begin(), wnd.end());
    hipError_t e = hipMalloc((void**)&h->dev, blob.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->dev, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("denoiser_create: upload failed: %s", hipGetErrorString(e));
        if (h->dev) (void)hipFree(h->dev);
        delete h;
        return TTSAMD_EHIP;
    }
    *out = h;
    return 0;
}

void denoiser_destroy(Denoiser* h) {
    if (!h) return;
    if (h->dev) (void)hipFree(h->dev);
    delete h;
}

// frames[b] = nsamples[b] / HOP + 1   (center=True)
__global__ void frame_counts_kernel(const int64_t* __restrict__ ns, int B, int64_t* __restrict__ frames) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b < B) frames[b] = ns[b] / HOP + 1;
}

// X[b][k][t] = x_reflect[b][t*HOP + k - NFFT/2]      (im2col of the centred, reflect-padded signal)
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ wave, int64_t wave_bs,
                                                          const int64_t* __restrict__ ns, int F, int Fs,
                                                          float* __restrict__ X) {
    const int b = blockIdx.z, k = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int n = (int)ns[b];
    const int fr = n / HOP + 1;
    if (t >= F) return;
    float v = 0.f;
    if (t < fr) {
        int i = t * HOP + k - NFFT / 2;
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
        // n <= NFFT/2 (an utterance of 0-2 frames) would reflect past the other end: torch's reflect pad raises there and
        // so do the Python wrappers; here the index is only kept inside the row
        i = min(max(i, 0), max(n - 1, 0));
        v = n > 0 ? wave[(int64_t)b * wave_bs + i] : 0.f;
    }
    X[((int64_t)b * NFFT + k) * Fs + t] = v;
}

// S[b][f | NBIN+f][t] *= max(0, |S| - strength*bias[f]) / |S|     (denoiser.py:68-71)
__global__ __launch_bounds__(256) void spec_gain_kernel(float* __restrict__ S, const float* __restrict__ bias,
                                                        float strength, int F, int Fs) {
    const int b = blockIdx.z, f = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= F) return;
    float* sb = S + (int64_t)b * SPEC_CP * Fs;
    const float re = sb[(int64_t)f * Fs + t], im = sb[(int64_t)(NBIN + f) * Fs + t];
    const float mag = sqrtf(re * re + im * im);
    const float md = fmaxf(mag - bias[f] * strength, 0.f);
    float ore, oim;
    if (mag > 0.f) {
        const float g = md / mag;
        ore = re * g;
        oim = im * g;
    } else {
        ore = md;      // angle(0) = 0
        oim = 0.f;
    }
    sb[(int64_t)f * Fs + t] = ore;
    sb[(int64_t)(NBIN + f) * Fs + t] = oim;
}

// out[b][m] = sum_t Y[b][m + pad - t*HOP][t] / sum_t w^2[m + pad - t*HOP],  m < HOP*frames - (1024 - 2*pad - HOP)...
// frames_b = frames[b]*frames_mul + frames_add; torch.istft(center) : pad = 512, n_out = HOP*(frames-1);
// Vocos 'same' (spectral_ops.py:47-75): pad = 384, n_out = HOP*frames.
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ Y, const float* __restrict__ win,
                                                          const int64_t* __restrict__ frames, int frames_mul,
                                                          int frames_add, int pad, int F,
                                                          float* __restrict__ wave, int64_t wave_bs) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int fr = (int)frames[b] * frames_mul + frames_add;
    const int n_out = (fr - 1) * HOP + NFFT - 2 * pad;
    if (m >= n_out) return;
    const int mp = m + pad;
    const int t_hi = min(fr - 1, mp / HOP);
    const int t_lo = mp >= NFFT ? (mp - NFFT) / HOP + 1 : 0;   // first frame with mp - t*HOP < NFFT
    float acc = 0.f, env = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int k = mp - t * HOP;
        if (k < 0 || k >= NFFT) continue;
        acc += Y[((int64_t)b * NFFT + k) * F + t];
        env += win[k] * win[k];
    }
    wave[(int64_t)b * wave_bs + m] = acc / env;
}

int32_t launch_overlap_add(const float* Y, const float* win, const int64_t* frames, int32_t frames_mul, int32_t frames_add,
                           int32_t pad, int32_t B, int32_t F, int32_t n_max, float* wave, int64_t wave_bs, hipStream_t s) {
    if (n_max <= 0 || B <= 0) return 0;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((n_max + 255) / 256, B), dim3(256), 0, s, Y, win, frames, frames_mul,
                       frames_add, pad, F, wave, wave_bs);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

// |STFT| of frame 0 (denoiser.py:60-64: bias_spec[:, :, 0])
__global__ void mag_frame0_kernel(const float* __restrict__ S, int F, float* __restrict__ out) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= NBIN) return;
    const float re = S[(int64_t)f * F], im = S[(int64_t)(NBIN + f) * F];   /* F = the row stride here */
    out[f] = sqrtf(re * re + im * im);
}

struct DnWs {
    float *X, *S;
    int64_t* frames;
};
// row stride of the frame / spectrum matrices: F rounded up to 4 so that every row is 16-byte aligned -- the conv engine then takes its
// float4 row epilogue and vector loads for the two DFT GEMMs (F = n_max / 256 + 1 is odd for whole-frame lengths: the per-lane epilogue)
static inline int dn_stride(int F) { return (F + 3) & ~3; }

static void carve(Arena& a, int B, int F, DnWs& w) {
    w.X = a.take<float>((int64_t)B * NFFT * dn_stride(F));        // frames matrix, reused for the time-domain frames
    w.S = a.take<float>((int64_t)B * SPEC_CP * dn_stride(F));
    w.frames = a.take<int64_t>(B);
}

int64_t denoiser_workspace_bytes(int32_t B, int32_t n_max) {
    Arena a(nullptr, 0);
    DnWs w;
    carve(a, B, n_max / HOP + 1, w);
    return a.off;
}

static int32_t dft_gemm(const Denoiser* h, bool inverse, const float* x, float* y, const int64_t* frames, int B,
                        int F, hipStream_t s) {
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    const int cin = inverse ? SPEC_CP : NFFT, cout = inverse ? NFFT : SPEC_CP;
    const int Fs = dn_stride(F);
    p.x = x; p.x_bs = (int64_t)cin * Fs; p.x_cs = Fs;
    p.w = h->dev + (inverse ? h->w_inv : h->w_fwd); p.bias = nullptr;
    p.y = y; p.y_bs = (int64_t)cout * Fs; p.y_cs = Fs; p.y_ts = 1;
    p.lens_in = frames; p.lens_out = frames; p.len_in_mul = 1; p.len_out_mul = 1;
    p.Lin = F; p.Nout = F; p.Cin = cin; p.Cout = cout; p.CoutP = cout; p.K = 1;
    p.dil = 1; p.pad = 0; p.n_phase = 1; p.in_slope = 1.f; p.mode = 0; p.div = 1.f; p.batch = B;
    prof_begin(s, 2.0 * cin * cout / HOP);
    const int32_t rc = launch_conv(p, s);
    prof_end(s);
    return rc;
}

static int32_t stft(const Denoiser* h, const float* wave, int64_t wave_bs, const int64_t* ns, int B, int F,
                    const DnWs& w, hipStream_t s) {
    hipLaunchKernelGGL(frame_counts_kernel, dim3((B + 63) / 64), dim3(64), 0, s, ns, B, w.frames);
    hipLaunchKernelGGL(stft_frames_kernel, dim3((F + 255) / 256, NFFT, B), dim3(256), 0, s, wave, wave_bs, ns, F, dn_stride(F), w.X);
    TTS_CHECK_HIP(hipGetLastError());
    return dft_gemm(h, false, w.X, w.S, w.frames, B, F, s);
}

int32_t denoiser_bias_spec(const Denoiser* h, const float* audio, const int64_t* n_dev, int32_t n, float* bias_out,
                           void* ws, int64_t ws_bytes, hipStream_t s) {
    TTS_REQUIRE(h && audio && n_dev && bias_out, "denoiser_bias_spec: null argument");
    TTS_REQUIRE(n > NFFT / 2, "denoiser: reflect padding needs more than %d samples (got %d)", NFFT / 2, n);
    const int F = n / HOP + 1;
    Arena a(ws, ws_bytes);
    DnWs w;
    carve(a, 1, F, w);
    if (!ws || !a.ok) {
        set_error("denoiser: workspace of %lld bytes needed, %lld given", (long long)a.off, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    TTS_TRY(stft(h, audio, n, n_dev, 1, F, w, s));
    hipLaunchKernelGGL(mag_frame0_kernel, dim3((NBIN + 63) / 64), dim3(64), 0, s, w.S, dn_stride(F), bias_out);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

int32_t denoise(const Denoiser* h, float* wave, int64_t wave_bs, const int64_t* nsamples, int32_t B, int32_t n_max,
                const float* bias_spec, float strength, void* ws, int64_t ws_bytes, hipStream_t s) {
    TTS_REQUIRE(h && wave && nsamples && bias_spec, "denoise: null argument");
    TTS_REQUIRE(B >= 1 && n_max > NFFT / 2, "denoise: needs more than %d samples per utterance", NFFT / 2);
    const int F = n_max / HOP + 1;
    Arena a(ws, ws_bytes);
    DnWs w;
    carve(a, B, F, w);
    if (!ws || !a.ok) {
        set_error("denoise: workspace of %lld bytes needed, %lld given", (long long)a.off, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    TTS_TRY(stft(h, wave, wave_bs, nsamples, B, F, w, s));
    hipLaunchKernelGGL(spec_gain_kernel, dim3((F + 255) / 256, NBIN, B), dim3(256), 0, s, w.S, bias_spec, strength, F, dn_stride(F));
    TTS_CHECK_HIP(hipGetLastError());
    TTS_TRY(dft_gemm(h, true, w.S, w.X, w.frames, B, F, s));
    // center=True: frames = n/HOP + 1 (w.frames), pad = NFFT/2, n_out = HOP*(frames-1)
    return launch_overlap_add(w.X, h->dev + h->window, w.frames, 1, 0, NFFT / 2, B, dn_stride(F), n_max, wave, wave_bs, s);
}

}  // namespace ttsamd
