// HiFi-GAN bias denoiser (vocoder/hifigan/denoiser.py:32-72) on the GPU:
//   STFT(n_fft 1024, hop 256, hann, center/reflect)  ->  |X| - strength*bias, clamp >= 0, keep
//   phase  ->  ISTFT (window, overlap-add, divide by the window envelope).
// denoise() runs ONE kernel per call for STFT -> gain -> ISTFT frames (denoise_fft_kernel: a block per frame, two radix-4 Stockham
// FFTs of 1024 points in LDS, 0.1 MFLOP per frame) + the overlap-add; rounds 1-5 ran both DFTs as GEMMs on the MFMA conv engine
// (a 1x1 "conv" over the frame axis with the windowed DFT matrix as weights: 4.2 MFLOP per frame, two 59 MB matrices through HBM,
// 0.96 ms per B = 32 call against 0.1x now) -- that path still serves denoiser_bias_spec (one frame, once per model) and Vocos' head.
// torchaudio itself is absent from the reference tree; semantics are those of
// torch.stft/istft(center=True, pad_mode='reflect', onesided, unnormalised).
#include <cmath>
#include <cstring>
#include <vector>

#include "kernels.hpp"
#include "fft1024.hpp"

namespace ttsamd {

constexpr int NFFT = 1024, HOP = 256, NBIN = NFFT / 2 + 1;        // 513
constexpr int SPEC_CP = 1152;                                     // padded to 9 x 128 (co tiles) and % 16
struct Denoiser {
    float* dev = nullptr;
    int64_t w_fwd = 0, window = 0, twiddle = 0;
};

static const double kTwoPi = 6.283185307179586476925286766559;

static void hann(std::vector<double>& win) {
    win.resize(NFFT);
    for (int k = 0; k < NFFT; ++k) win[k] = 0.5 - 0.5 * std::cos(kTwoPi * k / NFFT);   // periodic hann
}

// the periodic hann window of both STFT paths (also Vocos' ISTFT head, vocos.hip)
void hann_window_1024(std::vector<float>& window) {
    std::vector<double> win;
    hann(win);
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
    std::vector<float> wnd;
    hann_window_1024(wnd);
    h->window = (int64_t)blob.size();
    blob.insert(blob.end(), wnd.begin(), wnd.end());
    h->twiddle = fft1024_append_twiddles(blob);                 // denoise_fft_kernel
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

// out[b][m] = sum_t Y[b][m + pad - t*HOP][t] / sum_t w^2[m + pad - t*HOP],  m < HOP*frames - (1024 - 2*pad - HOP)...
// frames_b = frames[b]*frames_mul + frames_add; torch.istft(center) : pad = 512, n_out = HOP*(frames-1);
// Vocos 'same' (spectral_ops.py:47-75): pad = 384, n_out = HOP*frames.
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ Y, const float* __restrict__ win,
                                                          const int64_t* __restrict__ frames, int frames_mul,
                                                          int frames_add, int pad, int F, int ks, int ts,
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
        acc += Y[(int64_t)b * NFFT * F + (int64_t)k * ks + (int64_t)t * ts];      // [b][k][F] (ks = F, ts = 1) or [b][F][k] (ks = 1, ts = NFFT)
        env += win[k] * win[k];
    }
    wave[(int64_t)b * wave_bs + m] = acc / env;
}

int32_t launch_overlap_add(const float* Y, const float* win, const int64_t* frames, int32_t frames_mul, int32_t frames_add,
                           int32_t pad, int32_t B, int32_t F, int32_t n_max, float* wave, int64_t wave_bs, hipStream_t s,
                           int32_t frame_major) {
    if (n_max <= 0 || B <= 0) return 0;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((n_max + 255) / 256, B), dim3(256), 0, s, Y, win, frames, frames_mul,
                       frames_add, pad, F, frame_major ? 1 : F, frame_major ? NFFT : 1, wave, wave_bs);
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

// ---- denoise(): one block per frame.  x_w = window * reflect-padded frame -> X = FFT(x_w) -> X' = X * max(|X| - strength * bias, 0) / |X|
// (denoiser.py:68-71; the gain is real and depends on |X[k]| = |X[1024 - k]| only, so the Hermitian symmetry of a real signal's spectrum
// survives and the complex inverse transform of X' is real: exactly irfft of its one-sided half, whose DC / Nyquist bins are real already)
// -> y = Re(FFT(conj X')) / 1024 * window -> Y[b][t][k], frame-major for the overlap-add.  FFT = five radix-4 Stockham autosort passes
// over two LDS buffers of 1024 complex (thread i: inputs a[i + 256 r], twiddles tw[r k 256 / p], outputs b[4 (i - k) + k + p r], k = i % p;
// checked against numpy.fft in double before it was written: 4e-14).
__global__ __launch_bounds__(256) void denoise_fft_kernel(const float* __restrict__ wave, int64_t wave_bs, const int64_t* __restrict__ ns,
                                                          const float* __restrict__ bias, float strength,
                                                          const float* __restrict__ win, const float2* __restrict__ tw_g, int F,
                                                          float* __restrict__ Y) {
    __shared__ float2 buf[2][NFFT];
    __shared__ float2 tw[NFFT];
    const int b = blockIdx.y, t = blockIdx.x, i = threadIdx.x;
    const int n = (int)ns[b];
    if (t >= n / HOP + 1) return;                               // frames the overlap-add never reads
    const float* wb = wave + (int64_t)b * wave_bs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = i + 256 * r;
        tw[k] = tw_g[k];
        int m = t * HOP + k - NFFT / 2;                         // centred frame, reflect padding (stft_frames_kernel)
        if (m < 0) m = -m;
        if (m >= n) m = 2 * (n - 1) - m;
        m = min(max(m, 0), max(n - 1, 0));
        buf[0][k] = make_float2(n > 0 ? wb[m] * win[k] : 0.f, 0.f);
    }
    __syncthreads();
    fft1024_stockham(buf[0], buf[1], tw, i);                    // X in buf[1]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = i + 256 * r;
        const float2 x = buf[1][k];
        const float mag = sqrtf(x.x * x.x + x.y * x.y);
        const float md = fmaxf(mag - bias[k <= NFFT / 2 ? k : NFFT - k] * strength, 0.f);
        float2 o;
        if (mag > 0.f) {
            const float g = md / mag;
            o = make_float2(x.x * g, -(x.y * g));               // conj(X'): the inverse transform as a forward one
        } else {
            o = make_float2(md, 0.f);                           // angle(0) = 0
        }
        buf[0][k] = o;
    }
    __syncthreads();
    fft1024_stockham(buf[0], buf[1], tw, i);
    float* yb = Y + ((int64_t)b * F + t) * NFFT;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = i + 256 * r;
        yb[k] = buf[1][k].x * (1.0f / NFFT) * win[k];
    }
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

// forward DFT of the frame matrix as a GEMM on the conv engine (denoiser_bias_spec only: one utterance, once per model)
static int32_t dft_gemm(const Denoiser* h, const float* x, float* y, const int64_t* frames, int B,
                        int F, hipStream_t s) {
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    const int cin = NFFT, cout = SPEC_CP;
    const int Fs = dn_stride(F);
    p.x = x; p.x_bs = (int64_t)cin * Fs; p.x_cs = Fs;
    p.w = h->dev + h->w_fwd; p.bias = nullptr;
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
    return dft_gemm(h, w.X, w.S, w.frames, B, F, s);
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
    hipLaunchKernelGGL(frame_counts_kernel, dim3((B + 63) / 64), dim3(64), 0, s, nsamples, B, w.frames);
    // STFT -> gain -> ISTFT frames, one block per frame, time-domain frames into X as [b][frame][k]
    hipLaunchKernelGGL(denoise_fft_kernel, dim3(F, B), dim3(256), 0, s, wave, wave_bs, nsamples, bias_spec, strength, h->dev + h->window,
                       reinterpret_cast<const float2*>(h->dev + h->twiddle), F, w.X);
    TTS_CHECK_HIP(hipGetLastError());
    // center=True: frames = n/HOP + 1 (w.frames), pad = NFFT/2, n_out = HOP*(frames-1)
    return launch_overlap_add(w.X, h->dev + h->window, w.frames, 1, 0, NFFT / 2, B, F, n_max, wave, wave_bs, s, /*frame_major=*/1);
}

}  // namespace ttsamd
