// extern "C" surface of libttsamd.so (include/ttsamd.h): argument checks, error strings,
// the launch-timing hooks used by bench.py, and thin forwards into the model code.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace ttsamd {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

static int g_precision = 0;
int32_t default_precision() { return g_precision; }

// ---- launch timing (HIP events on the launch stream) -------------------------------
struct Prof {
    std::mutex mu;
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs
    size_t used = 0;
    size_t launches = 0;
    double flops_per_frame = 0.0;
} g_prof;

// Opens an event pair on `s` (a single launch, or a fork..join section of concurrent launches on several streams: the
// pair then measures the wall time of the section on the stream that forks and joins, never a sum of overlapping spans).
void prof_section_begin(hipStream_t s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.used + 2 > g_prof.ev.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        g_prof.ev.push_back(a);
        g_prof.ev.push_back(b);
    }
    (void)hipEventRecord(g_prof.ev[g_prof.used], s);
}

void prof_section_end(hipStream_t s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.used + 2 > g_prof.ev.size()) return;
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], s);
    g_prof.used += 2;
}

// Counts one conv launch (and its algorithmic FLOPs per frame) inside an open section.
void prof_add(double flops_per_frame) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.launches += 1;
    g_prof.flops_per_frame += flops_per_frame;
}

void prof_begin(hipStream_t s, double flops_per_frame) {
    prof_section_begin(s);
    prof_add(flops_per_frame);
}

void prof_end(hipStream_t s) { prof_section_end(s); }

struct HifiGan;
struct FastPitch;
struct Denoiser;
struct Vocos;
int32_t vocos_create(const ttsamd_tensor*, int32_t, int32_t, int32_t, int32_t, int32_t, Vocos**);
void vocos_destroy(Vocos*);
int64_t vocos_workspace_bytes(const Vocos*, int32_t, int32_t);
int64_t vocos_bias_workspace_bytes(const Vocos*);
int32_t vocos_bias_vec(const Vocos*, float*, void*, int64_t, hipStream_t);
int32_t vocos_forward(const Vocos*, const float*, const int64_t*, int32_t, int32_t, float, const float*, float*, void*,
                      int64_t, hipStream_t);
struct Taco2;
int32_t tacotron2_create(const ttsamd_tensor*, int32_t, const ttsamd_tacotron2_cfg*, Taco2**);
void tacotron2_destroy(Taco2*);
int64_t tacotron2_workspace_bytes(const Taco2*, int32_t, int32_t, int32_t);
int32_t tacotron2_infer(const Taco2*, const int64_t*, const int64_t*, const int64_t*, int32_t, int32_t, int32_t, int64_t,
                        float*, int32_t*, float*, float*, int32_t*, void*, int64_t, hipStream_t);
struct Tagger;
int32_t tagger_create(const ttsamd_tensor*, int32_t, const ttsamd_tagger_cfg*, Tagger**);
void tagger_destroy(Tagger*);
int64_t tagger_workspace_bytes(const Tagger*, int32_t, int32_t);
int32_t tagger_forward(const Tagger*, const int64_t*, int32_t, int32_t, float*, void*, int64_t, hipStream_t);
int32_t denoiser_create(Denoiser**);
void denoiser_destroy(Denoiser*);
int64_t denoiser_workspace_bytes(int32_t, int32_t);
int32_t denoiser_bias_spec(const Denoiser*, const float*, const int64_t*, int32_t, float*, void*, int64_t, hipStream_t);
int32_t denoise(const Denoiser*, float*, int64_t, const int64_t*, int32_t, int32_t, const float*, float, void*, int64_t,
                hipStream_t);
int32_t hifigan_create(const ttsamd_tensor*, int32_t, const ttsamd_hifigan_cfg*, HifiGan**);
void hifigan_destroy(HifiGan*);
int64_t hifigan_workspace_bytes(const HifiGan*, int32_t, int32_t);
int32_t hifigan_forward(const HifiGan*, const float*, const int64_t*, int32_t, int32_t, float*, void*, int64_t,
                        hipStream_t);
int32_t fastpitch_create(const ttsamd_tensor*, int32_t, const ttsamd_fastpitch_cfg*, int32_t, FastPitch**);
void fastpitch_destroy(FastPitch*);
int64_t fastpitch_encode_workspace_bytes(const FastPitch*, int32_t, int32_t);
int64_t fastpitch_decode_workspace_bytes(const FastPitch*, int32_t, int32_t);
int32_t fastpitch_encode(const FastPitch*, const int64_t*, int32_t, int32_t, int32_t, float, const float*,
                         const float*, const float*, float, float, float, float*, float*, float*, float*, int64_t*,
                         int64_t*, void*, int64_t, hipStream_t);
void fastpitch_set_batch_mode(const FastPitch*, int);
int32_t fastpitch_decode(const FastPitch*, float*, const int64_t*, int32_t, int32_t, float*, void*, int64_t,
                         hipStream_t);

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int cout, int cin, int k, int cp,
                                        float* __restrict__ out) {
    // same layout as pack_conv_weight (conv_mfma.hip): [cin/8][k][2][cp][4]
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)cin * k * cp;
    if (i >= n) return;
    const int pq = (int)(i % 4);
    const int co = (int)((i / 4) % cp);
    const int kk = (int)((i / (4 * (int64_t)cp)) % 2);
    const int t = (int)((i / (8 * (int64_t)cp)) % k);
    const int o = (int)(i / (8 * (int64_t)cp * k));
    out[i] = co < cout ? w[((int64_t)co * cin + (8 * o + 2 * pq + kk)) * k + t] : 0.f;
}

__global__ void split_bf16_kernel(const float* __restrict__ packed, int64_t n, unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float f = packed[i];
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    const unsigned h = u >> 16;
    const float r = f - __uint_as_float(h << 16);
    unsigned v = __float_as_uint(r);
    v += 0x7fffu + ((v >> 16) & 1u);
    out[i] = (unsigned short)h;
    out[n + i] = (unsigned short)(v >> 16);
}

// ---- run-time routing options (common.hpp: TTS_OPTIONS) --------------------------------------------------------------------------------
namespace {
struct OptDesc { const char* name; int kind; long long lo, hi; };
const OptDesc kOpts[OPT_COUNT] = {
#define TTS_OPT_DESC(name, kind, lo, hi) {"TTSAMD_" #name, kind, lo, hi},
    TTS_OPTIONS(TTS_OPT_DESC)
#undef TTS_OPT_DESC
};
std::atomic<const char*> g_opt[OPT_COUNT];       // interned strings, never freed (a few bytes per ttsamd_set_option call)
std::once_flag g_opt_once;
std::string g_opt_env_error;                     // first malformed TTSAMD_<NAME> of the environment (ttsamd_options_check)

bool opt_valid(const OptDesc& d, const char* v) {
    if (!v || !*v) return false;
    char* end = nullptr;
    errno = 0;
    const long long x = std::strtoll(v, &end, d.kind == 1 ? 16 : 10);
    return errno == 0 && end != v && *end == '\0' && x >= d.lo && x <= d.hi;
}
void opt_init() {
    std::call_once(g_opt_once, [] {
        for (int i = 0; i < OPT_COUNT; ++i) {
            const char* e = std::getenv(kOpts[i].name);        // the ONE getenv per option and process
            if (!e || !*e) continue;
            if (!opt_valid(kOpts[i], e)) {
                if (g_opt_env_error.empty()) {
                    char buf[256];
                    std::snprintf(buf, sizeof buf, "%s='%s' in the environment is not a valid value (%s in [%llx, %llx])", kOpts[i].name, e,
                                  kOpts[i].kind == 1 ? "hex mask" : "integer", kOpts[i].lo, kOpts[i].hi);
                    g_opt_env_error = buf;
                }
                continue;
            }
            g_opt[i].store(strdup(e), std::memory_order_release);
        }
    });
}
int opt_find(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (std::strcmp(name, kOpts[i].name) == 0 || std::strcmp(name, kOpts[i].name + 7) == 0) return i;
    return -1;
}
}  // namespace

const char* opt_str(Opt o) {
    opt_init();
    return g_opt[o].load(std::memory_order_acquire);
}

}  // namespace ttsamd

using namespace ttsamd;

extern "C" {

const char* ttsamd_last_error(void) { return g_err.c_str(); }
int32_t ttsamd_version(void) { return TTSAMD_ABI_VERSION; }

int32_t ttsamd_set_option(const char* name, const char* value) {
    opt_init();
    const int i = opt_find(name);
    TTS_REQUIRE(i >= 0, "set_option: unknown option '%s' (ttsamd_option_name lists them)", name ? name : "(null)");
    if (!value || !*value) {
        g_opt[i].store(nullptr, std::memory_order_release);
        return 0;
    }
    TTS_REQUIRE(opt_valid(kOpts[i], value), "set_option: %s='%s' is not a valid value (%s in [%llx, %llx])", kOpts[i].name, value,
                kOpts[i].kind == 1 ? "hex mask" : "integer", kOpts[i].lo, kOpts[i].hi);
    g_opt[i].store(strdup(value), std::memory_order_release);
    return 0;
}
int32_t ttsamd_get_option(const char* name, char* value, int32_t capacity) {
    opt_init();
    const int i = opt_find(name);
    TTS_REQUIRE(i >= 0 && value && capacity > 0, "get_option: unknown option '%s' or no buffer", name ? name : "(null)");
    const char* v = g_opt[i].load(std::memory_order_acquire);
    std::snprintf(value, (size_t)capacity, "%s", v ? v : "");
    return 0;
}
const char* ttsamd_option_name(int32_t index) { return (index >= 0 && index < OPT_COUNT) ? kOpts[index].name : nullptr; }
int32_t ttsamd_options_check(void) {
    opt_init();
    TTS_REQUIRE(g_opt_env_error.empty(), "%s", g_opt_env_error.c_str());
    return 0;
}

int32_t ttsamd_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return std::string(prop.gcnArchName).rfind("gfx950", 0) == 0 ? 1 : 0;
}

int32_t ttsamd_hifigan_create(const ttsamd_tensor* weights, int32_t n, const ttsamd_hifigan_cfg* cfg, void** handle) {
    HifiGan* h = nullptr;
    const int32_t rc = hifigan_create(weights, n, cfg, &h);
    if (rc == 0) *handle = h;
    return rc;
}
int32_t ttsamd_hifigan_destroy(void* handle) {
    hifigan_destroy((HifiGan*)handle);
    return 0;
}
int64_t ttsamd_hifigan_workspace_bytes(void* handle, int32_t batch, int32_t t_max) {
    if (!handle || batch < 1 || t_max < 1) return 0;
    return hifigan_workspace_bytes((HifiGan*)handle, batch, t_max);
}
int32_t ttsamd_hifigan_forward(void* handle, const float* mel, const int64_t* lens, int32_t batch, int32_t t_max,
                               float* wave, void* workspace, int64_t workspace_bytes, void* stream) {
    return hifigan_forward((HifiGan*)handle, mel, lens, batch, t_max, wave, workspace, workspace_bytes,
                           (hipStream_t)stream);
}

int32_t ttsamd_fastpitch_create(const ttsamd_tensor* weights, int32_t n, const ttsamd_fastpitch_cfg* cfg,
                                void** handle) {
    FastPitch* h = nullptr;
    const int32_t rc = fastpitch_create(weights, n, cfg, /*pos_cap=*/16384, &h);
    if (rc == 0) *handle = h;
    return rc;
}
int32_t ttsamd_fastpitch_destroy(void* handle) {
    fastpitch_destroy((FastPitch*)handle);
    return 0;
}
int64_t ttsamd_fastpitch_encode_workspace_bytes(void* handle, int32_t batch, int32_t n_tokens) {
    if (!handle || batch < 1 || n_tokens < 1) return 0;
    return fastpitch_encode_workspace_bytes((FastPitch*)handle, batch, n_tokens);
}
int64_t ttsamd_fastpitch_decode_workspace_bytes(void* handle, int32_t batch, int32_t t_max) {
    if (!handle || batch < 1 || t_max < 1) return 0;
    return fastpitch_decode_workspace_bytes((FastPitch*)handle, batch, t_max);
}
int32_t ttsamd_fastpitch_encode(void* handle, const int64_t* ids, int32_t batch, int32_t n_tokens, int32_t speaker,
                                float pace, const float* dur_tgt, const float* pitch_tgt, const float* energy_tgt,
                                float pitch_mul, float pitch_add, float max_duration, float* enc_cond,
                                float* dur_pred, float* pitch_pred, float* energy_pred, int64_t* reps,
                                int64_t* dec_lens, void* workspace, int64_t workspace_bytes, void* stream) {
    return fastpitch_encode((FastPitch*)handle, ids, batch, n_tokens, speaker, pace, dur_tgt, pitch_tgt, energy_tgt,
                            pitch_mul, pitch_add, max_duration, enc_cond, dur_pred, pitch_pred, energy_pred, reps,
                            dec_lens, workspace, workspace_bytes, (hipStream_t)stream);
}
int32_t ttsamd_length_regulate(const float* enc, const int64_t* reps, int32_t batch, int32_t n_tokens,
                               int32_t channels, int32_t t_max, float* out, int32_t* idx, void* stream) {
    TTS_REQUIRE(enc && reps && out && batch >= 1 && n_tokens >= 1 && channels >= 1 && t_max >= 0,
                "length_regulate: bad argument");
    return launch_regulate_gather(enc, reps, nullptr, 0, batch, n_tokens, channels, t_max, out, idx,
                                  (hipStream_t)stream);
}
int32_t ttsamd_fastpitch_set_batch_mode(void* handle, int32_t mode) {
    TTS_REQUIRE(handle != nullptr, "fastpitch_set_batch_mode: null handle");
    TTS_REQUIRE(mode == 0 || mode == 1, "fastpitch_set_batch_mode: mode %d (0 = padded-batch arithmetic, 1 = every utterance alone)", mode);
    fastpitch_set_batch_mode((const FastPitch*)handle, mode);
    return 0;
}
int32_t ttsamd_fastpitch_decode(void* handle, float* x, const int64_t* dec_lens, int32_t batch, int32_t t_max,
                                float* mel, void* workspace, int64_t workspace_bytes, void* stream) {
    return fastpitch_decode((FastPitch*)handle, x, dec_lens, batch, t_max, mel, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

int32_t ttsamd_denoiser_create(void** handle) {
    Denoiser* h = nullptr;
    const int32_t rc = denoiser_create(&h);
    if (rc == 0) *handle = h;
    return rc;
}
int32_t ttsamd_denoiser_destroy(void* handle) {
    denoiser_destroy((Denoiser*)handle);
    return 0;
}
int64_t ttsamd_denoiser_workspace_bytes(int32_t batch, int32_t n_max) {
    if (batch < 1 || n_max < 1) return 0;
    return denoiser_workspace_bytes(batch, n_max);
}
int32_t ttsamd_denoiser_bias_spec(void* handle, const float* audio, const int64_t* n_dev, int32_t n,
                                  float* bias_spec, void* workspace, int64_t workspace_bytes, void* stream) {
    return denoiser_bias_spec((Denoiser*)handle, audio, n_dev, n, bias_spec, workspace, workspace_bytes,
                              (hipStream_t)stream);
}
int32_t ttsamd_denoise(void* handle, float* wave, int64_t wave_stride, const int64_t* nsamples, int32_t batch,
                       int32_t n_max, const float* bias_spec, float strength, void* workspace,
                       int64_t workspace_bytes, void* stream) {
    return denoise((Denoiser*)handle, wave, wave_stride, nsamples, batch, n_max, bias_spec, strength, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

int32_t ttsamd_vocos_create(const ttsamd_tensor* weights, int32_t n, int32_t input_channels, int32_t dim,
                            int32_t intermediate_dim, int32_t num_layers, void** handle) {
    Vocos* h = nullptr;
    const int32_t rc = vocos_create(weights, n, input_channels, dim, intermediate_dim, num_layers, &h);
    if (rc == 0) *handle = h;
    return rc;
}
int32_t ttsamd_vocos_destroy(void* handle) {
    vocos_destroy((Vocos*)handle);
    return 0;
}
int64_t ttsamd_vocos_workspace_bytes(void* handle, int32_t batch, int32_t t_max) {
    if (!handle || batch < 1 || t_max < 1) return 0;
    const int64_t a = vocos_workspace_bytes((Vocos*)handle, batch, t_max), b = vocos_bias_workspace_bytes((Vocos*)handle);
    return a > b ? a : b;
}
int32_t ttsamd_vocos_bias_vec(void* handle, float* bias_vec, void* workspace, int64_t workspace_bytes, void* stream) {
    return vocos_bias_vec((Vocos*)handle, bias_vec, workspace, workspace_bytes, (hipStream_t)stream);
}
int32_t ttsamd_vocos_forward(void* handle, const float* mel, const int64_t* lens, int32_t batch, int32_t t_max,
                             float denoise, const float* bias_vec, float* wave, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    return vocos_forward((Vocos*)handle, mel, lens, batch, t_max, denoise, bias_vec, wave, workspace, workspace_bytes,
                         (hipStream_t)stream);
}

int32_t ttsamd_tacotron2_create(const ttsamd_tensor* weights, int32_t n, const ttsamd_tacotron2_cfg* cfg, void** handle) {
    TTS_REQUIRE(handle, "tacotron2_create: null handle");
    Taco2* h = nullptr;
    const int32_t rc = tacotron2_create(weights, n, cfg, &h);
    if (rc == 0) *handle = h;
    return rc;
}
int32_t ttsamd_tacotron2_destroy(void* handle) {
    tacotron2_destroy((Taco2*)handle);
    return 0;
}
int64_t ttsamd_tacotron2_workspace_bytes(void* handle, int32_t batch, int32_t n_tokens, int32_t max_step) {
    if (!handle || batch < 1 || n_tokens < 1 || max_step < 1) return 0;
    return tacotron2_workspace_bytes((Taco2*)handle, batch, n_tokens, max_step);
}
int32_t ttsamd_tacotron2_infer(void* handle, const int64_t* tokens, const int64_t* lengths, const int64_t* speaker_ids,
                               int32_t batch, int32_t n_tokens, int32_t max_step, int64_t dropout_seed, float* mel_post,
                               int32_t* mel_lens, float* alignments, float* mel_raw, int32_t* n_steps, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    return tacotron2_infer((Taco2*)handle, tokens, lengths, speaker_ids, batch, n_tokens, max_step, dropout_seed, mel_post,
                           mel_lens, alignments, mel_raw, n_steps, workspace, workspace_bytes, (hipStream_t)stream);
}

int32_t ttsamd_tagger_create(const ttsamd_tensor* weights, int32_t n, const ttsamd_tagger_cfg* cfg, void** handle) {
    TTS_REQUIRE(handle, "tagger_create: null handle");
    Tagger* h = nullptr;
    const int32_t rc = tagger_create(weights, n, cfg, &h);
    if (rc == 0) *handle = h;
    return rc;
}
int32_t ttsamd_tagger_destroy(void* handle) {
    tagger_destroy((Tagger*)handle);
    return 0;
}
int64_t ttsamd_tagger_workspace_bytes(void* handle, int32_t batch, int32_t n_chars) {
    if (!handle || batch < 1 || n_chars < 1) return 0;
    return tagger_workspace_bytes((Tagger*)handle, batch, n_chars);
}
int32_t ttsamd_tagger_forward(void* handle, const int64_t* ids, int32_t batch, int32_t n_chars, float* probs,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    return tagger_forward((Tagger*)handle, ids, batch, n_chars, probs, workspace, workspace_bytes, (hipStream_t)stream);
}

int64_t ttsamd_conv1d_packed_floats(int32_t cout, int32_t cin, int32_t k) {
    // fp32 packed + bf16 hi/lo planes (+ k = 3 / 7 / 11: the Winograd group filters, conv_wino.hip / conv_wino2.hip)
    return 2 * (int64_t)cin * k * cout_padded(cout) +
           ((k == 3 || k == 7 || k == 11) ? (int64_t)cin * (wino2_groups(k) + wino4_groups(k)) * cout_padded(cout) : 0);
}

// [Cout][Cin][k] -> the F(4,3) group filters in the packed layout [cin/8][NGQ][2][cp][4] (same values as pack_wino4_weight, conv_wino4.hip)
__global__ void pack_wino4_weight_kernel(const float* __restrict__ w, int cout, int cin, int k, int cp, float* __restrict__ out) {
    const int ng = k == 3 ? 6 : (k == 7 ? 16 : 24), nsf = k == 3 ? 1 : (k == 7 ? 2 : 4);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)cin * ng * cp;
    if (i >= n) return;
    const int pq = (int)(i & 3);
    const int64_t r = i >> 2;
    const int co = (int)(r % cp);
    const int64_t r2 = r / cp;
    const int kk = (int)(r2 & 1);
    const int64_t r3 = r2 >> 1;
    const int t = (int)(r3 % ng), o = (int)(r3 / ng);
    float v = 0.f;
    if (co < cout) {
        const float* g = w + ((int64_t)co * cin + (8 * o + 2 * pq + kk)) * k;
        if (t < 6 * nsf) {
            const int s = t / 6, j = t % 6;
            const double g0 = g[3 * s], g1 = 3 * s + 1 < k ? g[3 * s + 1] : 0.0, g2 = 3 * s + 2 < k ? g[3 * s + 2] : 0.0;
            v = j == 0 ? (float)(g0 / 4) : (j == 1 ? (float)(-(g0 + g1 + g2) / 6) : (j == 2 ? (float)(-(g0 - g1 + g2) / 6) :
                (j == 3 ? (float)(g0 / 24 + g1 / 12 + g2 / 6) : (j == 4 ? (float)(g0 / 24 - g1 / 12 + g2 / 6) : (float)g2))));
        } else {
            v = g[6];                       // k = 7: the single tap, four copies (planes P0 / P6 / P7 / P5)
        }
    }
    out[i] = v;
}

// [Cout][Cin][k] -> the Winograd group filters in the packed NG-tap layout [cin/8][NG][2][cp][4] (same values as pack_wino2_weight:
// sub-filter s: g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2; single tap t: g_t, -g_t)
__global__ void pack_wino2_weight_kernel(const float* __restrict__ w, int cout, int cin, int k, int cp, float* __restrict__ out) {
    const int ns = k / 3, ng = 4 * ns + 2 * (k - 3 * ns);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)cin * ng * cp;
    if (i >= n) return;
    const int pq = (int)(i & 3);
    const int64_t r = i >> 2;
    const int co = (int)(r % cp);
    const int64_t r2 = r / cp;
    const int kk = (int)(r2 & 1);
    const int64_t r3 = r2 >> 1;
    const int t = (int)(r3 % ng), o = (int)(r3 / ng);
    float v = 0.f;
    if (co < cout) {
        const float* g = w + ((int64_t)co * cin + (8 * o + 2 * pq + kk)) * k;
        if (t < 4 * ns) {
            const int s = t >> 2, j = t & 3;
            const double g0 = g[3 * s], g1 = g[3 * s + 1], g2 = g[3 * s + 2];
            v = j == 0 ? (float)g0 : (j == 1 ? (float)((g0 + g1 + g2) * 0.5) : (j == 2 ? (float)((g0 - g1 + g2) * 0.5) : (float)g2));
        } else {
            const int l = (t - 4 * ns) >> 1;
            v = ((t - 4 * ns) & 1) ? -g[3 * ns + l] : g[3 * ns + l];
        }
    }
    out[i] = v;
}

int32_t ttsamd_conv1d_ex(const float* x, const float* w, const float* bias, const float* res, const int64_t* lens, int32_t batch,
                         int32_t cin, int32_t cout, int32_t k, int32_t dilation, int32_t lin, float in_slope,
                         int32_t relu_out, int32_t mode, float div, float* y, float* packed, void* stream) {
    TTS_REQUIRE(x && w && y && packed, "conv1d: null argument");
    TTS_REQUIRE(mode >= 0 && mode <= 2 && (mode != 2 || div != 0.f), "conv1d: bad mode / div");
    // the residual-preload epilogues apply the activation to y_prev + res + conv + b: only mode 0 has the documented meaning with relu_out
    TTS_REQUIRE(relu_out == 0 || mode == 0, "conv1d: relu_out with an accumulate mode (mode %d) is not defined", mode);
    hipStream_t s = (hipStream_t)stream;
    const int cp = cout_padded(cout);
    const int64_t n = (int64_t)cin * k * cp;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, cout, cin, k,
                       cp, packed);
    TTS_CHECK_HIP(hipGetLastError());
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.x_bs = (int64_t)cin * lin; p.x_cs = lin;
    p.w = packed; p.bias = bias;
    p.precision = default_precision();
    if (p.precision != 0) {
        unsigned short* planes = reinterpret_cast<unsigned short*>(packed + n);
        hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, packed, n, planes);
        TTS_CHECK_HIP(hipGetLastError());
        p.w_bf16 = planes;
    }
    if ((k == 3 || k == 7 || k == 11) && p.precision == 0 && cin % 8 == 0) {
        float* wino = packed + 2 * n;
        const int64_t nw = (int64_t)cin * wino2_groups(k) * cp;
        hipLaunchKernelGGL(pack_wino2_weight_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, w, cout, cin, k, cp, wino);
        TTS_CHECK_HIP(hipGetLastError());
        p.w_wino = wino;
        float* wino4 = wino + nw;
        const int64_t nw4 = (int64_t)cin * wino4_groups(k) * cp;
        hipLaunchKernelGGL(pack_wino4_weight_kernel, dim3((unsigned)((nw4 + 255) / 256)), dim3(256), 0, s, w, cout, cin, k, cp, wino4);
        TTS_CHECK_HIP(hipGetLastError());
        p.w_wino4 = wino4;
    }
    p.y = y; p.y_bs = (int64_t)cout * lin; p.y_cs = lin; p.y_ts = 1;
    p.lens_in = lens; p.lens_out = lens; p.len_in_mul = 1; p.len_out_mul = 1;
    p.Lin = lin; p.Nout = lin; p.Cin = cin; p.Cout = cout; p.CoutP = cp; p.K = k;
    p.dil = dilation; p.pad = (k * dilation - dilation) / 2;
    p.res = res; p.r_bs = (int64_t)cout * lin; p.r_cs = lin;
    p.n_phase = 1; p.in_slope = in_slope; p.relu_out = relu_out; p.mode = mode; p.div = div; p.batch = batch;
    prof_begin(s, 2.0 * cout * cin * k);
    const int32_t rc = launch_conv(p, s);
    prof_end(s);
    return rc;
}

int32_t ttsamd_conv1d(const float* x, const float* w, const float* bias, const int64_t* lens, int32_t batch,
                      int32_t cin, int32_t cout, int32_t k, int32_t dilation, int32_t lin, float in_slope,
                      int32_t relu_out, float* y, float* packed, void* stream) {
    return ttsamd_conv1d_ex(x, w, bias, nullptr, lens, batch, cin, cout, k, dilation, lin, in_slope, relu_out, 0, 1.f, y, packed, stream);
}

int64_t ttsamd_resblock_pair_packed_floats(int32_t channels, int32_t k, int32_t variant) {
    // the two direct packings + (variant 4: conv 2, variant 5: both convs) as Winograd F(2,3) groups
    if (channels < 1 || k < 1) return 0;
    const int64_t n = 2 * (int64_t)channels * k * channels;
    const int nwino = (variant == 4 ? 1 : (variant == 5 ? 2 : 0));
    return n + ((k == 3 || k == 7 || k == 11) ? nwino * (int64_t)channels * wino2_groups(k) * channels : 0);
}

int32_t ttsamd_resblock_pair(const float* x, float* y, const float* w1, const float* b1, const float* w2, const float* b2,
                             int32_t channels, int32_t k, int32_t dil, const int64_t* lens, int32_t len_mul, int32_t L,
                             int32_t batch, int32_t mode, float div, float slope, int32_t variant, float* packed, int64_t packed_floats,
                             void* stream) {
    TTS_REQUIRE(x && y && w1 && b1 && w2 && b2 && packed && channels % 32 == 0 && k >= 1 && batch >= 1 && L >= 1 && len_mul >= 1 &&
                mode >= 0 && mode <= 2, "resblock_pair: bad argument");
    TTS_REQUIRE(packed_floats >= ttsamd_resblock_pair_packed_floats(channels, k, variant),
                "resblock_pair: `packed` holds %lld floats, variant %d at C = %d, k = %d needs %lld (ttsamd_resblock_pair_packed_floats)",
                (long long)packed_floats, variant, channels, k, (long long)ttsamd_resblock_pair_packed_floats(channels, k, variant));
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)channels * k * channels;
    for (int i = 0; i < 2; ++i) {
        hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, i == 0 ? w1 : w2, channels,
                           channels, k, channels, packed + i * n);
        TTS_CHECK_HIP(hipGetLastError());
    }
    const double fl = 2.0 * (2.0 * channels * channels * k);
    int32_t rc;
    prof_begin(s, fl);
    if (variant == 1) {
        rc = launch_fused_pair(channels, x, y, packed, b1, packed + n, b2, k, dil, lens, len_mul, L, batch, mode, div, slope, s);
    } else if (variant >= 2 && variant <= 5) {
        const float *w2w = nullptr, *w1w = nullptr;
        if (variant >= 4) {       // 256-column blocks with phase B (4) or both phases (5) on Winograd F(2,3): the groups behind the direct packings
            TTS_REQUIRE((k == 3 || k == 7 || k == 11) && (channels <= 64 || (variant == 5 && channels == 128)),
                        "resblock_pair: variants 4 / 5 are built for C = 32 / 64 (5: and 128), k = 3 / 7 / 11");
            const int64_t nw = (int64_t)channels * wino2_groups(k) * channels;
            for (int i = 0; i < (variant == 5 ? 2 : 1); ++i) {
                float* wino = packed + 2 * n + i * nw;
                hipLaunchKernelGGL(pack_wino2_weight_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, i == 0 ? w2 : w1, channels,
                                   channels, k, channels, wino);
                TTS_CHECK_HIP(hipGetLastError());
                (i == 0 ? w2w : w1w) = wino;
            }
        }
        rc = launch_fused_pair2(channels, x, y, packed, b1, packed + n, b2, k, dil, lens, len_mul, L, batch, mode, div, slope,
                                variant == 3 ? 1 : 2, s, w2w, w1w);
    } else if (variant == -1) {
        rc = 0;                                    // the two weight re-layout launches only (tools/fused_pair_bench.py subtracts them)
    } else {
        set_error("resblock_pair: variant %d (1: first generation, 2 / 3: second generation with 256- / 128-column blocks, 4 / 5: 256 columns + Winograd phase B / both phases)", variant);
        rc = TTSAMD_EINVAL;
    }
    prof_end(s);
    return rc;
}

int32_t ttsamd_set_precision(int32_t precision) {
    TTS_REQUIRE(precision >= 0 && precision <= 2, "set_precision: 0 = fp32 MFMA, 1 = bf16 MFMA, 2 = split-bf16 MFMA");
    g_precision = precision;
    return 0;
}
int32_t ttsamd_get_precision(void) { return g_precision; }

int32_t ttsamd_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = on != 0;
    g_prof.used = 0;
    g_prof.launches = 0;
    g_prof.flops_per_frame = 0.0;
    return 0;
}

int32_t ttsamd_profile_read(double* out3) {
    TTS_REQUIRE(out3, "profile_read: null argument");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    // busy time of the conv engine = length of the UNION of the sections' intervals: on one stream the sections follow each
    // other (union = sum); with the acoustic model of batch i + 1 on a second stream under the vocoder of batch i
    // (ttsamd.pipeline) they overlap, and a sum would count the shared time twice
    std::vector<std::pair<float, float>> iv;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        TTS_CHECK_HIP(hipEventSynchronize(g_prof.ev[i + 1]));
        float t0 = 0.f, t1 = 0.f;
        if (i > 0) TTS_CHECK_HIP(hipEventElapsedTime(&t0, g_prof.ev[0], g_prof.ev[i]));
        TTS_CHECK_HIP(hipEventElapsedTime(&t1, g_prof.ev[0], g_prof.ev[i + 1]));
        iv.emplace_back(t0, std::max(t0, t1));
    }
    std::sort(iv.begin(), iv.end());
    double ms = 0.0;
    float lo = 0.f, hi = -1.f;
    for (const auto& v : iv) {
        if (hi < lo || v.first > hi) {
            if (hi >= lo) ms += hi - lo;
            lo = v.first;
            hi = v.second;
        } else {
            hi = std::max(hi, v.second);
        }
    }
    if (hi >= lo) ms += hi - lo;
    out3[0] = ms;
    out3[1] = (double)g_prof.launches;
    out3[2] = (double)(g_prof.used / 2);
    return 0;
}

}  // extern "C"
