// FastPitch.infer on the MI355X: handle creation (weight re-layout, sinusoid table, upload)
// and the two phases either side of the data-dependent decoder length.
// Replaces models/fastpitch/fastpitch/model.py:351-409 (FastPitch.infer), :114-133
// (TemporalPredictor), :68-90 (regulate_len) and transformer.py:113-225 (FFTransformer).
// Internals are channel-first [B][C][S]; the k=3 conv-FF (97 % of the FLOPs), qkv/o_net/proj
// and predictor convs all run on the MFMA conv engine (conv_mfma.hip).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "bfo3.hpp"

namespace ttsamd {

struct PConv {
    int64_t w_off = 0, b_off = -1, w16_off = 0;
    int64_t wo_off = -1;   // bf16 octet engine weights [Cin/16][K][2][CoutP][8] in the uint16 blob (the conv-FF convs; -1: not packed)
    int64_t wo3_off = -1;  // ... and their split-bf16 twin [Cin/16][K][2][CoutP][hi 8 | lo 8] (bfo3.hpp)
    int64_t ww4_off = -1;  // ... and as Winograd F(4,3) groups (conv_wino4.hip)
    int64_t ww_off = -1;   // k = 3: Winograd F(2,3) filters as a 4-tap conv in the fp32 blob (conv_wino.hip; -1: none)
    int cin = 0, cout = 0, k = 0;
};
struct FftLayer {
    PConv qkv, o_net, ff0, ff2;
    int64_t ln1_g, ln1_b, ln2_g, ln2_b;
};
struct Predictor {
    std::vector<PConv> convs;
    std::vector<int64_t> ln_g, ln_b;
    int64_t fc_w, fc_b;
    int filter;
};

struct FastPitch {
    ttsamd_fastpitch_cfg cfg;
    float* dev = nullptr;
    uint16_t* dev16 = nullptr;
    int64_t dev_n = 0, dev16_n = 0;   // element counts of the two blobs (ttsamd_dp_broadcast_weights)
    std::vector<FftLayer> enc, dec;
    Predictor dur, pitch, energy;
    int64_t word_emb, pos_enc, pos_dec, spk_emb = -1, proj_b;
    PConv proj;
    int64_t pitch_emb_w, pitch_emb_b, energy_emb_w = -1, energy_emb_b = -1;
    int pos_cap = 0;
    // ttsamd_fastpitch_set_batch_mode: 0 = the reference's padded-batch arithmetic (hidden activations of conv-FF / the predictors are NOT
    // masked, so an utterance's result depends on the longest one of its batch: SURVEY 3.4-1); 1 = every utterance as if it were alone --
    // those two second convs read their input masked at the utterance's own length, which is all it takes: every other op already masks
    mutable std::atomic<int> alone{0};
};

using TensorMap = std::map<std::string, const ttsamd_tensor*>;

static int64_t numel(const ttsamd_tensor* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}

struct Builder {
    const TensorMap& tm;
    std::vector<float> blob;
    std::vector<uint16_t> blob16;
    int32_t rc = 0;
    explicit Builder(const TensorMap& t) : tm(t) {}

    const ttsamd_tensor* get(const std::string& name, int64_t n_expected) {
        if (rc) return nullptr;
        auto it = tm.find(name);
        if (it == tm.end()) {
            set_error("fastpitch: missing tensor '%s'", name.c_str());
            rc = TTSAMD_EINVAL;
            return nullptr;
        }
        if (n_expected >= 0 && numel(it->second) != n_expected) {
            set_error("fastpitch: tensor '%s' has %lld elements, expected %lld", name.c_str(),
                      (long long)numel(it->second), (long long)n_expected);
            rc = TTSAMD_EINVAL;
            return nullptr;
        }
        return it->second;
    }
    int64_t raw(const std::string& name, int64_t n_expected) {
        const ttsamd_tensor* t = get(name, n_expected);
        if (!t) return 0;
        const int64_t off = (int64_t)blob.size();
        blob.insert(blob.end(), t->data, t->data + numel(t));
        blob.resize(align_up((int64_t)blob.size(), 64));
        return off;
    }
    // Conv1d weight [cout][cin][k] or Linear weight [cout][cin] (k = 1)
    PConv conv(const std::string& base, int cin, int cout, int k, bool bias, bool octet = false) {
        PConv c;
        c.cin = cin; c.cout = cout; c.k = k;
        const ttsamd_tensor* w = get(base + ".weight", (int64_t)cin * cout * k);
        if (!w) return c;
        c.w_off = (int64_t)blob.size();
        blob.resize(blob.size() + (size_t)cin * k * cout_padded(cout));
        pack_conv_weight(w->data, cout, cin, k, blob.data() + c.w_off);
        {
            const int64_t nn = (int64_t)cin * k * cout_padded(cout);
            c.w16_off = (int64_t)blob16.size();
            blob16.resize(blob16.size() + 2 * nn);
            split_packed_bf16(blob.data() + c.w_off, nn, blob16.data() + c.w16_off);
        }
        if (k == 3 && cin % 8 == 0 && cout % 128 == 0) {
            blob.resize(align_up((int64_t)blob.size(), 64));
            c.ww_off = (int64_t)blob.size();
            blob.resize(blob.size() + (size_t)cin * 4 * cout_padded(cout));
            pack_wino_weight(w->data, cout, cin, blob.data() + c.ww_off);
            if (cin % 16 == 0) {
                blob.resize(align_up((int64_t)blob.size(), 64));
                c.ww4_off = (int64_t)blob.size();
                blob.resize(blob.size() + (size_t)cin * wino4_groups(3) * cout_padded(cout));
                pack_wino4_weight(w->data, cout, cin, 3, blob.data() + c.ww4_off);
            }
        }
        if (octet && cin % 8 == 0 && cout % 32 == 0 && cout >= 128 && (k == 1 || k == 3 || k == 7 || k == 11)) {
            // the conv-FF pair also runs on the bf16 octet engine (config 3): v_mfma_f32_32x32x16_bf16, bf16 intermediate
            blob16.resize(align_up((int64_t)blob16.size(), 64));
            c.wo_off = (int64_t)blob16.size();
            blob16.resize(blob16.size() + (size_t)bfo_packed_conv_elems(cout, cin, k));
            bfo_pack_conv_weight(w->data, cout, cin, k, blob16.data() + c.wo_off);
            blob16.resize(align_up((int64_t)blob16.size(), 64));
            c.wo3_off = (int64_t)blob16.size();
            blob16.resize(blob16.size() + (size_t)bfo3_packed_conv_elems(cout, cin, k));
            bfo3_pack_conv_weight(w->data, cout, cin, k, blob16.data() + c.wo3_off);
        }
        blob.resize(align_up((int64_t)blob.size(), 64));
        if (bias) c.b_off = raw(base + ".bias", cout);
        return c;
    }
};

static void build_fft(Builder& b, const std::string& prefix, int n_layers, int d_model, int d_head, int n_head,
                      int d_inner, int k, std::vector<FftLayer>& out) {
    for (int i = 0; i < n_layers && b.rc == 0; ++i) {
        const std::string p = prefix + ".layers." + std::to_string(i) + ".";
        FftLayer l;
        l.qkv = b.conv(p + "dec_attn.qkv_net", d_model, 3 * n_head * d_head, 1, true, true);
        l.o_net = b.conv(p + "dec_attn.o_net", n_head * d_head, d_model, 1, false, true);
        l.ln1_g = b.raw(p + "dec_attn.layer_norm.weight", d_model);
        l.ln1_b = b.raw(p + "dec_attn.layer_norm.bias", d_model);
        l.ff0 = b.conv(p + "pos_ff.CoreNet.0", d_model, d_inner, k, true, true);
        l.ff2 = b.conv(p + "pos_ff.CoreNet.2", d_inner, d_model, k, true, true);   // index 2: ReLU at 1 (transformer.py:59-65 with Dropout commented out -> Sequential index 2)
        l.ln2_g = b.raw(p + "pos_ff.layer_norm.weight", d_model);
        l.ln2_b = b.raw(p + "pos_ff.layer_norm.bias", d_model);
        out.push_back(l);
    }
}

static void build_predictor(Builder& b, const std::string& prefix, int d_in, int filter, int k, int n_layers,
                            Predictor& pr) {
    pr.filter = filter;
    for (int i = 0; i < n_layers && b.rc == 0; ++i) {
        const std::string p = prefix + ".layers." + std::to_string(i) + ".";
        pr.convs.push_back(b.conv(p + "conv", i == 0 ? d_in : filter, filter, k, true, true));
        pr.ln_g.push_back(b.raw(p + "norm.weight", filter));
        pr.ln_b.push_back(b.raw(p + "norm.bias", filter));
    }
    pr.fc_w = b.raw(prefix + ".fc.weight", filter);
    pr.fc_b = b.raw(prefix + ".fc.bias", 1);
}

// Channel-first sinusoid table [d_model][cap]: row c<half = sin(t*inv_freq[c]), else cos
// (transformer.py:41-44).  The product t*inv_freq is rounded to fp32 as torch.matmul does.
static int64_t build_pos_table(Builder& b, const std::string& name, int d_model, int cap) {
    const ttsamd_tensor* f = b.get(name, d_model / 2);
    if (!f) return 0;
    const int64_t off = (int64_t)b.blob.size();
    b.blob.resize(b.blob.size() + (size_t)d_model * cap);
    float* o = b.blob.data() + off;
    const int half = d_model / 2;
    for (int c = 0; c < half; ++c) {
        const float fr = f->data[c];
        for (int t = 0; t < cap; ++t) {
            const float arg = (float)t * fr;
            o[(int64_t)c * cap + t] = (float)std::sin((double)arg);
            o[(int64_t)(c + half) * cap + t] = (float)std::cos((double)arg);
        }
    }
    return off;
}

int32_t fastpitch_create(const ttsamd_tensor* weights, int32_t n, const ttsamd_fastpitch_cfg* cfg, int32_t pos_cap,
                         FastPitch** out) {
    TTS_REQUIRE(weights && cfg && out, "fastpitch_create: null argument");
    TTS_REQUIRE(cfg->in_fft_n_heads == 1 && cfg->out_fft_n_heads == 1, "fastpitch: only n_heads = 1 is built");
    TTS_REQUIRE(cfg->d_model % 16 == 0, "fastpitch: d_model must be a multiple of 16");
    TensorMap tm;
    for (int i = 0; i < n; ++i) tm[weights[i].name] = &weights[i];
    Builder b(tm);
    auto* h = new FastPitch();
    h->cfg = *cfg;
    h->pos_cap = pos_cap;
    const int d = cfg->d_model;
    h->word_emb = b.raw("encoder.word_emb.weight", (int64_t)cfg->n_symbols * d);
    h->pos_enc = build_pos_table(b, "encoder.pos_emb.inv_freq", d, pos_cap);
    h->pos_dec = build_pos_table(b, "decoder.pos_emb.inv_freq", d, pos_cap);
    build_fft(b, "encoder", cfg->in_fft_n_layers, d, cfg->in_fft_d_head, cfg->in_fft_n_heads, cfg->in_fft_filter,
              cfg->in_fft_kernel, h->enc);
    build_fft(b, "decoder", cfg->out_fft_n_layers, d, cfg->out_fft_d_head, cfg->out_fft_n_heads,
              cfg->out_fft_filter, cfg->out_fft_kernel, h->dec);
    build_predictor(b, "duration_predictor", d, cfg->dur_filter, cfg->dur_kernel, cfg->dur_n_layers, h->dur);
    build_predictor(b, "pitch_predictor", d, cfg->pitch_filter, cfg->pitch_kernel, cfg->pitch_n_layers, h->pitch);
    h->pitch_emb_w = b.raw("pitch_emb.weight", (int64_t)d * cfg->pitch_emb_kernel);
    h->pitch_emb_b = b.raw("pitch_emb.bias", d);
    if (cfg->energy_conditioning) {
        build_predictor(b, "energy_predictor", d, cfg->energy_filter, cfg->energy_kernel, cfg->energy_n_layers,
                        h->energy);
        h->energy_emb_w = b.raw("energy_emb.weight", (int64_t)d * cfg->energy_emb_kernel);
        h->energy_emb_b = b.raw("energy_emb.bias", d);
    }
    h->proj = b.conv("proj", d, cfg->n_mel_channels, 1, true);
    if (cfg->n_speakers > 1 && b.rc == 0) {
        const ttsamd_tensor* e = b.get("speaker_emb.weight", (int64_t)cfg->n_speakers * d);
        if (e) {
            h->spk_emb = (int64_t)b.blob.size();
            for (int64_t i = 0; i < numel(e); ++i) b.blob.push_back(e->data[i] * cfg->speaker_emb_weight);  // model.py:361
            b.blob.resize(align_up((int64_t)b.blob.size(), 64));
        }
    }
    int32_t rc = b.rc;
    if (rc == 0) {
        h->dev_n = (int64_t)b.blob.size();
        h->dev16_n = (int64_t)b.blob16.size();
        hipError_t e = hipMalloc((void**)&h->dev, b.blob.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->dev, b.blob.data(), b.blob.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&h->dev16, b.blob16.size() * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemcpy(h->dev16, b.blob16.data(), b.blob16.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("fastpitch_create: weight upload failed: %s", hipGetErrorString(e));
            rc = TTSAMD_EHIP;
        }
    }
    if (rc) {
        if (h->dev) (void)hipFree(h->dev);
        if (h->dev16) (void)hipFree(h->dev16);
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

void fastpitch_destroy(FastPitch* h) {
    if (!h) return;
    if (h->dev) (void)hipFree(h->dev);
    if (h->dev16) (void)hipFree(h->dev16);
    delete h;
}

// ------------------------------------------------------------------------------------

struct FftWs {
    float *q, *a, *y, *hid, *splitk;
    float* o3;          // split-bf16 mode: the x3 copies of x, y and the attention output ((2 d + d_head) x 4 bytes per position)
    int64_t* lens1;     // [B] min(len + 1, S), filled by encode / decode for batches of 2 and more (run_fft: ragged conv-FF)
    bool ragged;        // lens1 is valid
};

// split-K scratch of the call in flight on this thread (carved from the caller's workspace by encode / decode)
static thread_local float* t_splitk_ws = nullptr;
struct SplitKScope {
    explicit SplitKScope(float* p) { t_splitk_ws = p; }
    ~SplitKScope() { t_splitk_ws = nullptr; }
};
// split-bf16 mode, batch <= 2: FastPitch is ~135 dependent launches of a few microseconds and the exact-fp32 engine's are the shorter ones
// (tools/b1_parts.py: batch 1 1.17 ms against 1.58 on the x3 kernels, batch 2 1.50 / 1.60, batch 3 1.80 / 1.74, batch 4 1.89 / 1.73): the
// call then runs its convs in fp32 (at least as accurate; HiFi-GAN stays on the split-bf16 octet engine)
static thread_local bool t_small_f32 = false;
struct SmallBatchScope {
    explicit SmallBatchScope(int B) { t_small_f32 = default_precision() == 2 && B <= 2; }
    ~SmallBatchScope() { t_small_f32 = false; }
};

static int32_t run_conv(const FastPitch* h, const PConv& c, const float* x, float* y, const float* res, int B, int S,
                        const int64_t* lens_in, int relu, hipStream_t s, const int64_t* lens_out = nullptr) {
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.x_bs = (int64_t)c.cin * S; p.x_cs = S;
    p.w = h->dev + c.w_off; p.bias = c.b_off >= 0 ? h->dev + c.b_off : nullptr;
    p.w_bf16 = h->dev16 + c.w16_off; p.precision = t_small_f32 ? 0 : default_precision();
    p.w_wino = c.ww_off >= 0 ? h->dev + c.ww_off : nullptr;
    p.w_wino4 = c.ww4_off >= 0 ? h->dev + c.ww4_off : nullptr;
    p.y = y; p.y_bs = (int64_t)c.cout * S; p.y_cs = S; p.y_ts = 1;
    p.res = res; p.r_bs = (int64_t)c.cout * S; p.r_cs = S;
    p.lens_in = lens_in; p.lens_out = lens_out; p.len_in_mul = 1; p.len_out_mul = 1;
    p.Lin = S; p.Nout = S;
    p.Cin = c.cin; p.Cout = c.cout; p.CoutP = cout_padded(c.cout); p.K = c.k;
    p.dil = 1; p.pad = c.k / 2;
    p.n_phase = 1; p.in_slope = 1.0f; p.relu_out = relu; p.mode = 0; p.div = 1.f; p.batch = B;
    p.splitk_ws = t_splitk_ws; p.splitk_floats = t_splitk_ws ? kSplitKFloatsFp : 0;
    prof_begin(s, 2.0 * c.cout * c.cin * c.k);
    const int32_t rc = launch_conv(p, s);
    prof_end(s);
    return rc;
}

// transformer.py:172-177 x n_layers.  x is updated in place.
static int32_t run_fft(const FastPitch* h, const std::vector<FftLayer>& layers, int d_head, float* x,
                       const int64_t* lens, int B, int S, const FftWs& w, hipStream_t s) {
    const int d = h->cfg.d_model;
    const float scale = 1.0f / std::sqrt((float)d_head);
    const bool alone = h->alone.load(std::memory_order_relaxed) != 0;
    const PConv *ff2_of = nullptr, *ff0_of = nullptr;   // the layer's conv-FF convs (octet paths: set per layer below)
    const char* ffe = opt_str(OPT_BFO_FF);              // read per call: the tests and A/B runs flip it
    bool octet = default_precision() == 1 && !(ffe && ffe[0] == '0') && d % 64 == 0 && d <= 512 && d_head == 64;
    for (const FftLayer& l : layers)
        octet = octet && l.ff0.wo_off >= 0 && l.ff2.wo_off >= 0 && l.qkv.wo_off >= 0 && l.o_net.wo_off >= 0 && l.qkv.cout == 3 * d_head;
    if (octet) {
        // ---- config 3: the whole FFT block on the bf16 matrix cores.  The residual stream x / y stays fp32 channel-first (LayerNorm
        // statistics, residual adds); every GEMM reads a bf16 octet copy of its input (bfo.hpp) that its producer writes alongside:
        //   xo (LayerNorm 2 / the initial pack) -> qkv conv -> fp32 q|k|v -> bf16 MFMA attention -> ao (octet)
        //   -> o_net conv + x -> y (fp32) -> LayerNorm 1 -> y, yo -> Conv1d + ReLU -> hid (octet, 1536 channels)
        //   -> Conv1d + y -> x (fp32) -> LayerNorm 2 -> x, xo                                   (transformer.py:113-160, 72-90, 172-177)
        // All octet tensors live in the fp32-sized `hid` buffer: 2 B S (d_inner + 2 d + 64) bytes of its 4 B S d_inner.
        const int di = layers[0].ff0.cout;
        char* base = (char*)w.hid;
        void* hid_o = base;
        void* xo = base + (int64_t)B * di * S * 2;
        void* yo = (char*)xo + (int64_t)B * d * S * 2;
        void* ao = (char*)yo + (int64_t)B * d * S * 2;
        TTS_REQUIRE((int64_t)2 * (di + 2 * d + d_head) <= (int64_t)4 * di, "fastpitch: octet buffers do not fit the hidden buffer");
        TTS_TRY(bfo_launch_pack(x, B, d, S, 1.f, xo, s));
        BfoConvParams cp;
        // ln_g != nullptr: LayerNorm (masked with `lens`) over the fp32 result in place + its octet copy `ln_o` (BfoConvParams::ln_*)
        auto conv = [&](const PConv& c, const void* in, void* out_o, float* out_f, const float* res_f, float out_slope,
                        const float* ln_g = nullptr, const float* ln_b = nullptr, void* ln_o = nullptr) -> int32_t {
            std::memset(&cp, 0, sizeof(cp));
            cp.ln_g = ln_g; cp.ln_b = ln_b; cp.ln_octet = ln_o; cp.ln_lens = ln_g ? lens : nullptr;
            cp.batch = B; cp.len_mul = 1; cp.Lin = S; cp.dil = 1; cp.up = 1; cp.div = 1.f; cp.res_slope = 1.f;
            cp.x = in; cp.y = out_o; cp.y_f32 = out_f; cp.res_f32 = res_f;
            if (alone && &c == ff2_of) { cp.lens = lens; cp.out_all = 1; }      // batch mode 1: the hidden activation masked on load
            cp.w = h->dev16 + c.wo_off; cp.bias = c.b_off >= 0 ? h->dev + c.b_off : nullptr;
            cp.Cin = c.cin; cp.Cout = c.cout; cp.K = c.k; cp.out_slope = out_slope;
            cp.splitk_ws = t_splitk_ws; cp.splitk_floats = t_splitk_ws ? kSplitKFloatsFp : 0;
            prof_begin(s, 2.0 * c.cout * c.cin * c.k);
            const int32_t rc = bfo_launch_conv(cp, s);
            prof_end(s);
            return rc;
        };
        for (const FftLayer& l : layers) {
            ff2_of = &l.ff2;
            TTS_TRY(conv(l.qkv, xo, nullptr, w.q, nullptr, 1.f));
            TTS_TRY(launch_attention_bf16(w.q, lens, B, d_head, S, scale, nullptr, s, ao));
            TTS_TRY(conv(l.o_net, ao, nullptr, w.y, x, 1.f, h->dev + l.ln1_g, h->dev + l.ln1_b, yo));   // + LayerNorm 1 -> y, yo
            TTS_TRY(conv(l.ff0, yo, hid_o, nullptr, nullptr, 0.f));               // ReLU = leaky-relu with slope 0, applied by the producer
            TTS_TRY(conv(l.ff2, hid_o, nullptr, x, w.y, 1.f, h->dev + l.ln2_g, h->dev + l.ln2_b, xo));  // + LayerNorm 2 -> x, xo
        }
        return 0;
    }
    bool x3 = default_precision() == 2 && !t_small_f32 && !(ffe && ffe[0] == '0') && (d == 384 || d == 256 || d == 512) && d_head % 8 == 0 && w.o3;
    for (const FftLayer& l : layers)
        x3 = x3 && l.ff0.wo3_off >= 0 && l.ff2.wo3_off >= 0 && l.qkv.wo3_off >= 0 && l.o_net.wo3_off >= 0 && l.qkv.cout == 3 * d_head;
    if (x3) {
        // ---- split bf16: the block above with every GEMM operand = hi + lo (bfo3.hpp).  The residual stream stays fp32 channel-first;
        // each GEMM reads an x3 copy of its input written by its producer (LayerNorm, the ReLU conv, or a pack of the attention output):
        //   xo -> qkv conv -> fp32 q|k|v -> exact fp32 attention -> a -> pack -> ao -> o_net conv + x -> y -> LayerNorm 1 -> y, yo
        //   -> Conv1d + ReLU -> hid (x3, 1536 channels = the whole fp32-sized buffer) -> Conv1d + y -> x -> LayerNorm 2 -> x, xo
        const int di = layers[0].ff0.cout;
        (void)di;
        char* base = (char*)w.o3;
        void* xo = base;
        void* yo = base + (int64_t)B * d * S * 4;
        void* ao = (char*)yo + (int64_t)B * d * S * 4;
        void* hid_o = w.hid;
        TTS_TRY(bfo3_launch_pack(x, B, d, S, 1.f, xo, s));
        BfoConvParams cp;
        auto conv = [&](const PConv& c, const void* in, void* out_o, float* out_f, const float* res_f, float out_slope) -> int32_t {
            std::memset(&cp, 0, sizeof(cp));
            cp.batch = B; cp.len_mul = 1; cp.Lin = S; cp.dil = 1; cp.up = 1; cp.div = 1.f; cp.res_slope = 1.f;
            cp.x = in; cp.y = out_o; cp.y_f32 = out_f; cp.res_f32 = res_f;
            // ragged conv-FF as on the fp32 path below (one length per launch here -- it masks the input and bounds the output): hid on
            // [0, len], the second conv reads it masked past that and writes x on [0, len] (frame len: finite, zeroed by LayerNorm 2)
            if (w.ragged && (&c == ff2_of || &c == ff0_of)) cp.lens = (alone && &c == ff2_of) ? lens : w.lens1;
            else if (alone && &c == ff2_of) { cp.lens = lens; cp.out_all = 1; }
            cp.w = h->dev16 + c.wo3_off; cp.bias = c.b_off >= 0 ? h->dev + c.b_off : nullptr;
            cp.Cin = c.cin; cp.Cout = c.cout; cp.K = c.k; cp.out_slope = out_slope;
            prof_begin(s, 2.0 * c.cout * c.cin * c.k);
            const int32_t rc = bfo3_launch_conv(cp, s);
            prof_end(s);
            return rc;
        };
        for (const FftLayer& l : layers) {
            ff2_of = &l.ff2; ff0_of = &l.ff0;
            TTS_TRY(conv(l.qkv, xo, nullptr, w.q, nullptr, 1.f));
            TTS_TRY(launch_attention(w.q, lens, B, d_head, S, scale, w.a, s, t_splitk_ws, t_splitk_ws ? kSplitKFloatsFp : 0));
            TTS_TRY(bfo3_launch_pack(w.a, B, d_head, S, 1.f, ao, s));
            TTS_TRY(conv(l.o_net, ao, nullptr, w.y, x, 1.f));
            TTS_TRY(launch_layernorm_cf_x3(w.y, w.y, yo, h->dev + l.ln1_g, h->dev + l.ln1_b, lens, 1, B, d, S, s));
            TTS_TRY(conv(l.ff0, yo, hid_o, nullptr, nullptr, 0.f));               // ReLU = leaky-relu with slope 0, applied by the producer
            TTS_TRY(conv(l.ff2, hid_o, nullptr, x, w.y, 1.f));
            TTS_TRY(launch_layernorm_cf_x3(x, x, xo, h->dev + l.ln2_g, h->dev + l.ln2_b, lens, 1, B, d, S, s));
        }
        return 0;
    }
    for (const FftLayer& l : layers) {
        TTS_TRY(run_conv(h, l.qkv, x, w.q, nullptr, B, S, nullptr, 0, s));
        TTS_TRY(launch_attention(w.q, lens, B, d_head, S, scale, w.a, s, t_splitk_ws, t_splitk_ws ? kSplitKFloatsFp : 0));
        TTS_TRY(run_conv(h, l.o_net, w.a, w.y, x, B, S, nullptr, 0, s));
        TTS_TRY(launch_layernorm_cf(w.y, w.y, h->dev + l.ln1_g, h->dev + l.ln1_b, lens, 1, B, d, S, s));
        // conv-FF (97 % of the layer's FLOPs) only as far as a row's valid frames need it: y is zero past len (LayerNorm 1 masks), frame
        // len - 1 of the second conv reads hid[len] -- the one un-masked hidden frame that makes a padded batch composition-dependent
        //  -- so hid is computed on [0, len] (lens1 = min(len + 1, S)), read masked past that, and x written on [0, len):
        // the tiles past a row's end are never launched (ragged-batch block compaction, common.hpp: live_tile).  Every product the
        // reference's padded-batch arithmetic feeds into a valid frame is still there; what changes is rounding: an F(4,3) output quad
        // shares its transformed window, so the frames next to a row's end see zeros instead of the dead hidden frames in terms that
        // cancel only in exact arithmetic (mel of the B = 32 bench batch: 5e-6 max-abs against the all-frames schedule, tools/fp_digest.py;
        // the split-bf16 path, which has no transform, is bit-identical).  Frames >= len of x keep finite stale values that LayerNorm 2 zeroes
        const int64_t* l1 = w.ragged ? w.lens1 : nullptr;
        TTS_TRY(run_conv(h, l.ff0, w.y, w.hid, nullptr, B, S, nullptr, 1, s, l1));
        TTS_TRY(run_conv(h, l.ff2, w.hid, x, w.y, B, S, alone ? lens : l1, 0, s, w.ragged ? lens : nullptr));
        TTS_TRY(launch_layernorm_cf(x, x, h->dev + l.ln2_g, h->dev + l.ln2_b, lens, 1, B, d, S, s));
    }
    return 0;
}

// model.py:129-133; input masked on load (lens_in), hidden NOT masked
// (lens1 = min(len + 1, longest row) or nullptr: the second conv's input past that is zero PADDING in the reference, not a hidden frame --
// it matters when the caller's rows are wider than the longest utterance)
static int32_t run_predictor(const FastPitch* h, const Predictor& pr, const float* x, const int64_t* lens, int B,
                             int S, float* t0, float* t1, float* out, float* out2, float max_dur, float mul,
                             float add, hipStream_t s, void* px3 = nullptr, const int64_t* lens1 = nullptr) {
    const float* src = x;
    float* bufs[2] = {t0, t1};
    {
        // config 3: Conv1d + ReLU -> LayerNorm chain on the bf16 octet engine (input packed once, masked on load; LayerNorm writes the
        // next conv's octet copy).  The octet tensors sit behind the fp32 buffers' used part: t0 / t1 are sized for the widest filter.
        const char* ffe = opt_str(OPT_BFO_FF);
        bool octet = default_precision() == 1 && !(ffe && ffe[0] == '0') && pr.convs.size() == 2 && pr.filter % 64 == 0 && pr.filter <= 512 &&
                     pr.convs[0].cin % 8 == 0 && pr.convs[0].cin <= pr.filter * 2;
        for (const PConv& c : pr.convs) octet = octet && c.wo_off >= 0;
        if (octet) {
            // t1 ([B][filter][S] fp32) holds the octet tensors one after the other: the packed input ([B][cin/8][S][8] bf16, cin <= 2 filter),
            // then -- once the first conv has consumed it -- the octet copy of the first LayerNorm's output
            void* xo = t1;
            BfoConvParams cp;
            auto conv = [&](const PConv& c, const void* in, float* out_f, const int64_t* lens_in) -> int32_t {
                std::memset(&cp, 0, sizeof(cp));
                cp.batch = B; cp.len_mul = 1; cp.Lin = S; cp.dil = 1; cp.up = 1; cp.div = 1.f; cp.res_slope = 1.f;
                cp.x = in; cp.y_f32 = out_f; cp.lens = lens_in; cp.out_all = 1;
                cp.w = h->dev16 + c.wo_off; cp.bias = c.b_off >= 0 ? h->dev + c.b_off : nullptr;
                cp.Cin = c.cin; cp.Cout = c.cout; cp.K = c.k; cp.out_slope = 0.f;                      // ReLU
                cp.splitk_ws = t_splitk_ws; cp.splitk_floats = t_splitk_ws ? kSplitKFloatsFp : 0;
                prof_begin(s, 2.0 * c.cout * c.cin * c.k);
                const int32_t rc = bfo_launch_conv(cp, s);
                prof_end(s);
                return rc;
            };
            TTS_TRY(bfo_launch_pack(x, B, pr.convs[0].cin, S, 1.f, xo, s));
            TTS_TRY(conv(pr.convs[0], xo, t0, lens));
            // LayerNorm of layer 0 in place (fp32, t0) + its octet copy into t1 (the packed input there is dead now)
            TTS_TRY(launch_layernorm_cf_octet(t0, t0, t1, h->dev + pr.ln_g[0], h->dev + pr.ln_b[0], nullptr, 0, B, pr.filter, S, s));
            // the second conv reads the octet copy; the fp32 LayerNorm output in t0 is dead, so its result goes there
            TTS_TRY(conv(pr.convs[1], t1, t0, h->alone.load(std::memory_order_relaxed) ? lens : lens1));
            TTS_TRY(launch_layernorm_cf(t0, t0, h->dev + pr.ln_g[1], h->dev + pr.ln_b[1], nullptr, 0, B, pr.filter, S, s));
            return launch_pred_fc(t0, h->dev + pr.fc_w, h->dev + pr.fc_b, lens, B, pr.filter, S, out, out2, max_dur, mul, add, s);
        }
    }
    {
        // split bf16: the same chain on the x3 kernels.  The x3 tensors are as large as the fp32 ones: the packed input ([B][cin][S] x 4
        // bytes, cin <= 2 filter would not fit) and the copy of the first LayerNorm's output both go to `px3`
        const char* ffe = opt_str(OPT_BFO_FF);
        bool x3 = default_precision() == 2 && !t_small_f32 && !(ffe && ffe[0] == '0') && pr.convs.size() == 2 && (pr.filter == 256 || pr.filter == 384 || pr.filter == 512) &&
                  pr.convs[0].cin % 8 == 0 && pr.convs[0].cin <= pr.filter * 2 && px3 != nullptr;
        for (const PConv& c : pr.convs) x3 = x3 && c.wo3_off >= 0;
        if (x3) {
            BfoConvParams cp;
            auto conv = [&](const PConv& c, const void* in, float* out_f, const int64_t* lens_in) -> int32_t {
                std::memset(&cp, 0, sizeof(cp));
                cp.batch = B; cp.len_mul = 1; cp.Lin = S; cp.dil = 1; cp.up = 1; cp.div = 1.f; cp.res_slope = 1.f;
                cp.x = in; cp.y_f32 = out_f; cp.lens = lens_in; cp.out_all = 1;
                cp.w = h->dev16 + c.wo3_off; cp.bias = c.b_off >= 0 ? h->dev + c.b_off : nullptr;
                cp.Cin = c.cin; cp.Cout = c.cout; cp.K = c.k; cp.out_slope = 0.f;                      // ReLU
                prof_begin(s, 2.0 * c.cout * c.cin * c.k);
                const int32_t rc = bfo3_launch_conv(cp, s);
                prof_end(s);
                return rc;
            };
            TTS_TRY(bfo3_launch_pack(x, B, pr.convs[0].cin, S, 1.f, px3, s));
            TTS_TRY(conv(pr.convs[0], px3, t0, lens));
            TTS_TRY(launch_layernorm_cf_x3(t0, t0, px3, h->dev + pr.ln_g[0], h->dev + pr.ln_b[0], nullptr, 0, B, pr.filter, S, s));
            TTS_TRY(conv(pr.convs[1], px3, t1, h->alone.load(std::memory_order_relaxed) ? lens : lens1));
            TTS_TRY(launch_layernorm_cf(t1, t1, h->dev + pr.ln_g[1], h->dev + pr.ln_b[1], nullptr, 0, B, pr.filter, S, s));
            return launch_pred_fc(t1, h->dev + pr.fc_w, h->dev + pr.fc_b, lens, B, pr.filter, S, out, out2, max_dur, mul, add, s);
        }
    }
    for (size_t i = 0; i < pr.convs.size(); ++i) {
        float* dst = bufs[i & 1];
        TTS_TRY(run_conv(h, pr.convs[i], src, dst, nullptr, B, S, (i == 0 || h->alone.load(std::memory_order_relaxed)) ? lens : lens1, 1, s));
        TTS_TRY(launch_layernorm_cf(dst, dst, h->dev + pr.ln_g[i], h->dev + pr.ln_b[i], nullptr, 0, B, pr.filter, S, s));
        src = dst;
    }
    return launch_pred_fc(src, h->dev + pr.fc_w, h->dev + pr.fc_b, lens, B, pr.filter, S, out, out2, max_dur, mul,
                          add, s);
}

struct EncWs {
    FftWs f;
    float *p0, *p1, *log_dur;
    int64_t* lens;
};

static void carve_enc(const FastPitch* h, Arena& a, int B, int L, EncWs& w) {
    const ttsamd_fastpitch_cfg& c = h->cfg;
    const int d = c.d_model;
    const int filt = std::max(std::max(c.dur_filter, c.pitch_filter), c.energy_filter);
    w.f.q = a.take<float>((int64_t)B * 3 * c.in_fft_n_heads * c.in_fft_d_head * L);
    w.f.a = a.take<float>((int64_t)B * c.in_fft_n_heads * c.in_fft_d_head * L);
    w.f.y = a.take<float>((int64_t)B * d * L);
    w.f.hid = a.take<float>((int64_t)B * c.in_fft_filter * L);
    w.f.o3 = a.take<float>((int64_t)B * (2 * d + c.in_fft_n_heads * c.in_fft_d_head) * L);
    w.p0 = a.take<float>((int64_t)B * filt * L);
    w.p1 = a.take<float>((int64_t)B * filt * L);
    w.log_dur = a.take<float>((int64_t)B * L);
    w.lens = a.take<int64_t>(B);
    w.f.splitk = a.take<float>(kSplitKFloatsFp);
    w.f.lens1 = a.take<int64_t>(B);
    w.f.ragged = false;
}

int64_t fastpitch_encode_workspace_bytes(const FastPitch* h, int32_t B, int32_t L) {
    Arena a(nullptr, 0);
    EncWs w;
    carve_enc(h, a, B, L, w);
    return a.off;
}

int32_t fastpitch_encode(const FastPitch* h, const int64_t* ids, int32_t B, int32_t L, int32_t speaker, float pace,
                         const float* dur_tgt, const float* pitch_tgt, const float* energy_tgt, float pitch_mul,
                         float pitch_add, float max_duration, float* enc_cond, float* dur_pred, float* pitch_pred,
                         float* energy_pred, int64_t* reps, int64_t* dec_lens, void* ws, int64_t ws_bytes,
                         hipStream_t s) {
    TTS_REQUIRE(h && ids && enc_cond && dur_pred && pitch_pred && reps && dec_lens, "fastpitch_encode: null argument");
    TTS_REQUIRE(B >= 1 && L >= 1 && L <= h->pos_cap, "fastpitch_encode: bad batch/n_tokens (%d, %d; cap %d)", B, L,
                h->pos_cap);
    TTS_REQUIRE(pace > 0.f, "fastpitch_encode: pace must be > 0");
    const ttsamd_fastpitch_cfg& c = h->cfg;
    TTS_REQUIRE(c.n_speakers <= 1 || (speaker >= 0 && speaker < c.n_speakers), "fastpitch_encode: speaker %d out of range", speaker);
    TTS_REQUIRE(!c.energy_conditioning || energy_pred || energy_tgt, "fastpitch_encode: energy_pred is null");
    Arena a(ws, ws_bytes);
    EncWs w;
    carve_enc(h, a, B, L, w);
    if (!ws || !a.ok) {
        set_error("fastpitch_encode: workspace of %lld bytes needed, %lld given", (long long)a.off, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    const int d = c.d_model;
    SplitKScope splitk(w.f.splitk);
    SmallBatchScope small_f32(B);
    float* x = enc_cond;
    const float* spk = (c.n_speakers > 1 && h->spk_emb >= 0) ? h->dev + h->spk_emb + (int64_t)speaker * d : nullptr;
    TTS_TRY(launch_embed(ids, h->dev + h->word_emb, h->dev + h->pos_enc, h->pos_cap, spk, c.padding_idx, c.n_symbols, B, L, d, x,
                         w.lens, s));
    // (one frame past a row's end is what a k = 3 conv reads: the ragged schedule is built for the reference's kernel sizes)
    if (B >= 2 && c.in_fft_kernel == 3 && c.dur_kernel == 3 && c.pitch_kernel == 3 && (!c.energy_conditioning || c.energy_kernel == 3)) {
        TTS_TRY(launch_lens_plus1(w.lens, L, B, /*clamp_at_max=*/0, w.f.lens1, s));
        w.f.ragged = true;
    }
    TTS_TRY(run_fft(h, h->enc, c.in_fft_d_head, x, w.lens, B, L, w.f, s));
    // durations (model.py:367-368)
    TTS_TRY(run_predictor(h, h->dur, x, w.lens, B, L, w.p0, w.p1, w.log_dur, dur_pred, max_duration, 1.f, 0.f, s, w.f.o3, w.f.ragged ? w.f.lens1 : nullptr));
    // pitch (model.py:371-386); pitch_trf = mul*p + add (networks.py:38-42)
    TTS_TRY(run_predictor(h, h->pitch, x, w.lens, B, L, w.p0, w.p1, pitch_pred, nullptr, 0.f, pitch_mul, pitch_add, s, w.f.o3, w.f.ragged ? w.f.lens1 : nullptr));
    TTS_TRY(launch_scalar_emb_add(x, pitch_tgt ? pitch_tgt : pitch_pred, h->dev + h->pitch_emb_w,
                                  h->dev + h->pitch_emb_b, B, d, L, c.pitch_emb_kernel, s));
    // energy (model.py:389-399)
    if (c.energy_conditioning) {
        if (energy_pred)
            TTS_TRY(run_predictor(h, h->energy, x, w.lens, B, L, w.p0, w.p1, energy_pred, nullptr, 0.f, 1.f, 0.f, s, w.f.o3, w.f.ragged ? w.f.lens1 : nullptr));
        TTS_TRY(launch_scalar_emb_add(x, energy_tgt ? energy_tgt : energy_pred, h->dev + h->energy_emb_w,
                                      h->dev + h->energy_emb_b, B, d, L, c.energy_emb_kernel, s));
    }
    // integer half of regulate_len (model.py:72-76)
    return launch_durations_to_reps(dur_tgt ? dur_tgt : dur_pred, pace, B, L, reps, dec_lens, s);
}

static void carve_dec(const FastPitch* h, Arena& a, int B, int T, FftWs& w) {
    const ttsamd_fastpitch_cfg& c = h->cfg;
    w.q = a.take<float>((int64_t)B * 3 * c.out_fft_n_heads * c.out_fft_d_head * T);
    w.a = a.take<float>((int64_t)B * c.out_fft_n_heads * c.out_fft_d_head * T);
    w.y = a.take<float>((int64_t)B * c.d_model * T);
    w.hid = a.take<float>((int64_t)B * c.out_fft_filter * T);
    w.splitk = a.take<float>(kSplitKFloatsFp);
    w.o3 = a.take<float>((int64_t)B * (2 * c.d_model + c.out_fft_n_heads * c.out_fft_d_head) * T);
    w.lens1 = a.take<int64_t>(B);
    w.ragged = false;
}

int64_t fastpitch_decode_workspace_bytes(const FastPitch* h, int32_t B, int32_t T) {
    Arena a(nullptr, 0);
    FftWs w;
    carve_dec(h, a, B, T, w);
    return a.off;
}

int32_t fastpitch_decode(const FastPitch* h, float* x, const int64_t* dec_lens, int32_t B, int32_t T, float* mel,
                         void* ws, int64_t ws_bytes, hipStream_t s) {
    TTS_REQUIRE(h && x && dec_lens && mel, "fastpitch_decode: null argument");
    TTS_REQUIRE(B >= 1 && T >= 1 && T <= h->pos_cap, "fastpitch_decode: bad batch/t_max (%d, %d; cap %d)", B, T,
                h->pos_cap);
    Arena a(ws, ws_bytes);
    FftWs w;
    carve_dec(h, a, B, T, w);
    if (!ws || !a.ok) {
        set_error("fastpitch_decode: workspace of %lld bytes needed, %lld given", (long long)a.off, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    const ttsamd_fastpitch_cfg& c = h->cfg;
    SplitKScope splitk(w.splitk);
    SmallBatchScope small_f32(B);
    // decoder input = len_regulated + pos_emb*mask (transformer.py:215-219, embed_input=False)
    TTS_TRY(launch_add_pos(x, h->dev + h->pos_dec, h->pos_cap, dec_lens, B, c.d_model, T, s));
    if (B >= 2 && c.out_fft_kernel == 3) {
        TTS_TRY(launch_lens_plus1(dec_lens, T, B, /*clamp_at_max=*/1, w.lens1, s));     // t_max may be the caller's 16-byte-padded row width
        w.ragged = true;
    }
    TTS_TRY(run_fft(h, h->dec, c.out_fft_d_head, x, dec_lens, B, T, w, s));
    // proj + permute (model.py:406-408): channel-first output IS the permuted layout
    return run_conv(h, h->proj, x, mel, nullptr, B, T, nullptr, 0, s);
}

void fastpitch_set_batch_mode(const FastPitch* h, int mode) { h->alone.store(mode != 0 ? 1 : 0, std::memory_order_relaxed); }

void fastpitch_blobs(const void* hv, void** f32, int64_t* n_f32, void** b16, int64_t* n_b16) {
    const FastPitch* h = (const FastPitch*)hv;
    *f32 = h->dev; *n_f32 = h->dev_n; *b16 = h->dev16; *n_b16 = h->dev16_n;
}

}  // namespace ttsamd
