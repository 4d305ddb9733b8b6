// BiLSTM sequence taggers of the reference's models/diacritizers:
//   Shakkelha  (shakkelha/network.py:29-42)  Embedding -> 2 x nn.LSTM(bidirectional) -> 3 x Linear (+ReLU) -> softmax
//   Shakkala   (shakkala/network.py:31-43)   Embedding -> LSTMHardSigmoid -> BatchNorm1d -> 2 x LSTMHardSigmoid -> Linear -> softmax
// Every LSTM layer = one 1x1 conv on the MFMA engine (input projections of both directions, bias = b_ih + b_hh,
// the eval-mode BatchNorm folded into the following projection) + one recurrent kernel (block per (sequence,
// direction), thread per hidden unit, W_hh^T streamed from L2).  Activations stay channel-first [B][C][T].
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace ttsamd {

struct GConv {
    int64_t w_off = 0, b_off = 0, w16_off = 0;
    int cin = 0, cout = 0;
};

struct Tagger {
    ttsamd_tagger_cfg cfg;
    float* dev = nullptr;
    uint16_t* dev16 = nullptr;
    int64_t emb = 0;
    int emb_pad = 0;
    std::vector<GConv> xproj, dense;
    std::vector<int64_t> whhT_f, whhT_b;
    int max_c = 0;   // widest activation (channels)
};

static int64_t gnumel(const ttsamd_tensor* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}

int32_t tagger_create(const ttsamd_tensor* weights, int32_t n, const ttsamd_tagger_cfg* cfg, Tagger** out) {
    TTS_REQUIRE(weights && cfg && out, "tagger_create: null argument");
    TTS_REQUIRE(cfg->n_lstm >= 1 && cfg->n_lstm <= 4 && cfg->n_dense >= 1 && cfg->n_dense <= 4 && cfg->emb_dim >= 1 &&
                cfg->n_vocab >= 1, "tagger_create: bad geometry");
    std::map<std::string, const ttsamd_tensor*> tm;
    for (int i = 0; i < n; ++i) tm[weights[i].name] = &weights[i];
    int32_t rc = 0;
    auto get = [&](const std::string& name, int64_t cnt) -> const float* {
        if (rc) return nullptr;
        auto it = tm.find(name);
        if (it == tm.end() || gnumel(it->second) != cnt) {
            set_error("tagger: missing or mis-sized tensor '%s' (expected %lld elements)", name.c_str(), (long long)cnt);
            rc = TTSAMD_EINVAL;
            return nullptr;
        }
        return it->second->data;
    };
    std::vector<float> blob;
    std::vector<uint16_t> blob16;
    auto push = [&](const float* p, int64_t cnt) {
        const int64_t off = (int64_t)blob.size();
        blob.insert(blob.end(), p, p + cnt);
        blob.resize(align_up((int64_t)blob.size(), 64));
        return off;
    };
    auto pack = [&](const std::vector<float>& w, const std::vector<float>& bias, int cin, int cout) {
        GConv c;
        c.cin = cin; c.cout = cout;
        const int64_t nn = (int64_t)cin * cout_padded(cout);
        c.w_off = (int64_t)blob.size();
        blob.resize(blob.size() + nn);
        pack_conv_weight(w.data(), cout, cin, 1, blob.data() + c.w_off);
        c.w16_off = (int64_t)blob16.size();
        blob16.resize(blob16.size() + 2 * nn);
        split_packed_bf16(blob.data() + c.w_off, nn, blob16.data() + c.w16_off);
        blob.resize(align_up((int64_t)blob.size(), 64));
        c.b_off = push(bias.data(), cout);
        return c;
    };
    auto* h = new Tagger();
    h->cfg = *cfg;
    const int E = cfg->emb_dim, Ep = (int)align_up(E, 32);
    h->emb_pad = Ep;
    {
        const float* e = get("emb.weight", (int64_t)cfg->n_vocab * E);
        if (e) {
            std::vector<float> t((size_t)cfg->n_vocab * Ep, 0.f);
            for (int v = 0; v < cfg->n_vocab; ++v) std::memcpy(&t[(size_t)v * Ep], e + (size_t)v * E, E * sizeof(float));
            h->emb = push(t.data(), (int64_t)t.size());
        }
    }
    int cin = E, cin_pad = Ep;
    h->max_c = Ep;
    for (int l = 0; l < cfg->n_lstm && !rc; ++l) {
        const int H = cfg->lstm_hidden[l];
        const std::string p = "lstm" + std::to_string(l) + ".";
        TTS_REQUIRE(H >= 1 && H <= 1024 && cin_pad % 32 == 0, "tagger_create: lstm%d geometry (%d -> %d) not supported", l, cin, H);
        const float* wf = get(p + "weight_ih_l0", (int64_t)4 * H * cin);
        const float* wb = get(p + "weight_ih_l0_reverse", (int64_t)4 * H * cin);
        const float* bif = get(p + "bias_ih_l0", 4 * H);
        const float* bhf = get(p + "bias_hh_l0", 4 * H);
        const float* bib = get(p + "bias_ih_l0_reverse", 4 * H);
        const float* bhb = get(p + "bias_hh_l0_reverse", 4 * H);
        const float* hf = get(p + "weight_hh_l0", (int64_t)4 * H * H);
        const float* hb = get(p + "weight_hh_l0_reverse", (int64_t)4 * H * H);
        if (rc) break;
        std::vector<float> scale(cin, 1.f), shift(cin, 0.f);
        if (l == 1 && cfg->bn_after_lstm0) {   // y = (x - mean) * g / sqrt(var + eps) + beta, folded into this projection
            const float* g = get("bn0.weight", cin);
            const float* be = get("bn0.bias", cin);
            const float* mu = get("bn0.running_mean", cin);
            const float* var = get("bn0.running_var", cin);
            if (rc) break;
            for (int k = 0; k < cin; ++k) {
                scale[k] = g[k] / std::sqrt(var[k] + cfg->bn_eps);
                shift[k] = be[k] - mu[k] * scale[k];
            }
        }
        std::vector<float> w((size_t)8 * H * cin_pad, 0.f), bias(8 * H);
        for (int d = 0; d < 2; ++d) {
            const float* src = d ? wb : wf;
            for (int r = 0; r < 4 * H; ++r) {
                double extra = 0.0;
                for (int k = 0; k < cin; ++k) {
                    w[((size_t)d * 4 * H + r) * cin_pad + k] = src[(size_t)r * cin + k] * scale[k];
                    extra += (double)src[(size_t)r * cin + k] * shift[k];
                }
                bias[d * 4 * H + r] = (d ? bib[r] + bhb[r] : bif[r] + bhf[r]) + (float)extra;
            }
        }
        h->xproj.push_back(pack(w, bias, cin_pad, 8 * H));
        for (int d = 0; d < 2; ++d) {
            const float* src = d ? hb : hf;
            std::vector<float> t((size_t)4 * H * H);
            for (int r = 0; r < 4 * H; ++r)
                for (int k = 0; k < H; ++k) t[(size_t)k * 4 * H + r] = src[(size_t)r * H + k];
            (d ? h->whhT_b : h->whhT_f).push_back(push(t.data(), (int64_t)t.size()));
        }
        cin = 2 * H;
        cin_pad = (int)align_up(cin, 32);
        h->max_c = std::max(h->max_c, std::max(8 * H, cin_pad));
    }
    for (int i = 0; i < cfg->n_dense && !rc; ++i) {
        const int dout = cfg->dense_dim[i];
        const std::string p = "dense" + std::to_string(i) + ".";
        const float* w = get(p + "weight", (int64_t)dout * cin);
        const float* b = get(p + "bias", dout);
        if (rc) break;
        TTS_REQUIRE(cin_pad % 32 == 0, "tagger_create: dense%d input width %d not supported", i, cin);
        std::vector<float> wp((size_t)dout * cin_pad, 0.f), bv(b, b + dout);
        for (int r = 0; r < dout; ++r) std::memcpy(&wp[(size_t)r * cin_pad], w + (size_t)r * cin, cin * sizeof(float));
        h->dense.push_back(pack(wp, bv, cin_pad, dout));
        cin = dout;
        cin_pad = (int)align_up(cin, 32);
        h->max_c = std::max(h->max_c, cin_pad);
    }
    if (rc == 0) {
        hipError_t e = hipMalloc((void**)&h->dev, blob.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->dev, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&h->dev16, std::max<size_t>(1, blob16.size()) * sizeof(uint16_t));
        if (e == hipSuccess && !blob16.empty())
            e = hipMemcpy(h->dev16, blob16.data(), blob16.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("tagger_create: upload failed: %s", hipGetErrorString(e));
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

void tagger_destroy(Tagger* h) {
    if (!h) return;
    if (h->dev) (void)hipFree(h->dev);
    if (h->dev16) (void)hipFree(h->dev16);
    delete h;
}

// ------------------------------------------------------------------------------------ kernels

__global__ __launch_bounds__(256) void tagger_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb,
                                                           int n_vocab, int Ep, int T, float* __restrict__ x) {
    const int b = blockIdx.y, tl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    if (t >= T) return;
    const int64_t id = min(max(ids[(int64_t)b * T + t], (int64_t)0), (int64_t)n_vocab - 1);
    const float* er = emb + id * Ep;
    for (int c = g; c < Ep; c += 4) x[((int64_t)b * Ep + c) * T + t] = er[c];
}

__device__ __forceinline__ float tag_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tag_hard_sigmoid(float x) { return fminf(fmaxf(fmaf(0.2f, x, 0.5f), 0.f), 1.f); }

// One direction of one LSTM layer over the full length T (no packing: the reference feeds padded ids as they
// are).  xproj [B][8H][T] holds W_ih x + b_ih + b_hh for both directions; whhT [H][4H]; y [B][2H (+pad)][T].
template <bool HARD>
__global__ __launch_bounds__(1024) void tagger_bilstm_kernel(const float* __restrict__ xproj,
                                                             const float* __restrict__ whhT_f,
                                                             const float* __restrict__ whhT_b, int H, int T, int Cy,
                                                             float* __restrict__ y) {
    extern __shared__ float hs[];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const bool act = j < H;
    const float* __restrict__ wT = dir ? whhT_b : whhT_f;
    const float* xp = xproj + ((int64_t)b * 8 * H + (int64_t)dir * 4 * H) * T;
    float c = 0.f;
    if (act) hs[j] = 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        float hn = 0.f;
        if (act) {
            float g0 = xp[(int64_t)(0 * H + j) * T + t], g1 = xp[(int64_t)(1 * H + j) * T + t];
            float g2 = xp[(int64_t)(2 * H + j) * T + t], g3 = xp[(int64_t)(3 * H + j) * T + t];
            for (int k = 0; k < H; ++k) {
                const float hk = hs[k];
                const float* wr = wT + (int64_t)k * 4 * H + j;
                g0 = fmaf(wr[0], hk, g0);
                g1 = fmaf(wr[H], hk, g1);
                g2 = fmaf(wr[2 * H], hk, g2);
                g3 = fmaf(wr[3 * H], hk, g3);
            }
            const float ig = HARD ? tag_hard_sigmoid(g0) : tag_sigmoid(g0);
            const float fg = HARD ? tag_hard_sigmoid(g1) : tag_sigmoid(g1);
            const float og = HARD ? tag_hard_sigmoid(g3) : tag_sigmoid(g3);
            c = fg * c + ig * tanhf(g2);
            hn = og * tanhf(c);
        }
        __syncthreads();
        if (act) {
            hs[j] = hn;
            y[((int64_t)b * Cy + dir * H + j) * T + t] = hn;
        }
        __syncthreads();
    }
}

// probs[b][t][k] = softmax_k(logits[b][k][t])
__global__ __launch_bounds__(256) void tagger_softmax_kernel(const float* __restrict__ logits, int n_cls, int T,
                                                             float* __restrict__ probs) {
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const float* lp = logits + (int64_t)b * n_cls * T + t;
    float mx = -INFINITY;
    for (int k = 0; k < n_cls; ++k) mx = fmaxf(mx, lp[(int64_t)k * T]);
    float sm = 0.f;
    for (int k = 0; k < n_cls; ++k) sm += expf(lp[(int64_t)k * T] - mx);
    const float inv = 1.0f / sm;
    float* pp = probs + ((int64_t)b * T + t) * n_cls;
    for (int k = 0; k < n_cls; ++k) pp[k] = expf(lp[(int64_t)k * T] - mx) * inv;
}

// ------------------------------------------------------------------------------------ host

int64_t tagger_workspace_bytes(const Tagger* h, int32_t B, int32_t T) {
    return 2 * align_up((int64_t)B * h->max_c * T * (int64_t)sizeof(float), 256);
}

static int32_t gconv(const Tagger* h, const GConv& c, const float* x, float* y, int B, int T, int relu, hipStream_t s) {
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.x_bs = (int64_t)c.cin * T; p.x_cs = T;
    p.w = h->dev + c.w_off; p.bias = h->dev + c.b_off;
    p.w_bf16 = h->dev16 + c.w16_off; p.precision = 0;          // taggers decide by arg-max: always exact fp32
    p.y = y; p.y_bs = (int64_t)c.cout * T; p.y_cs = T; p.y_ts = 1;
    p.len_in_mul = 1; p.len_out_mul = 1; p.Lin = T; p.Nout = T;
    p.Cin = c.cin; p.Cout = c.cout; p.CoutP = cout_padded(c.cout); p.K = 1;
    p.dil = 1; p.pad = 0; p.n_phase = 1; p.in_slope = 1.f; p.relu_out = relu; p.mode = 0; p.div = 1.f; p.batch = B;
    return launch_conv(p, s);
}

int32_t tagger_forward(const Tagger* h, const int64_t* ids, int32_t B, int32_t T, float* probs, void* ws, int64_t ws_bytes,
                       hipStream_t s) {
    TTS_REQUIRE(h && ids && probs, "tagger_forward: null argument");
    TTS_REQUIRE(B >= 1 && T >= 1, "tagger_forward: bad batch/length (%d, %d)", B, T);
    const int64_t need = tagger_workspace_bytes(h, B, T);
    if (!ws || ws_bytes < need) {
        set_error("tagger_forward: workspace of %lld bytes needed, %lld given", (long long)need, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    float* buf0 = (float*)ws;
    float* buf1 = (float*)((char*)ws + need / 2);
    const ttsamd_tagger_cfg& c = h->cfg;
    hipLaunchKernelGGL(tagger_embed_kernel, dim3((T + 63) / 64, B), dim3(256), 0, s, ids, h->dev + h->emb, c.n_vocab,
                       h->emb_pad, T, buf0);
    TTS_CHECK_HIP(hipGetLastError());
    float *x = buf0, *o = buf1;
    for (int l = 0; l < c.n_lstm; ++l) {
        const int H = c.lstm_hidden[l], Cy = (int)align_up(2 * H, 32);
        TTS_TRY(gconv(h, h->xproj[l], x, o, B, T, 0, s));                       // o = [B][8H][T]
        if (Cy != 2 * H) TTS_CHECK_HIP(hipMemsetAsync(x, 0, (size_t)B * Cy * T * sizeof(float), s));
        const int threads = (int)align_up(H, 64);
        const size_t lds = (size_t)H * sizeof(float);
        if (c.hard_sigmoid)
            hipLaunchKernelGGL(tagger_bilstm_kernel<true>, dim3(B, 2), dim3(threads), lds, s, o, h->dev + h->whhT_f[l],
                               h->dev + h->whhT_b[l], H, T, Cy, x);
        else
            hipLaunchKernelGGL(tagger_bilstm_kernel<false>, dim3(B, 2), dim3(threads), lds, s, o, h->dev + h->whhT_f[l],
                               h->dev + h->whhT_b[l], H, T, Cy, x);
        TTS_CHECK_HIP(hipGetLastError());                                      // x = [B][2H][T]
    }
    for (int i = 0; i < c.n_dense; ++i) {
        const bool last = i == c.n_dense - 1;
        TTS_TRY(gconv(h, h->dense[i], x, o, B, T, last ? 0 : 1, s));
        if (!last && (h->dense[i].cout % 32) != 0) {
            set_error("tagger_forward: hidden dense width %d must be a multiple of 32", h->dense[i].cout);
            return TTSAMD_EINVAL;
        }
        std::swap(x, o);
    }
    const int n_cls = c.dense_dim[c.n_dense - 1];
    hipLaunchKernelGGL(tagger_softmax_kernel, dim3((T + 255) / 256, B), dim3(256), 0, s, x, n_cls, T, probs);
    TTS_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace ttsamd
