// Tacotron2MS.infer on the MI355X (BASELINE config 4).
// Replaces models/tacotron2/tacotron2_ms.py:279-332 and the torchaudio.models.tacotron2 private
// classes it instantiates (:188-207: _Encoder, _Decoder.infer, _Postnet) — restated from the
// published architecture because torchaudio is not vendored (parity unpinned).
//   encoder : embedding gather -> 3 x [Conv1d k5 + BatchNorm(eval, folded) + ReLU] on the MFMA conv
//             engine -> input projections of both LSTM directions as one 1x1 conv -> a persistent
//             BiLSTM kernel (one block per (utterance, direction), packed-sequence semantics)
//   decoder : one autoregressive step = prenet (dropout from a counter-based hash) -> attention
//             LSTMCell -> location-sensitive attention -> decoder LSTMCell -> mel/gate projection;
//             the LSTM cells are weight-streaming GEMVs (75 MB of fp32 weights per step, resident
//             in L2/MALL), one wave64 per hidden unit; the stop test is read back every 8 steps
//   postnet : 5 x [Conv1d k5 + BatchNorm folded (+ tanh)] on the conv engine, residual fused.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace ttsamd {

#ifndef TP_SKIP
#define TP_SKIP 0
#endif

struct TConv {
    int64_t w_off = 0, b_off = 0, w16_off = 0;
    int cin = 0, cout = 0, k = 0;
};

struct Taco2 {
    ttsamd_tacotron2_cfg cfg;
    float* dev = nullptr;
    uint16_t* dev16 = nullptr;
    int64_t emb = 0, spk = -1;
    std::vector<TConv> enc_convs, post_convs;
    TConv enc_xproj;
    int64_t enc_whhT[2];
    int64_t pre0, pre1;
    int64_t att_wih, att_whh, att_b, dec_wih, dec_whh, dec_b;
    int64_t wq, wmT, v, loc_conv, loc_denseT;
    int64_t proj_w, proj_b;
    int64_t loc_fold = 0, projx_w = 0, projx_b = 0;   // persistent decoder: folded location filter, projection + folded prenet layer 1
    int mem_dim = 0;
    // host side of the stop test (one infer call at a time per handle: guarded by mu)
    mutable std::mutex mu;
    mutable int32_t* pinned = nullptr;      // [TACO_RING][pinned_cap] finished flags copied back asynchronously
    mutable int pinned_cap = 0;
    // persistent decoder: after a hand-off time-out (the 256 blocks were not co-resident: another stream held CUs) the next
    // `persist_skip` calls go straight to the graph path instead of paying the 80 ms spin again; doubles up to 256 on every
    // further time-out, resets on the first success
    mutable int persist_skip = 0, persist_backoff = 0, persist_launch_failures = 0;
    mutable hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    mutable hipEvent_t ev_in = nullptr;
    mutable hipStream_t loop_stream = nullptr;   // capture is not allowed on the legacy default stream torch hands us
};

using TensorMap = std::map<std::string, const ttsamd_tensor*>;

static int64_t tnumel(const ttsamd_tensor* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}

struct TBuilder {
    const TensorMap& tm;
    std::vector<float> blob;
    std::vector<uint16_t> blob16;
    int32_t rc = 0;
    explicit TBuilder(const TensorMap& t) : tm(t) {}
    const float* get(const std::string& name, int64_t n) {
        if (rc) return nullptr;
        auto it = tm.find(name);
        if (it == tm.end() || tnumel(it->second) != n) {
            set_error("tacotron2: missing or mis-sized tensor '%s' (expected %lld elements)", name.c_str(), (long long)n);
            rc = TTSAMD_EINVAL;
            return nullptr;
        }
        return it->second->data;
    }
    int64_t push(const float* p, int64_t n) {
        const int64_t off = (int64_t)blob.size();
        if (p) blob.insert(blob.end(), p, p + n);
        blob.resize(align_up((int64_t)blob.size(), 64));
        return off;
    }
    int64_t raw(const std::string& name, int64_t n) { return push(get(name, n), n); }
    // [rows][cols] row-major -> [cols][rows]
    int64_t transposed(const std::string& name, int rows, int cols) {
        const float* p = get(name, (int64_t)rows * cols);
        if (!p) return 0;
        std::vector<float> t((size_t)rows * cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = p[(size_t)r * cols + c];
        return push(t.data(), (int64_t)t.size());
    }
    int64_t bias_sum(const std::string& a, const std::string& b2, int n) {
        const float *pa = get(a, n), *pb = get(b2, n);
        if (!pa || !pb) return 0;
        std::vector<float> t(n);
        for (int i = 0; i < n; ++i) t[i] = pa[i] + pb[i];
        return push(t.data(), n);
    }
    TConv conv_packed(const float* w, const float* bias, int cin, int cout, int k) {
        TConv c;
        c.cin = cin; c.cout = cout; c.k = k;
        const int64_t nn = (int64_t)cin * k * cout_padded(cout);
        c.w_off = (int64_t)blob.size();
        blob.resize(blob.size() + nn);
        pack_conv_weight(w, cout, cin, k, blob.data() + c.w_off);
        c.w16_off = (int64_t)blob16.size();
        blob16.resize(blob16.size() + 2 * nn);
        split_packed_bf16(blob.data() + c.w_off, nn, blob16.data() + c.w16_off);
        blob.resize(align_up((int64_t)blob.size(), 64));
        c.b_off = push(bias, cout);
        return c;
    }
    // Conv1d followed by BatchNorm1d in eval mode, folded: w' = w*s, b' = (b - mean)*s + beta, s = gamma/sqrt(var+eps)
    TConv conv_bn(const std::string& base, int cin, int cout, int k) {
        const float* w = get(base + ".0.weight", (int64_t)cout * cin * k);
        const float* b = get(base + ".0.bias", cout);
        const float* g = get(base + ".1.weight", cout);
        const float* be = get(base + ".1.bias", cout);
        const float* mu = get(base + ".1.running_mean", cout);
        const float* var = get(base + ".1.running_var", cout);
        if (rc) return TConv();
        std::vector<float> wf((size_t)cout * cin * k), bf(cout);
        for (int co = 0; co < cout; ++co) {
            const float s = g[co] / std::sqrt(var[co] + 1e-5f);
            for (int64_t i = 0; i < (int64_t)cin * k; ++i) wf[(size_t)co * cin * k + i] = w[(size_t)co * cin * k + i] * s;
            bf[co] = (b[co] - mu[co]) * s + be[co];
        }
        return conv_packed(wf.data(), bf.data(), cin, cout, k);
    }
};

int32_t tacotron2_create(const ttsamd_tensor* weights, int32_t n, const ttsamd_tacotron2_cfg* cfg, Taco2** out) {
    TTS_REQUIRE(weights && cfg && out, "tacotron2_create: null argument");
    const int E = cfg->encoder_embedding_dim, S = cfg->num_speakers > 1 ? cfg->speaker_embedding_dim : 0;
    const int M = E + S, P = cfg->prenet_dim, A = cfg->attention_rnn_dim, D = cfg->decoder_rnn_dim;
    const int Hd = cfg->attention_hidden_dim, NF = cfg->attention_location_n_filter, KS = cfg->attention_location_kernel_size;
    TTS_REQUIRE(cfg->symbol_embedding_dim == E && E % 16 == 0 && (E / 2) == 256 && A == 1024 && D % 4 == 0 && D <= 1024 && Hd == 128 &&
                NF == 32 && KS <= 63 && P == 256 && cfg->n_mels == 80,
                "tacotron2_create: only the shipped geometry is built (512/256/1024/1024/128/32/80)");
    TensorMap tm;
    for (int i = 0; i < n; ++i) tm[weights[i].name] = &weights[i];
    TBuilder b(tm);
    auto* h = new Taco2();
    h->cfg = *cfg;
    h->mem_dim = M;
    h->emb = b.raw("embedding.weight", (int64_t)cfg->n_symbol * E);
    if (S) h->spk = b.raw("speaker_embedding.weight", (int64_t)cfg->num_speakers * S);
    for (int i = 0; i < cfg->encoder_n_convolution && !b.rc; ++i)
        h->enc_convs.push_back(b.conv_bn("encoder.convolutions." + std::to_string(i), E, E, cfg->encoder_kernel_size));
    {   // both directions' input projections as one 1x1 conv: rows [fwd i,f,g,o | bwd i,f,g,o], bias = b_ih + b_hh
        const int Hh = E / 2;
        const float* wf = b.get("encoder.lstm.weight_ih_l0", (int64_t)4 * Hh * E);
        const float* wb = b.get("encoder.lstm.weight_ih_l0_reverse", (int64_t)4 * Hh * E);
        const float* bif = b.get("encoder.lstm.bias_ih_l0", 4 * Hh);
        const float* bhf = b.get("encoder.lstm.bias_hh_l0", 4 * Hh);
        const float* bib = b.get("encoder.lstm.bias_ih_l0_reverse", 4 * Hh);
        const float* bhb = b.get("encoder.lstm.bias_hh_l0_reverse", 4 * Hh);
        if (!b.rc) {
            std::vector<float> w((size_t)8 * Hh * E), bias(8 * Hh);
            std::memcpy(w.data(), wf, (size_t)4 * Hh * E * sizeof(float));
            std::memcpy(w.data() + (size_t)4 * Hh * E, wb, (size_t)4 * Hh * E * sizeof(float));
            for (int i = 0; i < 4 * Hh; ++i) {
                bias[i] = bif[i] + bhf[i];
                bias[4 * Hh + i] = bib[i] + bhb[i];
            }
            h->enc_xproj = b.conv_packed(w.data(), bias.data(), E, 8 * Hh, 1);
        }
        h->enc_whhT[0] = b.transposed("encoder.lstm.weight_hh_l0", 4 * Hh, Hh);
        h->enc_whhT[1] = b.transposed("encoder.lstm.weight_hh_l0_reverse", 4 * Hh, Hh);
    }
    h->pre0 = b.raw("decoder.prenet.layers.0.weight", (int64_t)P * cfg->n_mels);
    h->pre1 = b.raw("decoder.prenet.layers.1.weight", (int64_t)P * P);
    h->att_wih = b.raw("decoder.attention_rnn.weight_ih", (int64_t)4 * A * (P + M));
    h->att_whh = b.raw("decoder.attention_rnn.weight_hh", (int64_t)4 * A * A);
    h->att_b = b.bias_sum("decoder.attention_rnn.bias_ih", "decoder.attention_rnn.bias_hh", 4 * A);
    h->wq = b.raw("decoder.attention_layer.query_layer.weight", (int64_t)Hd * A);
    h->wmT = b.transposed("decoder.attention_layer.memory_layer.weight", Hd, M);
    h->v = b.raw("decoder.attention_layer.v.weight", Hd);
    h->loc_conv = b.raw("decoder.attention_layer.location_layer.location_conv.weight", (int64_t)NF * 2 * KS);
    h->loc_denseT = b.transposed("decoder.attention_layer.location_layer.location_dense.weight", Hd, NF);
    h->dec_wih = b.raw("decoder.decoder_rnn.weight_ih", (int64_t)4 * D * (A + M));
    h->dec_whh = b.raw("decoder.decoder_rnn.weight_hh", (int64_t)4 * D * D);
    h->dec_b = b.bias_sum("decoder.decoder_rnn.bias_ih", "decoder.decoder_rnn.bias_hh", 4 * D);
    {   // mel projection rows 0..79 and the gate as row 80
        const float* pw = b.get("decoder.linear_projection.weight", (int64_t)cfg->n_mels * (D + M));
        const float* pb = b.get("decoder.linear_projection.bias", cfg->n_mels);
        const float* gw = b.get("decoder.gate_layer.weight", D + M);
        const float* gb = b.get("decoder.gate_layer.bias", 1);
        if (!b.rc) {
            std::vector<float> w((size_t)(cfg->n_mels + 1) * (D + M)), bias(cfg->n_mels + 1);
            std::memcpy(w.data(), pw, (size_t)cfg->n_mels * (D + M) * sizeof(float));
            std::memcpy(w.data() + (size_t)cfg->n_mels * (D + M), gw, (size_t)(D + M) * sizeof(float));
            std::memcpy(bias.data(), pb, cfg->n_mels * sizeof(float));
            bias[cfg->n_mels] = gb[0];
            h->proj_w = b.push(w.data(), (int64_t)w.size());
            h->proj_b = b.push(bias.data(), (int64_t)bias.size());
            // persistent decoder (taco_decoder_persistent): rows 0..80 as above, rows 81..336 = prenet layer 1 folded
            // into the projection, W0 * Wp and W0 * bp (the prenet has no bias and nothing but the mel projection
            // between it and [dec_h | ctx]); accumulated in double so the fold adds no rounding of its own
            const float* w0 = b.get("decoder.prenet.layers.0.weight", (int64_t)P * cfg->n_mels);
            if (w0) {
                const int KP = D + M, NM = cfg->n_mels;
                std::vector<float> wx((size_t)(NM + 1 + P) * KP), bx(NM + 1 + P);
                std::memcpy(wx.data(), w.data(), w.size() * sizeof(float));
                std::memcpy(bx.data(), bias.data(), bias.size() * sizeof(float));
                std::vector<double> row(KP);
                for (int j = 0; j < P; ++j) {
                    std::fill(row.begin(), row.end(), 0.0);
                    double bj = 0.0;
                    for (int m = 0; m < NM; ++m) {
                        const double a = w0[(size_t)j * NM + m];
                        const float* pr = pw + (size_t)m * KP;
                        for (int k = 0; k < KP; ++k) row[k] += a * pr[k];
                        bj += a * pb[m];
                    }
                    for (int k = 0; k < KP; ++k) wx[(size_t)(NM + 1 + j) * KP + k] = (float)row[k];
                    bx[NM + 1 + j] = (float)bj;
                }
                h->projx_w = b.push(wx.data(), (int64_t)wx.size());
                h->projx_b = b.push(bx.data(), (int64_t)bx.size());
            }
        }
    }
    {   // location layer folded: G[h][c][k] = sum_q dense[h][q] * conv[q][c][k]   (no nonlinearity between the two)
        const float* lc = b.get("decoder.attention_layer.location_layer.location_conv.weight", (int64_t)NF * 2 * KS);
        const float* ld = b.get("decoder.attention_layer.location_layer.location_dense.weight", (int64_t)Hd * NF);
        if (!b.rc) {
            std::vector<float> g((size_t)Hd * 2 * KS);
            for (int hh = 0; hh < Hd; ++hh)
                for (int i = 0; i < 2 * KS; ++i) {
                    double a = 0.0;
                    for (int q = 0; q < NF; ++q) a += (double)ld[(size_t)hh * NF + q] * lc[(size_t)q * 2 * KS + i];
                    g[(size_t)hh * 2 * KS + i] = (float)a;
                }
            h->loc_fold = b.push(g.data(), (int64_t)g.size());
        }
    }
    for (int i = 0; i < cfg->postnet_n_convolution && !b.rc; ++i) {
        const int cin = i == 0 ? cfg->n_mels : cfg->postnet_embedding_dim;
        const int cout = i == cfg->postnet_n_convolution - 1 ? cfg->n_mels : cfg->postnet_embedding_dim;
        h->post_convs.push_back(b.conv_bn("postnet.convolutions." + std::to_string(i), cin, cout, cfg->postnet_kernel_size));
    }
    int32_t rc = b.rc;
    if (rc == 0) {
        hipError_t e = hipMalloc((void**)&h->dev, b.blob.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->dev, b.blob.data(), b.blob.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&h->dev16, b.blob16.size() * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemcpy(h->dev16, b.blob16.data(), b.blob16.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("tacotron2_create: upload failed: %s", hipGetErrorString(e));
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

void tacotron2_destroy(Taco2* h) {
    if (!h) return;
    if (h->pinned) (void)hipHostFree(h->pinned);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->loop_stream) (void)hipStreamDestroy(h->loop_stream);
    if (h->dev) (void)hipFree(h->dev);
    if (h->dev16) (void)hipFree(h->dev16);
    delete h;
}

// ------------------------------------------------------------------------------------ kernels

__global__ __launch_bounds__(256) void taco_embed_kernel(const int64_t* __restrict__ tok, const float* __restrict__ emb,
                                                         int n_symbol, int E, int L, float* __restrict__ x) {
    const int b = blockIdx.y, tl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    if (t >= L) return;
    const float* er = emb + min(max(tok[(int64_t)b * L + t], (int64_t)0), (int64_t)n_symbol - 1) * E;   // clamped: memory safety
    for (int c = g; c < E; c += 4) x[((int64_t)b * E + c) * L + t] = er[c];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Packed-sequence BiLSTM (hidden 256 per direction): block = (utterance, direction), thread = hidden unit.
// xproj [B][2048][L] already holds W_ih x + b_ih + b_hh; whhT [256][1024]; memory [B][L][M] (pre-zeroed).
__global__ __launch_bounds__(256) void taco_bilstm_kernel(const float* __restrict__ xproj,
                                                          const float* __restrict__ whhT_f,
                                                          const float* __restrict__ whhT_b,
                                                          const int64_t* __restrict__ lens, int L, int M,
                                                          float* __restrict__ memory) {
    __shared__ float hs[256];
    const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
    const float* __restrict__ wT = dir ? whhT_b : whhT_f;
    const int n = min((int)lens[b], L);
    float c = 0.f;
    hs[j] = 0.f;
    __syncthreads();
    const float* xp = xproj + ((int64_t)b * 2048 + dir * 1024) * L;
    for (int s = 0; s < n; ++s) {
        const int t = dir ? n - 1 - s : s;
        float g0 = xp[(int64_t)(0 * 256 + j) * L + t], g1 = xp[(int64_t)(1 * 256 + j) * L + t];
        float g2 = xp[(int64_t)(2 * 256 + j) * L + t], g3 = xp[(int64_t)(3 * 256 + j) * L + t];
        for (int k = 0; k < 256; ++k) {
            const float hk = hs[k];
            const float* wr = wT + (int64_t)k * 1024 + j;
            g0 = fmaf(wr[0], hk, g0);
            g1 = fmaf(wr[256], hk, g1);
            g2 = fmaf(wr[512], hk, g2);
            g3 = fmaf(wr[768], hk, g3);
        }
        c = sigmoidf_(g1) * c + sigmoidf_(g0) * tanhf(g2);
        const float hn = sigmoidf_(g3) * tanhf(c);
        __syncthreads();
        hs[j] = hn;
        memory[((int64_t)b * L + t) * M + dir * 256 + j] = hn;
        __syncthreads();
    }
}

__global__ void taco_spk_kernel(float* __restrict__ memory, const float* __restrict__ spk, const int64_t* __restrict__ sids,
                                int n_spk, int L, int M, int E, int S) {
    const int b = blockIdx.y, t = blockIdx.x, j = threadIdx.x;
    if (j < S) memory[((int64_t)b * L + t) * M + E + j] = spk[min(max(sids[b], (int64_t)0), (int64_t)n_spk - 1) * S + j];
}

// pm[b][t][h] = sum_m memory[b][t][m] * wmT[m][h]      (attention memory_layer, no bias)
__global__ __launch_bounds__(128) void taco_pm_kernel(const float* __restrict__ memory, const float* __restrict__ wmT,
                                                      int L, int M, float* __restrict__ pm) {
    const int b = blockIdx.y, t = blockIdx.x, hh = threadIdx.x;
    const float* mr = memory + ((int64_t)b * L + t) * M;
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc = fmaf(mr[m], wmT[(int64_t)m * 128 + hh], acc);
    pm[((int64_t)b * L + t) * 128 + hh] = acc;
}

__device__ __forceinline__ float taco_keep(unsigned seed, unsigned layer, unsigned step, unsigned b, unsigned j) {
    unsigned x = seed * 0x9E3779B1u + layer * 0x85EBCA77u + step * 0xC2B2AE3Du + b * 0x27D4EB2Fu + j * 0x165667B1u;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return (x & 1u) ? 2.0f : 0.0f;
}

// prenet: 2 x [Linear(no bias) + ReLU + dropout(p=0.5, always on upstream)]; grid (utterance, 8): every block
// redoes layer 1 (80 KB of weights, 4 threads per unit) and owns 32 units of layer 2 — one CU streams only
// ~25-50 GB/s from the MALL, so the 336 KB of prenet weights must not go through a single block per utterance
__global__ __launch_bounds__(1024) void taco_prenet_kernel(const float* __restrict__ dec_in, const float* __restrict__ w0,
                                                           const float* __restrict__ w1, int n_mels, long long seed,
                                                           const int* __restrict__ step_base, int step_off,
                                                           float* __restrict__ out) {
    __shared__ float xin[128], h0[256];
    const int b = blockIdx.x, tid = threadIdx.x, j = tid >> 2, sub = tid & 3;
    const int step = *step_base + step_off;
    if (tid < n_mels) xin[tid] = dec_in[(int64_t)b * n_mels + tid];
    __syncthreads();
    {
        const int kq = n_mels / 4;                   // 80 -> 20 inputs per thread
        const float* wr = w0 + (int64_t)j * n_mels + sub * kq;
        float a = 0.f;
        for (int k = 0; k < kq; k += 4) {
            const float4 wv = *reinterpret_cast<const float4*>(wr + k);
            const int kk = sub * kq + k;
            a = fmaf(wv.x, xin[kk], fmaf(wv.y, xin[kk + 1], fmaf(wv.z, xin[kk + 2], fmaf(wv.w, xin[kk + 3], a))));
        }
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        a = fmaxf(a, 0.f);
        if (seed >= 0) a *= taco_keep((unsigned)seed, 0u, (unsigned)step, (unsigned)b, (unsigned)j);
        if (sub == 0) h0[j] = a;
    }
    __syncthreads();
    {   // layer 2: this block's 32 of the 256 units, 32 lanes per unit (8 inputs each)
        const int jj = blockIdx.y * 32 + (tid >> 5), l32 = tid & 31;
        const float* wr = w1 + (int64_t)jj * 256 + l32 * 8;
        const float4 wa = *reinterpret_cast<const float4*>(wr), wb = *reinterpret_cast<const float4*>(wr + 4);
        const int kk = l32 * 8;
        float a = fmaf(wa.x, h0[kk], fmaf(wa.y, h0[kk + 1], fmaf(wa.z, h0[kk + 2], wa.w * h0[kk + 3])));
        a = fmaf(wb.x, h0[kk + 4], fmaf(wb.y, h0[kk + 5], fmaf(wb.z, h0[kk + 6], fmaf(wb.w, h0[kk + 7], a))));
        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o);
        a = fmaxf(a, 0.f);
        if (seed >= 0) a *= taco_keep((unsigned)seed, 1u, (unsigned)step, (unsigned)b, (unsigned)jj);
        if (l32 == 0) out[(int64_t)b * 256 + jj] = a;
    }
}

// ---- block-level GEMV pieces of the decoder step ------------------------------------------------
// The decoder's matrices (75 MB of fp32 LSTM weights, 0.5 MB query, 0.5 MB projection) are streamed once per
// step: a 256-thread block owns NR output rows, the K axis is split over the threads as float4 and the BC
// batch columns ride along in registers (x is tiny and L2-resident), then a shuffle + LDS reduction.
constexpr int TACO_BC = 8;

template <int NR>
__device__ __forceinline__ void block_dots_acc(const float* __restrict__ w, int64_t row_stride, int K,
                                               const float* __restrict__ x, int x_stride, int b0, int B,
                                               float (&acc)[NR][TACO_BC]) {
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        float4 wv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) wv[r] = *reinterpret_cast<const float4*>(w + r * row_stride + k);
#pragma unroll
        for (int bb = 0; bb < TACO_BC; ++bb) {
            const int b = min(b0 + bb, B - 1);
            const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)b * x_stride + k);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                acc[r][bb] = fmaf(wv[r].x, xv.x, fmaf(wv[r].y, xv.y, fmaf(wv[r].z, xv.z, fmaf(wv[r].w, xv.w, acc[r][bb]))));
        }
    }
}

// sums acc over the 256 threads; the totals land in red[r][bb] (valid after the trailing barrier)
template <int NR>
__device__ __forceinline__ void block_dots_reduce(float (&acc)[NR][TACO_BC], float (*red)[TACO_BC], float (*part)[NR][TACO_BC]) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int bb = 0; bb < TACO_BC; ++bb) {
            float v = acc[r][bb];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) part[wid][r][bb] = v;
        }
    __syncthreads();
    if (threadIdx.x < NR * TACO_BC) {
        const int r = threadIdx.x / TACO_BC, bb = threadIdx.x % TACO_BC;
        red[r][bb] = part[0][r][bb] + part[1][r][bb] + part[2][r][bb] + part[3][r][bb];
    }
    __syncthreads();
}

// LSTMCell as a column-parallel weight-streaming GEMV (tools/gemv_bench.hip: 24.5 us for the attention +
// decoder pair at B=8 vs 38.5 for a row-per-block layout; a bare read of the same 75.5 MB takes 11.5 us).
// thread = one float4 column group of [x1 | x2 | h] with its 8 batch columns of x in registers (loaded once);
// block = 2 hidden units = 8 gate rows, one float4 weight load per row and thread (a wave reads 1 KB of a
// row).  The thread's 64 (row, batch) partials are summed over the wave with a transposing butterfly
// (63 shuffles, lane l ends with partial l), over the waves through LDS; c updated in place, h ping-pong.
template <int NW>
__global__ __launch_bounds__(NW * 64) void taco_lstm_kernel(const float* __restrict__ x1, int n1,
                                                            const float* __restrict__ x2, int n2,
                                                            const float* __restrict__ h_in, float* __restrict__ c,
                                                            const float* __restrict__ wih, const float* __restrict__ whh,
                                                            const float* __restrict__ bias, float* __restrict__ h_out,
                                                            int B, int H) {
    __shared__ float part[NW][64], gates[64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int K1 = n1 + n2, K4 = (K1 + H) / 4;
    const bool act = tid < K4;
    const int k = 4 * min(tid, K4 - 1);
    const int u0 = blockIdx.x * 2;
    float4 w[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {          // row r = (unit r >> 2, gate r & 3)
        const int64_t row = (int64_t)(r & 3) * H + (u0 + (r >> 2));
        w[r] = *reinterpret_cast<const float4*>(k < K1 ? wih + row * K1 + k : whh + row * H + (k - K1));
    }
    for (int b0 = 0; b0 < B; b0 += TACO_BC) {
        float v[64];
#pragma unroll
        for (int bb = 0; bb < TACO_BC; ++bb) {
            const int b = min(b0 + bb, B - 1);
            const float* src = k < n1 ? x1 + (int64_t)b * n1 + k
                             : k < K1 ? x2 + (int64_t)b * n2 + (k - n1) : h_in + (int64_t)b * H + (k - K1);
            float4 xv = *reinterpret_cast<const float4*>(src);
            if (!act) xv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                v[r * 8 + bb] = fmaf(w[r].x, xv.x, fmaf(w[r].y, xv.y, fmaf(w[r].z, xv.z, w[r].w * xv.w)));
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int m = 32 >> s, half = 32 >> s;
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const float send = upper ? v[j] : v[j + half];
                const float keep = upper ? v[j + half] : v[j];
                v[j] = keep + __shfl_xor(send, m);
            }
        }
        part[wid][lane] = v[0];
        __syncthreads();
        if (tid < 64) {
            float g = 0.f;
#pragma unroll
            for (int q = 0; q < NW; ++q) g += part[q][tid];
            gates[tid] = g;                 // [unit (2)][gate (4)][batch (8)]
        }
        __syncthreads();
        if (tid < 2 * TACO_BC) {
            const int uu = tid / TACO_BC, bb = tid % TACO_BC, b = b0 + bb, u = u0 + uu;
            if (b < B) {
                const float* gp = gates + uu * 32 + bb;
                const float gi = gp[0] + bias[u], gf = gp[8] + bias[H + u];
                const float gg = gp[16] + bias[2 * H + u], go = gp[24] + bias[3 * H + u];
                const float cn = sigmoidf_(gf) * c[(int64_t)b * H + u] + sigmoidf_(gi) * tanhf(gg);
                c[(int64_t)b * H + u] = cn;
                h_out[(int64_t)b * H + u] = sigmoidf_(go) * tanhf(cn);
            }
        }
        __syncthreads();
    }
}

// processed query pq[b][r] = sum_k wq[r][k] att_h[b][k]   (query_layer, no bias), column-parallel like the
// LSTM kernel: thread = float4 column group (A/4 = 256 threads), block = 8 of the 128 rows
__global__ __launch_bounds__(256) void taco_query_kernel(const float* __restrict__ att_h, int A,
                                                         const float* __restrict__ wq, float* __restrict__ pq, int B) {
    __shared__ float part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r0 = blockIdx.x * 8, k = 4 * tid;
    float4 w[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r] = *reinterpret_cast<const float4*>(wq + (int64_t)(r0 + r) * A + k);
    for (int b0 = 0; b0 < B; b0 += TACO_BC) {
        float v[64];
#pragma unroll
        for (int bb = 0; bb < TACO_BC; ++bb) {
            const float4 xv = *reinterpret_cast<const float4*>(att_h + (int64_t)min(b0 + bb, B - 1) * A + k);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                v[r * 8 + bb] = fmaf(w[r].x, xv.x, fmaf(w[r].y, xv.y, fmaf(w[r].z, xv.z, w[r].w * xv.w)));
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int m = 32 >> s, half = 32 >> s;
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const float send = upper ? v[j] : v[j + half];
                const float keep = upper ? v[j + half] : v[j];
                v[j] = keep + __shfl_xor(send, m);
            }
        }
        part[wid][lane] = v[0];
        __syncthreads();
        if (tid < 64) {
            const int r = tid >> 3, bb = tid & 7;
            if (b0 + bb < B) pq[(int64_t)(b0 + bb) * 128 + r0 + r] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
        }
        __syncthreads();
    }
}

// Location-sensitive attention energies, one wave64 per (utterance, token):
//   f[q]  = sum_k lc[q][0][k] aw[t+k-h] + lc[q][1][k] cum[t+k-h]        (32 filters; lane halves = the 2 channels)
//   e[t]  = sum_h v[h] tanh(pq[h] + sum_q ldT[q][h] f[q] + pm[t][h])      (128 hidden units, 2 per lane)
constexpr int TACO_LMAX = 1024;
__global__ __launch_bounds__(256) void taco_energy_kernel(const float* __restrict__ pq, const float* __restrict__ pm,
                                                          const float* __restrict__ loc_conv, int KS,
                                                          const float* __restrict__ loc_denseT,
                                                          const float* __restrict__ v, const float* __restrict__ aw,
                                                          const float* __restrict__ aw_cum, int L,
                                                          float* __restrict__ e) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= L) return;
    const int q = lane & 31, ch = lane >> 5, half = (KS - 1) / 2;
    const float* src = (ch ? aw_cum : aw) + (int64_t)b * L;
    const float* lc = loc_conv + (q * 2 + ch) * KS;
    float f = 0.f;
    for (int k = 0; k < KS; ++k) {
        const int sidx = t + k - half;
        const float a = (sidx >= 0 && sidx < L) ? src[sidx] : 0.f;
        f = fmaf(lc[k], a, f);
    }
    f += __shfl_xor(f, 32);
    float pl0 = 0.f, pl1 = 0.f;
#pragma unroll
    for (int qq = 0; qq < 32; ++qq) {
        const float fq = __shfl(f, qq);
        pl0 = fmaf(loc_denseT[qq * 128 + lane], fq, pl0);
        pl1 = fmaf(loc_denseT[qq * 128 + 64 + lane], fq, pl1);
    }
    const float* pmr = pm + ((int64_t)b * L + t) * 128;
    const float* pqr = pq + (int64_t)b * 128;
    float acc = v[lane] * tanhf(pqr[lane] + pl0 + pmr[lane]) + v[64 + lane] * tanhf(pqr[64 + lane] + pl1 + pmr[64 + lane]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) e[(int64_t)b * L + t] = acc;
}

// masked softmax over the tokens + context vector; grid (B, ceil(M/128)): every block redoes the (tiny)
// softmax, block y == 0 also publishes aw, aw_cum += aw and the alignment row.
__global__ __launch_bounds__(128) void taco_context_kernel(const float* __restrict__ e, const float* __restrict__ memory,
                                                           int M, const int64_t* __restrict__ lens, int L,
                                                           float* __restrict__ aw, float* __restrict__ aw_cum,
                                                           float* __restrict__ ctx, float* __restrict__ align_out,
                                                           int Tcap, const int* __restrict__ step_base, int step_off) {
    __shared__ float ws[TACO_LMAX], red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int step = *step_base + step_off;
    const int n = min((int)lens[b], L);
    float mx = -INFINITY;
    for (int t = tid; t < n; t += 128) {
        const float x = e[(int64_t)b * L + t];
        ws[t] = x;
        mx = fmaxf(mx, x);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    float sm = 0.f;
    for (int t = tid; t < n; t += 128) {
        const float p = expf(ws[t] - mx);
        ws[t] = p;
        sm += p;
    }
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    if ((tid & 63) == 0) red[2 + (tid >> 6)] = sm;
    __syncthreads();
    const float inv = 1.0f / (red[2] + red[3]);
    for (int t = tid; t < n; t += 128) ws[t] *= inv;
    __syncthreads();
    if (blockIdx.y == 0) {
        for (int t = tid; t < L; t += 128) {
            const float w = t < n ? ws[t] : 0.f;
            aw[(int64_t)b * L + t] = w;
            aw_cum[(int64_t)b * L + t] += w;
            if (step < Tcap) align_out[((int64_t)b * Tcap + step) * L + t] = w;
        }
    }
    const int m = blockIdx.y * 128 + tid;
    if (m < M) {
        const float* mr = memory + (int64_t)b * L * M + m;
        float a = 0.f;
        for (int t = 0; t < n; ++t) a = fmaf(ws[t], mr[(int64_t)t * M], a);
        ctx[(int64_t)b * M + m] = a;
    }
}

// mel projection (rows 0..79) + gate (row 80) from [dec_h | ctx], one block per row; stop bookkeeping as
// torchaudio's _Decoder.infer: lengths[~finished] += 1, then finished |= sigmoid(gate) > threshold.
__global__ __launch_bounds__(256) void taco_proj_kernel(const float* __restrict__ dec_h, int D,
                                                        const float* __restrict__ ctx, int M,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int n_mels, float thr, const int* __restrict__ step_base,
                                                        int step_off, int Tcap, int B,
                                                        float* __restrict__ mel_out, float* __restrict__ dec_in,
                                                        int32_t* __restrict__ mel_lens, int32_t* __restrict__ finished) {
    __shared__ float red[1][TACO_BC], part[4][1][TACO_BC];
    const int r = blockIdx.x, K = D + M;
    const int step = *step_base + step_off;
    if (step >= Tcap) return;            // steps past max_step inside the last 8-step graph are no-ops
    for (int b0 = 0; b0 < B; b0 += TACO_BC) {
        float acc[1][TACO_BC];
#pragma unroll
        for (int bb = 0; bb < TACO_BC; ++bb) acc[0][bb] = 0.f;
        block_dots_acc<1>(w + (int64_t)r * K, 0, D, dec_h, D, b0, B, acc);
        block_dots_acc<1>(w + (int64_t)r * K + D, 0, M, ctx, M, b0, B, acc);
        block_dots_reduce<1>(acc, red, part);
        if (threadIdx.x < TACO_BC && b0 + threadIdx.x < B) {
            const int b = b0 + threadIdx.x;
            const float a = red[0][threadIdx.x] + bias[r];
            if (r < n_mels) {
                mel_out[((int64_t)b * n_mels + r) * Tcap + step] = a;
                dec_in[(int64_t)b * n_mels + r] = a;
            } else {
                if (!finished[b]) mel_lens[b] += 1;
                if (sigmoidf_(a) > thr) finished[b] = 1;
            }
        }
        __syncthreads();
    }
}

__global__ void taco_advance_kernel(int* step_base, int n) { *step_base += n; }

// ---- the whole decoder loop as ONE persistent cooperative kernel ------------------------------------------------------
// 256 blocks x 256 threads, one block per CU, resident for every step of the utterance batch (B <= 8, L <= 256):
//   * the 71 MB of LSTMCell weights never leave the chip: block i owns hidden units 4i..4i+3 of BOTH cells, the 16 gate
//     rows of the attention cell sit in its LDS (112-120 KB), the 16 rows of the decoder cell in its registers (160-192
//     VGPRs per thread, one wave per SIMD), the cell states c in the registers of the 32 threads that apply the gates;
//   * the query rows, the folded location filter, this block's slice of processed_memory and of the encoder memory, its
//     rows of the mel / gate projection and of the prenet are resident too (registers / LDS);
//   * the phases of a step are separated by a grid barrier WITHOUT cache maintenance: everything that crosses blocks
//     (h, ctx, alignment weights, partial energies, prenet activations: the `xch` arena) is written and read with
//     device-coherent buffer accesses (sc1), an s_waitcnt orders the data before the block's barrier slot.  A barrier
//     with __threadfence / acquire fences costs 27-33 us on the 8-XCD part (every fence walks the XCD's L2), this one
//     3.6 us (tools/grid_barrier_bench.hip);
//   * six barriers per step:  attention LSTM | query + partial energies (attention dims split over 16 block groups) |
//     softmax + context (32 column groups per utterance) | decoder LSTM | mel / gate projection + prenet layer 1 (folded
//     into the projection: W0 Wp) | prenet layer 2;  the stop test runs on the device.
struct TacoPersist {
    const float *pre1, *att_wih, *att_whh, *att_b, *dec_wih, *dec_whh, *dec_b, *wq, *loc_fold, *v, *pm, *memory, *projx_w, *projx_b;
    const int64_t* lens;
    float* xch;                 // (segment length + 1) regions of `step_floats`, then the barrier slots / error flag / step count
    int64_t tail_o;             // float offset of the tail (slots [256], err, steps)
    // The loop runs in SEGMENTS of at most TACO_SEG steps, one cooperative launch each (steps [s0, s1) of max_step): the arena holds one
    // segment (82 MB instead of 0.5-0.9 GB at the wrapper's decoder_max_step = 3000), region 0 of a later segment is the last region of
    // the one before (copied between the launches, where every XCD's L2 is coherent again -- a ring INSIDE a launch is not possible with
    // L2-cached first reads: a reused slot can sit in a reader XCD's L2 with its old, valid-looking content), and the per-thread state the
    // kernel keeps in registers across steps (cell states, cumulative attention, stop flags, frame counts) goes through `state`.
    int s0, s1;
    float* state;               // [256 blocks][256 threads][8]: c_att, c_dec, cum, fin, mlen
    int step_floats, Lp, PTp;   // region size; padded row / tile strides (multiples of 32 floats = one 128-byte line)
    float *mel_out, *align_out;
    int32_t* mel_lens;
    int B, L, KS, Tcap, max_step, n_mels;
    float thr;
    long long seed;
    int early_stop;             // 0: keep decoding to max_step after every utterance finished (decoder_early_stopping=False)
};

// One REGION of the exchange arena per decoder step.  Everything step s produces goes to region s + 1 (region 0 = the zero
// initial state), so no address is ever read before its final value is in memory: the readers use ordinary L2-cached loads
// (a first version read with sc1 loads that bypass L2 -- 256 blocks x 80-160 KB per phase through the fabric, 7-12 us per
// phase) and only the STORES are device-coherent write-throughs.  Every 128-byte line has exactly one writer block
// (layouts below), so no XCD's L2 ever holds a line that another XCD completes later.
//   att_h, dec_h : [block 256][utterance 8][unit 4]        a reader's float4 column group g of utterance b = line g, +4 b
//   ctx          : [utterance 8][column group 32][32]      (MC = 16 / 20 floats used per line)
//   pre          : [unit group 64][utterance 8][unit 4]
//   h0           : [unit 256][32]                          (8 utterances used)
//   aw, cum      : [utterance 8][Lp]
//   epart        : [attention-dim group 16][tile 16][PTp]
//   fin          : [32]
constexpr int TR_ATT = 0, TR_DEC = 8192, TR_CTX = 16384, TR_PRE = 24576, TR_H0 = 26624, TR_AW = 34816;
__host__ __device__ inline int taco_region_floats(int Lp, int PTp) { return TR_AW + 16 * Lp + 256 * PTp + 32; }

template <int M_>
struct PGeo {
    static constexpr int KA = 256 + M_ + 1024, K4A = KA / 4, NJA = (K4A + 127) / 128;     // attention cell  [pre | ctx | h]
    static constexpr int KD = 1024 + M_ + 1024, K4D = KD / 4, NJD = (K4D + 127) / 128;    // decoder cell    [att_h | ctx | h]
    static constexpr int KP = 1024 + M_, K4P = KP / 4;                                    // projection      [dec_h | ctx]
    static constexpr int MC = M_ / 32, NS = 256 / MC;                                     // context columns per block, t slices
};

// Device-coherent accesses to the exchange arena.  (clang's __builtin_amdgcn_raw_buffer_load_b128 of this ROCm lowers to an
// i32 load whose value is splatted over the vector; the LLVM intrinsics are declared directly, as composable_kernel does.)
typedef int taco_i4 __attribute__((ext_vector_type(4)));
typedef float taco_f4 __attribute__((ext_vector_type(4)));
__device__ taco_f4 taco_buffer_load_f4(taco_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float taco_buffer_load_f1(taco_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void taco_buffer_store_f1(float v, taco_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
#define XST(foff, val) taco_buffer_store_f1((float)(val), rs, (foff) * 4, 0, 16)      /* aux 16 = sc1: write-through, device scope */
#define XLD4(dst, ptr) dst = *reinterpret_cast<const float4*>(ptr);

typedef float taco_f2 __attribute__((ext_vector_type(2)));
// two fp32 FMAs per instruction (v_pk_fma_f32): the accumulator pair holds the even-k and the odd-k partial sums
__device__ __forceinline__ taco_f2 taco_pk_dot4(const taco_f4 w, const taco_f4 x, taco_f2 acc) {
    return __builtin_elementwise_fma(w.zw, x.zw, __builtin_elementwise_fma(w.xy, x.xy, acc));
}
__device__ __forceinline__ float taco_dot4(const float4 w, const float4 x, float acc) {
    return fmaf(w.x, x.x, fmaf(w.y, x.y, fmaf(w.z, x.z, fmaf(w.w, x.w, acc))));
}

// sums the 32 per-thread partials v[i] over the 64 lanes of the wave: lanes 2i and 2i+1 end with the total of partial i
__device__ __forceinline__ float taco_butterfly32(float (&v)[32], int lane) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int m = 32 >> s, half = 16 >> s;
        const bool upper = (lane & m) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            float lo = v[j], hi = v[j + half];
            asm volatile("" : "+v"(lo), "+v"(hi));       // values, not addresses: in this (large) kernel the optimizer otherwise
            const float send = upper ? lo : hi;          // turns the two selects into v[dynamic index] = 32 compare/select pairs
            const float keep = upper ? hi : lo;
            v[j] = keep + __shfl_xor(send, m);
        }
    }
    return v[0] + __shfl_xor(v[0], 1);
}

// grid barrier over the resident blocks: no cache maintenance (see above); false after a spin time-out
__device__ __forceinline__ bool taco_grid_barrier(unsigned* slots, unsigned epoch, int32_t* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool ok = true;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(slots + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        for (;;) {
            bool all = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sidx = 4 * (int)threadIdx.x + i;
                const unsigned val = sidx < (int)gridDim.x ? __hip_atomic_load(slots + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
                all = all && (int)(val - epoch) >= 0;
            }
            if (__all(all)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 20)) { ok = false; break; }
        }
        if (!ok && threadIdx.x == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");
    ok = __syncthreads_and(ok);
    asm volatile("" ::: "memory");
    return ok;
}

template <int M_, bool FLOW>
__global__ __launch_bounds__(256) void taco_decoder_persistent(const TacoPersist p) {
    using G = PGeo<M_>;
    extern __shared__ float4 taco_smem4[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, bid = blockIdx.x;
    const int B = p.B, L = p.L, KS = p.KS, half = (KS - 1) / 2, BL = B * L;
    const int PT = (BL + 15) / 16;                                   // (utterance, token) pairs per energy tile
    float4* sWa = taco_smem4;                                        // [16 rows][K4A] attention cell weights
    float* sf = reinterpret_cast<float*>(taco_smem4 + 16 * G::K4A);
    float* sMem = sf;  sf += L * G::MC;                              // [L][MC] this block's encoder-memory columns
    float* sPm = sf;   sf += PT * 8;                                 // [pair][8 dims] processed memory of this tile
    float* sG = sf;    sf += 8 * 2 * KS;                             // [8 dims][2][KS] folded location filter
    float* sAw = sf;   sf += PT + 2 * half;                          // alignment weights window of this tile
    float* sCum = sf;  sf += PT + 2 * half;
    float* part = sf;  sf += 256;
    float* gates = sf; sf += 128;
    float* sPq = sf;   sf += 64;
    float* sW = sf;    sf += 256;
    float* sRed = sf;  sf += G::NS * G::MC;
    float* red = sf;   sf += 8;
    float* sV = sf;    sf += 8;
    unsigned* slots = reinterpret_cast<unsigned*>(p.xch + p.tail_o);
    int32_t* err = reinterpret_cast<int32_t*>(p.xch + p.tail_o + 256);
    const int Lp = p.Lp, PTp = p.PTp, R_CUM = TR_AW + 8 * Lp, R_EP = TR_AW + 16 * Lp, R_FIN = R_EP + 256 * PTp;

    // ---------------- residency set-up (constant data: ordinary loads)
    const int u0 = bid * 4;                                          // hidden units of both cells
    if constexpr (FLOW) {
        // MFMA operand order (v_mfma_f32_16x16x4_f32, A = 16 gate rows x 4 k): entry (super-step i = 16 k values, lane l) holds
        // W[row l & 15][16 i + 4 (l >> 4) .. + 3], row r16 = (unit r16 >> 2, gate r16 & 3): component m feeds the m-th MFMA of the
        // super-step, whose B operand is component m of the lane's x float4 (same four consecutive k)
        for (int idx = tid; idx < G::KA * 4; idx += 256) {
            const int i = idx >> 6, l = idx & 63, r16 = l & 15, k = 16 * i + 4 * (l >> 4);
            const int64_t row = (int64_t)(r16 & 3) * 1024 + u0 + (r16 >> 2);
            sWa[idx] = *reinterpret_cast<const float4*>(k < 256 + M_ ? p.att_wih + row * (256 + M_) + k : p.att_whh + row * 1024 + (k - 256 - M_));
        }
    } else {
    for (int i = tid; i < 16 * G::K4A; i += 256) {
        const int r = i / G::K4A, g = i - r * G::K4A, k = 4 * g;
        const int64_t row = (int64_t)(r & 3) * 1024 + u0 + (r >> 2);  // row r = (unit r >> 2, gate r & 3)
        sWa[i] = *reinterpret_cast<const float4*>(k < 256 + M_ ? p.att_wih + row * (256 + M_) + k : p.att_whh + row * 1024 + (k - 256 - M_));
    }
    }
    const int hf = wid >> 1, tl = tid & 127;                         // wave pair hf owns gate rows 8 hf .. 8 hf + 7
    // decoder-cell weights: 8 rows x NJD float4 column groups per thread, parked in ACCUMULATION registers (the "a" constraint)
    // and moved to a VGPR where they are used -- left to the register allocator they compete with the x operands in flight
    float wd[8][G::NJD][4];
    constexpr int NSWA = G::KA / 64, NSWD = G::KD / 64;              // MFMA super-steps (16 k) per wave: attention / decoder cell
    float wm[NSWD][4];                                               // (dataflow schedule) decoder-cell weights in MFMA operand order
#define TACO_ACC_PUT(dst, val) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(val))
#define TACO_ACC_GET(dst, src) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(dst) : "a"(src))
    if constexpr (FLOW) {
#pragma unroll
        for (int n = 0; n < NSWD; ++n) {
            const int r16 = lane & 15, k = 16 * (wid + 4 * n) + 4 * (lane >> 4);
            const int64_t row = (int64_t)(r16 & 3) * 1024 + u0 + (r16 >> 2);
            const float4 wv = *reinterpret_cast<const float4*>(k < 1024 + M_ ? p.dec_wih + row * (1024 + M_) + k : p.dec_whh + row * 1024 + (k - 1024 - M_));
            TACO_ACC_PUT(wm[n][0], wv.x);
            TACO_ACC_PUT(wm[n][1], wv.y);
            TACO_ACC_PUT(wm[n][2], wv.z);
            TACO_ACC_PUT(wm[n][3], wv.w);
        }
    } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int R = 8 * hf + r;
        const int64_t row = (int64_t)(R & 3) * 1024 + u0 + (R >> 2);
#pragma unroll
        for (int j = 0; j < G::NJD; ++j) {
            const int g = tl + 128 * j, k = 4 * min(g, G::K4D - 1);
            float4 wv = *reinterpret_cast<const float4*>(k < 1024 + M_ ? p.dec_wih + row * (1024 + M_) + k : p.dec_whh + row * 1024 + (k - 1024 - M_));
            if (g >= G::K4D) wv = make_float4(0.f, 0.f, 0.f, 0.f);
            TACO_ACC_PUT(wd[r][j][0], wv.x);
            TACO_ACC_PUT(wd[r][j][1], wv.y);
            TACO_ACC_PUT(wd[r][j][2], wv.z);
            TACO_ACC_PUT(wd[r][j][3], wv.w);
        }
    }
    }
    const int g16 = bid & 15, tile = bid >> 4;                       // attention dims 8 g16 .. +7, energy tile
    for (int i = tid; i < 8 * 2 * KS; i += 256) sG[i] = p.loc_fold[(int64_t)8 * g16 * 2 * KS + i];
    if (tid < 8) sV[tid] = p.v[8 * g16 + tid];
    const int p0 = tile * PT, p1 = min(p0 + PT, BL);
    for (int i = tid; i < (p1 - p0) * 8; i += 256) sPm[i] = p.pm[(int64_t)(p0 + (i >> 3)) * 128 + 8 * g16 + (i & 7)];
    const int b4 = bid >> 5, cg = bid & 31;                          // softmax / context: utterance, column group
    if (b4 < B)
        for (int i = tid; i < L * G::MC; i += 256) {
            const int t = i / G::MC, c = i - t * G::MC;
            sMem[i] = p.memory[((int64_t)b4 * L + t) * M_ + cg * G::MC + c];
        }
    float4 wpA[2], wpB[2];                                           // projection rows: 81 + bid (prenet layer 1 folded), bid (mel / gate)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g = tid + 256 * j, k = 4 * min(g, G::K4P - 1);
        const float4 a = *reinterpret_cast<const float4*>(p.projx_w + (int64_t)(p.n_mels + 1 + bid) * G::KP + k);
        const float4 b = *reinterpret_cast<const float4*>(p.projx_w + (int64_t)min(bid, p.n_mels) * G::KP + k);
        const bool ok = g < G::K4P;
        wpA[j] = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f);
        wpB[j] = ok ? b : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float biasA = p.projx_b[p.n_mels + 1 + bid], biasB = p.projx_b[min(bid, p.n_mels)];
    float ba[4], bd[4];                                              // gate biases of this thread's unit (threads < 32)
    {
        const int uu = FLOW ? (lane >> 4) : ((tid >> 3) & 3);    // (dataflow: lane (unit l >> 4, utterance l & 15) of wave 0 applies the gates)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) {
            ba[gt] = p.att_b[gt * 1024 + u0 + uu];
            bd[gt] = p.dec_b[gt * 1024 + u0 + uu];
        }
    }
    float c_att = 0.f, c_dec = 0.f, cum = 0.f;
    int fin = 0, mlen = 0;
    unsigned epoch = 0;
    int steps = p.max_step;
    float* stp = p.state + ((int64_t)bid * 256 + tid) * 8;
    if (p.s0 > 0) {                                                  // a later segment: what the previous launch left in its registers
        c_att = stp[0]; c_dec = stp[1]; cum = stp[2];
        fin = __builtin_bit_cast(int, stp[3]); mlen = __builtin_bit_cast(int, stp[4]);
    }
    const int n4 = b4 < B ? min((int)p.lens[b4], L) : 0;
    __syncthreads();

    if constexpr (FLOW) {
    // ================= dataflow schedule: no grid barriers =================
    // The arena is pre-filled with a sentinel bit pattern no arithmetic produces (0xFFFFFFFF); consumers poll the very words
    // they need with device-coherent loads until none is the sentinel, so a hand-off costs one store -> visible -> load
    // chain instead of barrier + fetch.  Only SMALL fresh inputs are ever polled on the critical path: the big operands of the
    // two cells (previous h, context) were validated by this block in an earlier phase and their share of the gate sums is
    // computed ahead, while the fresh input is still on its way ("early" / "late" halves of a cell).
    constexpr unsigned SENT = 0xFFFFFFFFu;
    constexpr int POLL_LIM = 1 << 15;
#ifndef TACO_POLL_SLEEP
#define TACO_POLL_SLEEP 8
#endif
#ifndef TACO_POLL_GAP
#define TACO_POLL_GAP 0                     /* s_sleep units between the first two polls: swept 0...64 on the box, 0 is best (tools/taco_gap_sweep.sh) */
#endif
#define TACO_BACKOFF __builtin_amdgcn_s_sleep(TACO_POLL_SLEEP);   /* between polls: the fabric carries everybody's polls AND the stores they wait for */
    bool bad = false;
#define TACO_OK4(v) (__builtin_bit_cast(unsigned, (v).x) != SENT && __builtin_bit_cast(unsigned, (v).y) != SENT && \
                     __builtin_bit_cast(unsigned, (v).z) != SENT && __builtin_bit_cast(unsigned, (v).w) != SENT)
#define TACO_RSRC(rs_, base_) { const unsigned long long a_ = (unsigned long long)(base_); rs_.x = (int)(unsigned)a_; \
                                rs_.y = (int)(unsigned)(a_ >> 32); rs_.z = p.step_floats * 4; rs_.w = 0x00020000; }
#define TACO_LD4(rs_, foff_) taco_buffer_load_f4(rs_, (foff_) * 4, 0, 16)
#define TACO_LD1(rs_, foff_) taco_buffer_load_f1(rs_, (foff_) * 4, 0, 16)
    float* cred = sf;  sf += 1024;                                   // [wave][lane][4]: the four waves' partial gate tiles
    // The two LSTMCells run on the matrix pipe: one v_mfma_f32_16x16x4_f32 = 16 gate rows (4 units x 4 gates of this block) x
    // 16 columns (8 utterances, duplicated) x 4 k.  A "super-step" is 16 k = four MFMAs fed by ONE float4 of weights and ONE
    // float4 of x per lane (lane = (row or utterance l & 15, k quarter l >> 4)); wave w owns super-steps w, w + 4, ...  The
    // result tile has unit l >> 4's four gates for utterance l & 15 in the lane's four registers: exactly the pointwise update.
    // (Exact fp32 FMA chains, as on the vector pipe; at one wave per SIMD the vector version took 9.7 us per cell half.)
    const int c16 = lane & 15, kq = lane >> 4, bl = min(c16 & 7, B - 1);
    taco_f4 accA0 = {0.f, 0.f, 0.f, 0.f}, accA1 = accA0, accD0 = accA0, accD1 = accA0;
#define TACO_MFMA4(a0_, a1_, w_, x_)                                             \
    a0_ = __builtin_amdgcn_mfma_f32_16x16x4f32((w_).x, (x_).x, a0_, 0, 0, 0);    \
    a1_ = __builtin_amdgcn_mfma_f32_16x16x4f32((w_).y, (x_).y, a1_, 0, 0, 0);    \
    a0_ = __builtin_amdgcn_mfma_f32_16x16x4f32((w_).z, (x_).z, a0_, 0, 0, 0);    \
    a1_ = __builtin_amdgcn_mfma_f32_16x16x4f32((w_).w, (x_).w, a1_, 0, 0, 0);
    // x float4 of the attention cell's super-step i for this lane: [pre | ctx | att_h] column 16 i + 4 kq, utterance bl
#define TACO_ATT_XOFF(i_) ({ const int k_ = 16 * (i_) + 4 * kq, kc_ = k_ - 256, cg_ = kc_ / G::MC;                              \
        k_ < 256 ? TR_PRE + (k_ >> 2) * 32 + bl * 4 : k_ < 256 + M_ ? TR_CTX + (bl * 32 + cg_) * 32 + (kc_ - cg_ * G::MC)      \
                                                     : TR_ATT + ((k_ - 256 - M_) >> 2) * 32 + bl * 4; })
#define TACO_DEC_XOFF(i_) ({ const int k_ = 16 * (i_) + 4 * kq, kc_ = k_ - 1024, cg_ = kc_ / G::MC;                             \
        k_ < 1024 ? TR_ATT + (k_ >> 2) * 32 + bl * 4 : k_ < 1024 + M_ ? TR_CTX + (bl * 32 + cg_) * 32 + (kc_ - cg_ * G::MC)    \
                                                      : TR_DEC + ((k_ - 1024 - M_) >> 2) * 32 + bl * 4; })
    // Round 4: the operand tiles a lane polls for one phase are exactly the ones the NEXT phases' "early halves" of the two cells need (same
    // offsets: att_h of S2 = the decoder cell's and the next attention cell's att_h super-steps, the context of S5 late = the next attention
    // cell's and the projection's context super-steps, dec_h of S6 = the next decoder cell's dec_h super-steps).  Those super-steps now run from
    // the registers right behind the store of the phase that polled them ("tails") instead of re-loading the tiles through L2 two phases later
    // (S5 early + S1 early: 7.9 us of a 32 us step, most of it the re-load round trips).  Carried across phases: accA (next attention cell:
    // att_h part from the S2 tail, + context part from the S5-late tail, + prenet part in S1 late), accD (decoder cell: dec_h part from the
    // previous step's S6 tail, + att_h part from the S2 tail, + context part in S5 late), pj (projection's context part from the S5-late tail).
    // Before the first step of a launch the same parts are computed from region 0 with loads, in the same order (a later segment reproduces the
    // one-launch bits).  First attempt through L2, any sentinel sends the lane to the coherent path.
    constexpr int NEA = NSWA - 4, NCA = M_ / 64;      // early super-steps of the attention cell per wave: n = 0 .. NCA - 1 context, NCA .. NEA - 1 att_h
    constexpr int NCD = M_ / 64;                      // context super-steps of the decoder cell (n = 16 .. 16 + NCD - 1); n >= 16 + NCD: dec_h
#define TACO_ATT_LOAD_PART(rq, pq, N0, N1, INIT)                                                                            \
    {                                                                                                                       \
        constexpr int n0_ = (N0), cnt_ = (N1) - (N0);                                                                       \
        int vp = 0;                                                                                                         \
        asm volatile("" : "+v"(vp));                                                                                        \
        const int wz = wid + vp;                                                                                            \
        taco_f4 xs[cnt_];                                                                                                   \
        for (int spin = 0;; ++spin) {                                                                                       \
            bool okv = true;                                                                                                \
            _Pragma("unroll") for (int n = 0; n < cnt_; ++n) {                                                              \
                const int off_ = TACO_ATT_XOFF(wz + 4 * (n0_ + n + 4));                                                     \
                xs[n] = spin == 0 ? *reinterpret_cast<const taco_f4*>((pq) + off_) : TACO_LD4(rq, off_);                    \
                okv = okv && TACO_OK4(xs[n]);                                                                               \
            }                                                                                                               \
            if (okv) break;                                                                                                 \
            if (spin > POLL_LIM) { bad = true; break; }                                                                     \
            TACO_BACKOFF                                                                                                    \
        }                                                                                                                   \
        if (INIT) {                                                                                                         \
            accA0 = taco_f4{0.f, 0.f, 0.f, 0.f};                                                                            \
            accA1 = accA0;                                                                                                  \
        }                                                                                                                   \
        _Pragma("unroll") for (int n = 0; n < cnt_; ++n) {                                                                  \
            const taco_f4 wa = *reinterpret_cast<const taco_f4*>(&sWa[(wz + 4 * (n0_ + n + 4)) * 64 + lane]);               \
            TACO_MFMA4(accA0, accA1, wa, xs[n])                                                                             \
        }                                                                                                                   \
    }
    taco_i4 rs0;
    TACO_RSRC(rs0, p.xch)
    TACO_ATT_LOAD_PART(rs0, p.xch, NCA, NEA, true)                    // region 0 (zero state, or the previous segment's last region): att_h part ...
    TACO_ATT_LOAD_PART(rs0, p.xch, 0, NCA, false)                     // ... then the context part, as the tails below do it
    {                                                                 // ... and the decoder cell's dec_h part
        int vp = 0;
        asm volatile("" : "+v"(vp));
        const int wz = wid + vp;
        taco_f4 xs[16];
        for (int spin = 0;; ++spin) {
            bool okv = true;
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int off_ = TR_DEC + (4 * (wz + 4 * n) + kq) * 32 + bl * 4;
                xs[n] = spin == 0 ? *reinterpret_cast<const taco_f4*>(p.xch + off_) : TACO_LD4(rs0, off_);
                okv = okv && TACO_OK4(xs[n]);
            }
            if (okv) break;
            if (spin > POLL_LIM) { bad = true; break; }
            TACO_BACKOFF
        }
        accD0 = taco_f4{0.f, 0.f, 0.f, 0.f};
        accD1 = accD0;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            taco_f4 w;
            w.x = wm[16 + NCD + n][0];
            w.y = wm[16 + NCD + n][1];
            w.z = wm[16 + NCD + n][2];
            w.w = wm[16 + NCD + n][3];
            TACO_MFMA4(accD0, accD1, w, xs[n])
        }
    }
    taco_f4 pjc0 = {0.f, 0.f, 0.f, 0.f}, pjc1 = pjc0;                 // projection's context part (S5-late tail -> S6)
    float pre1w[4] = {0.f, 0.f, 0.f, 0.f};                            // prenet layer 2: this thread's column of the block's four rows (blocks 0..63)
    if (bid < 64) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pre1w[r] = p.pre1[(int64_t)(4 * bid + r) * 256 + tid];
    }
    // this thread's word of the alignment window of the block's energy tile (p1 - p0 + 2 half <= 158 words: one per thread)
    const int aw_n = (p1 - p0) + 2 * half;
    const int aw_fp = p0 - half + tid;
    const bool aw_ok = aw_fp >= 0 && aw_fp < BL;
    const int aw_fb = aw_ok ? aw_fp / L : 0, aw_fo = aw_fb * Lp + (aw_ok ? aw_fp - aw_fb * L : 0);
    float aw_pre = 0.f, cum_pre = 0.f;
    if (tid < aw_n) {
        aw_pre = TACO_LD1(rs0, TR_AW + aw_fo);
        cum_pre = TACO_LD1(rs0, R_CUM + aw_fo);
    }
    for (int s = p.s0; s < p.s1; ++s) {
        float* curw = p.xch + (int64_t)(s - p.s0 + 1) * p.step_floats;
        taco_i4 rs, rq;                                              // this step's region (stores, fresh reads); the previous step's
        TACO_RSRC(rs, curw)
        TACO_RSRC(rq, p.xch + (int64_t)(s - p.s0) * p.step_floats)
#ifdef TP_TIMING
        unsigned fst[20];
        int fsi = 0;
#define TF_STAMP() fst[fsi++] = (unsigned)wall_clock64();
#else
#define TF_STAMP()
#endif
        TF_STAMP()   /* 0 top */
        TF_STAMP()   /* 1 stop flags in */
        // ---------------- S1 late: the prenet super-steps of the attention cell, gates, new att_h
        {
            int vp = 0;
            asm volatile("" : "+v"(vp));
            const int wz = wid + vp;
            taco_f4 xp[4];
            int all_fin = 0;                                              // the gate block's stop flags of the previous step ride in the same poll
            taco_f4 xq[4];
            unsigned fv = 0u, fn = 0u;
#pragma unroll
            for (int n = 0; n < 4; ++n) xp[n] = TACO_LD4(rq, TACO_ATT_XOFF(wz + 4 * n));
            if (s > 0) fv = __builtin_bit_cast(unsigned, TACO_LD1(rq, R_FIN + (lane & 7)));
            // (no gap here: the prenet output has usually landed while this block did the early half of the cell)
            for (int spin = 0;; ++spin) {
                bool okv = true;
#pragma unroll
                for (int n = 0; n < 4; ++n) xq[n] = TACO_LD4(rq, TACO_ATT_XOFF(wz + 4 * n));
                if (s > 0) fn = __builtin_bit_cast(unsigned, TACO_LD1(rq, R_FIN + (lane & 7)));
#pragma unroll
                for (int n = 0; n < 4; ++n) okv = okv && TACO_OK4(xp[n]);
                if (s > 0) {
                    const unsigned long long bm = __ballot(fv != 0u || (lane & 7) >= B), vm = __ballot(fv != SENT || (lane & 7) >= B);
                    okv = okv && vm == ~0ull;
                    all_fin = bm == ~0ull;
                }
                if (__all(okv)) break;                                    // wave-uniform exit: the ballots above need every lane in the loop
                if (spin > POLL_LIM) { bad = true; break; }
#pragma unroll
                for (int n = 0; n < 4; ++n) xp[n] = xq[n];
                fv = fn;
            }
            const int every_fin = __syncthreads_or(all_fin);              // (every wave sees the same flags once they are valid)
            if (every_fin && p.early_stop) { steps = s; break; }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const taco_f4 wa = *reinterpret_cast<const taco_f4*>(&sWa[(wz + 4 * n) * 64 + lane]);
                TACO_MFMA4(accA0, accA1, wa, xp[n])
            }
            const taco_f4 tile = accA0 + accA1;
            *reinterpret_cast<taco_f4*>(cred + (wid * 64 + lane) * 4) = tile;
            __syncthreads();
            if (wid == 0 && c16 < 8) {                                    // lane = (unit kq, utterance c16): its four gates
                const taco_f4 g4 = *reinterpret_cast<const taco_f4*>(cred + lane * 4) + *reinterpret_cast<const taco_f4*>(cred + (64 + lane) * 4) +
                                   *reinterpret_cast<const taco_f4*>(cred + (128 + lane) * 4) + *reinterpret_cast<const taco_f4*>(cred + (192 + lane) * 4);
                const float gi = g4.x + ba[0], gf = g4.y + ba[1], gg = g4.z + ba[2], go = g4.w + ba[3];
                c_att = sigmoidf_(gf) * c_att + sigmoidf_(gi) * tanhf(gg);
                if (c16 < B) XST(TR_ATT + bid * 32 + c16 * 4 + kq, sigmoidf_(go) * tanhf(c_att));
            }
            __syncthreads();
        }
        TF_STAMP()   /* 2 S1 late done (pre polled, att_h stored) */
        // ---------------- S2+S3: query rows of this block, partial energies of its tile
        {
            int vp = 0;
            asm volatile("" : "+v"(vp));
            const int tid = (int)threadIdx.x + vp;
            // the previous step's alignment weights first: in memory since its softmax phase, and in this thread's registers since the
            // fetch behind its S5 late (aw_pre / cum_pre: the 1.5-2 us round trip used to sit here, in front of the att_h poll); a sentinel
            // (the fetch overtook the store) sends the thread to the polling path
            if (tid < aw_n) {
                float a = aw_pre, c = cum_pre;
                for (int spin = 0; __builtin_bit_cast(unsigned, a) == SENT || __builtin_bit_cast(unsigned, c) == SENT; ++spin) {
                    if (spin > POLL_LIM) { bad = true; break; }
                    TACO_BACKOFF
                    a = TACO_LD1(rq, TR_AW + aw_fo);
                    c = TACO_LD1(rq, R_CUM + aw_fo);
                }
                sAw[tid] = aw_ok ? a : 0.f;
                sCum[tid] = aw_ok ? c : 0.f;
            }
            __syncthreads();
            // the location term of this tile's energies depends on the previous alignment only: computed now, while att_h is on its way
            float locv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = tid + 256 * u;
                locv[u] = 0.f;
                if (it < (p1 - p0) * 8) {
                    const int pair = it >> 3, d = it & 7, pp = p0 + pair, b = pp / L, t = pp - b * L;
                    const float* gd = sG + d * 2 * KS;
                    float loc = 0.f;
                    if (KS == 31) {                                       // the published filter length, unrolled: its 124 LDS reads go out together
                                                                          // (rolled, every tap waited for its own four reads: 2.4 us at the head of S2)
#pragma unroll
                        for (int k = 0; k < 31; ++k) {
                            const int tt = t + k - 15;
                            const bool ok = tt >= 0 && tt < L;
                            const float av = sAw[pair + k], cv = sCum[pair + k];
                            loc = fmaf(gd[k], ok ? av : 0.f, loc);
                            loc = fmaf(gd[31 + k], ok ? cv : 0.f, loc);
                        }
                    } else {
                        for (int k = 0; k < KS; ++k) {
                            const int tt = t + k - half;
                            const bool ok = tt >= 0 && tt < L;
                            loc = fmaf(gd[k], ok ? sAw[pair + k] : 0.f, loc);
                            loc = fmaf(gd[KS + k], ok ? sCum[pair + k] : 0.f, loc);
                        }
                    }
                    locv[u] = loc;
                }
            }
            // query rows on the matrix pipe: rows 8 g16 .. + 7 (duplicated into the tile's rows 8..15), 64 super-steps over K = 1024,
            // 16 per wave; the weights are constants: fetched before the wait for att_h
            taco_f4 wqm[16], xa[16];
#pragma unroll
            for (int n = 0; n < 16; ++n)
                wqm[n] = *reinterpret_cast<const taco_f4*>(p.wq + (int64_t)(8 * g16 + (c16 & 7)) * 1024 + 16 * (wid + 4 * n + vp) + 4 * kq);
            {                                                             // two polls in flight, half a round trip apart
                taco_f4 xn[16];
#pragma unroll
                for (int n = 0; n < 16; ++n) xa[n] = TACO_LD4(rs, TR_ATT + (4 * (wid + 4 * n + vp) + kq) * 32 + bl * 4);
                __builtin_amdgcn_s_sleep(TACO_POLL_GAP);
                for (int spin = 0;; ++spin) {
                    bool okv = true;
#pragma unroll
                    for (int n = 0; n < 16; ++n) xn[n] = TACO_LD4(rs, TR_ATT + (4 * (wid + 4 * n + vp) + kq) * 32 + bl * 4);
#pragma unroll
                    for (int n = 0; n < 16; ++n) okv = okv && TACO_OK4(xa[n]);
                    if (okv) break;
                    if (spin > POLL_LIM) { bad = true; break; }
#pragma unroll
                    for (int n = 0; n < 16; ++n) xa[n] = xn[n];
                }
            }
            TF_STAMP()   /* 3 att_h in */
            {
                taco_f4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
#pragma unroll
                for (int n = 0; n < 16; ++n) { TACO_MFMA4(q0, q1, wqm[n], xa[n]) }
                *reinterpret_cast<taco_f4*>(cred + (wid * 64 + lane) * 4) = q0 + q1;
            }
            __syncthreads();
            if (tid < 64) {                                              // sPq[dim][utterance]: tile row = dim, lane = (dim >> 2, utterance), reg dim & 3
                const int d = tid >> 3, ub = tid & 7, src = ((d >> 2) * 16 + ub) * 4 + (d & 3);
                sPq[tid] = cred[src] + cred[256 + src] + cred[512 + src] + cred[768 + src];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = tid + 256 * u;
                if (it < (p1 - p0) * 8) {                                 // (whole groups of 8 lanes are in or out together)
                    const int pair = it >> 3, d = it & 7, pp = p0 + pair, b = pp / L;
                    float val = sV[d] * tanhf(sPq[d * 8 + b] + locv[u] + sPm[pair * 8 + d]);
                    val += __shfl_xor(val, 1);
                    val += __shfl_xor(val, 2);
                    val += __shfl_xor(val, 4);
                    if (d == 0) XST(R_EP + (g16 * 16 + tile) * PTp + pair, val);
                }
            }
            TF_STAMP()   /* 4 energies stored */
            // ---- tail: the att_h super-steps of the decoder cell (this step) and of the attention cell (next step), operands = xa
            {
                const int wz = wid + vp;
#pragma unroll
                for (int n = 0; n < 16; ++n) {
                    taco_f4 w;                                            // (a plain use: the compiler feeds the MFMA from the AGPR itself -- behind
                                                                          //  an inline-asm v_accvgpr_read it does not see the VALU -> MFMA hazard)
                    w.x = wm[n][0];
                    w.y = wm[n][1];
                    w.z = wm[n][2];
                    w.w = wm[n][3];
                    TACO_MFMA4(accD0, accD1, w, xa[n])
                }
                accA0 = taco_f4{0.f, 0.f, 0.f, 0.f};
                accA1 = accA0;
#pragma unroll
                for (int n = 0; n < 16; ++n) {
                    const taco_f4 wa = *reinterpret_cast<const taco_f4*>(&sWa[(wz + 4 * (NCA + n + 4)) * 64 + lane]);
                    TACO_MFMA4(accA0, accA1, wa, xa[n])
                }
            }
        }
        TF_STAMP()   /* 5 S2 tail done */
        // ---------------- S4: masked softmax over the tokens of utterance b4, context columns cg MC .. +MC
        if (b4 < B) {
            const int t = tid;
            float e = -INFINITY;
            if (t < n4) {
                const int pp = b4 * L + t, tp = pp / PT;
                const int eo = R_EP + tp * PTp + (pp - tp * PT);
                float ev[16], en[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) ev[g] = TACO_LD1(rs, eo + g * 16 * PTp);
                __builtin_amdgcn_s_sleep(TACO_POLL_GAP);
                for (int spin = 0;; ++spin) {
                    bool okv = true;
#pragma unroll
                    for (int g = 0; g < 16; ++g) en[g] = TACO_LD1(rs, eo + g * 16 * PTp);
#pragma unroll
                    for (int g = 0; g < 16; ++g) okv = okv && __builtin_bit_cast(unsigned, ev[g]) != SENT;
                    if (okv) break;
                    if (spin > POLL_LIM) { bad = true; break; }
#pragma unroll
                    for (int g = 0; g < 16; ++g) ev[g] = en[g];
                }
                e = 0.f;
#pragma unroll
                for (int g = 0; g < 16; ++g) e += ev[g];
            }
            float mx = e;
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) red[wid] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const float pr = t < n4 ? expf(e - mx) : 0.f;
            float sm = pr;
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            if (lane == 0) red[4 + wid] = sm;
            __syncthreads();
            const float w = pr * (1.0f / (red[4] + red[5] + red[6] + red[7]));
            sW[t] = w;
            if (cg == 0 && t < L) {
                cum += w;
                XST(TR_AW + b4 * Lp + t, w);
                XST(R_CUM + b4 * Lp + t, cum);
                if (s < p.Tcap) p.align_out[((int64_t)b4 * p.Tcap + s) * L + t] = w;
            }
            __syncthreads();
            if (tid < G::MC * G::NS) {
                const int col = tid % G::MC, sl = tid / G::MC;
                float a = 0.f;
                for (int t2 = sl; t2 < n4; t2 += G::NS) a = fmaf(sW[t2], sMem[t2 * G::MC + col], a);
                sRed[sl * G::MC + col] = a;
            }
            __syncthreads();
            if (tid < G::MC) {
                float a = 0.f;
#pragma unroll
                for (int sl = 0; sl < G::NS; ++sl) a += sRed[sl * G::MC + tid];
                XST(TR_CTX + (b4 * 32 + cg) * 32 + tid, a);
            }
        }
        __syncthreads();
        TF_STAMP()   /* 6 S4 done (epart polled, ctx stored) */
        // ---------------- S5 late: the context super-steps of the decoder cell, gates, new dec_h
        {
            int vp = 0;
            asm volatile("" : "+v"(vp));
            const int wz = wid + vp;
            constexpr int NC = M_ / 64;
            // two polls in flight, half a round trip apart: the expected wait after the data lands is a quarter round trip, not half
            taco_f4 xc[NC], xn[NC];
#pragma unroll
            for (int e = 0; e < NC; ++e) xc[e] = TACO_LD4(rs, TACO_DEC_XOFF(wz + 4 * (16 + e)));
            // (constants for the tail: the projection's context columns, tile rows alternate (row 81 + bid, row bid) as in S6)
            const float* wrowc = p.projx_w + (int64_t)((c16 & 1) ? min(bid, p.n_mels) : p.n_mels + 1 + bid) * G::KP + 4 * kq;
            taco_f4 wc[NC];
#pragma unroll
            for (int e = 0; e < NC; ++e) wc[e] = *reinterpret_cast<const taco_f4*>(wrowc + 16 * (64 + wz + 4 * e));
            __builtin_amdgcn_s_sleep(TACO_POLL_GAP);
            for (int spin = 0;; ++spin) {
                bool okv = true;
#pragma unroll
                for (int e = 0; e < NC; ++e) xn[e] = TACO_LD4(rs, TACO_DEC_XOFF(wz + 4 * (16 + e)));
#pragma unroll
                for (int e = 0; e < NC; ++e) okv = okv && TACO_OK4(xc[e]);
                if (okv) break;
                if (spin > POLL_LIM) { bad = true; break; }
#pragma unroll
                for (int e = 0; e < NC; ++e) xc[e] = xn[e];
            }
#pragma unroll
            for (int e = 0; e < NC; ++e) {
                taco_f4 w;
                w.x = wm[16 + e][0];
                w.y = wm[16 + e][1];
                w.z = wm[16 + e][2];
                w.w = wm[16 + e][3];
                TACO_MFMA4(accD0, accD1, w, xc[e])
            }
            const taco_f4 tile = accD0 + accD1;
            *reinterpret_cast<taco_f4*>(cred + (wid * 64 + lane) * 4) = tile;
            __syncthreads();
            if (wid == 0 && c16 < 8) {
                const taco_f4 g4 = *reinterpret_cast<const taco_f4*>(cred + lane * 4) + *reinterpret_cast<const taco_f4*>(cred + (64 + lane) * 4) +
                                   *reinterpret_cast<const taco_f4*>(cred + (128 + lane) * 4) + *reinterpret_cast<const taco_f4*>(cred + (192 + lane) * 4);
                const float gi = g4.x + bd[0], gf = g4.y + bd[1], gg = g4.z + bd[2], go = g4.w + bd[3];
                c_dec = sigmoidf_(gf) * c_dec + sigmoidf_(gi) * tanhf(gg);
                if (c16 < B) XST(TR_DEC + bid * 32 + c16 * 4 + kq, sigmoidf_(go) * tanhf(c_dec));
            }
            // ---- tail: the context super-steps of the next attention cell and of the projection, operands = xc
#pragma unroll
            for (int e = 0; e < NC; ++e) {
                const taco_f4 wa = *reinterpret_cast<const taco_f4*>(&sWa[(wz + 4 * (e + 4)) * 64 + lane]);
                TACO_MFMA4(accA0, accA1, wa, xc[e])
            }
            pjc0 = taco_f4{0.f, 0.f, 0.f, 0.f};
            pjc1 = pjc0;
#pragma unroll
            for (int e = 0; e < NC; ++e) { TACO_MFMA4(pjc0, pjc1, wc[e], xc[e]) }
            // ---- and the fetch of this step's alignment window for the NEXT step's location term (stored in S4, before the context this
            // block has just seen complete)
            if ((int)threadIdx.x < aw_n) {
                aw_pre = TACO_LD1(rs, TR_AW + aw_fo);
                cum_pre = TACO_LD1(rs, R_CUM + aw_fo);
            }
        }
        __syncthreads();
        TF_STAMP()   /* 7 S5 late done (ctx polled, dec_h stored, context tails) */
        // ---------------- S6: mel / gate row bid (< 81) and prenet layer-1 unit bid (folded) from [dec_h | ctx], on the matrix pipe:
        // tile rows alternate (row 81 + bid, row bid)
        {
            int vp = 0;
            asm volatile("" : "+v"(vp));
            const int wz = wid + vp;
            const float* wrow = p.projx_w + (int64_t)((c16 & 1) ? min(bid, p.n_mels) : p.n_mels + 1 + bid) * G::KP + 4 * kq;
            taco_f4 pj0 = pjc0, pj1 = pjc1;                              // the context columns: folded in by the S5-late tail
            taco_f4 wp[16], xd[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) wp[n] = *reinterpret_cast<const taco_f4*>(wrow + 16 * (wz + 4 * n));
            {
                taco_f4 xn[16];
#pragma unroll
                for (int n = 0; n < 16; ++n) xd[n] = TACO_LD4(rs, TR_DEC + (4 * (wz + 4 * n) + kq) * 32 + bl * 4);
                __builtin_amdgcn_s_sleep(TACO_POLL_GAP);
                for (int spin = 0;; ++spin) {
                    bool okv = true;
#pragma unroll
                    for (int n = 0; n < 16; ++n) xn[n] = TACO_LD4(rs, TR_DEC + (4 * (wz + 4 * n) + kq) * 32 + bl * 4);
#pragma unroll
                    for (int n = 0; n < 16; ++n) okv = okv && TACO_OK4(xd[n]);
                    if (okv) break;
                    if (spin > POLL_LIM) { bad = true; break; }
#pragma unroll
                    for (int n = 0; n < 16; ++n) xd[n] = xn[n];
                }
            }
#pragma unroll
            for (int n = 0; n < 16; ++n) { TACO_MFMA4(pj0, pj1, wp[n], xd[n]) }
            *reinterpret_cast<taco_f4*>(cred + (wid * 64 + lane) * 4) = pj0 + pj1;
            __syncthreads();
            if (tid < 16) {                                               // gates[0..7]: row 81 + bid (tile row 0), [8..15]: row bid (tile row 1), per utterance
                const int src = (tid & 7) * 4 + (tid >> 3);
                gates[tid] = cred[src] + cred[256 + src] + cred[512 + src] + cred[768 + src];
            }
            __syncthreads();
            if (tid < 8 && tid < B) {
                const int bb = tid;
                float a = fmaxf(gates[bb] + biasA, 0.f);
                if (p.seed >= 0) a *= taco_keep((unsigned)p.seed, 0u, (unsigned)(s + 1), (unsigned)bb, (unsigned)bid);
                XST(TR_H0 + bid * 32 + bb, a);
                const float m = gates[8 + bb] + biasB;
                if (bid < p.n_mels) {
                    p.mel_out[((int64_t)bb * p.n_mels + bid) * p.Tcap + s] = m;
                } else if (bid == p.n_mels) {
                    if (!fin) mlen += 1;
                    if (sigmoidf_(m) > p.thr) fin = 1;
                    p.mel_lens[bb] = mlen;
                    XST(R_FIN + bb, __builtin_bit_cast(float, fin));
                }
            }
            TF_STAMP()   /* 8 S6 done (dec_h polled, h0 stored) */
            // ---- tail: the dec_h super-steps of the NEXT step's decoder cell, operands = xd
            accD0 = taco_f4{0.f, 0.f, 0.f, 0.f};
            accD1 = accD0;
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                taco_f4 w;
                w.x = wm[16 + NCD + n][0];
                w.y = wm[16 + NCD + n][1];
                w.z = wm[16 + NCD + n][2];
                w.w = wm[16 + NCD + n][3];
                TACO_MFMA4(accD0, accD1, w, xd[n])
            }
        }
        __syncthreads();
        TF_STAMP()   /* 9 S6 tail done */
        // ---------------- S7: prenet layer 2, units 4 bid .. 4 bid + 3 on blocks 0..63
        if (bid < 64) {
            int vp = 0;
            asm volatile("" : "+v"(vp));
            const int tid = (int)threadIdx.x + vp;
            taco_f4 xa = TACO_LD4(rs, TR_H0 + tid * 32), xb = TACO_LD4(rs, TR_H0 + tid * 32 + 4);
            __builtin_amdgcn_s_sleep(TACO_POLL_GAP);
            for (int spin = 0;; ++spin) {
                const taco_f4 na = TACO_LD4(rs, TR_H0 + tid * 32), nb = TACO_LD4(rs, TR_H0 + tid * 32 + 4);
                bool okv = true;                                      // (utterances >= B of a line are never written)
                const float xs8[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) okv = okv && (bb >= B || __builtin_bit_cast(unsigned, xs8[bb]) != SENT);
                if (okv) break;
                if (spin > POLL_LIM) { bad = true; break; }
                xa = na;
                xb = nb;
            }
            const float xs8[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            float v[32];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float wr = pre1w[r];                               // (resident since the set-up: a load here sat between the h0 poll and the store of pre)
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) v[r * 8 + bb] = wr * xs8[min(bb, B - 1)];
            }
            const float tot = taco_butterfly32(v, lane);
            if (!(lane & 1)) part[wid * 32 + (lane >> 1)] = tot;
            __syncthreads();
            if (tid < 32) {
                const int r = tid >> 3, bb = tid & 7;
                float a = fmaxf(part[tid] + part[32 + tid] + part[64 + tid] + part[96 + tid], 0.f);
                if (p.seed >= 0) a *= taco_keep((unsigned)p.seed, 1u, (unsigned)(s + 1), (unsigned)bb, (unsigned)(4 * bid + r));
                if (bb < B) XST(TR_PRE + bid * 32 + bb * 4 + r, a);
            }
        }
        TF_STAMP()   /* 10 S7 done */
#ifdef TP_TIMING
        if (threadIdx.x == 0)
            for (int i = 0; i < 11; ++i) curw[TR_H0 + bid * 32 + 8 + i] = __builtin_bit_cast(float, fst[i]);
#endif
        if (__syncthreads_or(bad)) {
            if (tid == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
#undef TACO_ATT_LOAD_PART
#undef TACO_MFMA4
#undef TACO_ATT_XOFF
#undef TACO_DEC_XOFF
#undef TACO_OK4
#undef TACO_RSRC
#undef TACO_LD4
#undef TACO_LD1
    } else {
    for (int s = p.s0; s < p.s1; ++s) {
        const float* prv = p.xch + (int64_t)(s - p.s0) * p.step_floats;      // what step s - 1 left (region 0: zeros, or the previous segment's last region)
        float* curw = p.xch + (int64_t)(s - p.s0 + 1) * p.step_floats;       // what this step produces
        const float* cur = curw;
#ifdef TP_TIMING
        unsigned tstamp[13];
        int tsi = 0;
        const unsigned long long tp_c0 = clock64();
#define TP_STAMP() tstamp[tsi++] = (unsigned)wall_clock64();
#else
#define TP_STAMP()
#endif
        TP_STAMP()
        taco_i4 rs;                                                  // raw buffer over this step's region: the coherent stores
        {
            const unsigned long long a = (unsigned long long)curw;
            rs.x = (int)(unsigned)a; rs.y = (int)(unsigned)(a >> 32); rs.z = p.step_floats * 4; rs.w = 0x00020000;
        }
        // ---------------- S1: attention LSTMCell on [pre | ctx | att_h]
#if !(TP_SKIP & 1)
        {
            int vz = 0;
            asm volatile("" : "+v"(vz));                 // opaque zero: keeps the address arithmetic of this phase inside the
            (void)vz;                                    // step loop (hoisted out of it, it costs more registers than there are)
#pragma unroll 1
            for (int ps = 0; ps < 2; ++ps) {             // utterances 4 ps .. 4 ps + 3: 32 partial sums per thread at a time
                int vp = 0;
                asm volatile("" : "+v"(vp));             // (and keeps the two passes' LDS reads apart: merged, they hold 128 registers)
                const int tlz = tl + vz + vp;
                taco_f4 xs[G::NJA][4];                   // all of this pass's operands in flight at once: one memory round trip
#pragma unroll
                for (int j = 0; j < G::NJA; ++j) {
                    const int g = tlz + 128 * j, gc = min(g, G::K4A - 1), k = 4 * gc;
                    const int kc = k - 256, cgx = kc / G::MC;                                 // (ctx: column group, column)
                    const int xbase = k < 256 ? TR_PRE + gc * 32 : k < 256 + M_ ? TR_CTX + cgx * 32 + (kc - cgx * G::MC) : TR_ATT + (gc - (256 + M_) / 4) * 32;
                    const int xstride = k >= 256 && k < 256 + M_ ? 1024 : 4;
#pragma unroll
                    for (int bq = 0; bq < 4; ++bq) {
                        xs[j][bq] = *reinterpret_cast<const taco_f4*>(prv + xbase + min(4 * ps + bq, B - 1) * xstride);
                        if (g >= G::K4A) xs[j][bq] = taco_f4{0.f, 0.f, 0.f, 0.f};
                    }
                }
                taco_f2 vv[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) vv[i] = taco_f2{0.f, 0.f};
#pragma unroll
                for (int j = 0; j < G::NJA; ++j) {
                    const int gc = min(tlz + 128 * j, G::K4A - 1);
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const taco_f4 wa = *reinterpret_cast<const taco_f4*>(&sWa[(8 * hf + r) * G::K4A + gc]);
#pragma unroll
                        for (int bq = 0; bq < 4; ++bq) vv[r * 4 + bq] = taco_pk_dot4(wa, xs[j][bq], vv[r * 4 + bq]);
                    }
                }
                float v[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = vv[i].x + vv[i].y;
                const float tot = taco_butterfly32(v, lane);
                if (!(lane & 1)) part[wid * 64 + (lane >> 3) * 8 + 4 * ps + ((lane >> 1) & 3)] = tot;   // [row][batch]
            }
            __syncthreads();
            if (tid < 128) gates[tid] = part[(tid >> 6) * 128 + (tid & 63)] + part[(tid >> 6) * 128 + 64 + (tid & 63)];
            __syncthreads();
            if (tid < 32) {
                const int uu = tid >> 3, bb = tid & 7;
                const float* gp = gates + (uu >> 1) * 64 + (uu & 1) * 32 + bb;       // [half][unit (2)][gate (4)][batch (8)]
                const float gi = gp[0] + ba[0], gf = gp[8] + ba[1], gg = gp[16] + ba[2], go = gp[24] + ba[3];
                c_att = sigmoidf_(gf) * c_att + sigmoidf_(gi) * tanhf(gg);
                if (bb < B) XST(TR_ATT + bid * 32 + bb * 4 + uu, sigmoidf_(go) * tanhf(c_att));
            }
        }
#endif
        TP_STAMP()
        if (!taco_grid_barrier(slots, ++epoch, err)) return;
        TP_STAMP()
#if !(TP_SKIP & 2)
        // ---------------- S2+S3: processed query of this block's 8 attention dims, partial energies of its tile
        {
            int vz = 0;
            asm volatile("" : "+v"(vz));
            const int tidz = tid + vz;
            float4 wq[8];                                // this block's 8 query rows (L2-resident constants, re-read per step)
#pragma unroll
            for (int r = 0; r < 8; ++r) wq[r] = *reinterpret_cast<const float4*>(p.wq + (int64_t)(8 * g16 + r) * 1024 + 4 * tidz);
#pragma unroll 1
            for (int ps = 0; ps < 2; ++ps) {
                float v[32];
#pragma unroll
                for (int bq = 0; bq < 4; ++bq) {
                    float4 xv;
                    XLD4(xv, cur + TR_ATT + tidz * 32 + min(4 * ps + bq, B - 1) * 4)
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r * 4 + bq] = fmaf(wq[r].x, xv.x, fmaf(wq[r].y, xv.y, fmaf(wq[r].z, xv.z, wq[r].w * xv.w)));
                }
                const float tot = taco_butterfly32(v, lane);
                if (!(lane & 1)) part[wid * 64 + (lane >> 3) * 8 + 4 * ps + ((lane >> 1) & 3)] = tot;   // [dim][batch]
            }
            const int nw = (p1 - p0) + 2 * half;
            for (int i = tid; i < nw; i += 256) {
                const int fp = p0 - half + i;
                const bool ok = fp >= 0 && fp < BL;
                const int fb = ok ? fp / L : 0, fo = fb * Lp + (ok ? fp - fb * L : 0);
                const float a = prv[TR_AW + fo], c = prv[R_CUM + fo];
                sAw[i] = ok ? a : 0.f;
                sCum[i] = ok ? c : 0.f;
            }
            __syncthreads();
            if (tid < 64) sPq[tid] = part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid];     // [dim][batch]
            __syncthreads();
            for (int it = tid; it < (p1 - p0) * 8; it += 256) {
                const int pair = it >> 3, d = it & 7, pp = p0 + pair, b = pp / L, t = pp - b * L;
                const float* gd = sG + d * 2 * KS;
                float loc = 0.f;
                for (int k = 0; k < KS; ++k) {
                    const int tt = t + k - half;
                    const bool ok = tt >= 0 && tt < L;
                    loc = fmaf(gd[k], ok ? sAw[pair + k] : 0.f, loc);
                    loc = fmaf(gd[KS + k], ok ? sCum[pair + k] : 0.f, loc);
                }
                float val = sV[d] * tanhf(sPq[d * 8 + b] + loc + sPm[pair * 8 + d]);
                val += __shfl_xor(val, 1);
                val += __shfl_xor(val, 2);
                val += __shfl_xor(val, 4);
                if (d == 0) XST(R_EP + (g16 * 16 + tile) * PTp + pair, val);
            }
        }
#endif
        TP_STAMP()
        if (!taco_grid_barrier(slots, ++epoch, err)) return;
        TP_STAMP()
#if !(TP_SKIP & 4)
        // ---------------- S4: masked softmax over the tokens of utterance b4, context columns cg MC .. +MC
        if (b4 < B) {
            const int t = tid;
            float e = -INFINITY;
            if (t < n4) {
                const int pp = b4 * L + t, tp = pp / PT;
                const float* ep = cur + R_EP + tp * PTp + (pp - tp * PT);
                e = 0.f;
#pragma unroll
                for (int g = 0; g < 16; ++g) e += ep[g * 16 * PTp];
            }
            float mx = e;
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) red[wid] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            const float pr = t < n4 ? expf(e - mx) : 0.f;
            float sm = pr;
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            if (lane == 0) red[4 + wid] = sm;
            __syncthreads();
            const float w = pr * (1.0f / (red[4] + red[5] + red[6] + red[7]));
            sW[t] = w;
            if (cg == 0 && t < L) {
                cum += w;
                XST(TR_AW + b4 * Lp + t, w);
                XST(R_CUM + b4 * Lp + t, cum);
                if (s < p.Tcap) p.align_out[((int64_t)b4 * p.Tcap + s) * L + t] = w;
            }
            __syncthreads();
            if (tid < G::MC * G::NS) {
                const int col = tid % G::MC, sl = tid / G::MC;
                float a = 0.f;
                for (int t2 = sl; t2 < n4; t2 += G::NS) a = fmaf(sW[t2], sMem[t2 * G::MC + col], a);
                sRed[sl * G::MC + col] = a;
            }
            __syncthreads();
            if (tid < G::MC) {
                float a = 0.f;
#pragma unroll
                for (int sl = 0; sl < G::NS; ++sl) a += sRed[sl * G::MC + tid];
                XST(TR_CTX + (b4 * 32 + cg) * 32 + tid, a);
            }
        }
#endif
        TP_STAMP()
        if (!taco_grid_barrier(slots, ++epoch, err)) return;
        TP_STAMP()
#if !(TP_SKIP & 8)
        // ---------------- S5: decoder LSTMCell on [att_h | ctx | dec_h], weights in registers
        {
            int vz = 0;
            asm volatile("" : "+v"(vz));
            const int tlz = tl + vz;
#pragma unroll 1
            for (int ps = 0; ps < 2; ++ps) {
                taco_f4 xs[G::NJD][4];
#pragma unroll
                for (int j = 0; j < G::NJD; ++j) {
                    const int g = tlz + 128 * j, k = 4 * min(g, G::K4D - 1);
                    const int gk = k >> 2, kc = k - 1024, cgx = kc / G::MC;
                    const bool rec = k >= 1024 + M_;                                      // the recurrent part comes from the previous step
                    const float* xsrc = rec ? prv : cur;
                    const int xbase = k < 1024 ? TR_ATT + gk * 32 : !rec ? TR_CTX + cgx * 32 + (kc - cgx * G::MC) : TR_DEC + (gk - (1024 + M_) / 4) * 32;
                    const int xstride = k >= 1024 && !rec ? 1024 : 4;
#pragma unroll
                    for (int bq = 0; bq < 4; ++bq) xs[j][bq] = *reinterpret_cast<const taco_f4*>(xsrc + xbase + min(4 * ps + bq, B - 1) * xstride);
                }
                taco_f2 vv[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) vv[i] = taco_f2{0.f, 0.f};
#pragma unroll
                for (int j = 0; j < G::NJD; ++j)
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        taco_f4 w;                                                         // zero past K4D
                        TACO_ACC_GET(w.x, wd[r][j][0]);
                        TACO_ACC_GET(w.y, wd[r][j][1]);
                        TACO_ACC_GET(w.z, wd[r][j][2]);
                        TACO_ACC_GET(w.w, wd[r][j][3]);
#pragma unroll
                        for (int bq = 0; bq < 4; ++bq) vv[r * 4 + bq] = taco_pk_dot4(w, xs[j][bq], vv[r * 4 + bq]);
                    }
                float v[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = vv[i].x + vv[i].y;
                const float tot = taco_butterfly32(v, lane);
                if (!(lane & 1)) part[wid * 64 + (lane >> 3) * 8 + 4 * ps + ((lane >> 1) & 3)] = tot;
            }
            __syncthreads();
            if (tid < 128) gates[tid] = part[(tid >> 6) * 128 + (tid & 63)] + part[(tid >> 6) * 128 + 64 + (tid & 63)];
            __syncthreads();
            if (tid < 32) {
                const int uu = tid >> 3, bb = tid & 7;
                const float* gp = gates + (uu >> 1) * 64 + (uu & 1) * 32 + bb;
                const float gi = gp[0] + bd[0], gf = gp[8] + bd[1], gg = gp[16] + bd[2], go = gp[24] + bd[3];
                c_dec = sigmoidf_(gf) * c_dec + sigmoidf_(gi) * tanhf(gg);
                if (bb < B) XST(TR_DEC + bid * 32 + bb * 4 + uu, sigmoidf_(go) * tanhf(c_dec));
            }
        }
#endif
        TP_STAMP()
        if (!taco_grid_barrier(slots, ++epoch, err)) return;
        TP_STAMP()
#if !(TP_SKIP & 16)
        // ---------------- S6: mel / gate row bid (< 81) and prenet layer-1 unit bid (folded) from [dec_h | ctx]
        {
            float acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            int vz = 0;
            asm volatile("" : "+v"(vz));
            const int tidz = tid + vz;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int g = tidz + 256 * j, k = 4 * min(g, G::K4P - 1);
                const int kc = k - 1024, cgx = kc / G::MC;
                const int xbase = k < 1024 ? TR_DEC + (k >> 2) * 32 : TR_CTX + cgx * 32 + (kc - cgx * G::MC);
                const int xstride = k < 1024 ? 4 : 1024;
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) {
                    float4 xv;
                    XLD4(xv, cur + xbase + min(bb, B - 1) * xstride)
                    acc[bb] = taco_dot4(wpA[j], xv, acc[bb]);                    // weights are zero past K4P
                    acc[8 + bb] = taco_dot4(wpB[j], xv, acc[8 + bb]);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float a = acc[i];
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
                if (lane == 0) part[wid * 16 + i] = a;
            }
            __syncthreads();
            if (tid < 16) gates[tid] = part[tid] + part[16 + tid] + part[32 + tid] + part[48 + tid];
            __syncthreads();
            if (tid < 8 && tid < B) {
                const int bb = tid;
                float a = fmaxf(gates[bb] + biasA, 0.f);
                if (p.seed >= 0) a *= taco_keep((unsigned)p.seed, 0u, (unsigned)(s + 1), (unsigned)bb, (unsigned)bid);
                XST(TR_H0 + bid * 32 + bb, a);
                const float m = gates[8 + bb] + biasB;
                if (bid < p.n_mels) {
                    p.mel_out[((int64_t)bb * p.n_mels + bid) * p.Tcap + s] = m;
                } else if (bid == p.n_mels) {                                    // torchaudio _Decoder.infer stop bookkeeping
                    if (!fin) mlen += 1;
                    if (sigmoidf_(m) > p.thr) fin = 1;
                    p.mel_lens[bb] = mlen;
                    XST(R_FIN + bb, __builtin_bit_cast(float, fin));
                }
            }
        }
#endif
        TP_STAMP()
        if (!taco_grid_barrier(slots, ++epoch, err)) return;
        TP_STAMP()
        {
            int all = 1;
            for (int b = 0; b < B; ++b) all &= __builtin_bit_cast(int, cur[R_FIN + b]) != 0;
            if (all && p.early_stop) { steps = s + 1; break; }
        }
        // ---------------- S7: prenet layer 2, units 4 bid .. 4 bid + 3 on blocks 0..63 (input of the next step's attention cell)
        if (bid < 64) {
            float v[32];
            float4 xa, xb;                                            // h0[utterance 0..7][unit tid]
            XLD4(xa, cur + TR_H0 + tid * 32)
            XLD4(xb, cur + TR_H0 + tid * 32 + 4)
            const float xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float wr = p.pre1[(int64_t)(4 * bid + r) * 256 + tid];
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) v[r * 8 + bb] = wr * xs[bb];
            }
            const float tot = taco_butterfly32(v, lane);
            if (!(lane & 1)) part[wid * 32 + (lane >> 1)] = tot;
            __syncthreads();
            if (tid < 32) {
                const int r = tid >> 3, bb = tid & 7;
                float a = fmaxf(part[tid] + part[32 + tid] + part[64 + tid] + part[96 + tid], 0.f);
                if (p.seed >= 0) a *= taco_keep((unsigned)p.seed, 1u, (unsigned)(s + 1), (unsigned)bb, (unsigned)(4 * bid + r));
                if (bb < B) XST(TR_PRE + bid * 32 + bb * 4 + r, a);
            }
        }
        TP_STAMP()
        if (!taco_grid_barrier(slots, ++epoch, err)) return;
        TP_STAMP()
#ifdef TP_TIMING
        if (tid == 0)
        {
            for (int i = 0; i < 13; ++i) curw[TR_H0 + bid * 32 + 8 + i] = __builtin_bit_cast(float, tstamp[i]);
            curw[TR_H0 + bid * 32 + 21] = __builtin_bit_cast(float, (unsigned)(clock64() - tp_c0));      // shader-clock cycles of this step
        }
#endif
    }
    }   // barrier schedule
    stp[0] = c_att; stp[1] = c_dec; stp[2] = cum;                    // for the next segment (if any)
    stp[3] = __builtin_bit_cast(float, fin); stp[4] = __builtin_bit_cast(float, mlen);
    if (bid == 0 && tid == 0) *reinterpret_cast<int32_t*>(p.xch + p.tail_o + 320) = steps;
}
#undef XST
#undef TACO_ACC_PUT
#undef TACO_ACC_GET
#undef XLD4

// ------------------------------------------------------------------------------------ host

// TTSAMD_TACO_PERSISTENT (read per call): see taco_persistent_mode()
static int taco_persistent_mode() {          // 0 graph path (8-step hipGraph replay), 1 persistent kernel with grid barriers, 2 persistent kernel with
    const char* e = opt_str(OPT_TACO_PERSISTENT);   // dataflow hand-offs and MFMA cells (default where it fits: 47.5 vs 57.5 us per step)
    if (!e || !e[0]) return 2;
    return (e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2;
}
static bool taco_persistent_wanted() { return taco_persistent_mode() != 0; }

struct TWs {
    float *x0, *x1, *xproj, *memory, *pm, *pre, *pq, *energy, *att_h[2], *att_c, *dec_h[2], *dec_c, *aw, *aw_cum, *ctx, *dec_in;
    float *post0, *post1;
    int32_t *finished, *step;
    // persistent decoder: exchange arena of (segment + 1) per-step regions + tail (barrier slots, error flag, step count)
    float* xch;
    int64_t xch_floats, tail_o;
    int step_floats, Lp, PTp, seg;
    float* pstate;              // persistent decoder: per-thread state carried from one segment's launch to the next
};

// steps per launch of the persistent decoder = regions of its exchange arena (TTSAMD_TACO_SEG: tests drive several segments on short runs).
// Read by workspace_bytes and by infer alike; a value that grows between the two calls is caught by infer's own carve of the arena it was
// given ("workspace of N bytes needed"), never written past.
static int taco_segment_steps() {
    const char* e = opt_str(OPT_TACO_SEG);
    const int v = e ? atoi(e) : 512;
    return std::max(8, v);
}

static void tcarve(const Taco2* h, Arena& a, int B, int L, int Tcap, TWs& w) {
    const ttsamd_tacotron2_cfg& c = h->cfg;
    const int E = c.encoder_embedding_dim, M = h->mem_dim;
    w.x0 = a.take<float>((int64_t)B * E * L);
    w.x1 = a.take<float>((int64_t)B * E * L);
    w.xproj = a.take<float>((int64_t)B * 4 * E * L);
    w.memory = a.take<float>((int64_t)B * L * M);
    w.pm = a.take<float>((int64_t)B * L * 128);
    w.pq = a.take<float>((int64_t)B * 128);
    w.energy = a.take<float>((int64_t)B * L);
    w.att_c = a.take<float>((int64_t)B * c.attention_rnn_dim);
    w.dec_c = a.take<float>((int64_t)B * c.decoder_rnn_dim);
    w.dec_in = a.take<float>((int64_t)B * c.n_mels);
    w.pre = a.take<float>((int64_t)B * c.prenet_dim);
    for (int i = 0; i < 2; ++i) w.att_h[i] = a.take<float>((int64_t)B * c.attention_rnn_dim);
    for (int i = 0; i < 2; ++i) w.dec_h[i] = a.take<float>((int64_t)B * c.decoder_rnn_dim);
    w.aw = a.take<float>((int64_t)B * L);
    w.aw_cum = a.take<float>((int64_t)B * L);
    w.ctx = a.take<float>((int64_t)B * M);
    w.finished = a.take<int32_t>(B);
    w.Lp = (int)align_up(L, 32);
    w.PTp = (int)align_up(((int64_t)std::min(B, 8) * L + 15) / 16, 32);
    w.step_floats = taco_region_floats(w.Lp, w.PTp);
    const bool persist_geo = taco_persistent_wanted() && B <= 8 && L <= 256;   // (taco_decoder_persistent's residency plan; else no arena)
    w.seg = std::min(Tcap, taco_segment_steps());
    w.tail_o = persist_geo ? (int64_t)(w.seg + 1) * w.step_floats : 0;
    w.xch_floats = w.tail_o + 384;
    w.xch = a.take<float>(w.xch_floats);
    w.pstate = a.take<float>(persist_geo ? (int64_t)256 * 256 * 8 : 0);
    w.post0 = a.take<float>((int64_t)B * c.postnet_embedding_dim * Tcap);
    w.post1 = a.take<float>((int64_t)B * c.postnet_embedding_dim * Tcap);
    w.step = a.take<int32_t>(1);
}

int64_t tacotron2_workspace_bytes(const Taco2* h, int32_t B, int32_t L, int32_t Tcap) {
    Arena a(nullptr, 0);
    TWs w;
    tcarve(h, a, B, L, Tcap, w);
    return a.off;
}

static int32_t tconv(const Taco2* h, const TConv& c, const float* x, int64_t x_bs, int x_cs, float* y, int64_t y_bs,
                     int y_cs, const float* res, int B, int T, int act, hipStream_t s) {
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.x_bs = x_bs; p.x_cs = x_cs;
    p.w = h->dev + c.w_off; p.bias = h->dev + c.b_off;
    p.w_bf16 = h->dev16 + c.w16_off; p.precision = default_precision();
    p.y = y; p.y_bs = y_bs; p.y_cs = y_cs; p.y_ts = 1;
    p.res = res; p.r_bs = y_bs; p.r_cs = y_cs;
    p.len_in_mul = 1; p.len_out_mul = 1; p.Lin = T; p.Nout = T;
    p.Cin = c.cin; p.Cout = c.cout; p.CoutP = cout_padded(c.cout); p.K = c.k;
    p.dil = 1; p.pad = c.k / 2; p.n_phase = 1; p.in_slope = 1.f; p.relu_out = act; p.mode = 0; p.div = 1.f; p.batch = B;
    prof_begin(s, 2.0 * c.cout * c.cin * c.k);
    const int32_t rc = launch_conv(p, s);
    prof_end(s);
    return rc;
}

static int32_t launch_lstm(const float* x1, int n1, const float* x2, int n2, const float* h_in, float* c,
                           const float* wih, const float* whh, const float* bias, float* h_out, int B, int H,
                           hipStream_t s) {
    const int K4 = (n1 + n2 + H) / 4, NW = (K4 + 63) / 64;
    TTS_REQUIRE(n1 % 4 == 0 && n2 % 4 == 0 && H % 4 == 0 && H % 2 == 0 && NW <= 16,
                "tacotron2: LSTM geometry %d+%d+%d not supported", n1, n2, H);
#define TTS_LSTM_CASE(N)                                                                                         \
    if (NW <= N) {                                                                                               \
        hipLaunchKernelGGL(taco_lstm_kernel<N>, dim3(H / 2), dim3(N * 64), 0, s, x1, n1, x2, n2, h_in, c, wih, whh, \
                           bias, h_out, B, H);                                                                   \
        return 0;                                                                                                \
    }
    TTS_LSTM_CASE(8)
    TTS_LSTM_CASE(11)
    TTS_LSTM_CASE(16)
#undef TTS_LSTM_CASE
    return 0;
}

int32_t tacotron2_infer(const Taco2* h, const int64_t* tokens, const int64_t* lengths, const int64_t* speaker_ids,
                        int32_t B, int32_t L, int32_t max_step, int64_t dropout_seed, float* mel_post,
                        int32_t* mel_lens, float* alignments, float* mel_raw, int32_t* n_steps_out, void* ws,
                        int64_t ws_bytes, hipStream_t s) {
    TTS_REQUIRE(h && tokens && lengths && mel_post && mel_lens && alignments && mel_raw && n_steps_out,
                "tacotron2_infer: null argument");
    TTS_REQUIRE(B >= 1 && L >= 1 && L <= TACO_LMAX && max_step >= 1, "tacotron2_infer: bad batch/length/max_step");
    const ttsamd_tacotron2_cfg& c = h->cfg;
    TTS_REQUIRE(c.num_speakers <= 1 || speaker_ids, "tacotron2_infer: speaker_ids is null");
    const double t_enter = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    Arena a(ws, ws_bytes);
    TWs w;
    const int Tcap = max_step;
    tcarve(h, a, B, L, Tcap, w);
    if (!ws || !a.ok) {
        set_error("tacotron2_infer: workspace of %lld bytes needed, %lld given", (long long)a.off, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    const int E = c.encoder_embedding_dim, M = h->mem_dim, A = c.attention_rnn_dim, D = c.decoder_rnn_dim, P = c.prenet_dim;
    const float* W = h->dev;
    // ---- encoder
    hipLaunchKernelGGL(taco_embed_kernel, dim3((L + 63) / 64, B), dim3(256), 0, s, tokens, W + h->emb, h->cfg.n_symbol, E, L, w.x0);
    TTS_CHECK_HIP(hipGetLastError());
    float *cur = w.x0, *nxt = w.x1;
    for (const TConv& cv : h->enc_convs) {
        TTS_TRY(tconv(h, cv, cur, (int64_t)E * L, L, nxt, (int64_t)E * L, L, nullptr, B, L, 1, s));
        std::swap(cur, nxt);
    }
    TTS_TRY(tconv(h, h->enc_xproj, cur, (int64_t)E * L, L, w.xproj, (int64_t)4 * E * L, L, nullptr, B, L, 0, s));
    TTS_CHECK_HIP(hipMemsetAsync(w.memory, 0, (size_t)B * L * M * sizeof(float), s));
    hipLaunchKernelGGL(taco_bilstm_kernel, dim3(B, 2), dim3(256), 0, s, w.xproj, W + h->enc_whhT[0], W + h->enc_whhT[1],
                       lengths, L, M, w.memory);
    if (c.num_speakers > 1)
        hipLaunchKernelGGL(taco_spk_kernel, dim3(L, B), dim3(128), 0, s, w.memory, W + h->spk, speaker_ids, c.num_speakers, L, M,
                           E, c.speaker_embedding_dim);
    hipLaunchKernelGGL(taco_pm_kernel, dim3(L, B), dim3(128), 0, s, w.memory, W + h->wmT, L, M, w.pm);
    TTS_CHECK_HIP(hipGetLastError());
    // ---- decoder state
    TTS_CHECK_HIP(hipMemsetAsync(w.att_h[0], 0, (size_t)B * A * sizeof(float), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.dec_h[0], 0, (size_t)B * D * sizeof(float), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.aw, 0, (size_t)B * L * sizeof(float), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.aw_cum, 0, (size_t)B * L * sizeof(float), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.ctx, 0, (size_t)B * M * sizeof(float), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.finished, 0, (size_t)B * sizeof(int32_t), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.att_c, 0, (size_t)B * A * sizeof(float), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.dec_c, 0, (size_t)B * D * sizeof(float), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.dec_in, 0, (size_t)B * c.n_mels * sizeof(float), s));
    TTS_CHECK_HIP(hipMemsetAsync(mel_lens, 0, (size_t)B * sizeof(int32_t), s));
    TTS_CHECK_HIP(hipMemsetAsync(w.step, 0, sizeof(int32_t), s));
    static const bool dbg = exp_env("TTSAMD_TACO_DEBUG") != nullptr;
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    int steps = 0;
    bool done = false;
    // ---- the persistent decoder (one cooperative launch for the whole loop) when the geometry fits its residency plan
    {
        const bool want = taco_persistent_wanted();
        const int KS = c.attention_location_kernel_size, half = (KS - 1) / 2, PT = (B * L + 15) / 16;
        const int MC = M / 32, NS = MC ? 256 / MC : 0, K4A = (P + M + A) / 4;
        const size_t lds = (size_t)16 * K4A * 16 +
                           sizeof(float) * ((size_t)L * MC + (size_t)PT * 8 + 16 * KS + 2 * (PT + 2 * half) + 256 + 128 + 64 + 256 + NS * MC + 16 + 1024);
        int dev_id = 0, n_cu = 0, coop = 0, lds_max = 0;
        (void)hipGetDevice(&dev_id);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_id);
        (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev_id);
        (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev_id);
        const char* pe = opt_str(OPT_TACO_PERSISTENT);
        const bool explicit_req = pe && (pe[0] == '1' || pe[0] == '2');          // an explicit request must run the persistent kernel or fail
        bool fits = want && B <= 8 && L <= 256 && (M == 512 || M == 640) && A == 1024 && D == 1024 && P == 256 && KS % 2 == 1 &&
                    (B * L + 15) / 16 + KS - 1 <= 256 &&      /* the alignment window of an energy tile: one word per thread */
                    lds <= (size_t)std::max(lds_max, 64 * 1024) && n_cu >= 256 && coop && w.tail_o > 0;
        if (explicit_req && !fits) {
            set_error("tacotron2_infer: TTSAMD_TACO_PERSISTENT=%c but the persistent decoder does not fit (B=%d <= 8, L=%d <= 256, memory dim %d in "
                      "{512, 640}, %zu B of LDS <= %d, %d CUs >= 256, cooperative launch %d)", pe[0], B, L, M, lds, lds_max, n_cu, coop);
            return TTSAMD_EINVAL;
        }
        if (fits && !explicit_req) {
            std::lock_guard<std::mutex> lock(h->mu);
            if (h->persist_skip > 0) { --h->persist_skip; fits = false; }        // a recent time-out: do not pay the spin again
        }
        if (fits) {
            TacoPersist q;
            q.pre1 = W + h->pre1;
            q.att_wih = W + h->att_wih; q.att_whh = W + h->att_whh; q.att_b = W + h->att_b;
            q.dec_wih = W + h->dec_wih; q.dec_whh = W + h->dec_whh; q.dec_b = W + h->dec_b;
            q.wq = W + h->wq; q.loc_fold = W + h->loc_fold; q.v = W + h->v;
            q.pm = w.pm; q.memory = w.memory;
            q.projx_w = W + h->projx_w; q.projx_b = W + h->projx_b;
            q.lens = lengths;
            q.xch = w.xch; q.tail_o = w.tail_o; q.step_floats = w.step_floats; q.Lp = w.Lp; q.PTp = w.PTp;
            q.mel_out = mel_raw; q.align_out = alignments; q.mel_lens = mel_lens;
            q.B = B; q.L = L; q.KS = KS; q.Tcap = Tcap; q.max_step = max_step; q.n_mels = c.n_mels;
            q.thr = c.gate_threshold; q.seed = (long long)dropout_seed; q.early_stop = c.decoder_early_stopping != 0;
            const bool flow = taco_persistent_mode() == 2;
            const void* fn = flow ? (M == 512 ? (const void*)taco_decoder_persistent<512, true> : (const void*)taco_decoder_persistent<640, true>)
                                  : (M == 512 ? (const void*)taco_decoder_persistent<512, false> : (const void*)taco_decoder_persistent<640, false>);
            std::lock_guard<std::mutex> lock(h->mu);
            q.state = w.pstate;
            void* args[] = {&q};
            const double t0 = now_us();
            int32_t tail[2] = {0, 0};                                   // err_o and steps_o are 64 floats apart: two copies
            // one cooperative launch per segment of at most w.seg steps (TacoPersist: why); a whole bench-size decode (448 steps) is one
            for (int s0 = 0; s0 < max_step; s0 += w.seg) {
                q.s0 = s0;
                q.s1 = std::min(max_step, s0 + w.seg);
                if (s0 > 0)    // region 0 of this segment = what the last step of the previous one produced
                    TTS_CHECK_HIP(hipMemcpyAsync(w.xch, w.xch + (int64_t)w.seg * w.step_floats, (size_t)w.step_floats * sizeof(float),
                                                 hipMemcpyDeviceToDevice, s));
                if (flow)    // every word a consumer may poll starts as the sentinel 0xFFFFFFFF (regions 1 ... of this segment)
                    TTS_CHECK_HIP(hipMemsetAsync(w.xch + w.step_floats, 0xFF, (size_t)(w.tail_o - w.step_floats) * sizeof(float), s));
                if (s0 == 0) TTS_CHECK_HIP(hipMemsetAsync(w.xch, 0, (size_t)w.step_floats * sizeof(float), s));    // region 0: the zero initial state
                TTS_CHECK_HIP(hipMemsetAsync(w.xch + w.tail_o, 0, 384 * sizeof(float), s));                        // barrier slots, error flag, step count
                // LDS opt-in or cooperative launch rejected (another partitioning / device, hipErrorCooperativeLaunchTooLarge): the
                // graph path below still works, so only an explicit request turns this into an error
                hipError_t le = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (le == hipSuccess) le = hipLaunchCooperativeKernel(fn, dim3(256), dim3(256), args, (unsigned)lds, s);
                if (le != hipSuccess) {
                    (void)hipGetLastError();
                    if (explicit_req) {
                        set_error("tacotron2_infer: launching the persistent decoder failed: %s", hipGetErrorString(le));
                        return TTSAMD_EHIP;
                    }
                    // hipErrorCooperativeLaunchTooLarge can be transient (another stream held CUs at that moment): same bounded back-off
                    // as a hand-off time-out; anything else (LDS opt-in rejected, no cooperative launch on this partitioning) is permanent
                    if (le == hipErrorCooperativeLaunchTooLarge) {
                        h->persist_backoff = std::min(256, std::max(8, 2 * h->persist_backoff));
                        h->persist_skip = h->persist_backoff;
                    } else {
                        h->persist_skip = 1 << 30;
                    }
                    ++h->persist_launch_failures;
                    fprintf(stderr, "ttsamd: tacotron2 persistent decoder could not be launched (%s), using the graph path%s\n", hipGetErrorString(le),
                            le == hipErrorCooperativeLaunchTooLarge ? " (retrying after a back-off)" : " for the life of this handle");
                    tail[0] = -1;
                    break;
                }
                TTS_CHECK_HIP(hipMemcpyAsync(&tail[0], w.xch + w.tail_o + 256, sizeof(int32_t), hipMemcpyDeviceToHost, s));
                TTS_CHECK_HIP(hipMemcpyAsync(&tail[1], w.xch + w.tail_o + 320, sizeof(int32_t), hipMemcpyDeviceToHost, s));
                TTS_CHECK_HIP(hipStreamSynchronize(s));
                // a time-out, or every utterance's gate fired inside this segment -- also on its LAST step (the kernel then reports exactly s1;
                // an unfinished segment reports max_step): no extra segment (a region copy, an 82 MB memset and a cooperative launch for nothing)
                if (tail[0] != 0 || (int64_t)tail[1] <= (int64_t)q.s1) break;
            }
            if (dbg) fprintf(stderr, "[taco] persistent decoder: %.0f us for %d steps (%zu B of LDS per block)\n", now_us() - t0, (int)tail[1], lds);
            if (tail[0] != 0) {
                // a block waited > 80 ms for another one: the 256 blocks were not all resident (CUs held by other streams).  The
                // graph path below recomputes the whole loop from the same state; only an explicit request is an error.
                if (explicit_req) {
                    set_error("tacotron2_infer: the persistent decoder timed out waiting for another block (are 256 CUs free for it?)");
                    return TTSAMD_EHIP;
                }
                if (tail[0] > 0) {
                    h->persist_backoff = std::min(256, std::max(8, 2 * h->persist_backoff));
                    h->persist_skip = h->persist_backoff;
                    fprintf(stderr, "ttsamd: tacotron2 persistent decoder timed out, using the graph path (and for the next %d calls)\n",
                            h->persist_skip);
                }
                TTS_CHECK_HIP(hipMemsetAsync(mel_lens, 0, (size_t)B * sizeof(int32_t), s));     // (the kernel had started counting)
            } else {
                steps = tail[1];
                done = true;
                h->persist_backoff = 0;
            }
            if (const char* dump = done ? exp_env("TTSAMD_TACO_DUMP") : nullptr) {   // debugging aid: the region the last step produced
                std::vector<float> hx((size_t)w.step_floats);
                TTS_CHECK_HIP(hipMemcpy(hx.data(), w.xch + (int64_t)(steps - (steps > 0 ? (steps - 1) / w.seg * w.seg : 0)) * w.step_floats, hx.size() * sizeof(float), hipMemcpyDeviceToHost));
                if (FILE* f = fopen(dump, "wb")) {
                    const int32_t hdr[8] = {B, L, M, steps, w.Lp, w.PTp, w.step_floats, 0};
                    fwrite(hdr, sizeof(hdr), 1, f);
                    fwrite(hx.data(), sizeof(float), hx.size(), f);
                    fclose(f);
                }
            }
        }
    }
    if (!done) {
    // The loop is launch-bound when issued kernel by kernel (7 dependent launches of 4-14 us per step), so
    // 8 steps are captured once into a hipGraph (the step index lives in device memory and is advanced by the
    // graph's last node) and replayed.  The stop flags come back through pinned memory two replays late, so
    // the host never drains the queue; frames computed past the stop are sliced off (the reference breaks
    // as soon as every utterance has finished, tacotron2_ms.py:327 -> torchaudio _Decoder.infer).
    constexpr int GSTEPS = 8, RING = 4;
    std::lock_guard<std::mutex> lock(h->mu);
    if (h->pinned_cap < B) {
        if (h->pinned) TTS_CHECK_HIP(hipHostFree(h->pinned));
        h->pinned = nullptr;
        TTS_CHECK_HIP(hipHostMalloc((void**)&h->pinned, (size_t)RING * B * sizeof(int32_t), hipHostMallocDefault));
        h->pinned_cap = B;
    }
    for (auto& e : h->ev)
        if (!e) TTS_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!h->ev_in) TTS_CHECK_HIP(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    if (!h->loop_stream) TTS_CHECK_HIP(hipStreamCreateWithFlags(&h->loop_stream, hipStreamNonBlocking));
    hipStream_t caller = s;
    TTS_CHECK_HIP(hipEventRecord(h->ev_in, caller));          // encoder + state resets were queued on the caller's stream
    s = h->loop_stream;
    TTS_CHECK_HIP(hipStreamWaitEvent(s, h->ev_in, 0));
    if (dbg) (void)hipStreamSynchronize(s);
    const double t_cap0 = now_us();
    if (dbg) fprintf(stderr, "[taco] encoder + state reset (synced): %.0f us\n", t_cap0 - t_enter);
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    TTS_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    int32_t cap_rc = 0;
    for (int i = 0; i < GSTEPS && cap_rc == 0; ++i) {
        const int pi = i & 1, po = pi ^ 1;
        hipLaunchKernelGGL(taco_prenet_kernel, dim3(B, 8), dim3(1024), 0, s, w.dec_in, W + h->pre0, W + h->pre1, c.n_mels,
                           (long long)dropout_seed, w.step, i, w.pre);
        cap_rc = launch_lstm(w.pre, P, w.ctx, M, w.att_h[pi], w.att_c, W + h->att_wih, W + h->att_whh, W + h->att_b,
                             w.att_h[po], B, A, s);
        hipLaunchKernelGGL(taco_query_kernel, dim3(128 / 8), dim3(256), 0, s, w.att_h[po], A, W + h->wq, w.pq, B);
        hipLaunchKernelGGL(taco_energy_kernel, dim3((L + 3) / 4, B), dim3(256), 0, s, w.pq, w.pm, W + h->loc_conv,
                           c.attention_location_kernel_size, W + h->loc_denseT, W + h->v, w.aw, w.aw_cum, L, w.energy);
        hipLaunchKernelGGL(taco_context_kernel, dim3(B, (M + 127) / 128), dim3(128), 0, s, w.energy, w.memory, M, lengths,
                           L, w.aw, w.aw_cum, w.ctx, alignments, Tcap, w.step, i);
        if (cap_rc == 0)
            cap_rc = launch_lstm(w.att_h[po], A, w.ctx, M, w.dec_h[pi], w.dec_c, W + h->dec_wih, W + h->dec_whh,
                                 W + h->dec_b, w.dec_h[po], B, D, s);
        hipLaunchKernelGGL(taco_proj_kernel, dim3(c.n_mels + 1), dim3(256), 0, s, w.dec_h[po], D, w.ctx, M, W + h->proj_w,
                           W + h->proj_b, c.n_mels, c.gate_threshold, w.step, i, Tcap, B, mel_raw, w.dec_in, mel_lens,
                           w.finished);
    }
    hipLaunchKernelGGL(taco_advance_kernel, dim3(1), dim3(1), 0, s, w.step, GSTEPS);
    hipError_t cap_err = hipStreamEndCapture(s, &graph);
    if (cap_err == hipSuccess && cap_rc == 0) cap_err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (cap_rc != 0) return cap_rc;
    if (cap_err != hipSuccess) {
        set_error("tacotron2_infer: capturing the decoder step graph failed: %s", hipGetErrorString(cap_err));
        return TTSAMD_EHIP;
    }
    const double t_loop0 = now_us();
    hipError_t run_err = hipSuccess;
    for (int g = 0; steps < max_step && run_err == hipSuccess; ++g) {
        run_err = hipGraphLaunch(exec, s);
        steps = std::min(steps + GSTEPS, (int)max_step);
        const int slot = g % RING;
        if (run_err == hipSuccess)
            run_err = hipMemcpyAsync(h->pinned + (size_t)slot * B, w.finished, B * sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (run_err == hipSuccess) run_err = hipEventRecord(h->ev[slot], s);
        if (g >= 2 && run_err == hipSuccess) {               // flags as of replay g-2: keeps two replays queued
            const int old = (g - 2) % RING;
            run_err = hipEventSynchronize(h->ev[old]);
            bool all = true;
            for (int b = 0; b < B; ++b) all = all && h->pinned[(size_t)old * B + b];
            if (all && c.decoder_early_stopping != 0) break;
        }
    }
    const double t_loop1 = now_us();
    if (run_err == hipSuccess) run_err = hipStreamSynchronize(s);   // host-blocking: the caller's stream may go on
    if (dbg) fprintf(stderr, "[taco] encoder+reset %.0f us (synced), capture+instantiate %.0f us, loop host %.0f us, loop total %.0f us for %d steps\n", 0.0, t_loop0 - t_cap0, t_loop1 - t_loop0, now_us() - t_loop0, steps);
    (void)hipGraphExecDestroy(exec);
    s = caller;
    if (run_err != hipSuccess) {
        set_error("tacotron2_infer: decoder loop failed: %s", hipGetErrorString(run_err));
        return TTSAMD_EHIP;
    }
    }   // graph path
    // number of frames the reference would have produced: it stops at the step the last utterance finishes
    std::vector<int32_t> lens_h(B);
    TTS_CHECK_HIP(hipMemcpyAsync(lens_h.data(), mel_lens, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    TTS_CHECK_HIP(hipStreamSynchronize(s));
    int T = 0;
    for (int b = 0; b < B; ++b) T = std::max(T, (int)lens_h[b]);
    T = c.decoder_early_stopping != 0 ? std::min(T, steps) : steps;   // without early stopping the reference returns every step it ran
    *n_steps_out = T;
    // ---- postnet on [B][80][T] (row stride Tcap), residual fused into the last conv
    const float* px = mel_raw;
    int64_t px_bs = (int64_t)c.n_mels * Tcap;
    int px_cs = Tcap;
    float* bufs[2] = {w.post0, w.post1};
    const int np_ = (int)h->post_convs.size();
    for (int i = 0; i < np_; ++i) {
        const TConv& cv = h->post_convs[i];
        const bool last = i == np_ - 1;
        float* y = last ? mel_post : bufs[i & 1];
        const int64_t y_bs = last ? (int64_t)c.n_mels * Tcap : (int64_t)cv.cout * T;
        const int y_cs = last ? Tcap : T;
        TTS_TRY(tconv(h, cv, px, px_bs, px_cs, y, y_bs, y_cs, last ? mel_raw : nullptr, B, T, last ? 0 : 3, s));
        px = y; px_bs = y_bs; px_cs = y_cs;
    }
    if (dbg) {
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "[taco] whole call (synced): %.0f us, T=%d\n", now_us() - t_enter, T);
    }
    return 0;
}

}  // namespace ttsamd
