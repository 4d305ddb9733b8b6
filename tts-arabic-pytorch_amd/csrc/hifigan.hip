// HiFi-GAN V1 generator on the MFMA conv engine: handle creation (weight-norm fold, weight
// re-layout, upload) and the batched ragged forward.
// Replaces vocoder.load_hifigan (vocoder/__init__.py:3-20) and Generator.forward
// (vocoder/hifigan/models.py:111-127) incl. ResBlock1.forward (:46-53).
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "bfo3.hpp"

namespace ttsamd {

// default routing of the second-generation fused pair (fused2_choice below), set from same-box A/B runs of the bench workload
constexpr unsigned kFused2Mask = 0x00F;       // C = 32: k = 3 / 7 / 11; C = 64: k = 3.  Both convs of these pairs run on Winograd F(2,3) inside the launch
                                              // (resblock_pair2<..., WM = 2>).  Round 6: every other pair runs un-fused on the F(4,3) kernel (conv_wino4.hip,
                                              // 64-row blocks): same-box A/B of the bench step (tools/ab_env.sh) 56.72 (07f, k = 7 / 11 only on F(4,3)) / 56.15
                                              // (06f) / 55.00 (05f) / 54.81 ms (04f); then with k = 3 on F(4,3) too: 54.68 (04f) / 55.11 (047: C = 64 k = 3
                                              // un-fused) / 54.59 (007) / 54.10 ms (00f: C = 128 k = 3 un-fused).  Round 5 on F(2,3): C = 64 k = 11 fused 61.93 vs
                                              // 62.45; C = 128 k = 3 fused 63.90 vs 64.06
constexpr unsigned kFused2MaskN1 = 0x000;     // 128-column blocks (the direct-arithmetic kernels only): none by default
constexpr int64_t kFused2SmallColumns = 2 * 256 * 252;   // batch x positions under which a stage counts as a small problem

struct ConvW {
    int64_t w_off = 0, b_off = 0;  // float offsets into the device weight blob
    int64_t w16_off = 0, w_n = 0;  // bf16 planes (hi, lo) in the uint16 blob; packed element count
    int64_t wo_off = -1;           // bf16 octet engine (bfo.hpp): [Cin/16][K][2][CoutP][8] in the same uint16 blob (-1: not packed)
    int64_t wo3_off = -1;          // its split-bf16 mode (bfo3.hpp): [Cin/16][K][2][CoutP][hi 8 | lo 8]
    int64_t ww4_off = -1;          // ... and as Winograd F(4,3) groups (conv_wino4.hip; the un-fused convs: Cout >= 128)
    int64_t ww_off = -1;           // k = 3 / 7 / 11: Winograd F(2,3) (sub-)filters + single taps as an NG-tap conv in the fp32 blob (conv_wino2.hip; -1: none)
    int cin = 0, cout = 0, k = 0;
};

struct HifiGan {
    ttsamd_hifigan_cfg cfg;
    float* dev = nullptr;  // one blob with every packed weight and bias
    uint16_t* dev16 = nullptr;  // bf16 hi/lo planes of the same packed weights
    int64_t dev_n = 0, dev16_n = 0;  // element counts of the two blobs (ttsamd_dp_broadcast_weights)
    ConvW conv_pre, conv_post;
    std::vector<ConvW> ups;
    std::vector<ConvW> c1, c2;  // [stage*n_kernels + j][m]
    int hop = 1;
    int64_t max_cl = 0;  // max over stages of C * (L / T)
    bool bfo_ok = false; // every layer fits the bf16 octet engine (config 3 path, hifigan_forward_bfo)
    // the three ResBlocks of a stage run on three streams (created and first dispatched in hifigan_create; guarded by mu)
    mutable std::mutex mu;
    mutable hipStream_t side[2] = {nullptr, nullptr};
    mutable hipEvent_t ev_fork = nullptr, ev_done[3] = {nullptr, nullptr, nullptr};
};

// The ResBlocks of one stage are independent until their sum.  A launch is a few rounds of equally long blocks, so up to
// a third of the chip idles in its tail (stage 1 at batch 32: 3.5 rounds of 200 us blocks; batch 1..8: 1-3 rounds
// everywhere); the three branches are therefore issued on three streams and fill each other's tails.  Only the last conv
// of each branch, which accumulates into the stage output, is chained j = 0 -> 1 -> 2 by events, so the sum is formed in
// the reference's order (bit-identical results, test_hifigan_branch_streams_bit_identical).  Measured gain: +6 % at batch 1,
// +7 % at 8, +4-5 % at 12-16, +2 % at 32.  Profiling (ttsamd_profile_*) brackets each fork..join section with ONE event
// pair on the caller's stream -- wall time of the section, never a sum of overlapping launches.
// TTSAMD_HIFIGAN_STREAMS=0/1 forces either schedule.
static bool use_branch_streams(const HifiGan* h, int32_t B, int32_t T) {
    if (h->cfg.n_kernels != 3) return false;
    const char* env = opt_str(OPT_HIFIGAN_STREAMS);   // read per call: the tests flip it
    if (env) return env[0] == '1';
    // bf16 octet engine: the launches are power-bound at batch 32 and latency-bound below; the fork / join events cost more than the
    // overlap returns under ~8 k frames (batch 1: 2.09 -> 1.92 ms, batch 8: 4.35 -> 4.26 on one stream; batch 32: 10.02 -> 9.88 ms with
    // three under the two-stream pipeline)
    // (split bf16, round 6: three streams win at every size -- batch 1 1.18 vs 1.25 ms, 2: 1.87 / 1.94, 4: 3.37 / 3.53, 8: 6.15 / 6.40,
    // tools/b1_parts.py -- so the rule below is the plain-bf16 engine's only)
    if (default_precision() == 1 && h->bfo_ok) {
        const char* bfo_env = opt_str(OPT_BFO);
        if (!(bfo_env && bfo_env[0] == '0')) return (int64_t)B * T >= 8192;
    }
    return true;
}

// Which ResBlock pairs of the fp32 engine go out as ONE launch of the second-generation fused kernel (resblock_fused2.hip), and with
// which block width: 0 = not this pair, 2 = 256-column blocks, 1 = 128-column blocks.  Bit 3 * ci + ki of the masks, ci = 0 / 1 / 2
// for C = 32 / 64 / 128, ki = 0 / 1 / 2 for k = 3 / 7 / 11.  TTSAMD_FUSED2=0 turns the kernel off, TTSAMD_FUSED2_MASK / _MASK_N1 (hex)
// replace the defaults (A/B runs and the forced-kernel parity tests); all read per call.
// The routing switches are read ONCE per forward call (snapshot below) and handed down: a forward used to make several hundred
// getenv calls, each a window for a concurrent setenv from another host thread (os.environ writes in Python do exactly that).
struct Fused2Switches {
    bool on = true, mask_forced = false, small_on = false;
    bool wino_b = true, wino_a = true;   // phase B / phase A + B of the C = 32 / 64 pairs as Winograd F(2,3) (TTSAMD_FUSED2_WB=0: both direct, =1: phase B only)
    unsigned mask = kFused2Mask, mask_n1 = kFused2MaskN1;
};
static bool parse_hex_mask(const char* txt, unsigned& out) {
    if (!txt || !*txt) return false;
    char* end = nullptr;
    const unsigned long v = std::strtoul(txt, &end, 16);
    if (end == txt || *end != '\0' || v > 0x1ff) return false;
    out = (unsigned)v;
    return true;
}
static int32_t read_fused2_switches(Fused2Switches& sw) {
    const char* e = opt_str(OPT_FUSED2);
    sw.on = !(e && e[0] == '0');
    if (const char* m = opt_str(OPT_FUSED2_MASK)) {
        TTS_REQUIRE(parse_hex_mask(m, sw.mask), "TTSAMD_FUSED2_MASK='%s' is not a hex mask of 9 bits (bit 3 * ci + ki; 1ff = every pair)", m);
        sw.mask_forced = true;
    }
    if (const char* m = opt_str(OPT_FUSED2_MASK_N1))
        TTS_REQUIRE(parse_hex_mask(m, sw.mask_n1), "TTSAMD_FUSED2_MASK_N1='%s' is not a hex mask of 9 bits", m);
    const char* wb = opt_str(OPT_FUSED2_WB);
    sw.wino_b = !(wb && wb[0] == '0');
    sw.wino_a = sw.wino_b && !(wb && wb[0] == '1');
    const char* se = exp_env("TTSAMD_FUSED2_SMALL");
    sw.small_on = se && se[0] == '1';
    return 0;
}

static int fused2_choice(const Fused2Switches& sw, int32_t channels, int32_t k, int32_t dil, int32_t L, const float* x, const float* y,
                         int64_t columns) {
    if (!sw.on) return 0;
    const int ci = channels == 32 ? 0 : (channels == 64 ? 1 : (channels == 128 ? 2 : -1));
    const int ki = k == 3 ? 0 : (k == 7 ? 1 : (k == 11 ? 2 : -1));
    if (ci < 0 || ki < 0) return 0;
    unsigned mask = sw.mask, mask_n1 = sw.mask_n1;
    const unsigned bit = 1u << (3 * ci + ki);
    // small problems (batch 1 ... 4: under two rounds of 256-column blocks).  Measured (tools/f2_small.sh): every pair as ONE launch of
    // 128-column blocks -- half the launches of the un-fused engine -- is SLOWER there (batch 1: 5.27 vs 5.04 ms per step, batch 4: 13.35
    // vs 12.88; batch 8 equal): the un-fused engine's 64 x 64 tiles with split K put 3-4x more blocks on the chip.  So small problems
    // keep the un-fused engine (a forced TTSAMD_FUSED2_MASK overrides the rule: parity tests of the fused kernels on small inputs).
    const bool small = columns < (int64_t)kFused2SmallColumns && !sw.mask_forced;
    if (small) {
        if (!sw.small_on) return 0;
        mask = 0x1ff; mask_n1 = 0x1ff;
    }
    if (!(mask & bit)) return 0;
    int ntw = (mask_n1 & bit) ? 1 : 2;
    if (!fused_pair2_supported(channels, k, dil, L, x, y, ntw)) {
        ntw = 3 - ntw;
        if (!fused_pair2_supported(channels, k, dil, L, x, y, ntw)) return 0;
    }
    return ntw;
}

using TensorMap = std::map<std::string, const ttsamd_tensor*>;

static int64_t numel(const ttsamd_tensor* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}

// Folded weight of `<base>` : either `<base>.weight`, or g*v/||v|| from the weight-norm pair
// (norm over all dims but 0, as torch._weight_norm(v, g, 0); vocoder/hifigan/models.py:129-136).
static int32_t folded_weight(const TensorMap& tm, const std::string& base, int ndim_expected,
                             std::vector<float>& out, int64_t shape[3]) {
    const ttsamd_tensor *w = nullptr, *g = nullptr, *v = nullptr;
    auto it = tm.find(base + ".weight");
    if (it != tm.end()) w = it->second;
    if (!w) {
        auto ig = tm.find(base + ".parametrizations.weight.original0");
        auto iv = tm.find(base + ".parametrizations.weight.original1");
        if (ig == tm.end() || iv == tm.end()) {
            ig = tm.find(base + ".weight_g");
            iv = tm.find(base + ".weight_v");
        }
        TTS_REQUIRE(ig != tm.end() && iv != tm.end(), "hifigan: no weight for layer '%s'", base.c_str());
        g = ig->second;
        v = iv->second;
    }
    const ttsamd_tensor* src = w ? w : v;
    TTS_REQUIRE(src->ndim == ndim_expected, "hifigan: '%s' has ndim %d, expected %d", base.c_str(), src->ndim,
                ndim_expected);
    for (int i = 0; i < 3; ++i) shape[i] = src->shape[i];
    const int64_t n = numel(src);
    out.resize(n);
    if (w) {
        std::memcpy(out.data(), w->data, n * sizeof(float));
        return 0;
    }
    const int64_t d0 = v->shape[0], inner = n / d0;
    TTS_REQUIRE(numel(g) == d0, "hifigan: '%s' weight_g has %lld elements, expected %lld", base.c_str(),
                (long long)numel(g), (long long)d0);
    for (int64_t i = 0; i < d0; ++i) {
        double ss = 0.0;
        const float* vr = v->data + i * inner;
        for (int64_t j = 0; j < inner; ++j) ss += (double)vr[j] * vr[j];
        const float scale = g->data[i] / (float)std::sqrt(ss);
        for (int64_t j = 0; j < inner; ++j) out[i * inner + j] = vr[j] * scale;
    }
    return 0;
}

static int32_t get_bias(const TensorMap& tm, const std::string& base, int n, std::vector<float>& blob,
                        int64_t& off) {
    auto it = tm.find(base + ".bias");
    TTS_REQUIRE(it != tm.end() && numel(it->second) == n, "hifigan: missing/mis-sized '%s.bias'", base.c_str());
    off = (int64_t)blob.size();
    blob.insert(blob.end(), it->second->data, it->second->data + n);
    blob.resize(align_up((int64_t)blob.size(), 64));
    return 0;
}

static void add_bf16(std::vector<float>& blob, std::vector<uint16_t>& blob16, ConvW& cw, int64_t n) {
    cw.w_n = n;
    cw.w16_off = (int64_t)blob16.size();
    blob16.resize(blob16.size() + 2 * n);
    split_packed_bf16(blob.data() + cw.w_off, n, blob16.data() + cw.w16_off);
}

static int32_t add_conv(const TensorMap& tm, const std::string& base, int cin, int cout, int k,
                        std::vector<float>& blob, std::vector<uint16_t>& blob16, ConvW& cw) {
    std::vector<float> w;
    int64_t shp[3];
    TTS_TRY(folded_weight(tm, base, 3, w, shp));
    TTS_REQUIRE(shp[0] == cout && shp[1] == cin && shp[2] == k, "hifigan: '%s' has shape [%lld,%lld,%lld], expected [%d,%d,%d]",
                base.c_str(), (long long)shp[0], (long long)shp[1], (long long)shp[2], cout, cin, k);
    cw.cin = cin; cw.cout = cout; cw.k = k;
    cw.w_off = (int64_t)blob.size();
    blob.resize(blob.size() + (size_t)cin * k * cout_padded(cout));
    pack_conv_weight(w.data(), cout, cin, k, blob.data() + cw.w_off);
    add_bf16(blob, blob16, cw, (int64_t)cin * k * cout_padded(cout));
    if ((k == 3 || k == 7 || k == 11) && cin % 8 == 0 && cout % 32 == 0) {      // (cout = 32: the fused pairs' Winograd phase B)
        blob.resize(align_up((int64_t)blob.size(), 64));
        cw.ww_off = (int64_t)blob.size();
        blob.resize(blob.size() + (size_t)cin * wino2_groups(k) * cout_padded(cout));
        pack_wino2_weight(w.data(), cout, cin, k, blob.data() + cw.ww_off);
        if (cout >= 64) {             // the un-fused ResBlock convs (stages with 64 / 128 / 256 channels): F(4,3) decomposition
            blob.resize(align_up((int64_t)blob.size(), 64));
            cw.ww4_off = (int64_t)blob.size();
            blob.resize(blob.size() + (size_t)cin * wino4_groups(k) * cout_padded(cout));
            pack_wino4_weight(w.data(), cout, cin, k, blob.data() + cw.ww4_off);
        }
    }
    if (cin % 8 == 0 && cout % 32 == 0) {
        blob16.resize(align_up((int64_t)blob16.size(), 64));
        cw.wo_off = (int64_t)blob16.size();
        blob16.resize(blob16.size() + (size_t)bfo_packed_conv_elems(cout, cin, k));
        bfo_pack_conv_weight(w.data(), cout, cin, k, blob16.data() + cw.wo_off);
        blob16.resize(align_up((int64_t)blob16.size(), 64));
        cw.wo3_off = (int64_t)blob16.size();
        blob16.resize(blob16.size() + (size_t)bfo3_packed_conv_elems(cout, cin, k));
        bfo3_pack_conv_weight(w.data(), cout, cin, k, blob16.data() + cw.wo3_off);
    }
    blob.resize(align_up((int64_t)blob.size(), 64));
    return get_bias(tm, base, cout, blob, cw.b_off);
}

void hifigan_destroy(HifiGan* h);

__global__ void hifigan_touch_kernel() {}   // first dispatch of a branch stream (hifigan_create)

int32_t hifigan_create(const ttsamd_tensor* weights, int32_t n, const ttsamd_hifigan_cfg* cfg, HifiGan** out) {
    TTS_REQUIRE(weights && cfg && out, "hifigan_create: null argument");
    TTS_REQUIRE(cfg->n_ups >= 1 && cfg->n_ups <= 8 && cfg->n_kernels >= 1 && cfg->n_kernels <= 8 &&
                cfg->n_dilations >= 1 && cfg->n_dilations <= 8, "hifigan_create: bad config counts");
    TensorMap tm;
    for (int i = 0; i < n; ++i) tm[weights[i].name] = &weights[i];
    auto* h = new HifiGan();
    h->cfg = *cfg;
    std::vector<float> blob;
    std::vector<uint16_t> blob16;
    const int c0 = cfg->upsample_initial_channel;
    int32_t rc = add_conv(tm, "conv_pre", cfg->num_mels, c0, 7, blob, blob16, h->conv_pre);
    int ch = c0, mul = 1;
    h->max_cl = c0;
    for (int i = 0; rc == 0 && i < cfg->n_ups; ++i) {
        const int u = cfg->upsample_rates[i], kt = cfg->upsample_kernel_sizes[i];
        const int cin = ch, cout = ch / 2;
        if (kt != 2 * u || (kt - u) % 2 != 0) {
            set_error("hifigan: upsample kernel %d / rate %d: only kernel = 2*rate is built", kt, u);
            rc = TTSAMD_EINVAL;
            break;
        }
        std::vector<float> w;
        int64_t shp[3];
        rc = folded_weight(tm, "ups." + std::to_string(i), 3, w, shp);
        if (rc) break;
        if (shp[0] != cin || shp[1] != cout || shp[2] != kt) {
            set_error("hifigan: ups.%d has shape [%lld,%lld,%lld], expected [%d,%d,%d]", i, (long long)shp[0],
                      (long long)shp[1], (long long)shp[2], cin, cout, kt);
            rc = TTSAMD_EINVAL;
            break;
        }
        ConvW cw;
        cw.cin = cin; cw.cout = cout; cw.k = kt;
        cw.w_off = (int64_t)blob.size();
        blob.resize(blob.size() + (size_t)u * cin * 2 * cout_padded(cout));
        pack_convt_weight(w.data(), cin, cout, kt, u, (kt - u) / 2, blob.data() + cw.w_off);
        add_bf16(blob, blob16, cw, (int64_t)u * cin * 2 * cout_padded(cout));
        if (cin % 16 == 0 && cout % 32 == 0 && (u == 8 || u == 2)) {
            blob16.resize(align_up((int64_t)blob16.size(), 64));
            cw.wo_off = (int64_t)blob16.size();
            blob16.resize(blob16.size() + (size_t)bfo_packed_convt_elems(cin, cout, u));
            bfo_pack_convt_weight(w.data(), cin, cout, u, blob16.data() + cw.wo_off);
            blob16.resize(align_up((int64_t)blob16.size(), 64));
            cw.wo3_off = (int64_t)blob16.size();
            blob16.resize(blob16.size() + (size_t)bfo3_packed_convt_elems(cin, cout, u));
            bfo3_pack_convt_weight(w.data(), cin, cout, u, blob16.data() + cw.wo3_off);
        }
        blob.resize(align_up((int64_t)blob.size(), 64));
        rc = get_bias(tm, "ups." + std::to_string(i), cout, blob, cw.b_off);
        if (rc) break;
        h->ups.push_back(cw);
        ch = cout;
        mul *= u;
        h->max_cl = std::max<int64_t>(h->max_cl, (int64_t)ch * mul);
        for (int j = 0; rc == 0 && j < cfg->n_kernels; ++j) {
            const int r = i * cfg->n_kernels + j, kk = cfg->resblock_kernel_sizes[j];
            for (int m = 0; rc == 0 && m < cfg->n_dilations; ++m) {
                ConvW a, b;
                rc = add_conv(tm, "resblocks." + std::to_string(r) + ".convs1." + std::to_string(m), ch, ch, kk, blob, blob16, a);
                if (rc) break;
                rc = add_conv(tm, "resblocks." + std::to_string(r) + ".convs2." + std::to_string(m), ch, ch, kk, blob, blob16, b);
                h->c1.push_back(a);
                h->c2.push_back(b);
            }
        }
    }
    h->hop = mul;
    if (rc == 0) {
        // the bf16 octet engine covers this generator if every layer was packed for it
        bool ok = h->conv_pre.wo_off >= 0 && ch == 32 && cfg->num_mels % 8 == 0;
        for (const ConvW& cw : h->ups) ok = ok && cw.wo_off >= 0;
        for (size_t i = 0; i < h->c1.size(); ++i) ok = ok && h->c1[i].wo_off >= 0 && h->c2[i].wo_off >= 0;
        for (int j = 0; j < cfg->n_kernels; ++j) {
            const int kk = cfg->resblock_kernel_sizes[j];
            ok = ok && (kk == 3 || kk == 7 || kk == 11);
            for (int m = 0; m < cfg->n_dilations; ++m) ok = ok && cfg->resblock_dilations[j][m] >= 1 && cfg->resblock_dilations[j][m] <= BFO_DMAX;
        }
        h->bfo_ok = ok;
    }
    if (rc == 0) {
        // conv_post: [1][C][7] -> plain [C][7]
        std::vector<float> w;
        int64_t shp[3];
        rc = folded_weight(tm, "conv_post", 3, w, shp);
        if (rc == 0 && (shp[0] != 1 || shp[1] != ch || shp[2] != 7)) {
            set_error("hifigan: conv_post has shape [%lld,%lld,%lld], expected [1,%d,7]", (long long)shp[0],
                      (long long)shp[1], (long long)shp[2], ch);
            rc = TTSAMD_EINVAL;
        }
        if (rc == 0) {
            h->conv_post.cin = ch; h->conv_post.cout = 1; h->conv_post.k = 7;
            h->conv_post.w_off = (int64_t)blob.size();
            blob.insert(blob.end(), w.begin(), w.end());
            blob.resize(align_up((int64_t)blob.size(), 64));
            rc = get_bias(tm, "conv_post", 1, blob, h->conv_post.b_off);
        }
    }
    if (rc == 0) {
        h->dev_n = (int64_t)blob.size();
        h->dev16_n = (int64_t)blob16.size();
        hipError_t e = hipMalloc((void**)&h->dev, blob.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(h->dev, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&h->dev16, blob16.size() * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMemcpy(h->dev16, blob16.data(), blob16.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("hifigan_create: weight upload failed: %s", hipGetErrorString(e));
            rc = TTSAMD_EHIP;
        }
        // The two ResBlock branch streams exist and have dispatched once BEFORE the caller's own side streams do (the runtime
        // binds a stream to a hardware queue at its first dispatch).  Created lazily inside the first forward they were bound
        // after the two ttsamd.pipeline streams whenever that schedule ran first, and every later one-stream call paid 1.2-1.9 ms
        // for its fork / join events (batch 1: 3.3 instead of 2.1 ms, tools/pipe_debug2.py).
        if (e == hipSuccess && cfg->n_kernels == 3) {
            for (auto& st : h->side)
                if (e == hipSuccess) {
                    // highest stream priority: under the two-stream pipeline (ttsamd.pipeline: its vocoder stream is created the same way)
                    // the acoustic model of the next batch fills what the vocoder leaves instead of competing with it -- same-box A/B,
                    // three alternations: fp32 52.86 -> 52.59 ms per step, split bf16 25.40 -> 25.25
                    int lo = 0, hi = 0;
                    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi);
                }
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
            for (auto& ev : h->ev_done)
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            for (auto& st : h->side)
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(hifigan_touch_kernel, dim3(1), dim3(64), 0, st);
                    e = hipGetLastError();
                }
            for (auto& st : h->side)
                if (e == hipSuccess && st) e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                set_error("hifigan_create: branch streams: %s", hipGetErrorString(e));
                rc = TTSAMD_EHIP;
            }
        }
    }
    if (rc != 0) {
        hifigan_destroy(h);            // blobs, branch streams and events alike
        return rc;
    }
    *out = h;
    return 0;
}

void hifigan_destroy(HifiGan* h) {
    if (!h) return;
    if (h->dev) (void)hipFree(h->dev);
    if (h->dev16) (void)hipFree(h->dev16);
    for (hipStream_t st : h->side) if (st) (void)hipStreamDestroy(st);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->ev_done) if (e) (void)hipEventDestroy(e);
    delete h;
}

int64_t hifigan_workspace_bytes(const HifiGan* h, int32_t B, int32_t T) {
    Arena a(nullptr, 0);
    const int nb = use_branch_streams(h, B, T) ? 3 : 1;
    for (int i = 0; i < 2 + 2 * nb; ++i) a.take<float>((int64_t)B * h->max_cl * T);
    for (int i = 0; i < nb; ++i) a.take<float>(kSplitKFloats);
    a.take<uint16_t>((int64_t)B * align_up(h->cfg.num_mels, 8) * T * 2);   // octet engine: the mel in its layout (x3: 4 bytes per element)
    return a.off;
}

int32_t hifigan_forward(const HifiGan* h, const float* mel, const int64_t* lens, int32_t B, int32_t T, float* wave,
                        void* ws, int64_t ws_bytes, hipStream_t s) {
    TTS_REQUIRE(h && mel && wave && B >= 1 && T >= 1, "hifigan_forward: bad argument");
    Arena a(ws, ws_bytes);
    const bool multi = use_branch_streams(h, B, T);
    const int nb = multi ? 3 : 1;
    float* cur = a.take<float>((int64_t)B * h->max_cl * T);
    float* ups_out = a.take<float>((int64_t)B * h->max_cl * T);
    float *Tbs[3], *Rs[3], *splitks[3];
    for (int i = 0; i < nb; ++i) {
        Tbs[i] = a.take<float>((int64_t)B * h->max_cl * T);
        Rs[i] = a.take<float>((int64_t)B * h->max_cl * T);
    }
    for (int i = 0; i < nb; ++i) splitks[i] = a.take<float>(kSplitKFloats);
    for (int i = nb; i < 3; ++i) { Tbs[i] = Tbs[0]; Rs[i] = Rs[0]; splitks[i] = splitks[0]; }
    uint16_t* mel_o = a.take<uint16_t>((int64_t)B * align_up(h->cfg.num_mels, 8) * T * 2);
    if (!ws || !a.ok) {
        set_error("hifigan_forward: workspace of %lld bytes needed, %lld given", (long long)a.off, (long long)ws_bytes);
        return TTSAMD_ENOMEM;
    }
    const ttsamd_hifigan_cfg& cfg = h->cfg;
    std::unique_lock<std::mutex> lk(h->mu, std::defer_lock);
    hipStream_t bs[3] = {s, s, s};
    if (multi) {
        lk.lock();
        TTS_REQUIRE(h->ev_fork && h->side[0] && h->side[1], "hifigan_forward: the branch streams were not created (hifigan_create)");
        bs[1] = h->side[0];
        bs[2] = h->side[1];
    }

    // any failure below must not leave the side streams running on a workspace the caller is about to recycle
    struct Join {
        bool on; hipStream_t a, b;
        ~Join() { if (on) { (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); } }
    } join{false, bs[1], bs[2]};
    const auto fail = [&](int32_t rc) { join.on = multi && rc != 0; return rc; };

    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.batch = B;
    p.lens_in = lens; p.lens_out = lens;
    p.n_phase = 1; p.div = 1.f;
    p.splitk_ws = splitks[0]; p.splitk_floats = kSplitKFloats;     // batch 1: stage-1 launches have < 256 tiles
    // plain bf16 mode: the c1 -> c2 intermediate of a ResBlock only feeds c2, so it crosses HBM as packed bf16
    // (ConvParams::y_packed / x_packed; same rounding point as the fp32 buffer + round-on-load, bit-identical)
    const char* pk_env = opt_str(OPT_BF16_PACKED_T);
    const bool pack_t = default_precision() == 1 && !(pk_env && pk_env[0] == '0');
    Fused2Switches f2sw;
    TTS_TRY(read_fused2_switches(f2sw));
    const char* fz_env = opt_str(OPT_FUSED_PAIR);
    const bool fused_ok = default_precision() == 0 && !(fz_env && fz_env[0] == '0');
    const char* ct_env = opt_str(OPT_CONVT);
    const bool convt_ok = !(ct_env && ct_env[0] == '0');   // all-phases-per-wave transposed conv (convt_mfma.hip)
    bool in_section = false;   // inside a multi-stream fork..join section (profiling brackets the section)
    int pack_io = 0;   // bit 0: x is packed, bit 1: write y packed (set around the c1 / c2 launches below)
    auto conv = [&](const ConvW& cw, const float* x, hipStream_t st, float* y, const float* res, int L, int mul,
                    int dil, float slope, int mode, float div) -> int32_t {
        p.x = x; p.x_bs = (int64_t)cw.cin * L; p.x_cs = L;
        p.w = h->dev + cw.w_off; p.bias = h->dev + cw.b_off;
        p.w_bf16 = h->dev16 + cw.w16_off; p.precision = default_precision();
        p.w_wino = cw.ww_off >= 0 ? h->dev + cw.ww_off : nullptr;
        p.w_wino4 = cw.ww4_off >= 0 ? h->dev + cw.ww4_off : nullptr;
        p.y = y; p.y_bs = (int64_t)cw.cout * L; p.y_cs = L; p.y_ts = 1;
        p.res = res; p.r_bs = (int64_t)cw.cout * L; p.r_cs = L;
        p.len_in_mul = mul; p.len_out_mul = mul; p.Lin = L; p.Nout = L;
        p.Cin = cw.cin; p.Cout = cw.cout; p.CoutP = cout_padded(cw.cout); p.K = cw.k;
        p.dil = dil; p.pad = (cw.k * dil - dil) / 2;
        p.n_phase = 1; p.phase_p = 0;
        p.in_slope = slope; p.relu_out = 0; p.mode = mode; p.div = div;
        p.x_packed = (pack_io & 1) ? 1 : 0; p.y_packed = (pack_io & 2) ? 1 : 0; p.pack_slope = 0.1f;
        // in the three-stream schedule the whole fork..join section is bracketed once (below): per-launch spans overlap
        if (in_section) prof_add(2.0 * cw.cout * cw.cin * cw.k * mul);
        else prof_begin(st, 2.0 * cw.cout * cw.cin * cw.k * mul);
        const int32_t rc = launch_conv(p, st);
        if (!in_section) prof_end(st);
        return rc;
    };

#define HG_TRY(expr) do { int32_t rc_ = (expr); if (rc_ != 0) return fail(rc_); } while (0)
#define HG_CHECK_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return fail(TTSAMD_EHIP); } } while (0)
    // ---- config 3: plain bf16 (precision 1) and split bf16 (precision 2) run on the octet engine (bfo.hpp / bfo3.hpp):
    // v_mfma_f32_32x32x16_bf16, activations in HBM as octet entries (bf16, or hi + lo) stored pre-activated for their consumer,
    // fused c1 -> c2 pairs for C <= 128.  TTSAMD_BFO=0 keeps the round-2 bf16 engine (fp32 activations in HBM).
    const char* bfo_env = opt_str(OPT_BFO);
    const int prec = default_precision();
    if ((prec == 1 || prec == 2) && h->bfo_ok && !(bfo_env && bfo_env[0] == '0')) {
        const bool x3 = prec == 2;
        const uint16_t* W16 = h->dev16;
        const auto woff = [x3](const ConvW& cw) { return x3 ? cw.wo3_off : cw.wo_off; };
        const auto l_conv = x3 ? bfo3_launch_conv : bfo_launch_conv;
        const auto l_convt = x3 ? bfo3_launch_convt : bfo_launch_convt;
        const auto l_pair = x3 ? bfo3_launch_pair : bfo_launch_pair;
        const auto pair_ok = x3 ? bfo3_pair_supported : bfo_pair_supported;
        const char* c3e = opt_str(OPT_BFO_CHAIN);      // 0: three pair launches per k = 3 ResBlock (bit-identical; A/B and parity runs)
        const bool chain3_on = !(c3e && c3e[0] == '0');
        void *curo = cur, *upso = ups_out;                 // the fp32-sized buffers hold bf16 / x3 tensors of the same element count
        HG_TRY((x3 ? bfo3_launch_pack : bfo_launch_pack)(mel, B, cfg.num_mels, T, 1.f, mel_o, s));
        BfoConvParams cp;
        std::memset(&cp, 0, sizeof(cp));
        cp.batch = B; cp.lens = lens; cp.div = 1.f; cp.res_slope = 1.f;
        // conv_pre (models.py:112); its consumer, the first upsampler, applies leaky_relu(0.1) (models.py:114)
        cp.x = mel_o; cp.y = curo; cp.w = W16 + woff(h->conv_pre); cp.bias = h->dev + h->conv_pre.b_off;
        cp.len_mul = 1; cp.Lin = T; cp.Cin = h->conv_pre.cin; cp.Cout = h->conv_pre.cout; cp.K = 7; cp.dil = 1; cp.up = 1;
        cp.mode = 0; cp.out_slope = 0.1f;
        prof_begin(s, 2.0 * cp.Cout * cp.Cin * 7);
        int32_t rc = l_conv(cp, s);
        prof_end(s);
        HG_TRY(rc);
        int L = T, mul = 1;
        for (int i = 0; i < cfg.n_ups; ++i) {
            const int u = cfg.upsample_rates[i];
            const ConvW& uw = h->ups[i];
            // ConvTranspose1d on the activated stage input; the ResBlocks read its output through leaky_relu(0.1)
            cp.x = curo; cp.y = upso; cp.w = W16 + woff(uw); cp.bias = h->dev + uw.b_off; cp.res = nullptr; cp.sum_in = nullptr;
            cp.len_mul = mul; cp.Lin = L; cp.Cin = uw.cin; cp.Cout = uw.cout; cp.K = 2; cp.dil = 1; cp.up = u;
            cp.mode = 0; cp.out_slope = 0.1f;
            prof_begin(s, 2.0 * uw.cout * uw.cin * 2 * u * mul);
            rc = l_convt(cp, s);
            prof_end(s);
            HG_TRY(rc);
            L *= u; mul *= u;
            // what reads the stage sum: the next upsampler through leaky_relu(0.1), conv_post through leaky_relu(0.01)
            const float next_slope = i + 1 < cfg.n_ups ? 0.1f : 0.01f;
            if (multi) {
                prof_section_begin(s);
                in_section = true;
                HG_CHECK_HIP(hipEventRecord(h->ev_fork, s));
            }
            for (int j = 0; j < cfg.n_kernels; ++j) {
                hipStream_t st = bs[j % 3];
                void *Tb = Tbs[j % 3], *R = Rs[j % 3];
                if (multi && j > 0) HG_CHECK_HIP(hipStreamWaitEvent(st, h->ev_fork, 0));
                const void* src = upso;
                {
                    // a ResBlock whose three pairs fit the chained kernel (k = 3; k = 7 at C <= 64) goes out as ONE launch (bfo_chain.hip; bit-identical)
                    const int li0 = (i * cfg.n_kernels + j) * cfg.n_dilations;
                    int32_t dl[3] = {0, 0, 0};
                    for (int m = 0; m < cfg.n_dilations && m < 3; ++m) dl[m] = cfg.resblock_dilations[j][m];
                    const bool chain3 = x3 && chain3_on && bfo3_chain_supported(h->c1[li0].cin, h->c1[li0].k, dl, cfg.n_dilations, L);
                    if (chain3 || (!x3 && bfo_chain_wanted(h->c1[li0].cin, h->c1[li0].k, dl, cfg.n_dilations, L, B))) {
                        BfoChainParams cc;
                        std::memset(&cc, 0, sizeof(cc));
                        cc.x = src; cc.y = curo; cc.sum_in = curo;
                        for (int m = 0; m < 3; ++m) {
                            const ConvW &w1 = h->c1[li0 + m], &w2 = h->c2[li0 + m];
                            cc.w1[m] = W16 + woff(w1); cc.w2[m] = W16 + woff(w2); cc.b1[m] = h->dev + w1.b_off; cc.b2[m] = h->dev + w2.b_off;
                            cc.dil[m] = dl[m];
                        }
                        cc.lens = lens; cc.len_mul = mul; cc.L = L; cc.batch = B; cc.k = h->c1[li0].k;
                        cc.mode = cfg.n_kernels == 1 ? 0 : (j == 0 ? 0 : (j + 1 < cfg.n_kernels ? 1 : 2));
                        cc.div = (float)cfg.n_kernels; cc.in_slope = 0.1f; cc.mid_slope = 0.1f;
                        cc.out_slope = j + 1 == cfg.n_kernels ? next_slope : 1.f;
                        if (multi && j > 0) HG_CHECK_HIP(hipStreamWaitEvent(st, h->ev_done[j - 1], 0));
                        const double fl = 3 * 2.0 * (2.0 * h->c1[li0].cin * h->c1[li0].cin * h->c1[li0].k) * mul;
                        if (in_section) prof_add(fl); else prof_begin(st, fl);
                        rc = x3 ? bfo3_launch_chain(h->c1[li0].cin, cc, st) : bfo_launch_chain(h->c1[li0].cin, cc, st);
                        if (!in_section) prof_end(st);
                        HG_TRY(rc);
                        if (multi) HG_CHECK_HIP(hipEventRecord(h->ev_done[j], st));
                        continue;
                    }
                }
                for (int m = 0; m < cfg.n_dilations; ++m) {
                    const int li = (i * cfg.n_kernels + j) * cfg.n_dilations + m;
                    const int d = cfg.resblock_dilations[j][m];
                    const bool last = m + 1 == cfg.n_dilations;
                    const ConvW &w1 = h->c1[li], &w2 = h->c2[li];
                    void* dst = last ? curo : (src == R ? Tb : R);
                    // running ResBlock sum in `cur`: raw bf16 until the last branch stores it activated for its consumer
                    const int mode = !last || cfg.n_kernels == 1 ? 0 : (j == 0 ? 0 : (j + 1 < cfg.n_kernels ? 1 : 2));
                    const float out_slope = !last ? 0.1f : (j + 1 == cfg.n_kernels ? next_slope : 1.f);
                    if (multi && last && j > 0) HG_CHECK_HIP(hipStreamWaitEvent(st, h->ev_done[j - 1], 0));
                    const double fl = 2.0 * (2.0 * w1.cin * w1.cin * w1.k) * mul;
                    if (pair_ok(w1.cin, w1.k, d, L)) {
                        BfoPairParams pp;
                        std::memset(&pp, 0, sizeof(pp));
                        pp.x = src; pp.y = dst; pp.sum_in = curo;
                        pp.w1 = W16 + woff(w1); pp.w2 = W16 + woff(w2); pp.b1 = h->dev + w1.b_off; pp.b2 = h->dev + w2.b_off;
                        pp.lens = lens; pp.len_mul = mul; pp.L = L; pp.dil = d; pp.batch = B;
                        pp.mode = mode; pp.div = (float)cfg.n_kernels; pp.in_slope = 0.1f; pp.mid_slope = 0.1f; pp.out_slope = out_slope;
                        if (in_section) prof_add(fl); else prof_begin(st, fl);
                        rc = l_pair(w1.cin, w1.k, pp, st);
                        if (!in_section) prof_end(st);
                        HG_TRY(rc);
                    } else {
                        // C = 256 (stage 1): c1 and c2 as two launches, the intermediate stored activated for c2.  c2 reads its
                        // residual only at its own output positions, so it may run in place (dst == src == R)
                        void* t1 = Tb;
                        if (!last) dst = R;
                        cp.x = src; cp.y = t1; cp.w = W16 + woff(w1); cp.bias = h->dev + w1.b_off; cp.res = nullptr; cp.sum_in = nullptr;
                        cp.len_mul = mul; cp.Lin = L; cp.Cin = w1.cin; cp.Cout = w1.cout; cp.K = w1.k; cp.dil = d; cp.up = 1;
                        cp.mode = 0; cp.out_slope = 0.1f; cp.res_slope = 1.f;
                        cp.splitk_ws = splitks[j % 3]; cp.splitk_floats = kSplitKFloats;     // batch 1: 30 blocks per stage-1 conv
                        if (in_section) prof_add(fl); else prof_begin(st, fl);
                        rc = l_conv(cp, st);
                        if (rc == 0) {
                            cp.x = t1; cp.y = dst; cp.w = W16 + woff(w2); cp.bias = h->dev + w2.b_off; cp.res = src; cp.sum_in = curo;
                            cp.dil = 1; cp.mode = mode; cp.div = (float)cfg.n_kernels; cp.out_slope = out_slope; cp.res_slope = 0.1f;
                            rc = l_conv(cp, st);
                        }
                        if (!in_section) prof_end(st);
                        HG_TRY(rc);
                    }
                    if (multi && last) HG_CHECK_HIP(hipEventRecord(h->ev_done[j], st));
                    src = dst;
                }
            }
            if (multi) {
                HG_CHECK_HIP(hipStreamWaitEvent(s, h->ev_done[cfg.n_kernels - 1], 0));
                in_section = false;
                prof_section_end(s);
            }
        }
        HG_TRY((x3 ? bfo3_launch_conv_post : bfo_launch_conv_post)(curo, h->dev + h->conv_post.w_off, h->dev + h->conv_post.b_off, lens, mul, B, h->conv_post.cin,
                                    L, wave, (int64_t)L, s));
        return 0;
    }

    // conv_pre (models.py:112)
    HG_TRY(conv(h->conv_pre, mel, s, cur, nullptr, T, 1, 1, 1.0f, 0, 1.f));
    int L = T, mul = 1;
    for (int i = 0; i < cfg.n_ups; ++i) {
        const int u = cfg.upsample_rates[i], kt = cfg.upsample_kernel_sizes[i];
        const ConvW& uw = h->ups[i];
        // leaky_relu(0.1) + ConvTranspose1d as u polyphase 2-tap convs (models.py:114-115)
        p.x = cur; p.x_bs = (int64_t)uw.cin * L; p.x_cs = L;
        p.w = h->dev + uw.w_off; p.bias = h->dev + uw.b_off;
        p.w_bf16 = h->dev16 + uw.w16_off; p.precision = default_precision(); p.w_wino = nullptr; p.w_wino4 = nullptr;
        p.y = ups_out; p.y_bs = (int64_t)uw.cout * L * u; p.y_cs = L * u; p.y_ts = u;
        p.res = nullptr;
        p.len_in_mul = mul; p.len_out_mul = mul; p.Lin = L; p.Nout = L;
        p.Cin = uw.cin; p.Cout = uw.cout; p.CoutP = cout_padded(uw.cout); p.K = 2;
        p.dil = -1; p.pad = 0; p.n_phase = u; p.phase_p = (kt - u) / 2;
        p.in_slope = 0.1f; p.relu_out = 0; p.mode = 0; p.div = 1.f;
        p.x_packed = 0; p.y_packed = 0;
        prof_begin(s, 2.0 * uw.cout * uw.cin * 2 * u * mul);
        int32_t rc = (convt_ok && convt_supported(p)) ? launch_convt(p, s) : launch_conv(p, s);
        prof_end(s);
        HG_TRY(rc);
        L *= u; mul *= u;
        // 3 ResBlock1 on the same input, averaged (models.py:116-122, 46-53)
        if (multi) {
            prof_section_begin(s);
            in_section = true;
            HG_CHECK_HIP(hipEventRecord(h->ev_fork, s));
        }
        for (int j = 0; j < cfg.n_kernels; ++j) {
            hipStream_t st = bs[j % 3];
            float *Tb = Tbs[j % 3], *R = Rs[j % 3];
            p.splitk_ws = splitks[j % 3];
            if (multi && j > 0) HG_CHECK_HIP(hipStreamWaitEvent(st, h->ev_fork, 0));
            const float* src = ups_out;
            for (int m = 0; m < cfg.n_dilations; ++m) {
                const int li = (i * cfg.n_kernels + j) * cfg.n_dilations + m;
                const int d = cfg.resblock_dilations[j][m];
                // C = 32 stage (and k = 3 of the C = 64 stage), fp32: the c1 -> c2 pair in one launch with the intermediate in LDS (resblock_fused.hip);
                // x and y must differ (halo reads), so the pair outputs alternate between R and the unused c1 buffer
                {
                    const bool last = m + 1 == cfg.n_dilations;
                    float* dst = last ? cur : (src == R ? Tb : R);
                    const ConvW &w1 = h->c1[li], &w2 = h->c2[li];
                    const bool square = w1.cin == w1.cout && w2.cin == w1.cin && w2.cout == w1.cin && w1.k == w2.k;
                    const int ntw2 = (fused_ok && square) ? fused2_choice(f2sw, w1.cin, w1.k, d, L, src, dst, (int64_t)B * L) : 0;
                    if (ntw2 != 0 || (fused_ok && square && fused_pair_supported(w1.cin, w1.k, d, L, src, dst))) {
                        const int mode = !last ? 0 : (cfg.n_kernels == 1 ? 0 : (j == 0 ? 0 : (j + 1 < cfg.n_kernels ? 1 : 2)));
                        if (multi && last && j > 0) HG_CHECK_HIP(hipStreamWaitEvent(st, h->ev_done[j - 1], 0));
                        const double fl = 2.0 * (2.0 * w1.cin * w1.cin * w1.k) * mul;
                        if (in_section) prof_add(fl); else prof_begin(st, fl);
                        const int32_t frc = ntw2 != 0
                            ? launch_fused_pair2(w1.cin, src, dst, h->dev + w1.w_off, h->dev + w1.b_off, h->dev + w2.w_off,
                                                 h->dev + w2.b_off, w1.k, d, lens, mul, L, B, mode, (float)cfg.n_kernels, 0.1f, ntw2, st,
                                                 (f2sw.wino_b && w2.ww_off >= 0) ? h->dev + w2.ww_off : nullptr,
                                                 (f2sw.wino_a && w1.ww_off >= 0) ? h->dev + w1.ww_off : nullptr)
                            : launch_fused_pair(w1.cin, src, dst, h->dev + w1.w_off, h->dev + w1.b_off, h->dev + w2.w_off,
                                                h->dev + w2.b_off, w1.k, d, lens, mul, L, B, mode, (float)cfg.n_kernels, 0.1f, st);
                        if (!in_section) prof_end(st);
                        HG_TRY(frc);
                        if (multi && last) HG_CHECK_HIP(hipEventRecord(h->ev_done[j], st));
                        src = dst;
                        continue;
                    }
                }
                pack_io = pack_t ? 2 : 0;
                HG_TRY(conv(h->c1[li], src, st, Tb, nullptr, L, mul, d, 0.1f, 0, 1.f));
                pack_io = pack_t ? 1 : 0;
                if (m + 1 < cfg.n_dilations) {
                    HG_TRY(conv(h->c2[li], Tb, st, R, src, L, mul, 1, 0.1f, 0, 1.f));
                    pack_io = 0;
                    src = R;
                } else {
                    const int mode = (j == 0) ? 0 : (j + 1 < cfg.n_kernels ? 1 : 2);
                    const int md = cfg.n_kernels == 1 ? 0 : mode;
                    // the accumulation into `cur` follows branch j-1's
                    if (multi && j > 0) HG_CHECK_HIP(hipStreamWaitEvent(st, h->ev_done[j - 1], 0));
                    HG_TRY(conv(h->c2[li], Tb, st, cur, src, L, mul, 1, 0.1f, md, (float)cfg.n_kernels));
                    pack_io = 0;
                    if (multi) HG_CHECK_HIP(hipEventRecord(h->ev_done[j], st));
                }
            }
        }
        p.splitk_ws = splitks[0];
        if (multi) {
            HG_CHECK_HIP(hipStreamWaitEvent(s, h->ev_done[cfg.n_kernels - 1], 0));
            in_section = false;
            prof_section_end(s);
        }
    }
    // leaky_relu (default slope 0.01) + conv_post + tanh (models.py:123-125)
    HG_TRY(launch_conv_post(cur, (int64_t)h->conv_post.cin * L, L, h->dev + h->conv_post.w_off,
                             h->dev + h->conv_post.b_off, lens, mul, B, h->conv_post.cin, L, 0.01f, wave,
                             (int64_t)L, s));
#undef HG_TRY
#undef HG_CHECK_HIP
    return 0;
}

void hifigan_blobs(const void* hv, void** f32, int64_t* n_f32, void** b16, int64_t* n_b16) {
    const HifiGan* h = (const HifiGan*)hv;
    *f32 = h->dev; *n_f32 = h->dev_n; *b16 = h->dev16; *n_b16 = h->dev16_n;
}

}  // namespace ttsamd
