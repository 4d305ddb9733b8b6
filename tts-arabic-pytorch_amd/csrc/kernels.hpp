// Launchers of the small (HBM-bound / latency-bound) kernels around the MFMA conv engine.
#pragma once
#include <vector>

#include "common.hpp"

namespace ttsamd {

// HiFi-GAN tail: wave[b][t] = tanh(b0 + sum_{c,k} w[c][k] * lrelu_0.01(x[b][c][t+k-3]))
// (vocoder/hifigan/models.py:123-125).  x [B][C][L] (row stride L), w [C][7] device.
int32_t launch_conv_post(const float* x, int64_t x_bs, int32_t x_cs, const float* w, const float* bias,
                         const int64_t* lens, int32_t len_mul, int32_t B, int32_t C, int32_t L,
                         float in_slope, float* wave, int64_t wave_bs, hipStream_t s);

// LayerNorm over the channel axis of a channel-first tensor, in place or out of place:
// y[b][c][t] = ((x - mean_t) * rstd_t * gamma[c] + beta[c]) * (t < lens[b] || !mask)
// (transformer.py:88,158,174,176; model.py:56).  eps = 1e-5 (torch default).
int32_t launch_layernorm_cf(const float* x, float* y, const float* gamma, const float* beta,
                            const int64_t* lens, int32_t apply_mask, int32_t B, int32_t C, int32_t S,
                            hipStream_t s, float eps = 1e-5f);
// the same, also writing the result as an octet bf16 tensor [B][C/8][S][8] (bfo.hpp): the next conv's input copy (C % 64 == 0, C <= 512)
int32_t launch_layernorm_cf_octet(const float* x, float* y, void* y_octet, const float* gamma, const float* beta,
                                  const int64_t* lens, int32_t apply_mask, int32_t B, int32_t C, int32_t S,
                                  hipStream_t s, float eps = 1e-5f);

// Encoder input: x[b][c][t] = word_emb[ids[b][t]][c] + pos[t][c]*(ids!=pad) + spk[c]
// (transformer.py:212-219; model.py:355-361).  Also writes lens[b] = #non-pad tokens.
int32_t launch_embed(const int64_t* ids, const float* word_emb, const float* pos_table, int32_t pos_stride,
                     const float* spk, int32_t pad_idx, int32_t n_symbols, int32_t B, int32_t L, int32_t C, float* x, int64_t* lens,
                     hipStream_t s);

// 1-head self attention over channel-first q,k,v = rows [0,D),[D,2D),[2D,3D) of qkv [B][3D][S]
// (transformer.py:131-141): out[b][d][i] = sum_j softmax_j(q_i.k_j * scale | j < lens[b]) v_j[d]
// ws / ws_floats: scratch for the small-batch schedule (one block per key tile + a merge launch; same bits), or nullptr
int32_t launch_attention(const float* qkv, const int64_t* lens, int32_t B, int32_t D, int32_t S,
                         float scale, float* out, hipStream_t s, float* ws = nullptr, int64_t ws_floats = 0);

// the same on the bf16 matrix cores (attention_bf16.hip; config 3): launch_attention routes here under ttsamd_set_precision(1)
// out_octet != nullptr: the result leaves as an octet bf16 tensor [B][D/8][S][8] (bfo.hpp) instead of fp32 channel-first
int32_t launch_attention_bf16(const float* qkv, const int64_t* lens, int32_t B, int32_t D, int32_t S, float scale, float* out,
                              hipStream_t s, void* out_octet = nullptr);

// Predictor head (model.py:132): out[b][t] = (bias + sum_c w[c]*x[b][c][t]) * (t < lens[b]);
// mode 1 additionally writes dur = clamp(exp(out)-1, 0, max_dur) (model.py:368) to out2.
int32_t launch_pred_fc(const float* x, const float* w, const float* bias, const int64_t* lens, int32_t B,
                       int32_t C, int32_t S, float* out, float* out2, float max_dur, float mul, float add,
                       hipStream_t s);

// enc[b][c][t] += bias[c] + sum_k w[c][k] * src[b][t+k-K/2]   (Conv1d(1->C,k) embeddings, model.py:382-397)
int32_t launch_scalar_emb_add(float* enc, const float* src, const float* w, const float* bias, int32_t B,
                              int32_t C, int32_t S, int32_t K, hipStream_t s);

// Integer half of regulate_len (model.py:72-76): reps=(dur/pace+0.5).long(), dec_lens=sum.
int32_t launch_durations_to_reps(const float* dur, float pace, int32_t B, int32_t L, int64_t* reps,
                                 int64_t* dec_lens, hipStream_t s);

// Gather half (model.py:77-85) + optional decoder positional embedding (transformer.py:215-219).
int32_t launch_regulate_gather(const float* enc, const int64_t* reps, const float* pos_table, int32_t pos_stride, int32_t B,
                               int32_t L, int32_t C, int32_t T, float* out, int32_t* idx, hipStream_t s);

// x[b][c][t] += pos[t][c] * (t < lens[b])
int32_t launch_lens_plus1(const int64_t* lens, int32_t S, int32_t B, int32_t clamp_at_max, int64_t* out, hipStream_t s);   // out[b] = min(lens[b] + 1, clamp_at_max ? max_b lens[b] : S)
int32_t launch_add_pos(float* x, const float* pos_table, int32_t pos_stride, const int64_t* lens, int32_t B, int32_t C,
                       int32_t S, hipStream_t s);

// Depthwise Conv1d(C,C,k7,p3,groups=C) (vocoder/vocos/modules.py:31,45); input read as zero past lens[b]
int32_t launch_dwconv7(const float* x, const float* w, const float* bias, const int64_t* lens, int32_t B, int32_t C,
                       int32_t S, float* y, hipStream_t s);

// the window and the overlap-add shared by the denoiser and the Vocos ISTFT head
void hann_window_1024(std::vector<float>& window);   // periodic hann, n = 1024 (denoiser.hip)
int32_t launch_overlap_add(const float* Y, const float* win, const int64_t* frames, int32_t frames_mul, int32_t frames_add,
                           int32_t pad, int32_t B, int32_t F, int32_t n_max, float* wave, int64_t wave_bs, hipStream_t s,
                           int32_t frame_major = 0);   // Y as [b][k][F] (0) or [b][F][k] (1)

// Profiling of conv launches (bench roofline)
void prof_begin(hipStream_t s, double flops);
void prof_end(hipStream_t s);
// fork..join sections of concurrent launches: ONE event pair on the forking stream + per-launch counting
void prof_section_begin(hipStream_t s);
void prof_section_end(hipStream_t s);
void prof_add(double flops);

}  // namespace ttsamd
