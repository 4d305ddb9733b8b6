/*
 * ttsamd.h — C ABI of libttsamd.so: the MI355X (gfx950) FastPitch -> HiFi-GAN hot path.
 *
 * The reference (nipponjo/tts-arabic-pytorch) has no FFI/plugin interface; its boundary is
 * the Python class surface models.fastpitch.FastPitch2Wave / FastPitch.
 * Each entry point below replaces one reference *function* on that path and is what a
 * ctypes binding inside the reference's own modules would call (INTEGRATION.md shows the
 * stub).  Conventions:
 *   - plain C types only; every data pointer is a DEVICE pointer (tensor.data_ptr()) to
 *     contiguous row-major memory unless marked "host";
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     0/NULL = the default stream.  Calls are asynchronous on that stream;
 *   - return 0 on success, a negative TTSAMD_E* code otherwise; ttsamd_last_error()
 *     returns a thread-local message;
 *   - no hidden device allocation on hot calls: the caller passes a workspace sized by the
 *     matching *_workspace_bytes query.  Handles are immutable after create, so concurrent
 *     calls with distinct workspaces/streams are allowed.
 */
#ifndef TTSAMD_H
#define TTSAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTSAMD_OK 0
#define TTSAMD_EINVAL (-1)   /* bad argument / shape / missing tensor */
#define TTSAMD_EHIP (-2)     /* HIP runtime error */
#define TTSAMD_ENOMEM (-3)   /* workspace too small / allocation failure */

/* One named host tensor of a checkpoint (fp32, row-major).  Names are the reference's
 * state_dict keys (models/fastpitch/networks.py:52-71; vocoder/__init__.py:15-16). */
typedef struct ttsamd_tensor {
    const char* name;
    const float* data;   /* host */
    int32_t ndim;
    int64_t shape[4];
} ttsamd_tensor;

/* pretrained/hifigan-asc-v1/config.json:2,11-15 */
typedef struct ttsamd_hifigan_cfg {
    int32_t num_mels;                 /* 80 */
    int32_t upsample_initial_channel; /* 512 */
    int32_t n_ups;                    /* 4 */
    int32_t upsample_rates[8];        /* 8,8,2,2 */
    int32_t upsample_kernel_sizes[8]; /* 16,16,4,4 */
    int32_t n_kernels;                /* 3 resblocks per stage */
    int32_t resblock_kernel_sizes[8]; /* 3,7,11 */
    int32_t n_dilations;              /* 3 */
    int32_t resblock_dilations[8][8]; /* (1,3,5) each */
} ttsamd_hifigan_cfg;

/* models/fastpitch/__init__.py:3-41 (net_config), only the fields inference reads */
typedef struct ttsamd_fastpitch_cfg {
    int32_t n_mel_channels;   /* 80 */
    int32_t n_symbols;        /* 148 */
    int32_t padding_idx;      /* 0 */
    int32_t d_model;          /* symbols_embedding_dim 384 */
    int32_t in_fft_n_layers, in_fft_n_heads, in_fft_d_head, in_fft_kernel, in_fft_filter;
    int32_t out_fft_n_layers, out_fft_n_heads, out_fft_d_head, out_fft_kernel, out_fft_filter;
    int32_t dur_kernel, dur_filter, dur_n_layers;
    int32_t pitch_kernel, pitch_filter, pitch_n_layers, pitch_emb_kernel;
    int32_t energy_conditioning, energy_kernel, energy_filter, energy_n_layers, energy_emb_kernel;
    int32_t n_speakers;
    float speaker_emb_weight;
} ttsamd_fastpitch_cfg;

/* models/tacotron2/tacotron2_ms.py:121-153 (Tacotron2MS.__init__ defaults), inference fields only */
typedef struct ttsamd_tacotron2_cfg {
    int32_t n_symbol;                       /* 40 */
    int32_t num_speakers;                   /* 1: no speaker embedding */
    int32_t speaker_embedding_dim;          /* 128 (used when num_speakers > 1) */
    int32_t symbol_embedding_dim;           /* 512 */
    int32_t encoder_embedding_dim;          /* 512 */
    int32_t encoder_n_convolution;          /* 3 */
    int32_t encoder_kernel_size;            /* 5 */
    int32_t n_mels;                         /* 80 */
    int32_t prenet_dim;                     /* 256 */
    int32_t attention_rnn_dim;              /* 1024 */
    int32_t decoder_rnn_dim;                /* 1024 */
    int32_t attention_hidden_dim;           /* 128 */
    int32_t attention_location_n_filter;    /* 32 */
    int32_t attention_location_kernel_size; /* 31 */
    int32_t postnet_n_convolution;          /* 5 */
    int32_t postnet_kernel_size;            /* 5 */
    int32_t postnet_embedding_dim;          /* 512 */
    float gate_threshold;                   /* 0.5 */
    int32_t decoder_early_stopping;         /* 1: stop once every utterance's gate fired; 0: always decode max_step frames
                                             *    (tacotron2_ms.py:139,169,197 -> torchaudio _Decoder.infer) */
} ttsamd_tacotron2_cfg;

/* models/diacritizers/shakkelha/network.py:10-27, shakkala/network.py:9-24 as one tagger geometry */
typedef struct ttsamd_tagger_cfg {
    int32_t n_vocab;          /* embedding rows (91 / 149) */
    int32_t emb_dim;          /* 25 / 288 */
    int32_t n_lstm;           /* bidirectional LSTM layers (2 / 3) */
    int32_t lstm_hidden[4];   /* per direction (256,256 / 288,144,96) */
    int32_t hard_sigmoid;     /* 1: gates i,f,o = clamp(0.2x+0.5,0,1) (shakkala/lstm_hsm.py:352-379) */
    int32_t bn_after_lstm0;   /* 1: eval BatchNorm1d between lstm0 and lstm1 (shakkala/network.py:35) */
    float bn_eps;
    int32_t n_dense;          /* Linear layers after the LSTMs, ReLU on all but the last (3 / 1) */
    int32_t dense_dim[4];     /* 512,512,19 / 28; the last one is the number of classes */
} ttsamd_tagger_cfg;

const char* ttsamd_last_error(void);
/* ABI revision of this header.  Bumped whenever a struct gains a field or an argument changes meaning (2: ttsamd_tacotron2_cfg
 * gained decoder_early_stopping, ttsamd_profile_read's third value became the number of timed sections; 3: ttsamd_dp_* may be
 * bound twice per process, one communicator per stream; 4: ttsamd_bfo_resblock_chain takes the kernel size as its last argument; 5: the ttsamd_bfo3_* entries, ttsamd_conv1d_ex; 6: ttsamd_set_option / ttsamd_get_option / ttsamd_option_name / ttsamd_options_check replace the per-call
 * environment reads, ttsamd_resblock_pair takes the size of `packed`, ttsamd_resblock_pair_packed_floats).  ttsamd_version() returns the value the library was BUILT with: a caller
 * compiled against another revision must refuse to run (ttsamd/lib.py does). */
#define TTSAMD_ABI_VERSION 7
int32_t ttsamd_version(void);
/* Run-time routing options: every switch that routes between kernels / schedules that both ship (INTEGRATION.md lists them with their
 * defaults).  An option's value is seeded ONCE from the environment variable of the same name (TTSAMD_<NAME>) when the library is first
 * used and changed afterwards only here; `name` with or without the TTSAMD_ prefix, `value` as the variable would hold it (decimal, hex
 * for the masks), NULL or "" = unset (the default applies).  An unknown name or a value outside the option's range returns TTSAMD_EINVAL
 * and changes nothing.  ttsamd_get_option writes the current text ("" = unset); ttsamd_option_name(i) enumerates the names (NULL past the
 * last); ttsamd_options_check reports a malformed TTSAMD_<NAME> found in the environment at load (ttsamd/lib.py raises on it).  No
 * reference counterpart. */
int32_t ttsamd_set_option(const char* name, const char* value);
int32_t ttsamd_get_option(const char* name, char* value, int32_t capacity);
const char* ttsamd_option_name(int32_t index);
int32_t ttsamd_options_check(void);
/* 1 if a gfx950 device is visible to the HIP runtime, else 0 (never throws). */
int32_t ttsamd_device_ok(void);

/* ---- HiFi-GAN generator: replaces vocoder.load_hifigan + Generator.forward
 *      (vocoder/__init__.py:3-20, vocoder/hifigan/models.py:86-136) ------------------- */

/* Accepts weight-normalised checkpoints (`*.parametrizations.weight.original0/1`,
 * `*.weight_g/_v`) and folded ones (`*.weight`); folds w = g*v/||v|| (norm over all dims
 * but 0) on the host exactly as remove_weight_norm does (models.py:129-136). */
int32_t ttsamd_hifigan_create(const ttsamd_tensor* weights, int32_t n_weights,
                              const ttsamd_hifigan_cfg* cfg, void** handle);
int32_t ttsamd_hifigan_destroy(void* handle);
int64_t ttsamd_hifigan_workspace_bytes(void* handle, int32_t batch, int32_t t_max);
/* mel  [B][num_mels][t_max] (frames >= lens[b] are ignored), lens int64 [B] or NULL (all
 * t_max).  wave [B][hop*t_max]; samples >= hop*lens[b] are left untouched.  Every layer
 * zero-pads at the TRUE utterance edge, i.e. utterance b's result equals the reference's
 * unbatched Generator.forward(mel[b,:,:lens[b]]) (models/fastpitch/networks.py:340-341). */
int32_t ttsamd_hifigan_forward(void* handle, const float* mel, const int64_t* lens,
                               int32_t batch, int32_t t_max, float* wave,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* ---- FastPitch.infer split at its one data-dependent size (dec_lens):
 *      models/fastpitch/fastpitch/model.py:351-409 ------------------------------------ */

int32_t ttsamd_fastpitch_create(const ttsamd_tensor* weights, int32_t n_weights,
                                const ttsamd_fastpitch_cfg* cfg, void** handle);
int32_t ttsamd_fastpitch_destroy(void* handle);
int64_t ttsamd_fastpitch_encode_workspace_bytes(void* handle, int32_t batch, int32_t n_tokens);
int64_t ttsamd_fastpitch_decode_workspace_bytes(void* handle, int32_t batch, int32_t t_max);

/* Phase A (model.py:355-399 + the integer half of regulate_len :68-76).
 * ids int64 [B][L], zero-padded at the END of each row (text_collate_fn,
 * models/fastpitch/networks.py:16-35).  dur_tgt [B][L] / pitch_tgt [B][1][L] /
 * energy_tgt [B][1][L] may be NULL (use predictions).  pitch_mul/pitch_add apply the
 * reference's pitch_trf (networks.py:38-42) when != (1,0).
 * Outputs: enc_cond [B][d_model][L] (CHANNEL-FIRST conditioned encoder output, the input
 * of ttsamd_length_regulate), dur_pred [B][L], pitch_pred [B][1][L], energy_pred [B][L],
 * reps int64 [B][L] = (dur/pace+0.5).long(), dec_lens int64 [B].  The host reads
 * dec_lens (the reference syncs here too, model.py:76) to size phase B. */
int32_t ttsamd_fastpitch_encode(void* handle, const int64_t* ids, int32_t batch, int32_t n_tokens,
                                int32_t speaker, float pace, const float* dur_tgt,
                                const float* pitch_tgt, const float* energy_tgt,
                                float pitch_mul, float pitch_add, float max_duration,
                                float* enc_cond, float* dur_pred, float* pitch_pred,
                                float* energy_pred, int64_t* reps, int64_t* dec_lens,
                                void* workspace, int64_t workspace_bytes, void* stream);

/* Float half of regulate_len (model.py:77-85) as a gather instead of the reference's
 * one-hot matmul: out[b][c][t] = enc[b][c][j] with cumsum[j] <= t < cumsum[j+1], zero for
 * t >= dec_len.  enc [B][C][L], reps int64 [B][L], out [B][C][t_max], idx int32 [B][t_max]
 * (token index per frame, -1 past the end) may be NULL.  Bit-exact indices. */
int32_t ttsamd_length_regulate(const float* enc, const int64_t* reps, int32_t batch,
                               int32_t n_tokens, int32_t channels, int32_t t_max,
                               float* out, int32_t* idx, void* stream);

/* Phase B (model.py:405-408): decoder FFT + proj.  x [B][d_model][t_max] channel-first
 * (output of ttsamd_length_regulate; clobbered), dec_lens int64 [B], mel [B][80][t_max].
 * t_max is the ROW WIDTH of x and mel and may exceed max(dec_lens): the length of the reference's padded batch is taken from
 * dec_lens (batches of 2 and more), the extra columns are padding.  Pass a multiple of 4 for batches: the Winograd and float4-epilogue
 * paths of the conv engine need 16-byte-aligned rows (ttsamd/engine.py: FastPitchEngine.infer rounds t_max up and returns a view). */
int32_t ttsamd_fastpitch_decode(void* handle, float* x, const int64_t* dec_lens, int32_t batch,
                                int32_t t_max, float* mel, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* How a BATCH of utterances goes through ttsamd_fastpitch_encode / _decode of this handle (default 0; not a per-call argument: set it
 * between calls, not under a running one).
 *   0  the reference's padded-batch arithmetic: the hidden activations of the conv-FF blocks and of the predictors are not masked, so
 *      the last frames of an utterance depend on the longest one of its batch (FastPitch.infer on a padded batch,
 *      models/fastpitch/fastpitch/transformer.py:72-90, model.py:129-133) -- what tts(list, batch_size > 1) returns.
 *   1  every utterance as if it were alone in the call: those two convs read their input masked at the utterance's own length (every
 *      other op already masks), so row b equals FastPitch.infer(ids[b:b+1, :len_b]) -- the reference's batch_size = 1 loop
 *      (models/fastpitch/networks.py:402-411) as ONE ragged call.  Equal to the one-by-one calls within fp32 summation order. */
int32_t ttsamd_fastpitch_set_batch_mode(void* handle, int32_t mode);

/* ---- HiFi-GAN bias denoiser: replaces vocoder.hifigan.denoiser.Denoiser
 *      (vocoder/hifigan/denoiser.py:32-64 __init__, :66-72 forward).  STFT/ISTFT with
 *      n_fft = win = 1024, hop 256, periodic hann, center/reflect, onesided, unnormalised. --- */
int32_t ttsamd_denoiser_create(void** handle);
int32_t ttsamd_denoiser_destroy(void* handle);
int64_t ttsamd_denoiser_workspace_bytes(int32_t batch, int32_t n_max);
/* bias_spec[513] = |STFT(audio)|[:, frame 0]; audio [n] is the vocoder output for a zero mel
 * (denoiser.py:50-64); n_dev = device int64 holding n. */
int32_t ttsamd_denoiser_bias_spec(void* handle, const float* audio, const int64_t* n_dev, int32_t n,
                                  float* bias_spec, void* workspace, int64_t workspace_bytes,
                                  void* stream);
/* In place: wave[b][0 : 256*(nsamples[b]/256)] <- ISTFT(max(|X|-strength*bias,0) * e^{i arg X}).
 * wave [B][wave_stride], nsamples int64 [B] (device), every nsamples[b] > 512. */
int32_t ttsamd_denoise(void* handle, float* wave, int64_t wave_stride, const int64_t* nsamples,
                       int32_t batch, int32_t n_max, const float* bias_spec, float strength,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* ---- MelVocos('22k') vocoder: replaces vocoder.vocos.pretrained.MelVocos
 *      (vocoder/vocos/pretrained.py:34-93; backbone models.py:26-89, ConvNeXtBlock modules.py:8-60,
 *      ISTFTHead heads.py:26-41, ISTFT "same" spectral_ops.py:33-75).  Weight names are the keys of
 *      MelVocos.state_dict() (backbone.*, head.out.*). ------------------------------------------- */
int32_t ttsamd_vocos_create(const ttsamd_tensor* weights, int32_t n_weights, int32_t input_channels,
                            int32_t dim, int32_t intermediate_dim, int32_t num_layers, void** handle);
int32_t ttsamd_vocos_destroy(void* handle);
int64_t ttsamd_vocos_workspace_bytes(void* handle, int32_t batch, int32_t t_max);
/* bias_vec[513] = clip(exp(log-magnitude of a zero mel [1,80,88]), max 100)[:, frame 0]
 * (make_denoising_vector, pretrained.py:59-71). */
int32_t ttsamd_vocos_bias_vec(void* handle, float* bias_vec, void* workspace, int64_t workspace_bytes,
                              void* stream);
/* mel [B][80][t_max], lens int64 [B] (device) -> wave [B][256*t_max]; samples >= 256*lens[b] untouched.
 * mag = clamp(exp(.) - denoise*bias_vec, 0, 100) (pretrained.py:79-88). */
int32_t ttsamd_vocos_forward(void* handle, const float* mel, const int64_t* lens, int32_t batch,
                             int32_t t_max, float denoise, const float* bias_vec, float* wave,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* ---- Tacotron2MS.infer: replaces models/tacotron2/tacotron2_ms.py:279-332 (encoder, speaker
 *      concat, autoregressive _Decoder.infer, postnet).  Weight names are the keys of
 *      Tacotron2MS.state_dict() (embedding.weight, speaker_embedding.weight, encoder.*, decoder.*,
 *      postnet.*); BatchNorm layers are folded on the host (eval mode). --------------------------- */
int32_t ttsamd_tacotron2_create(const ttsamd_tensor* weights, int32_t n_weights,
                                const ttsamd_tacotron2_cfg* cfg, void** handle);
int32_t ttsamd_tacotron2_destroy(void* handle);
int64_t ttsamd_tacotron2_workspace_bytes(void* handle, int32_t batch, int32_t n_tokens, int32_t max_step);
/* tokens int64 [B][n_tokens] (zero-padded), lengths int64 [B] sorted descending or not (packed-sequence
 * semantics per utterance), speaker_ids int64 [B] or NULL (num_speakers == 1); all DEVICE pointers.
 * Outputs (device): mel_post / mel_raw [B][n_mels][max_step] (row stride max_step; frames >= *n_steps
 * are untouched), mel_lens int32 [B], alignments [B][max_step][n_tokens].  *n_steps (HOST) = number of
 * decoder steps the reference loop would have run (it stops once every utterance's gate fired, or at
 * max_step).  dropout_seed < 0 disables the prenet dropout; >= 0 applies the always-on p=0.5 dropout of
 * torchaudio's _Prenet with a counter-based hash (seed, layer, step, b, j) instead of torch's RNG. */
int32_t ttsamd_tacotron2_infer(void* handle, const int64_t* tokens, const int64_t* lengths,
                               const int64_t* speaker_ids, int32_t batch, int32_t n_tokens, int32_t max_step,
                               int64_t dropout_seed, float* mel_post, int32_t* mel_lens, float* alignments,
                               float* mel_raw, int32_t* n_steps, void* workspace, int64_t workspace_bytes,
                               void* stream);

/* ---- diacritizer taggers: replaces Shakkelha.forward / Shakkala.forward
 *      (models/diacritizers/shakkelha/network.py:29-42, shakkala/network.py:31-43).  Weight names are
 *      canonical: emb.weight, lstm<i>.{weight,bias}_{ih,hh}_l0[_reverse], bn0.{weight,bias,running_mean,
 *      running_var}, dense<i>.{weight,bias} (the Python classes rename emb0 / emb_input). ------------------ */
int32_t ttsamd_tagger_create(const ttsamd_tensor* weights, int32_t n_weights, const ttsamd_tagger_cfg* cfg,
                             void** handle);
int32_t ttsamd_tagger_destroy(void* handle);
int64_t ttsamd_tagger_workspace_bytes(void* handle, int32_t batch, int32_t n_chars);
/* ids int64 [B][n_chars] (device) -> probs [B][n_chars][n_classes] (device), softmax over the classes.
 * Sequences run over the full n_chars in both directions (no packing), as the reference modules do. */
int32_t ttsamd_tagger_forward(void* handle, const int64_t* ids, int32_t batch, int32_t n_chars, float* probs,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* ---- kernel-level entry used by the parity tests and the roofline bench ------------- */

/* One Conv1d through the implicit-GEMM MFMA kernel: y = conv1d(lrelu_slope(x), w) + b.
 * x [B][Cin][Lin], w [Cout][Cin][K] (torch layout, DEVICE), y [B][Cout][Lin] ("same"
 * padding (K*dil-dil)/2), lens int64 [B] or NULL.  Allocates nothing; `packed` must hold
 * ttsamd_conv1d_packed_floats(Cout,Cin,K) floats of scratch for the re-laid-out weights. */
int64_t ttsamd_conv1d_packed_floats(int32_t cout, int32_t cin, int32_t k);
int32_t ttsamd_conv1d(const float* x, const float* w, const float* bias, const int64_t* lens,
                      int32_t batch, int32_t cin, int32_t cout, int32_t k, int32_t dilation,
                      int32_t lin, float in_slope, int32_t relu_out, float* y, float* packed,
                      void* stream);
/* the same with the rest of the conv engine's epilogue: y = act(conv + bias + res) | y + ... | (y + ...) / div  (mode 0 | 1 | 2); res
 * [B][Cout][lin] or NULL.  Drives the residual-preload epilogues of the direct and the Winograd F(2,3) kernel (csrc/conv_wino.hip:
 * k = 3, dilation 1 launches of at least one 128 x 128 tile per CU; TTSAMD_WINO=0 keeps the direct kernel) in the parity tests. */
/* (relu_out != 0 is defined for mode 0 only: with an accumulate mode the call returns TTSAMD_EINVAL) */
int32_t ttsamd_conv1d_ex(const float* x, const float* w, const float* bias, const float* res, const int64_t* lens, int32_t batch,
                         int32_t cin, int32_t cout, int32_t k, int32_t dilation, int32_t lin, float in_slope,
                         int32_t relu_out, int32_t mode, float div, float* y, float* packed, void* stream);

/* One c1 -> c2 pair of a ResBlock1 (vocoder/hifigan/models.py:46-53) in exact fp32, intermediate in LDS:
 *   v = x + conv1d(lrelu(conv1d(lrelu(x, slope), w1, dilation dil) + b1, slope), w2) + b2;  y = v | y + v | (y + v) / div  (mode 0 | 1 | 2)
 * x, y [B][C][L] (y != x), w1 / w2 [C][C][K] (torch layout, DEVICE), lens int64 [B] or NULL (valid length lens[b] * len_mul:
 * every conv pads at the true edge).  variant 1: first-generation kernel (weights through an LDS ring; C = 32, or C = 64 with
 * k = 3), 2 / 3: second generation (weights from L2 into a register queue, raw window; 256- / 128-column blocks; C = 32 / 64 /
 * 128, k = 3 / 7 / 11), 4 / 5: 256-column blocks with conv 2 (4) or both convs (5) on Winograd F(2,3) (C = 32 / 64, k = 3 / 7 / 11;
 * 5 also C = 128).  `packed` is scratch for the re-laid-out weights: ttsamd_resblock_pair_packed_floats(C, K, variant) floats (the two
 * direct packings, + the Winograd group filters of variants 4 / 5); `packed_floats` = what the caller allocated, a smaller buffer is
 * TTSAMD_EINVAL (nothing is written). */
int64_t ttsamd_resblock_pair_packed_floats(int32_t channels, int32_t k, int32_t variant);
int32_t ttsamd_resblock_pair(const float* x, float* y, const float* w1, const float* b1, const float* w2, const float* b2,
                             int32_t channels, int32_t k, int32_t dil, const int64_t* lens, int32_t len_mul, int32_t L,
                             int32_t batch, int32_t mode, float div, float slope, int32_t variant, float* packed, int64_t packed_floats,
                             void* stream);

/* ---- bf16 octet engine (BASELINE config 3), kernel-level entries used by the parity tests and the roofline bench.
 *      Activations are [B][C/8][L][8] bf16 ("octet" layout: one 16-byte entry = 8 channels of one position = one lane's B
 *      operand of v_mfma_f32_32x32x16_bf16), stored PRE-ACTIVATED: a = leaky_relu(x, slope of the consumer).  Replaces
 *      ResBlock1.forward / Generator.forward's convs (vocoder/hifigan/models.py:46-53, 111-127) when
 *      ttsamd_set_precision(1) is in force; ttsamd_hifigan_forward routes through it by itself. ------------------------- */
/* fp32 channel-first [B][C][len] -> octet bf16 with leaky_relu(slope) applied (slope 1 = raw); and back (inverse applied) */
int32_t ttsamd_bfo_pack(const float* x, int32_t batch, int32_t channels, int32_t len, float slope, void* out, void* stream);
int32_t ttsamd_bfo_unpack(const void* in, int32_t batch, int32_t channels, int32_t len, float slope, float* out, void* stream);
/* HOST: torch Conv1d weight [Cout][Cin][K] (up = 1) or ConvTranspose1d weight [Cin][Cout][2*up] (stride up, padding up/2)
 * -> bf16 [phases][Cin/16][K][2][CoutP][8]; `out` holds ttsamd_bfo_weight_elems(...) uint16 */
int64_t ttsamd_bfo_weight_elems(int32_t cout, int32_t cin, int32_t k, int32_t up);
int32_t ttsamd_bfo_pack_weight(const float* w, int32_t cout, int32_t cin, int32_t k, int32_t up, uint16_t* out);
/* y = act_out(([sum_in +] conv(x) + bias [+ raw(res)]) [/ div]): x, res activated tensors, sum_in raw; mode as below;
 * out_slope 0 = ReLU.  up > 1: ConvTranspose1d(stride up), y has len_in * up positions (no res / sum).
 * y_f32 != NULL: the result leaves as fp32 channel-first [B][Cout][len] instead of y, res_f32 (NULL = none) is an fp32
 * channel-first residual: FastPitch's Conv1d + ReLU -> Conv1d + residual block (models/fastpitch/fastpitch/transformer.py:72-90)
 * keeps its residual stream in fp32 and only the 1536-channel intermediate in bf16. */
int32_t ttsamd_bfo_conv1d(const void* x, const void* w_packed, const float* bias, const void* res, const void* sum_in,
                          const int64_t* lens, int32_t len_mul, int32_t batch, int32_t cin, int32_t cout, int32_t k,
                          int32_t dilation, int32_t up, int32_t len_in, int32_t mode, float div, float res_slope,
                          float out_slope, void* y, float* y_f32, const float* res_f32, void* stream);
/* one c1 -> c2 pair of ResBlock1 (models.py:46-53) in one launch, C in {32, 64, 128}, k in {3, 7, 11}:
 *   v = raw(x) + conv(lrelu(conv(x, w1, dilation) + b1, mid_slope), w2) + b2
 *   mode 0: y = act(v)   1: y = act(sum_in + v)   2: y = act((sum_in + v) / div);   act = leaky_relu(out_slope), 1 = raw.
 * x is stored activated with in_slope; y must not alias x. */
int32_t ttsamd_bfo_resblock_pair(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                 const void* sum_in, const int64_t* lens, int32_t len_mul, int32_t batch, int32_t channels,
                                 int32_t k, int32_t dilation, int32_t len, int32_t mode, float div, float in_slope,
                                 float mid_slope, float out_slope, void* y, void* stream);
/* a whole k = 3 ResBlock1 (the three pairs of models.py:46-53 with dilations[0..2], C in {32, 64, 128}) in one launch; w1 / b1 / w2 /
 * b2 are arrays of three device pointers.  Equals three ttsamd_bfo_resblock_pair calls (out_slope = in_slope between them, mode /
 * sum_in / out_slope on the last) bit for bit. */
int32_t ttsamd_bfo_resblock_chain(const void* x, const void* const* w1, const float* const* b1, const void* const* w2,
                                  const float* const* b2, const int32_t* dilations, const void* sum_in, const int64_t* lens,
                                  int32_t len_mul, int32_t batch, int32_t channels, int32_t len, int32_t mode, float div,
                                  float in_slope, float mid_slope, float out_slope, void* y, void* stream, int32_t k);
/* wave[b][t] = tanh(bias + conv7(x)); x = 32-channel octet tensor already activated with slope 0.01 (models.py:123-125) */
int32_t ttsamd_bfo_conv_post(const void* x, const float* w, const float* bias, const int64_t* lens, int32_t len_mul,
                             int32_t batch, int32_t channels, int32_t len, float* wave, int64_t wave_stride, void* stream);

/* ---- split-bf16 ("x3") mode of the octet engine: the same layers with fp32-class results (north_star's 1e-3 / 1e-4), every
 *      value = hi + lo (two bf16), three v_mfma_f32_32x32x16_bf16 per product (Wh xh + Wh xl + Wl xh, fp32 accumulate).
 *      Activations ("x3 tensor"): [B][C/8][L][2][hi 4 bf16 | lo 4 bf16] = 32 bytes per (octet, position) -- the 16-byte half kk
 *      holds channels 8o + 4kk + {0..3}, one lane's slice of the MFMA C layout -- stored PRE-ACTIVATED like the bf16 tensors.
 *      Weights: [phases][Cin/16][K][2][CoutP][hi 8 | lo 8] bf16.  Same argument meanings as the ttsamd_bfo_* twins above;
 *      ttsamd_hifigan_forward / ttsamd_fastpitch_* route through these kernels under ttsamd_set_precision(2).
 *      Ops: vocoder/hifigan/models.py:46-53, 96-99, 111-127. ------------------------------------------------------------------- */
int32_t ttsamd_bfo3_pack(const float* x, int32_t batch, int32_t channels, int32_t len, float slope, void* out, void* stream);
int32_t ttsamd_bfo3_unpack(const void* in, int32_t batch, int32_t channels, int32_t len, float slope, float* out, void* stream);
int64_t ttsamd_bfo3_weight_elems(int32_t cout, int32_t cin, int32_t k, int32_t up);
int32_t ttsamd_bfo3_pack_weight(const float* w, int32_t cout, int32_t cin, int32_t k, int32_t up, uint16_t* out);
int32_t ttsamd_bfo3_conv1d(const void* x, const void* w_packed, const float* bias, const void* res, const void* sum_in,
                           const int64_t* lens, int32_t len_mul, int32_t batch, int32_t cin, int32_t cout, int32_t k,
                           int32_t dilation, int32_t up, int32_t len_in, int32_t mode, float div, float res_slope,
                           float out_slope, void* y, float* y_f32, const float* res_f32, void* stream);
int32_t ttsamd_bfo3_resblock_pair(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                  const void* sum_in, const int64_t* lens, int32_t len_mul, int32_t batch, int32_t channels,
                                  int32_t k, int32_t dilation, int32_t len, int32_t mode, float div, float in_slope,
                                  float mid_slope, float out_slope, void* y, void* stream);
/* a whole k = 3 ResBlock1 (three pairs, dilations[0..2]) in one launch; equals three ttsamd_bfo3_resblock_pair calls bit for bit */
int32_t ttsamd_bfo3_resblock_chain(const void* x, const void* const* w1, const float* const* b1, const void* const* w2,
                                   const float* const* b2, const int32_t* dilations, const void* sum_in, const int64_t* lens,
                                   int32_t len_mul, int32_t batch, int32_t channels, int32_t len, int32_t mode, float div,
                                   float in_slope, float mid_slope, float out_slope, void* y, void* stream);
int32_t ttsamd_bfo3_conv_post(const void* x, const float* w, const float* bias, const int64_t* lens, int32_t len_mul,
                              int32_t batch, int32_t channels, int32_t len, float* wave, int64_t wave_stride, void* stream);

/* MFMA operand precision of every conv/linear GEMM launched by the model forwards (process-wide):
 *   0 (default) exact fp32 (v_mfma_f32_32x32x2_f32) — BASELINE config 2;
 *   1 bf16 operands, fp32 accumulate — config 3: HiFi-GAN on the octet engine above (v_mfma_f32_32x32x16_bf16, bf16
 *     activations in HBM), the other models on v_mfma_f32_32x32x8_bf16_1k with fp32 activations;
 *   2 split bf16 (x = hi + lo, 3 MFMAs per product): fp32-class accuracy at bf16 MFMA rate.
 * Activations, LayerNorm, softmax, tanh, the FFTs and all integer work stay fp32/int64. */
int32_t ttsamd_set_precision(int32_t precision);
int32_t ttsamd_get_precision(void);

/* ---- data-parallel sharding over the GPUs of one node: RCCL over xGMI.
 *      No reference counterpart: the reference is single-device (inference.py:23-24); utterances are
 *      independent, so the only exchanges are C1 (weights, once) and C2 (lengths + audio fan-in per call).
 *      One communicator per process (= per GPU); librccl.so.1 is bound at run time by the first call.
 *      Host binding (ctypes / cgo / JNI alike): rank 0 calls ttsamd_dp_unique_id and ships the 128 bytes
 *      to the other ranks out of band (env, file, torch.distributed store); every rank then calls
 *      ttsamd_dp_init after hipSetDevice(local_rank). ------------------------------------------------- */
#define TTSAMD_DP_ID_BYTES 128
#define TTSAMD_DP_HIFIGAN 0
#define TTSAMD_DP_FASTPITCH 1
int32_t ttsamd_dp_unique_id(void* id128 /* host, out */);
int32_t ttsamd_dp_init(int32_t rank, int32_t world, const void* id128 /* host */, void** comm);
int32_t ttsamd_dp_destroy(void* comm);
int32_t ttsamd_dp_rank(void* comm);
int32_t ttsamd_dp_world(void* comm);
/* C1, generic: in-place broadcast of a device buffer from `root`. */
int32_t ttsamd_dp_broadcast(void* comm, void* buf, int64_t nbytes, int32_t root, void* stream);
/* C1, handle form: overwrites this rank's packed weight blobs (fp32 + bf16 planes) of a
 * ttsamd_hifigan / ttsamd_fastpitch handle (kind = TTSAMD_DP_*) with the root's.  Non-root ranks create
 * their handle from tensors of the same names and shapes (any values): only rank 0 reads the checkpoint
 * (replaces one torch.load + remove_weight_norm per GPU; vocoder/__init__.py:15-18). */
int32_t ttsamd_dp_broadcast_weights(void* comm, int32_t kind, void* handle, int32_t root, void* stream);
/* C2a: recv[r*nbytes .. (r+1)*nbytes) = rank r's send[0 .. nbytes) on every rank (the lengths). */
int32_t ttsamd_dp_allgather(void* comm, const void* send, void* recv, int64_t nbytes_per_rank, void* stream);
/* C2b: packed[off(b) + t] = wave[b][t] for t < min(nsamples[b], n_max), off(b) = sum of the shorter-indexed
 * utterances' sample counts: the valid samples of a padded ragged batch back to back (no padding crosses
 * xGMI or PCIe).  wave [B][wave_stride], nsamples int64 [B] (device).  Needs no communicator. */
int32_t ttsamd_dp_pack_audio(const float* wave, int64_t wave_stride, const int64_t* nsamples, int32_t batch,
                             int64_t n_max, float* packed, void* stream);
/* C2c: fan-in to `root`: rank r's packed[0 .. counts[r]) lands at recv[offsets[r] ..) on the root
 * (grouped ncclSend / ncclRecv: world-1 independent point-to-point transfers, one per xGMI link).
 * counts / offsets are HOST int64 [world] (floats), identical on every rank — the caller knows them from
 * the all-gathered lengths; recv / offsets are ignored on the other ranks. */
int32_t ttsamd_dp_gather_audio(void* comm, const float* packed, float* recv, const int64_t* counts,
                               const int64_t* offsets, int32_t root, void* stream);

/* Timing hooks for bench.py (roofline of the dominant kernel): when enabled, hifigan
 * forward brackets its ResBlock conv launches with HIP events on the launch stream. */
int32_t ttsamd_profile_enable(int32_t on);
/* Fills: [0] = conv-kernel ms (length of the union of the timed sections' intervals), [1] = number of conv launches,
 * [2] = number of timed sections (event pairs: a single launch, or one fork..join group of concurrent launches) */
int32_t ttsamd_profile_read(double* out3);

#ifdef __cplusplus
}
#endif
#endif /* TTSAMD_H */
