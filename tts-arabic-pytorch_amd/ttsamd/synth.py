"""Deterministic synthetic weights / inputs (no pretrained weights exist in the
reference tree: pretrained/ holds only hifigan-asc-v1/config.json, download_files.py:7-53
needs the network).  A counter-based generator keyed by (seed, crc32(tensor name)) so the
same tensors are reproduced bit-identically in the build container (where they are loaded
into the *reference* modules to make the golden fixtures) and on the GPU box.

Keys follow the reference checkpoints exactly:
  FastPitch  {'model': state_dict, 'config': net_config, 'symbols': [...]}
             (models/fastpitch/networks.py:52-71)
  HiFi-GAN   {'generator': weight-normed state_dict}            (vocoder/__init__.py:15-16)
"""
import zlib

import numpy as np

from .config import NET_CONFIG, HIFIGAN_CONFIG, VOCOS_22K_CONFIG, TACOTRON2_CONFIG, SHAKKELHA_CONFIG, SHAKKALA_CONFIG


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def _normal(seed, name, shape, std):
    return (_rng(seed, name).standard_normal(shape) * std).astype(np.float32)


def _fft_stack(sd, prefix, n_layers, d_model, d_head, n_head, d_inner, k, seed):
    for i in range(n_layers):
        p = f'{prefix}.layers.{i}.'
        sd[p + 'dec_attn.qkv_net.weight'] = _normal(seed, p + 'qkv.w', (3 * n_head * d_head, d_model), 1.2 / np.sqrt(d_model))
        sd[p + 'dec_attn.qkv_net.bias'] = _normal(seed, p + 'qkv.b', (3 * n_head * d_head,), 0.1)
        sd[p + 'dec_attn.o_net.weight'] = _normal(seed, p + 'o.w', (d_model, n_head * d_head), 1.0 / np.sqrt(d_head))
        sd[p + 'dec_attn.layer_norm.weight'] = 1.0 + _normal(seed, p + 'ln1.w', (d_model,), 0.1)
        sd[p + 'dec_attn.layer_norm.bias'] = _normal(seed, p + 'ln1.b', (d_model,), 0.1)
        sd[p + 'pos_ff.CoreNet.0.weight'] = _normal(seed, p + 'ff0.w', (d_inner, d_model, k), 1.0 / np.sqrt(d_model * k))
        sd[p + 'pos_ff.CoreNet.0.bias'] = _normal(seed, p + 'ff0.b', (d_inner,), 0.1)
        sd[p + 'pos_ff.CoreNet.2.weight'] = _normal(seed, p + 'ff2.w', (d_model, d_inner, k), 1.0 / np.sqrt(d_inner * k))
        sd[p + 'pos_ff.CoreNet.2.bias'] = _normal(seed, p + 'ff2.b', (d_model,), 0.1)
        sd[p + 'pos_ff.layer_norm.weight'] = 1.0 + _normal(seed, p + 'ln2.w', (d_model,), 0.1)
        sd[p + 'pos_ff.layer_norm.bias'] = _normal(seed, p + 'ln2.b', (d_model,), 0.1)


def _predictor(sd, prefix, d_in, d_f, k, n_layers, seed, fc_bias):
    for i in range(n_layers):
        p = f'{prefix}.layers.{i}.'
        cin = d_in if i == 0 else d_f
        sd[p + 'conv.weight'] = _normal(seed, p + 'conv.w', (d_f, cin, k), 1.4 / np.sqrt(cin * k))
        sd[p + 'conv.bias'] = _normal(seed, p + 'conv.b', (d_f,), 0.1)
        sd[p + 'norm.weight'] = 1.0 + _normal(seed, p + 'norm.w', (d_f,), 0.1)
        sd[p + 'norm.bias'] = _normal(seed, p + 'norm.b', (d_f,), 0.1)
    sd[prefix + '.fc.weight'] = _normal(seed, prefix + '.fc.w', (1, d_f), 0.3 / np.sqrt(d_f))
    sd[prefix + '.fc.bias'] = np.full((1,), fc_bias, np.float32)


def fastpitch_state_dict(config=None, seed=0):
    """name -> np.float32 array, same keys/shapes as reference FastPitch.state_dict()
    minus the training-only `attention.*` sub-module (models/fastpitch/fastpitch/model.py:234)."""
    c = dict(NET_CONFIG if config is None else config)
    d = c['symbols_embedding_dim']
    sd = {}
    sd['pitch_mean'] = np.zeros((1,), np.float32)
    sd['pitch_std'] = np.zeros((1,), np.float32)
    emb = _normal(seed, 'encoder.word_emb', (c['n_symbols'], d), 1.0)
    emb[c['padding_idx']] = 0.0
    sd['encoder.word_emb.weight'] = emb
    inv_freq = _inv_freq(d)
    sd['encoder.pos_emb.inv_freq'] = inv_freq
    _fft_stack(sd, 'encoder', c['in_fft_n_layers'], d, c['in_fft_d_head'], c['in_fft_n_heads'],
               c['in_fft_conv1d_filter_size'], c['in_fft_conv1d_kernel_size'], seed)
    _predictor(sd, 'duration_predictor', c['in_fft_output_size'], c['dur_predictor_filter_size'],
               c['dur_predictor_kernel_size'], c['dur_predictor_n_layers'], seed, float(np.log(8.0)))
    sd['decoder.pos_emb.inv_freq'] = inv_freq.copy()
    _fft_stack(sd, 'decoder', c['out_fft_n_layers'], d, c['out_fft_d_head'], c['out_fft_n_heads'],
               c['out_fft_conv1d_filter_size'], c['out_fft_conv1d_kernel_size'], seed)
    _predictor(sd, 'pitch_predictor', c['in_fft_output_size'], c['pitch_predictor_filter_size'],
               c['pitch_predictor_kernel_size'], c['pitch_predictor_n_layers'], seed, 0.0)
    kp = c['pitch_embedding_kernel_size']
    sd['pitch_emb.weight'] = _normal(seed, 'pitch_emb.w', (d, 1, kp), 0.3)
    sd['pitch_emb.bias'] = _normal(seed, 'pitch_emb.b', (d,), 0.1)
    if c['energy_conditioning']:
        _predictor(sd, 'energy_predictor', c['in_fft_output_size'], c['energy_predictor_filter_size'],
                   c['energy_predictor_kernel_size'], c['energy_predictor_n_layers'], seed, 0.0)
        ke = c['energy_embedding_kernel_size']
        sd['energy_emb.weight'] = _normal(seed, 'energy_emb.w', (d, 1, ke), 0.3)
        sd['energy_emb.bias'] = _normal(seed, 'energy_emb.b', (d,), 0.1)
    sd['proj.weight'] = _normal(seed, 'proj.w', (c['n_mel_channels'], c['out_fft_output_size']), 1.0 / np.sqrt(d))
    sd['proj.bias'] = _normal(seed, 'proj.b', (c['n_mel_channels'],), 0.5)
    if c['n_speakers'] > 1:
        sd['speaker_emb.weight'] = _normal(seed, 'speaker_emb', (c['n_speakers'], d), 0.5)
    return sd


def _inv_freq(demb):
    # models/fastpitch/fastpitch/transformer.py:37 — evaluated with torch so the buffer is
    # bit-identical to what the reference module registers.
    import torch
    return (1 / (10000 ** (torch.arange(0.0, demb, 2.0) / demb))).numpy().astype(np.float32)


def hifigan_state_dict(config=None, seed=0, weight_norm=True):
    """Weight-normalised generator checkpoint (keys as torch>=2.1 parametrizations:
    `<layer>.parametrizations.weight.original0` = g, `original1` = v; §3.4-6) or, with
    weight_norm=False, the folded `<layer>.weight` form."""
    h = dict(HIFIGAN_CONFIG if config is None else config)
    sd = {}

    def put(name, shape, fan_in, gain):
        v = _normal(seed, name + '.v', shape, 1.0)
        target = gain / np.sqrt(fan_in)                       # per-element std of folded w
        nv = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
        n_per = np.prod(shape[1:])
        g = (target * np.sqrt(n_per) * (1.0 + 0.1 * _rng(seed, name + '.g').standard_normal(nv.shape))).astype(np.float32)
        if weight_norm:
            sd[name + '.parametrizations.weight.original0'] = g
            sd[name + '.parametrizations.weight.original1'] = v
        else:
            sd[name + '.weight'] = (g.astype(np.float64) * v / nv).astype(np.float32)
        return g, v

    c0 = h['upsample_initial_channel']
    put('conv_pre', (c0, h['num_mels'], 7), h['num_mels'] * 7, 0.25)
    sd['conv_pre.bias'] = _normal(seed, 'conv_pre.b', (c0,), 0.1)
    ch = c0
    for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
        cin, ch = c0 // (2 ** i), c0 // (2 ** (i + 1))
        # ConvTranspose1d weight is [Cin, Cout, k]; each output sample sees Cin*k/u taps
        put(f'ups.{i}', (cin, ch, k), cin * k / u, 1.1)
        sd[f'ups.{i}.bias'] = _normal(seed, f'ups.{i}.b', (ch,), 0.1)
        for j, kk in enumerate(h['resblock_kernel_sizes']):
            r = i * len(h['resblock_kernel_sizes']) + j
            for m in range(len(h['resblock_dilation_sizes'][j])):
                put(f'resblocks.{r}.convs1.{m}', (ch, ch, kk), ch * kk, 1.2)
                sd[f'resblocks.{r}.convs1.{m}.bias'] = _normal(seed, f'rb{r}.c1.{m}.b', (ch,), 0.05)
                put(f'resblocks.{r}.convs2.{m}', (ch, ch, kk), ch * kk, 0.6)
                sd[f'resblocks.{r}.convs2.{m}.bias'] = _normal(seed, f'rb{r}.c2.{m}.b', (ch,), 0.05)
    put('conv_post', (1, ch, 7), ch * 7, 0.4)
    sd['conv_post.bias'] = _normal(seed, 'conv_post.b', (1,), 0.05)
    return sd


def vocos_state_dict(config=None, seed=0):
    """MelVocos('22k') backbone + head (vocoder/vocos/pretrained.py:34-45; keys as its state_dict).
    Scaled so that the log-magnitudes stay O(1) and the waveform is O(0.1)."""
    c = dict(VOCOS_22K_CONFIG if config is None else config)
    d, di, cin, nl = c['dim'], c['intermediate_dim'], c['input_channels'], c['num_layers']
    sd = {}
    sd['backbone.embed.weight'] = _normal(seed, 'vocos.embed.w', (d, cin, 7), 0.25 / np.sqrt(cin * 7))
    sd['backbone.embed.bias'] = _normal(seed, 'vocos.embed.b', (d,), 0.1)
    sd['backbone.norm.weight'] = 1.0 + _normal(seed, 'vocos.norm.w', (d,), 0.1)
    sd['backbone.norm.bias'] = _normal(seed, 'vocos.norm.b', (d,), 0.1)
    for i in range(nl):
        p = f'backbone.convnext.{i}.'
        sd[p + 'gamma'] = (0.5 + 0.1 * _rng(seed, p + 'gamma').standard_normal(d)).astype(np.float32)
        sd[p + 'dwconv.weight'] = _normal(seed, p + 'dw.w', (d, 1, 7), 1.0 / np.sqrt(7))
        sd[p + 'dwconv.bias'] = _normal(seed, p + 'dw.b', (d,), 0.1)
        sd[p + 'norm.weight'] = 1.0 + _normal(seed, p + 'norm.w', (d,), 0.1)
        sd[p + 'norm.bias'] = _normal(seed, p + 'norm.b', (d,), 0.1)
        sd[p + 'pwconv1.weight'] = _normal(seed, p + 'pw1.w', (di, d), 1.2 / np.sqrt(d))
        sd[p + 'pwconv1.bias'] = _normal(seed, p + 'pw1.b', (di,), 0.1)
        sd[p + 'pwconv2.weight'] = _normal(seed, p + 'pw2.w', (d, di), 1.0 / np.sqrt(di))
        sd[p + 'pwconv2.bias'] = _normal(seed, p + 'pw2.b', (d,), 0.1)
    sd['backbone.final_layer_norm.weight'] = 1.0 + _normal(seed, 'vocos.fln.w', (d,), 0.1)
    sd['backbone.final_layer_norm.bias'] = _normal(seed, 'vocos.fln.b', (d,), 0.1)
    nout = c['n_fft'] + 2
    w = _normal(seed, 'vocos.head.w', (nout, d), 1.0 / np.sqrt(d))
    w[:nout // 2] *= 0.6                                   # log-magnitude rows
    w[nout // 2:] *= 2.0                                   # phase rows
    sd['head.out.weight'] = w
    b = _normal(seed, 'vocos.head.b', (nout,), 0.3)
    b[:nout // 2] -= 0.3
    sd['head.out.bias'] = b
    return sd


def _bn(sd, seed, name, n):
    sd[name + '.weight'] = 1.0 + _normal(seed, name + '.w', (n,), 0.1)
    sd[name + '.bias'] = _normal(seed, name + '.b', (n,), 0.1)
    sd[name + '.running_mean'] = _normal(seed, name + '.rm', (n,), 0.1)
    sd[name + '.running_var'] = (1.0 + 0.2 * _rng(seed, name + '.rv').random(n)).astype(np.float32)


def _lstm(sd, seed, name, n_in, n_h, suffixes=('',), gain=1.0):
    k = gain / np.sqrt(n_h)
    for sfx in suffixes:
        sd[f'{name}.weight_ih{sfx}'] = ((_rng(seed, f'{name}.wih{sfx}').random((4 * n_h, n_in)) * 2 - 1) * k).astype(np.float32)
        sd[f'{name}.weight_hh{sfx}'] = ((_rng(seed, f'{name}.whh{sfx}').random((4 * n_h, n_h)) * 2 - 1) * k).astype(np.float32)
        sd[f'{name}.bias_ih{sfx}'] = ((_rng(seed, f'{name}.bih{sfx}').random(4 * n_h) * 2 - 1) * k).astype(np.float32)
        sd[f'{name}.bias_hh{sfx}'] = ((_rng(seed, f'{name}.bhh{sfx}').random(4 * n_h) * 2 - 1) * k).astype(np.float32)


def tacotron2_state_dict(config=None, seed=0, gate_bias=-2.0):
    """Tacotron2MS state_dict (models/tacotron2/tacotron2_ms.py:152-212) with the parameter names of
    torchaudio.models.tacotron2's private _Encoder/_Decoder/_Postnet as publicly documented
    (embedding, encoder.convolutions.i.{0,1}, encoder.lstm.*_l0[_reverse], decoder.prenet.layers.i,
    decoder.attention_rnn, decoder.attention_layer.{query_layer,memory_layer,v,location_layer.*},
    decoder.decoder_rnn, decoder.linear_projection, decoder.gate_layer, postnet.convolutions.i.{0,1},
    speaker_embedding).  torchaudio is absent here, so these names are NOT verified against it."""
    c = dict(TACOTRON2_CONFIG if config is None else config)
    sd = {}
    E, S = c['encoder_embedding_dim'], c['speaker_embedding_dim']
    M = E + (S if c['num_speakers'] > 1 else 0)
    sd['embedding.weight'] = _normal(seed, 'taco.emb', (c['n_symbol'], c['symbol_embedding_dim']), 0.5)
    if c['num_speakers'] > 1:
        sd['speaker_embedding.weight'] = _normal(seed, 'taco.spk', (c['num_speakers'], S), 0.5)
    k = c['encoder_kernel_size']
    for i in range(c['encoder_n_convolution']):
        sd[f'encoder.convolutions.{i}.0.weight'] = _normal(seed, f'taco.enc.conv{i}.w', (E, E, k), 1.4 / np.sqrt(E * k))
        sd[f'encoder.convolutions.{i}.0.bias'] = _normal(seed, f'taco.enc.conv{i}.b', (E,), 0.1)
        _bn(sd, seed, f'encoder.convolutions.{i}.1', E)
    _lstm(sd, seed, 'encoder.lstm', E, E // 2, ('_l0', '_l0_reverse'))
    P, A, D, H = c['prenet_dim'], c['attention_rnn_dim'], c['decoder_rnn_dim'], c['attention_hidden_dim']
    sd['decoder.prenet.layers.0.weight'] = _normal(seed, 'taco.pre0', (P, c['n_mels']), 1.4 / np.sqrt(c['n_mels']))
    sd['decoder.prenet.layers.1.weight'] = _normal(seed, 'taco.pre1', (P, P), 1.4 / np.sqrt(P))
    _lstm(sd, seed, 'decoder.attention_rnn', P + M, A)
    sd['decoder.attention_layer.query_layer.weight'] = _normal(seed, 'taco.att.q', (H, A), 1.0 / np.sqrt(A))
    sd['decoder.attention_layer.memory_layer.weight'] = _normal(seed, 'taco.att.m', (H, M), 1.0 / np.sqrt(M))
    sd['decoder.attention_layer.v.weight'] = _normal(seed, 'taco.att.v', (1, H), 3.0 / np.sqrt(H))
    nf, ks = c['attention_location_n_filter'], c['attention_location_kernel_size']
    sd['decoder.attention_layer.location_layer.location_conv.weight'] = _normal(seed, 'taco.att.lc', (nf, 2, ks), 1.0)
    sd['decoder.attention_layer.location_layer.location_dense.weight'] = _normal(seed, 'taco.att.ld', (H, nf), 1.0 / np.sqrt(nf))
    _lstm(sd, seed, 'decoder.decoder_rnn', A + M, D)
    sd['decoder.linear_projection.weight'] = _normal(seed, 'taco.proj.w', (c['n_mels'], D + M), 2.0 / np.sqrt(D + M))
    sd['decoder.linear_projection.bias'] = _normal(seed, 'taco.proj.b', (c['n_mels'],), 0.5)
    sd['decoder.gate_layer.weight'] = _normal(seed, 'taco.gate.w', (1, D + M), 6.0 / np.sqrt(D + M))
    sd['decoder.gate_layer.bias'] = np.full((1,), gate_bias, np.float32)
    PE, pk, n = c['postnet_embedding_dim'], c['postnet_kernel_size'], c['postnet_n_convolution']
    for i in range(n):
        cin = c['n_mels'] if i == 0 else PE
        cout = c['n_mels'] if i == n - 1 else PE
        sd[f'postnet.convolutions.{i}.0.weight'] = _normal(seed, f'taco.post{i}.w', (cout, cin, pk), 1.0 / np.sqrt(cin * pk))
        sd[f'postnet.convolutions.{i}.0.bias'] = _normal(seed, f'taco.post{i}.b', (cout,), 0.1)
        _bn(sd, seed, f'postnet.convolutions.{i}.1', cout)
    return sd


def _dense(sd, seed, name, n_in, n_out, gain=1.4):
    sd[name + '.weight'] = _normal(seed, name + '.w', (n_out, n_in), gain / np.sqrt(n_in))
    sd[name + '.bias'] = _normal(seed, name + '.b', (n_out,), 0.1)


TAGGER_LSTM_GAIN, TAGGER_OUT_GAIN = 4.0, 8.0   # make the synthetic taggers input-sensitive (varied classes)


def shakkelha_state_dict(seed=0):
    """Shakkelha.state_dict() (models/diacritizers/shakkelha/network.py:14-24): emb0, lstm0, lstm1 (nn.LSTM,
    bidirectional), dense0-2.  LSTM weights are scaled up a little so the synthetic tagger is not constant."""
    c, sd = SHAKKELHA_CONFIG, {}
    sd['emb0.weight'] = _normal(seed, 'shakkelha.emb', (c['n_vocab'], c['emb_dim']), 1.0)
    n_in = c['emb_dim']
    for i, h in enumerate(c['lstm_hidden']):
        _lstm(sd, seed, f'lstm{i}', n_in, h, ('_l0', '_l0_reverse'), gain=TAGGER_LSTM_GAIN)
        n_in = 2 * h
    for i, d in enumerate(c['dense_dim']):
        _dense(sd, seed, f'dense{i}', n_in, d, 1.4 if i < len(c['dense_dim']) - 1 else TAGGER_OUT_GAIN)
        n_in = d
    return sd


def shakkala_state_dict(seed=0):
    """Shakkala.state_dict() (models/diacritizers/shakkala/network.py:13-22): emb_input, lstm0-2
    (LSTMHardSigmoid, bidirectional), bn0 (BatchNorm1d eps 1e-3), dense0."""
    c, sd = SHAKKALA_CONFIG, {}
    sd['emb_input.weight'] = _normal(seed, 'shakkala.emb', (c['n_vocab'], c['emb_dim']), 1.0)
    n_in = c['emb_dim']
    for i, h in enumerate(c['lstm_hidden']):
        _lstm(sd, seed, f'lstm{i}', n_in, h, ('_l0', '_l0_reverse'), gain=TAGGER_LSTM_GAIN)
        n_in = 2 * h
    _bn(sd, seed, 'bn0', 2 * c['lstm_hidden'][0])
    sd['bn0.num_batches_tracked'] = np.zeros((), np.int64)
    _dense(sd, seed, 'dense0', n_in, c['dense_dim'][0], TAGGER_OUT_GAIN)
    return sd


def synth_ids(batch, n_tokens, seed=1234):
    """ids = 1 + rng mod 39 -> int64 [B, L] in [1, 39] (never padding_idx 0); SURVEY §8(d)."""
    r = np.random.default_rng([int(seed), 1])
    return (1 + r.integers(0, 39, size=(batch, n_tokens))).astype(np.int64)


def synth_durations(batch, n_tokens, seed=1234):
    """dur_tgt in [2, 12], mean 7 frames/token -> E[T_i] = 448 at 64 tokens; SURVEY §8(d)."""
    r = np.random.default_rng([int(seed), 2])
    return (2 + r.integers(0, 11, size=(batch, n_tokens))).astype(np.float32)
