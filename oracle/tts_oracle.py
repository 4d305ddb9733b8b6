"""CPU oracle for the FastPitch -> HiFi-GAN hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module; the product path (tts-arabic-pytorch_amd/) never does and fails loudly when the
HIP library is missing.

What it is: an own-code restatement of the reference's algorithm for the path, written
against `torch.nn.functional` on CPU tensors (the reference itself is nothing but a
sequence of stock ATen ops, SURVEY.md §2.2, so the ATen op *is* the arithmetic being
restated; the integer part — the length regulator — is additionally restated in pure
numpy integer arithmetic, `regulate_len_indices`).  Weights come in as a plain
{name: array} dict with the reference's checkpoint keys.  Every function cites the
reference file:line it follows (paths relative to the reference root).

Pinning: the reference has no tests / golden vectors, so the oracle is
pinned against outputs of the reference itself, run in the build container by
oracle/gen_golden.py (real reference modules + the same synthetic weights) and committed
as tests/golden/*.npz; tests/test_oracle_golden.py checks this file against them.
The denoiser part is pinned only against a torch.stft-based stand-in for torchaudio
("parity unpinned at the torchaudio boundary").
"""
import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1          # vocoder/hifigan/models.py:11


def _t(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.from_numpy(np.ascontiguousarray(x)).to(dtype)


def to_torch(sd, dtype=torch.float32):
    return {k: _t(v, dtype) for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# HiFi-GAN V1 generator
# --------------------------------------------------------------------------------------

def fold_weight_norm(sd):
    """vocoder/__init__.py:16-19 + vocoder/hifigan/models.py:129-136: load the normed
    state dict, then remove_parametrizations -> w = g * v / ||v|| with the norm over all
    dims but 0 (for ConvTranspose1d dim 0 is the in-channel axis).
    Accepts `parametrizations.weight.original0/1` and legacy `weight_g/weight_v` keys."""
    out = {}
    for k, v in sd.items():
        if k.endswith('.parametrizations.weight.original0') or k.endswith('.weight_g'):
            continue
        if k.endswith('.parametrizations.weight.original1') or k.endswith('.weight_v'):
            if k.endswith('.weight_v'):
                base, gk = k[:-len('.weight_v')], k[:-len('.weight_v')] + '.weight_g'
            else:
                base = k[:-len('.parametrizations.weight.original1')]
                gk = base + '.parametrizations.weight.original0'
            vv = _t(v, torch.float32)
            g = _t(sd[gk], torch.float32)
            out[base + '.weight'] = torch._weight_norm(vv, g, 0)
        else:
            out[k] = _t(v, torch.float32)
    return out


def _get_padding(k, d=1):
    return int((k * d - d) / 2)          # vocoder/hifigan/models.py:18-19


def hifigan_forward(w, mel, cfg, dtype=torch.float32, stages=None):
    """vocoder/hifigan/models.py:111-127 (Generator.forward) with ResBlock1 (:46-53).
    `w`: folded weights ({'conv_pre.weight', ...}); `mel`: [80,T] or [B,80,T].
    Returns wave [1, 256*T] (2-D in, as the reference's unbatched Conv1d does) or
    [B,1,256*T].  If `stages` is a list, stage outputs are appended to it."""
    W = {k: v.to(dtype) for k, v in w.items()}
    x = _t(mel, dtype)
    unb = x.dim() == 2
    if unb:
        x = x[None]
    nk = len(cfg['resblock_kernel_sizes'])
    x = F.conv1d(x, W['conv_pre.weight'], W['conv_pre.bias'], padding=3)
    if stages is not None:
        stages.append(x)
    for i, (u, k) in enumerate(zip(cfg['upsample_rates'], cfg['upsample_kernel_sizes'])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, W[f'ups.{i}.weight'], W[f'ups.{i}.bias'], stride=u,
                               padding=(k - u) // 2)
        if stages is not None:
            stages.append(x)
        xs = None
        for j, (kk, dil) in enumerate(zip(cfg['resblock_kernel_sizes'],
                                          cfg['resblock_dilation_sizes'])):
            r = i * nk + j
            y = x
            for m, d in enumerate(dil):
                xt = F.leaky_relu(y, LRELU_SLOPE)
                xt = F.conv1d(xt, W[f'resblocks.{r}.convs1.{m}.weight'],
                              W[f'resblocks.{r}.convs1.{m}.bias'], dilation=d,
                              padding=_get_padding(kk, d))
                xt = F.leaky_relu(xt, LRELU_SLOPE)
                xt = F.conv1d(xt, W[f'resblocks.{r}.convs2.{m}.weight'],
                              W[f'resblocks.{r}.convs2.{m}.bias'], padding=_get_padding(kk, 1))
                y = xt + y
            xs = y if xs is None else xs + y
        x = xs / nk
        if stages is not None:
            stages.append(x)
    x = F.leaky_relu(x)                       # default slope 0.01 (:123), not LRELU_SLOPE
    x = F.conv1d(x, W['conv_post.weight'], W['conv_post.bias'], padding=3)
    x = torch.tanh(x)
    return x[0] if unb else x


# --------------------------------------------------------------------------------------
# FastPitch
# --------------------------------------------------------------------------------------

def regulate_len_indices(durations, pace=1.0):
    """Integer restatement of models/fastpitch/fastpitch/model.py:68-90 (regulate_len).
    reps = (dur/pace + 0.5).long(); dec_lens = sum; frame t of utterance b copies token
    j with cumsum[j] <= t < cumsum[j+1], frames >= dec_len are zero rows (idx -1).
    Pure numpy; float step uses float32 like `durations.float() / pace`.
    Returns reps int64 [B,L], dec_lens int64 [B], idx int32 [B,T_max]."""
    d = np.asarray(durations, dtype=np.float32)
    reps = (d / np.float32(pace) + np.float32(0.5)).astype(np.int64)   # trunc toward 0 = .long()
    dec_lens = reps.sum(axis=1)
    tmax = int(dec_lens.max()) if dec_lens.size else 0
    B, L = reps.shape
    idx = np.full((B, tmax), -1, np.int32)
    for b in range(B):
        cs = np.concatenate([[0], np.cumsum(reps[b])])
        t = np.arange(tmax)
        j = np.searchsorted(cs, t, side='right') - 1        # cs[j] <= t < cs[j+1]
        ok = t < dec_lens[b]
        idx[b, ok] = j[ok]
    return reps, dec_lens, idx


def _pos_emb(pos_seq, inv_freq):
    # transformer.py:41-44: outer product, then [sin | cos] concatenated
    s = torch.matmul(pos_seq[:, None], inv_freq[None, :])
    return torch.cat([s.sin(), s.cos()], dim=1)[None]


def _fft(W, prefix, n_layers, inp, mask, d_head, n_head, trace=None):
    """transformer.py:207-225 body after the embedding: pos-emb add done by caller.
    Layers: TransformerLayer.forward (:172-177) = MultiHeadAttn (:113-160, post-LN) ->
    *= mask -> PositionwiseConvFF (:72-90, post-LN) -> *= mask."""
    out = inp
    scale = 1 / (d_head ** 0.5)
    for i in range(n_layers):
        p = f'{prefix}.layers.{i}.'
        B, S, _ = out.shape
        qkv = F.linear(out, W[p + 'dec_attn.qkv_net.weight'], W[p + 'dec_attn.qkv_net.bias'])
        hq, hk, hv = torch.chunk(qkv, 3, dim=2)
        q = hq.view(B, S, n_head, d_head).permute(2, 0, 1, 3).reshape(-1, S, d_head)
        k = hk.view(B, S, n_head, d_head).permute(2, 0, 1, 3).reshape(-1, S, d_head)
        v = hv.view(B, S, n_head, d_head).permute(2, 0, 1, 3).reshape(-1, S, d_head)
        score = torch.bmm(q, k.transpose(1, 2)) * scale
        am = (~mask.squeeze(2)).unsqueeze(1)                      # pad KEYS masked (:134-137)
        am = am.repeat(n_head, S, 1)
        score = score.masked_fill(am, -float('inf'))
        prob = F.softmax(score, dim=2)
        vec = torch.bmm(prob, v)
        vec = vec.view(n_head, B, S, d_head).permute(1, 2, 0, 3).contiguous().view(B, S, n_head * d_head)
        att = F.linear(vec, W[p + 'dec_attn.o_net.weight'])
        d_model = out.shape[2]
        out1 = F.layer_norm(out + att, (d_model,), W[p + 'dec_attn.layer_norm.weight'],
                            W[p + 'dec_attn.layer_norm.bias'])
        out1 = out1 * mask
        h = F.conv1d(out1.transpose(1, 2), W[p + 'pos_ff.CoreNet.0.weight'],
                     W[p + 'pos_ff.CoreNet.0.bias'], padding=W[p + 'pos_ff.CoreNet.0.weight'].shape[2] // 2)
        h = F.relu(h)
        h2 = F.conv1d(h, W[p + 'pos_ff.CoreNet.2.weight'], W[p + 'pos_ff.CoreNet.2.bias'],
                      padding=W[p + 'pos_ff.CoreNet.2.weight'].shape[2] // 2).transpose(1, 2)
        out2 = F.layer_norm(out1 + h2, (d_model,), W[p + 'pos_ff.layer_norm.weight'],
                            W[p + 'pos_ff.layer_norm.bias'])
        out = out2 * mask
        if trace is not None:
            trace[p + 'attn_out'] = out1
            trace[p + 'out'] = out
    return out


def _predictor(W, prefix, n_layers, enc_out, mask):
    """model.py:129-133 TemporalPredictor.forward over ConvReLUNorm (:54-57)."""
    out = (enc_out * mask).transpose(1, 2)
    for i in range(n_layers):
        p = f'{prefix}.layers.{i}.'
        w = W[p + 'conv.weight']
        out = F.relu(F.conv1d(out, w, W[p + 'conv.bias'], padding=w.shape[2] // 2))
        out = F.layer_norm(out.transpose(1, 2), (w.shape[0],), W[p + 'norm.weight'],
                           W[p + 'norm.bias']).transpose(1, 2)
    out = out.transpose(1, 2)
    return F.linear(out, W[prefix + '.fc.weight'], W[prefix + '.fc.bias']) * mask


def fastpitch_infer(w, cfg, ids, pace=1.0, dur_tgt=None, pitch_tgt=None, energy_tgt=None,
                    pitch_transform=None, max_duration=75, speaker=0, dtype=torch.float32,
                    trace=None):
    """models/fastpitch/fastpitch/model.py:351-409 (FastPitch.infer).
    ids int64 [B,L] (0 = padding).  Returns (mel [B,80,T_max], dec_lens int64 [B],
    dur_pred [B,L], pitch_pred [B,1,L], energy_pred [B,L])."""
    W = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in w.items()}
    ids = torch.as_tensor(np.asarray(ids)).long()
    d = cfg['symbols_embedding_dim']
    B, L = ids.shape
    if cfg['n_speakers'] > 1:                                   # :355-361
        spk = W['speaker_emb.weight'][torch.ones(B).long() * speaker].unsqueeze(1)
        spk = spk * cfg['speaker_emb_weight']
    else:
        spk = 0
    # encoder: transformer.py:207-219
    inp = F.embedding(ids, W['encoder.word_emb.weight'], padding_idx=cfg['padding_idx'])
    mask = (ids != cfg['padding_idx']).unsqueeze(2)
    pos = _pos_emb(torch.arange(L).to(dtype), W['encoder.pos_emb.inv_freq']) * mask
    enc_out = _fft(W, 'encoder', cfg['in_fft_n_layers'], inp + pos + spk, mask,
                   cfg['in_fft_d_head'], cfg['in_fft_n_heads'], trace)
    if trace is not None:
        trace['enc_out'] = enc_out
    # :367-368
    log_dur = _predictor(W, 'duration_predictor', cfg['dur_predictor_n_layers'], enc_out, mask).squeeze(-1)
    dur_pred = torch.clamp(torch.exp(log_dur) - 1, 0, max_duration)
    # :371-386
    pitch_pred = _predictor(W, 'pitch_predictor', cfg['pitch_predictor_n_layers'], enc_out, mask).permute(0, 2, 1)
    if pitch_transform is not None:
        if W['pitch_std'][0] == 0.0:
            mean, std = 218.14, 67.24
        else:
            mean, std = W['pitch_mean'][0], W['pitch_std'][0]
        pitch_pred = pitch_transform(pitch_pred, mask.sum(dim=(1, 2)), mean, std)
    kp = W['pitch_emb.weight'].shape[2]
    src = pitch_pred if pitch_tgt is None else _t(pitch_tgt, dtype)
    enc_out = enc_out + F.conv1d(src, W['pitch_emb.weight'], W['pitch_emb.bias'],
                                 padding=int((kp - 1) / 2)).transpose(1, 2)
    # :389-399
    if cfg['energy_conditioning']:
        ke = W['energy_emb.weight'].shape[2]
        if energy_tgt is None:
            energy_pred = _predictor(W, 'energy_predictor', cfg['energy_predictor_n_layers'], enc_out, mask).squeeze(-1)
            e_src = energy_pred.unsqueeze(1)
        else:
            energy_pred = None
            e_src = _t(energy_tgt, dtype)
        enc_out = enc_out + F.conv1d(e_src, W['energy_emb.weight'], W['energy_emb.bias'],
                                     padding=int((ke - 1) / 2)).transpose(1, 2)
    else:
        energy_pred = None
    if trace is not None:
        trace['enc_cond'] = enc_out
    # :401-403 regulate_len (model.py:68-90), same dense formulation as the reference
    durs = dur_pred if dur_tgt is None else _t(dur_tgt, dtype)
    reps = (durs.float() / pace + 0.5).long()
    dec_lens = reps.sum(dim=1)
    max_len = int(dec_lens.max())
    cs = torch.cumsum(F.pad(reps, (1, 0, 0, 0), value=0.0), dim=1)[:, None, :].to(dtype)
    rng = torch.arange(max_len)[None, :, None]
    mult = ((cs[:, :, :-1] <= rng) & (cs[:, :, 1:] > rng)).to(dtype)
    len_regulated = torch.matmul(mult, enc_out)
    if trace is not None:
        trace['len_regulated'] = len_regulated
        trace['reps'] = reps
    # decoder: transformer.py:208-219 with embed_input=False
    dmask = (torch.arange(max_len)[None, :] < dec_lens[:, None]).unsqueeze(2)
    dpos = _pos_emb(torch.arange(max_len).to(dtype), W['decoder.pos_emb.inv_freq']) * dmask
    dec_out = _fft(W, 'decoder', cfg['out_fft_n_layers'], len_regulated + dpos, dmask,
                   cfg['out_fft_d_head'], cfg['out_fft_n_heads'], trace)
    mel = F.linear(dec_out, W['proj.weight'], W['proj.bias']).permute(0, 2, 1)   # :406-408
    return mel, dec_lens, dur_pred, pitch_pred, energy_pred


# --------------------------------------------------------------------------------------
# Denoiser (torch.stft-based; torchaudio itself is absent — parity unpinned there)
# --------------------------------------------------------------------------------------

def denoiser_bias_spec(w_folded, cfg, dtype=torch.float32):
    """vocoder/hifigan/denoiser.py:50-64: vocoder(zeros[1,80,88]) -> |STFT| frame 0."""
    wave = hifigan_forward(w_folded, torch.zeros(1, 80, 88), cfg, dtype)[:, 0]   # [1, n]
    win = torch.hann_window(1024, dtype=wave.dtype)
    spec = torch.stft(wave, 1024, 256, 1024, win, center=True, pad_mode='reflect',
                      normalized=False, onesided=True, return_complex=True).abs()
    return spec[:, :, 0][:, :, None]


def denoise(wave, bias_spec, strength):
    """vocoder/hifigan/denoiser.py:66-72.  wave [1,n]."""
    wave = wave.float()
    win = torch.hann_window(1024)
    spec = torch.stft(wave, 1024, 256, 1024, win, center=True, pad_mode='reflect',
                      normalized=False, onesided=True, return_complex=True)
    mag, ph = spec.abs(), spec.angle()
    mag = torch.clamp(mag - bias_spec * strength, 0.0)
    return torch.istft(mag * torch.exp(1j * ph), 1024, 256, 1024, win, center=True,
                       normalized=False, onesided=True)


# --------------------------------------------------------------------------------------
# MelVocos('22k'): ConvNeXt backbone + ISTFT head ("same" padding)
# --------------------------------------------------------------------------------------

def _vocos_backbone(W, x, n_layers):
    """vocoder/vocos/models.py:77-89 (VocosBackbone.forward) over ConvNeXtBlock.forward
    (modules.py:43-60).  x [B,80,T] -> [B,T,512]."""
    x = F.conv1d(x, W['backbone.embed.weight'], W['backbone.embed.bias'], padding=3)
    d = x.shape[1]
    x = F.layer_norm(x.transpose(1, 2), (d,), W['backbone.norm.weight'], W['backbone.norm.bias'], eps=1e-6).transpose(1, 2)
    for i in range(n_layers):
        p = f'backbone.convnext.{i}.'
        res = x
        y = F.conv1d(x, W[p + 'dwconv.weight'], W[p + 'dwconv.bias'], padding=3, groups=d).transpose(1, 2)
        y = F.layer_norm(y, (d,), W[p + 'norm.weight'], W[p + 'norm.bias'], eps=1e-6)
        y = F.linear(y, W[p + 'pwconv1.weight'], W[p + 'pwconv1.bias'])
        y = F.gelu(y)
        y = F.linear(y, W[p + 'pwconv2.weight'], W[p + 'pwconv2.bias'])
        y = W[p + 'gamma'] * y
        x = res + y.transpose(1, 2)
    return F.layer_norm(x.transpose(1, 2), (d,), W['backbone.final_layer_norm.weight'],
                        W['backbone.final_layer_norm.bias'], eps=1e-6)


def vocos_bias_vec(w, cfg, dtype=torch.float32):
    """vocoder/vocos/pretrained.py:59-71 (make_denoising_vector): exp(log-mag) of a zero mel, frame 0."""
    W = {k: _t(v, dtype) for k, v in w.items()}
    feats = _vocos_backbone(W, torch.zeros(1, cfg['input_channels'], 88, dtype=dtype), cfg['num_layers'])
    xb = F.linear(feats, W['head.out.weight'], W['head.out.bias']).transpose(1, 2)
    mag, _ = xb.chunk(2, dim=1)
    return torch.clip(torch.exp(mag), max=1e2)[:, :, 0:1]


def vocos_forward(w, mel, cfg, denoise=0.0, bias_vec=None, dtype=torch.float32):
    """vocoder/vocos/pretrained.py:73-93 (MelVocos.forward) + ISTFT 'same' (spectral_ops.py:33-75).
    mel [B,80,T] -> wave [B, 256*T]."""
    W = {k: _t(v, dtype) for k, v in w.items()}
    x = _t(mel, dtype)
    feats = _vocos_backbone(W, x, cfg['num_layers'])
    xo = F.linear(feats, W['head.out.weight'], W['head.out.bias']).transpose(1, 2)
    mag, ph = xo.chunk(2, dim=1)
    mag = torch.exp(mag)
    if bias_vec is None:
        bias_vec = vocos_bias_vec(w, cfg, dtype)
    mag = torch.clamp(mag - denoise * bias_vec.to(dtype), min=0., max=1e2)
    S = mag * (torch.cos(ph) + 1j * torch.sin(ph))
    n_fft, hop = cfg['n_fft'], cfg['hop_length']
    pad = (n_fft - hop) // 2
    win = torch.hann_window(n_fft, dtype=dtype)
    B, N, T = S.shape
    ifft = torch.fft.irfft(S, n_fft, dim=1, norm='backward') * win[None, :, None]
    out_size = (T - 1) * hop + n_fft
    y = F.fold(ifft, output_size=(1, out_size), kernel_size=(1, n_fft), stride=(1, hop))[:, 0, 0, pad:-pad]
    env = F.fold(win.square().expand(1, T, -1).transpose(1, 2), output_size=(1, out_size), kernel_size=(1, n_fft),
                 stride=(1, hop)).squeeze()[pad:-pad]
    return y / env


def hifigan_forward_ragged(w, mel, lens, cfg, dtype=torch.float32):
    """The reference's per-utterance vocoder loop (models/fastpitch/networks.py:340-345: `Generator.forward` on each exact-length
    mel) restated on ONE padded batch: mel [B,80,T_max], lens [B] -> wave [B, 256*T_max] (zeros past 256*lens[b]).
    Before every conv the positions at or past the utterance's own length (lens[b] x upsampling so far) are set to zero, which is
    exactly what the conv's zero padding shows it in the unbatched call -- every valid output is the same sum of
    the same products as `hifigan_forward(mel[b, :, :lens[b]])`; only the library's choice of algorithm for the larger shape can
    differ (tests/test_oracle_golden.py pins the two to each other).  One shape per layer for the whole batch: the full-size GPU
    checks run it once instead of once per distinct length."""
    W = {k: v.to(dtype) for k, v in w.items()}
    x = _t(mel, dtype)
    lens = torch.as_tensor(lens, device=x.device).to(torch.int64)
    nk = len(cfg['resblock_kernel_sizes'])

    def masked(t, mul):
        m = torch.arange(t.shape[-1], device=t.device)[None, None, :] < (lens * mul)[:, None, None]
        return t * m
    mul = 1
    x = F.conv1d(masked(x, mul), W['conv_pre.weight'], W['conv_pre.bias'], padding=3)
    for i, (u, k) in enumerate(zip(cfg['upsample_rates'], cfg['upsample_kernel_sizes'])):
        x = F.conv_transpose1d(masked(F.leaky_relu(x, LRELU_SLOPE), mul), W[f'ups.{i}.weight'], W[f'ups.{i}.bias'], stride=u,
                               padding=(k - u) // 2)
        mul *= u
        xs = None
        for j, (kk, dil) in enumerate(zip(cfg['resblock_kernel_sizes'], cfg['resblock_dilation_sizes'])):
            r = i * nk + j
            y = x
            for m, d in enumerate(dil):
                xt = F.conv1d(masked(F.leaky_relu(y, LRELU_SLOPE), mul), W[f'resblocks.{r}.convs1.{m}.weight'],
                              W[f'resblocks.{r}.convs1.{m}.bias'], dilation=d, padding=_get_padding(kk, d))
                xt = F.conv1d(masked(F.leaky_relu(xt, LRELU_SLOPE), mul), W[f'resblocks.{r}.convs2.{m}.weight'],
                              W[f'resblocks.{r}.convs2.{m}.bias'], padding=_get_padding(kk, 1))
                y = xt + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.conv1d(masked(F.leaky_relu(x), mul), W['conv_post.weight'], W['conv_post.bias'], padding=3)
    return masked(torch.tanh(x), mul)[:, 0]


# --------------------------------------------------------------------------------------
# The whole .tts_batch-equivalent (used for goldens and as bench.py's cpu_baseline)
# --------------------------------------------------------------------------------------

def tts_batch(fp_w, fp_cfg, hg_w, hg_cfg, ids, dur_tgt=None, pace=1.0, speaker=0,
              denoise_strength=0.0, bias_spec=None):
    """models/fastpitch/networks.py:322-350 after tokenisation: batched FastPitch, then the
    vocoder looped per utterance on exact-length 2-D mels (:340-345)."""
    mel, dec_lens, *_ = fastpitch_infer(fp_w, fp_cfg, ids, pace=pace, dur_tgt=dur_tgt, speaker=speaker)
    waves = []
    for b in range(mel.shape[0]):
        wv = hifigan_forward(hg_w, mel[b, :, :int(dec_lens[b])], hg_cfg)
        if denoise_strength > 0:
            wv = denoise(wv, bias_spec, denoise_strength)
        waves.append(wv[0])
    return mel, dec_lens, waves
