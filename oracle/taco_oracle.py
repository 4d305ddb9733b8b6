"""CPU oracle for the Tacotron2 path (BASELINE config 4).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The arithmetic of this model lives in `torchaudio.models.tacotron2`
(_Encoder, _Decoder, _Postnet; imported at models/tacotron2/tacotron2_ms.py:113), a third-party
dependency that is absent from /root/reference and from this image, with no pinned version
(README.md:66 lists a bare `torchaudio`) and no test or golden vector in the reference.  This file
restates the *published* architecture (Shen et al. 2018 as implemented by NVIDIA
DeepLearningExamples, which torchaudio's docstring names as its origin) with the constructor
arguments fixed at tacotron2_ms.py:152-205, and the reference's own glue around it
(tacotron2_ms.py:279-332).  The HIP path is checked against THIS restatement only
(self-consistency), and DESIGN.md says so.

Prenet dropout: upstream applies F.dropout(p=0.5, training=True) at inference, i.e. the reference
output is a random variable.  Both this oracle and the HIP kernels draw the keep-mask from the same
counter-based hash (`keep_mask`), so they can be compared bit-for-bit per (seed, step, b, unit);
seed < 0 disables dropout (mask = 1, no 1/(1-p) scaling).
"""
import numpy as np
import torch
import torch.nn.functional as F


def keep_mask(seed, layer, step, batch, n):
    """uint32 hash -> keep bit (p = 0.5).  Returns float32 [batch, n] of {0., 2.}."""
    b = np.arange(batch, dtype=np.uint64)[:, None]
    j = np.arange(n, dtype=np.uint64)[None, :]
    M = np.uint64(0xFFFFFFFF)
    x = (np.uint64(seed) * np.uint64(0x9E3779B1) + np.uint64(layer) * np.uint64(0x85EBCA77)
         + np.uint64(step) * np.uint64(0xC2B2AE3D) + b * np.uint64(0x27D4EB2F) + j * np.uint64(0x165667B1)) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & M
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & M
    x ^= x >> np.uint64(16)
    return torch.from_numpy(((x & np.uint64(1)).astype(np.float32)) * 2.0)


def _bn_fold(W, name):
    """BatchNorm1d in eval mode as y = x * s + t."""
    s = W[name + '.weight'] / torch.sqrt(W[name + '.running_var'] + 1e-5)
    return s, W[name + '.bias'] - W[name + '.running_mean'] * s


def _lstm_cell(x, h, c, W, name, sfx=''):
    g = F.linear(x, W[f'{name}.weight_ih{sfx}'], W[f'{name}.bias_ih{sfx}']) + \
        F.linear(h, W[f'{name}.weight_hh{sfx}'], W[f'{name}.bias_hh{sfx}'])
    i, f, gg, o = g.chunk(4, dim=1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c), c


def tacotron2_infer(w, cfg, tokens, speaker_ids=None, lengths=None, max_step=None, seed=-1, dtype=torch.float32, trace=None):
    """Tacotron2MS.infer (models/tacotron2/tacotron2_ms.py:279-332).  tokens int64 [B, L] sorted by
    length descending (text_collate_fn), lengths int64 [B].  Returns (mel_postnet [B,80,T],
    mel_lengths int32 [B], alignments [B,T,L])."""
    W = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in w.items()}
    tokens = torch.as_tensor(np.asarray(tokens)).long()
    B, L = tokens.shape
    lengths = torch.full((B,), L, dtype=torch.long) if lengths is None else torch.as_tensor(np.asarray(lengths)).long()
    speaker_ids = torch.zeros(B, dtype=torch.long) if speaker_ids is None else torch.as_tensor(np.asarray(speaker_ids)).long()
    max_step = cfg['decoder_max_step'] if max_step is None else max_step
    # ---- encoder: embedding -> 3 x (conv k5 + BN + ReLU) on the padded batch -> packed BiLSTM
    x = F.embedding(tokens, W['embedding.weight']).transpose(1, 2)
    for i in range(cfg['encoder_n_convolution']):
        p = f'encoder.convolutions.{i}.'
        x = F.conv1d(x, W[p + '0.weight'], W[p + '0.bias'], padding=(cfg['encoder_kernel_size'] - 1) // 2)
        s, t = _bn_fold(W, p + '1')
        x = F.relu(x * s[None, :, None] + t[None, :, None])
    x = x.transpose(1, 2)                                       # [B, L, 512]
    Hh = cfg['encoder_embedding_dim'] // 2
    enc = torch.zeros(B, L, 2 * Hh, dtype=dtype)
    for b in range(B):
        n = int(lengths[b])
        h = c = torch.zeros(1, Hh, dtype=dtype)
        for t in range(n):
            h, c = _lstm_cell(x[b:b + 1, t], h, c, W, 'encoder.lstm', '_l0')
            enc[b, t, :Hh] = h[0]
        h = c = torch.zeros(1, Hh, dtype=dtype)
        for t in range(n - 1, -1, -1):
            h, c = _lstm_cell(x[b:b + 1, t], h, c, W, 'encoder.lstm', '_l0_reverse')
            enc[b, t, Hh:] = h[0]
    if cfg['num_speakers'] > 1:
        spk = W['speaker_embedding.weight'][speaker_ids].unsqueeze(1).repeat(1, L, 1)
        memory = torch.cat((enc, spk), dim=2)
    else:
        memory = enc
    if trace is not None:
        trace['conv_out'], trace['enc'] = x.clone(), enc.clone()           # test aid: encoder blocks vs torch.nn modules
    # ---- decoder.infer
    A, D = cfg['attention_rnn_dim'], cfg['decoder_rnn_dim']
    Mdim = memory.shape[2]
    mask = torch.arange(L)[None, :] >= lengths[:, None]         # True = padded
    pm = F.linear(memory, W['decoder.attention_layer.memory_layer.weight'])
    att_h = torch.zeros(B, A, dtype=dtype); att_c = torch.zeros(B, A, dtype=dtype)
    dec_h = torch.zeros(B, D, dtype=dtype); dec_c = torch.zeros(B, D, dtype=dtype)
    aw = torch.zeros(B, L, dtype=dtype); aw_cum = torch.zeros(B, L, dtype=dtype)
    ctx = torch.zeros(B, Mdim, dtype=dtype)
    dec_in = torch.zeros(B, cfg['n_mels'], dtype=dtype)
    mel_lens = torch.zeros(B, dtype=torch.int32)
    finished = torch.zeros(B, dtype=torch.bool)
    mels, aligns = [], []
    ks = cfg['attention_location_kernel_size']
    for step in range(max_step):
        p = dec_in
        for li in range(2):
            p = F.relu(F.linear(p, W[f'decoder.prenet.layers.{li}.weight']))
            if seed >= 0:
                p = p * keep_mask(seed, li, step, B, p.shape[1]).to(dtype)
        att_h, att_c = _lstm_cell(torch.cat((p, ctx), -1), att_h, att_c, W, 'decoder.attention_rnn')
        cat = torch.cat((aw.unsqueeze(1), aw_cum.unsqueeze(1)), dim=1)
        pq = F.linear(att_h.unsqueeze(1), W['decoder.attention_layer.query_layer.weight'])
        loc = F.conv1d(cat, W['decoder.attention_layer.location_layer.location_conv.weight'], padding=(ks - 1) // 2)
        pl = F.linear(loc.transpose(1, 2), W['decoder.attention_layer.location_layer.location_dense.weight'])
        e = F.linear(torch.tanh(pq + pl + pm), W['decoder.attention_layer.v.weight']).squeeze(2)
        e = e.masked_fill(mask, -float('inf'))
        aw = F.softmax(e, dim=1)
        ctx = torch.bmm(aw.unsqueeze(1), memory).squeeze(1)
        aw_cum = aw_cum + aw
        dec_h, dec_c = _lstm_cell(torch.cat((att_h, ctx), -1), dec_h, dec_c, W, 'decoder.decoder_rnn')
        hc = torch.cat((dec_h, ctx), dim=1)
        mel = F.linear(hc, W['decoder.linear_projection.weight'], W['decoder.linear_projection.bias'])
        gate = F.linear(hc, W['decoder.gate_layer.weight'], W['decoder.gate_layer.bias'])
        mels.append(mel); aligns.append(aw)
        if trace is not None:
            trace.setdefault('gate', []).append(gate.squeeze(1).clone())   # test aid: stop-token logits per step
            trace.setdefault('hc', []).append(hc.clone())                  # and the gate layer's input
            trace.setdefault('att_h', []).append(att_h.clone())
            trace.setdefault('prenet', []).append(p.clone())
        mel_lens[~finished] += 1
        finished |= torch.sigmoid(gate.squeeze(1)) > cfg['gate_threshold']
        if cfg.get('decoder_early_stopping', True) and bool(torch.all(finished)):
            break
        dec_in = mel
    mel = torch.stack(mels, dim=2)                              # [B, 80, T]
    align = torch.stack(aligns, dim=1)                          # [B, T, L]
    # ---- postnet (eval: no dropout) on the whole padded batch, no masking
    y = mel
    n = cfg['postnet_n_convolution']
    for i in range(n):
        p = f'postnet.convolutions.{i}.'
        y = F.conv1d(y, W[p + '0.weight'], W[p + '0.bias'], padding=(cfg['postnet_kernel_size'] - 1) // 2)
        s, t = _bn_fold(W, p + '1')
        y = y * s[None, :, None] + t[None, :, None]
        if i < n - 1:
            y = torch.tanh(y)
    return mel + y, mel_lens, align
