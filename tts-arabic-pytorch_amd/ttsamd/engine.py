"""Engines = C-ABI handles + workspaces.  PyTorch here is plumbing only (device memory,
current stream); every FLOP of the path runs in libttsamd.so."""
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .config import NET_CONFIG, HIFIGAN_CONFIG, VOCOS_22K_CONFIG, TACOTRON2_CONFIG


def _require_gpu():
    if not torch.cuda.is_available():
        raise L.TtsAmdError('no ROCm device visible: the ttsamd engines run only on an MI355X (gfx950); '
                            'there is no CPU fallback')
    lib = L.load()
    if not lib.ttsamd_device_ok():
        raise L.TtsAmdError('device 0 is not gfx950')
    return lib


PRECISIONS = {'f32': 0, 'bf16': 1, 'bf16x3': 2}


def set_precision(name):
    """MFMA operand precision of all conv/linear GEMMs: 'f32' (default, exact), 'bf16' (config 3),
    'bf16x3' (split bf16: fp32-class accuracy at bf16 MFMA rate)."""
    lib = L.load()
    L.check(lib.ttsamd_set_precision(PRECISIONS[name]), 'set_precision')


def get_precision():
    code = L.load().ttsamd_get_precision()
    return {v: k for k, v in PRECISIONS.items()}[code]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, device):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    return t.to(device=device, dtype=torch.float32).contiguous()


def _check_ids(t, n, what):
    """nn.Embedding raises IndexError on an out-of-range index; the HIP gathers only clamp.  Checked where it is free:
    on tensors that still live on the host (the drop-in wrappers tokenise on the CPU)."""
    if isinstance(t, torch.Tensor) and t.device.type == 'cpu' and t.numel():
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= n:
            raise IndexError(f'{what}: index out of range [0, {n}) (got min {lo}, max {hi})')


class _Workspace:
    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


class HifiGanEngine:
    """Handle over ttsamd_hifigan_* (replaces vocoder.load_hifigan + Generator.forward)."""

    def __init__(self, state_dict, config=None, device='cuda'):
        self.lib = _require_gpu()
        self.device = torch.device(device if device != 'cuda' else 'cuda:0')
        h = dict(HIFIGAN_CONFIG if config is None else config)
        if str(h.get('resblock', '1')) != '1':
            raise L.TtsAmdError('only ResBlock1 generators are built (config.json:2)')
        cfg = L.HifiGanCfg()
        cfg.num_mels = h.get('num_mels', 80)
        cfg.upsample_initial_channel = h['upsample_initial_channel']
        cfg.n_ups = len(h['upsample_rates'])
        for i, (u, k) in enumerate(zip(h['upsample_rates'], h['upsample_kernel_sizes'])):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        cfg.n_kernels = len(h['resblock_kernel_sizes'])
        cfg.n_dilations = len(h['resblock_dilation_sizes'][0])
        for j, (k, ds) in enumerate(zip(h['resblock_kernel_sizes'], h['resblock_dilation_sizes'])):
            cfg.resblock_kernel_sizes[j] = k
            for m, d in enumerate(ds):
                cfg.resblock_dilations[j][m] = d
        self.hop = int(np.prod(h['upsample_rates']))
        self.num_mels = cfg.num_mels
        arr, keep = L.make_tensors(state_dict)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ttsamd_hifigan_create(arr, len(arr), C.byref(cfg), C.byref(handle)), 'hifigan_create')
        self.handle = handle
        self.ws = _Workspace()

    def __del__(self):
        if getattr(self, 'handle', None):
            self.lib.ttsamd_hifigan_destroy(self.handle)
            self.handle = None

    def forward(self, mel, lens=None):
        """mel [B,80,T] float32 on the GPU, lens int64 [B] (device) or None -> wave [B, hop*T].
        Samples past hop*lens[b] are zero."""
        mel = _f32(mel, self.device)
        B, M, T = mel.shape
        assert M == self.num_mels
        if lens is not None:
            lens = lens.to(device=self.device, dtype=torch.int64).contiguous()
        wave = torch.zeros(B, self.hop * T, dtype=torch.float32, device=self.device)
        if T == 0:
            return wave
        with torch.cuda.device(self.device):
            nbytes = self.lib.ttsamd_hifigan_workspace_bytes(self.handle, B, T)
            ws = self.ws.get(nbytes, self.device)
            L.check(self.lib.ttsamd_hifigan_forward(self.handle, _ptr(mel), _ptr(lens), B, T, _ptr(wave), _ptr(ws),
                                                    nbytes, _stream()), 'hifigan_forward')
        return wave


class FastPitchEngine:
    """Handle over ttsamd_fastpitch_* (replaces FastPitch.infer, model.py:351-409)."""

    def __init__(self, state_dict, config=None, device='cuda'):
        self.lib = _require_gpu()
        self.device = torch.device(device if device != 'cuda' else 'cuda:0')
        c = dict(NET_CONFIG if config is None else config)
        self.config = c
        cfg = L.FastPitchCfg()
        cfg.n_mel_channels, cfg.n_symbols, cfg.padding_idx = c['n_mel_channels'], c['n_symbols'], c['padding_idx']
        cfg.d_model = c['symbols_embedding_dim']
        cfg.in_fft_n_layers, cfg.in_fft_n_heads, cfg.in_fft_d_head = c['in_fft_n_layers'], c['in_fft_n_heads'], c['in_fft_d_head']
        cfg.in_fft_kernel, cfg.in_fft_filter = c['in_fft_conv1d_kernel_size'], c['in_fft_conv1d_filter_size']
        cfg.out_fft_n_layers, cfg.out_fft_n_heads, cfg.out_fft_d_head = c['out_fft_n_layers'], c['out_fft_n_heads'], c['out_fft_d_head']
        cfg.out_fft_kernel, cfg.out_fft_filter = c['out_fft_conv1d_kernel_size'], c['out_fft_conv1d_filter_size']
        cfg.dur_kernel, cfg.dur_filter, cfg.dur_n_layers = c['dur_predictor_kernel_size'], c['dur_predictor_filter_size'], c['dur_predictor_n_layers']
        cfg.pitch_kernel, cfg.pitch_filter, cfg.pitch_n_layers = c['pitch_predictor_kernel_size'], c['pitch_predictor_filter_size'], c['pitch_predictor_n_layers']
        cfg.pitch_emb_kernel = c['pitch_embedding_kernel_size']
        cfg.energy_conditioning = int(bool(c['energy_conditioning']))
        cfg.energy_kernel, cfg.energy_filter, cfg.energy_n_layers = c['energy_predictor_kernel_size'], c['energy_predictor_filter_size'], c['energy_predictor_n_layers']
        cfg.energy_emb_kernel = c['energy_embedding_kernel_size']
        cfg.n_speakers, cfg.speaker_emb_weight = c['n_speakers'], float(c['speaker_emb_weight'])
        assert c.get('pitch_conditioning_formants', 1) == 1
        self.d_model, self.n_mel = cfg.d_model, cfg.n_mel_channels
        arr, keep = L.make_tensors(state_dict)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ttsamd_fastpitch_create(arr, len(arr), C.byref(cfg), C.byref(handle)), 'fastpitch_create')
        self.handle = handle
        self.ws = _Workspace()
        self._alone = False
        # rows padded to 16 bytes (see infer): the library's ragged schedule handles the reference's kernel size 3 only
        self._pad_ok = all(int(c[k]) == 3 for k in ('in_fft_conv1d_kernel_size', 'out_fft_conv1d_kernel_size', 'dur_predictor_kernel_size',
                                                    'pitch_predictor_kernel_size')) and \
            (not c['energy_conditioning'] or int(c['energy_predictor_kernel_size']) == 3)

    def __del__(self):
        if getattr(self, 'handle', None):
            self.lib.ttsamd_fastpitch_destroy(self.handle)
            self.handle = None

    def infer(self, ids, pace=1.0, dur_tgt=None, pitch_tgt=None, energy_tgt=None, pitch_mul=1.0, pitch_add=0.0,
              max_duration=75, speaker=0, return_idx=False, lens_hook=None, alone=False):
        """Same contract as FastPitch.infer (model.py:351-353) with pitch_transform restricted to
        the affine pitch_trf the reference wrappers install (networks.py:38-42,121-122).
        ids int64 [B,L] zero-padded at the end.  Returns (mel [B,80,T_max], dec_lens int64 [B],
        dur_pred [B,L], pitch_pred [B,1,L], energy_pred [B,L] or None).
        `lens_hook(dec_lens_device) -> host ints [B]` replaces the one device->host read of the call (the
        data-parallel path all-gathers every rank's lengths in that same synchronisation, ttsamd.dp).
        `alone=True`: every row as if it were the only utterance of the call (ttsamd_fastpitch_set_batch_mode 1) -- row b equals
        infer(ids[b:b+1, :len_b]) within fp32 summation order: the reference's batch_size = 1 loop as one ragged call."""
        if bool(alone) != self._alone:
            L.check(self.lib.ttsamd_fastpitch_set_batch_mode(self.handle, int(bool(alone))), 'fastpitch_set_batch_mode')
            self._alone = bool(alone)
        dev = self.device
        ids = torch.as_tensor(ids).to(device=dev, dtype=torch.int64).contiguous()
        B, Lt0 = ids.shape
        d = self.d_model
        dur_tgt, pitch_tgt, energy_tgt = _f32(dur_tgt, dev), _f32(pitch_tgt, dev), _f32(energy_tgt, dev)
        # Batches: the DECODER's frame rows padded to a multiple of 4 (16 bytes).  The conv engine's fast paths -- the Winograd F(4,3) kernel,
        # the float4 row epilogue -- need 16-byte-aligned rows, and a real batch's longest utterance is a multiple of 4 one time in four (config
        # 1: three of the four FastPitch calls ran the decoder conv-FF on the direct kernel's per-lane epilogue, 2x the time).  The decoder
        # takes the batch's length from dec_lens, not from the row width (lens_plus1_kernel), so the extra columns are plain padding; mel is
        # returned as a view of the un-padded shape.  (Batch 1 keeps its exact shape: one launch more would cost what alignment gains; the
        # bf16 octet path has no ragged schedule; the encoder's rows stay as given -- the caller's ids tensor IS the reference's padded batch.)
        pad = B >= 2 and self._pad_ok and get_precision() != 'bf16'
        Lt = Lt0
        enc = torch.empty(B, d, Lt, dtype=torch.float32, device=dev)
        dur_pred = torch.empty(B, Lt, dtype=torch.float32, device=dev)
        pitch_pred = torch.empty(B, 1, Lt, dtype=torch.float32, device=dev)
        energy_pred = torch.empty(B, Lt, dtype=torch.float32, device=dev) if self.config['energy_conditioning'] else None
        reps = torch.empty(B, Lt, dtype=torch.int64, device=dev)
        dec_lens = torch.empty(B, dtype=torch.int64, device=dev)
        lib = self.lib
        with torch.cuda.device(dev):
            nb = lib.ttsamd_fastpitch_encode_workspace_bytes(self.handle, B, Lt)
            ws = self.ws.get(nb, dev)
            L.check(lib.ttsamd_fastpitch_encode(self.handle, _ptr(ids), B, Lt, int(speaker), float(pace), _ptr(dur_tgt),
                                                _ptr(pitch_tgt), _ptr(energy_tgt), float(pitch_mul), float(pitch_add),
                                                float(max_duration), _ptr(enc), _ptr(dur_pred), _ptr(pitch_pred),
                                                _ptr(energy_pred), _ptr(reps), _ptr(dec_lens), _ptr(ws), nb, _stream()),
                    'fastpitch_encode')
            if lens_hook is None:
                t_max0 = int(dec_lens.max().item())     # the reference syncs here too (model.py:76)
            else:
                t_max0 = int(max(lens_hook(dec_lens), default=0))
            t_max = (t_max0 + 3) & ~3 if pad else t_max0
            x = torch.empty(B, d, t_max, dtype=torch.float32, device=dev)
            idx = torch.empty(B, t_max, dtype=torch.int32, device=dev) if return_idx else None
            mel = torch.empty(B, self.n_mel, t_max, dtype=torch.float32, device=dev)
            if t_max > 0:
                L.check(lib.ttsamd_length_regulate(_ptr(enc), _ptr(reps), B, Lt, d, t_max, _ptr(x), _ptr(idx), _stream()),
                        'length_regulate')
                nb = lib.ttsamd_fastpitch_decode_workspace_bytes(self.handle, B, t_max)
                ws = self.ws.get(nb, dev)
                L.check(lib.ttsamd_fastpitch_decode(self.handle, _ptr(x), _ptr(dec_lens), B, t_max, _ptr(mel), _ptr(ws), nb,
                                                    _stream()), 'fastpitch_decode')
        if t_max != t_max0:
            mel = mel[:, :, :t_max0]
            idx = None if idx is None else idx[:, :t_max0]
        out = (mel, dec_lens, dur_pred, pitch_pred, energy_pred)
        return out + (idx,) if return_idx else out


class DenoiserEngine:
    """Handle over ttsamd_denoiser_* (replaces vocoder.hifigan.denoiser.Denoiser)."""

    def __init__(self, device='cuda'):
        self.lib = _require_gpu()
        self.device = torch.device(device if device != 'cuda' else 'cuda:0')
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ttsamd_denoiser_create(C.byref(handle)), 'denoiser_create')
        self.handle = handle
        self.ws = _Workspace()

    def __del__(self):
        if getattr(self, 'handle', None):
            self.lib.ttsamd_denoiser_destroy(self.handle)
            self.handle = None

    def bias_spec(self, audio):
        """audio [n] (vocoder output for a zero mel) -> |STFT| of frame 0, shape [1, 513, 1]."""
        audio = _f32(audio, self.device).reshape(-1)
        n = audio.numel()
        out = torch.empty(513, dtype=torch.float32, device=self.device)
        n_dev = torch.tensor([n], dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            nb = self.lib.ttsamd_denoiser_workspace_bytes(1, n)
            ws = self.ws.get(nb, self.device)
            L.check(self.lib.ttsamd_denoiser_bias_spec(self.handle, _ptr(audio), _ptr(n_dev), n, _ptr(out), _ptr(ws), nb,
                                                       _stream()), 'denoiser_bias_spec')
        return out.reshape(1, 513, 1)

    def denoise(self, wave, nsamples, bias_spec, strength):
        """wave [B, n_max] (modified in place and returned), nsamples int64 [B] on the device."""
        assert wave.is_contiguous() and wave.dtype == torch.float32
        B, n_max = wave.shape
        nsamples = nsamples.to(device=self.device, dtype=torch.int64).contiguous()
        bias = _f32(bias_spec, self.device).reshape(-1)
        with torch.cuda.device(self.device):
            nb = self.lib.ttsamd_denoiser_workspace_bytes(B, n_max)
            ws = self.ws.get(nb, self.device)
            L.check(self.lib.ttsamd_denoise(self.handle, _ptr(wave), n_max, _ptr(nsamples), B, n_max, _ptr(bias),
                                            float(strength), _ptr(ws), nb, _stream()), 'denoise')
        return wave


class VocosEngine:
    """Handle over ttsamd_vocos_* (replaces vocoder.vocos.pretrained.MelVocos('22k'))."""

    def __init__(self, state_dict, config=None, device='cuda'):
        self.lib = _require_gpu()
        self.device = torch.device(device if device != 'cuda' else 'cuda:0')
        c = dict(VOCOS_22K_CONFIG if config is None else config)
        assert c['n_fft'] == 1024 and c['hop_length'] == 256 and c['padding'] == 'same'
        self.hop, self.n_mels = c['hop_length'], c['input_channels']
        arr, keep = L.make_tensors(state_dict)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ttsamd_vocos_create(arr, len(arr), c['input_channels'], c['dim'], c['intermediate_dim'],
                                                 c['num_layers'], C.byref(handle)), 'vocos_create')
        self.handle = handle
        self.ws = _Workspace()
        self._bias = None

    def __del__(self):
        if getattr(self, 'handle', None):
            self.lib.ttsamd_vocos_destroy(self.handle)
            self.handle = None

    def bias_vec(self):
        if self._bias is None:
            out = torch.empty(513, dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                nb = self.lib.ttsamd_vocos_workspace_bytes(self.handle, 1, 88)
                ws = self.ws.get(nb, self.device)
                L.check(self.lib.ttsamd_vocos_bias_vec(self.handle, _ptr(out), _ptr(ws), nb, _stream()), 'vocos_bias_vec')
            self._bias = out.reshape(1, 513, 1)
        return self._bias

    def forward(self, mel, lens=None, denoise=0.0):
        """mel [B,80,T] on the GPU, lens int64 [B] or None -> wave [B, 256*T] (zeros past 256*lens[b])."""
        mel = _f32(mel, self.device)
        B, M, T0 = mel.shape
        assert M == self.n_mels
        if lens is None:
            lens = torch.full((B,), T0, dtype=torch.int64, device=self.device)
        lens = lens.to(device=self.device, dtype=torch.int64).contiguous()
        if T0 == 0:
            return torch.zeros(B, 0, dtype=torch.float32, device=self.device)
        # frame rows padded to a multiple of 4 (16 bytes): the conv engine's fast paths (the k = 1 GEMM route, float4 row epilogues) need aligned
        # rows and a real T is a multiple of 4 one time in four; every layer reads frames >= lens[b] as zero, so the extra columns change nothing
        T = (T0 + 3) & ~3
        if T != T0:
            mel = torch.nn.functional.pad(mel, (0, T - T0))
        wave = torch.zeros(B, self.hop * T, dtype=torch.float32, device=self.device)
        bias = self.bias_vec().reshape(-1) if denoise != 0 else None
        with torch.cuda.device(self.device):
            nb = self.lib.ttsamd_vocos_workspace_bytes(self.handle, B, T)
            ws = self.ws.get(nb, self.device)
            L.check(self.lib.ttsamd_vocos_forward(self.handle, _ptr(mel), _ptr(lens), B, T, float(denoise), _ptr(bias),
                                                  _ptr(wave), _ptr(ws), nb, _stream()), 'vocos_forward')
        return wave if T == T0 else wave[:, :self.hop * T0]


class Tacotron2Engine:
    """Handle over ttsamd_tacotron2_* (replaces Tacotron2MS.infer, models/tacotron2/tacotron2_ms.py:279-332)."""

    def __init__(self, state_dict, config=None, device='cuda'):
        self.lib = _require_gpu()
        self.device = torch.device(device if device != 'cuda' else 'cuda:0')
        c = dict(TACOTRON2_CONFIG if config is None else config)
        self.config = c
        cfg = L.Tacotron2Cfg()
        for name, _ in L.Tacotron2Cfg._fields_:
            setattr(cfg, name, int(bool(c.get(name, True))) if name == 'decoder_early_stopping' else c[name])
        arr, keep = L.make_tensors(state_dict)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ttsamd_tacotron2_create(arr, len(arr), C.byref(cfg), C.byref(handle)), 'tacotron2_create')
        self.handle = handle
        self.ws = _Workspace()
        self.n_mels = c['n_mels']
        self.max_decoder_steps = c.get('decoder_max_step', 2000)

    def __del__(self):
        if getattr(self, 'handle', None):
            self.lib.ttsamd_tacotron2_destroy(self.handle)
            self.handle = None

    def infer(self, tokens, speaker_ids=None, lengths=None, max_step=None, dropout_seed=-1):
        """tokens int64 [B,L] -> (mel_postnet [B,80,T], mel_lens int32 [B], alignments [B,T,L]);
        dropout_seed None draws a fresh seed per call (the reference's always-on prenet dropout)."""
        _check_ids(tokens, self.config['n_symbol'], 'Tacotron2 tokens')
        if speaker_ids is not None and self.config['num_speakers'] > 1:
            _check_ids(speaker_ids, self.config['num_speakers'], 'Tacotron2 speaker_ids')
        tokens = tokens.to(device=self.device, dtype=torch.int64).contiguous()
        B, Ltok = tokens.shape
        if lengths is None:
            lengths = torch.full((B,), Ltok, dtype=torch.int64)
        lengths = lengths.to(device=self.device, dtype=torch.int64).contiguous()
        if self.config['num_speakers'] > 1:
            if speaker_ids is None:
                speaker_ids = torch.zeros(B, dtype=torch.int64)
            speaker_ids = speaker_ids.to(device=self.device, dtype=torch.int64).contiguous()
        else:
            speaker_ids = None
        max_step = int(self.max_decoder_steps if max_step is None else max_step)
        if dropout_seed is None:
            dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        mel_post = torch.zeros(B, self.n_mels, max_step, dtype=torch.float32, device=self.device)
        mel_raw = torch.zeros(B, self.n_mels, max_step, dtype=torch.float32, device=self.device)
        mel_lens = torch.zeros(B, dtype=torch.int32, device=self.device)
        align = torch.zeros(B, max_step, Ltok, dtype=torch.float32, device=self.device)
        n_steps = C.c_int32(0)
        with torch.cuda.device(self.device):
            nb = self.lib.ttsamd_tacotron2_workspace_bytes(self.handle, B, Ltok, max_step)
            ws = self.ws.get(nb, self.device)
            L.check(self.lib.ttsamd_tacotron2_infer(self.handle, _ptr(tokens), _ptr(lengths), _ptr(speaker_ids), B, Ltok,
                                                    max_step, int(dropout_seed), _ptr(mel_post), _ptr(mel_lens),
                                                    _ptr(align), _ptr(mel_raw), C.byref(n_steps), _ptr(ws), nb,
                                                    _stream()), 'tacotron2_infer')
        T = n_steps.value
        return mel_post[:, :, :T], mel_lens, align[:, :T]


class TaggerEngine:
    """Handle over ttsamd_tagger_* (replaces Shakkelha.forward / Shakkala.forward of models/diacritizers)."""

    RENAME = {'emb0.weight': 'emb.weight', 'emb_input.weight': 'emb.weight'}

    def __init__(self, state_dict, config, device='cuda'):
        self.lib = _require_gpu()
        self.device = torch.device(device if device != 'cuda' else 'cuda:0')
        c = dict(config)
        cfg = L.TaggerCfg()
        cfg.n_vocab, cfg.emb_dim = c['n_vocab'], c['emb_dim']
        cfg.n_lstm, cfg.n_dense = len(c['lstm_hidden']), len(c['dense_dim'])
        for i, v in enumerate(c['lstm_hidden']):
            cfg.lstm_hidden[i] = v
        for i, v in enumerate(c['dense_dim']):
            cfg.dense_dim[i] = v
        cfg.hard_sigmoid, cfg.bn_after_lstm0, cfg.bn_eps = c['hard_sigmoid'], c['bn_after_lstm0'], c['bn_eps']
        self.n_classes = c['dense_dim'][-1]
        self.n_vocab = c['n_vocab']
        arr, keep = L.make_tensors({self.RENAME.get(k, k): v for k, v in state_dict.items()})
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.ttsamd_tagger_create(arr, len(arr), C.byref(cfg), C.byref(handle)), 'tagger_create')
        self.handle = handle
        self.ws = _Workspace()

    def __del__(self):
        if getattr(self, 'handle', None):
            self.lib.ttsamd_tagger_destroy(self.handle)
            self.handle = None

    def forward(self, ids):
        """ids int64 [B, T] -> probs [B, T, n_classes] on the device"""
        ids = torch.as_tensor(ids)
        _check_ids(ids, self.n_vocab, 'tagger ids')
        ids = ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, T = ids.shape
        probs = torch.empty(B, T, self.n_classes, dtype=torch.float32, device=self.device)
        if T == 0:
            return probs
        with torch.cuda.device(self.device):
            nb = self.lib.ttsamd_tagger_workspace_bytes(self.handle, B, T)
            ws = self.ws.get(nb, self.device)
            L.check(self.lib.ttsamd_tagger_forward(self.handle, _ptr(ids), B, T, _ptr(probs), _ptr(ws), nb, _stream()),
                    'tagger_forward')
        return probs


def conv1d(x, w, bias=None, lens=None, dilation=1, in_slope=1.0, relu_out=False, res=None, mode=0, div=1.0, y=None):
    """Kernel-level entry (parity tests / roofline bench): y = act(conv1d(lrelu(x), w) + b [+ res]), 'same' padding;
    mode 1 / 2: y <- y + that / (y + that) / div (the ResBlock sum of HiFi-GAN); res may be y itself (in place)."""
    lib = _require_gpu()
    x = x.contiguous().float()
    w = w.contiguous().float()
    B, cin, lin = x.shape
    cout, cin2, k = w.shape
    assert cin == cin2
    if y is None:
        y = torch.zeros(B, cout, lin, dtype=torch.float32, device=x.device)
    packed = torch.empty(lib.ttsamd_conv1d_packed_floats(cout, cin, k), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.ttsamd_conv1d_ex(_ptr(x), _ptr(w), _ptr(bias), _ptr(res), _ptr(lens), B, cin, cout, k, dilation, lin,
                                     float(in_slope), int(relu_out), int(mode), float(div), _ptr(y), _ptr(packed), _stream()), 'conv1d')
    return y


def resblock_pair(x, w1, b1, w2, b2, dil, lens=None, len_mul=1, y=None, mode=0, div=1.0, slope=0.1, variant=2):
    """Kernel-level entry (parity tests / roofline bench): one fused c1 -> c2 pair of a ResBlock1 in exact fp32,
    v = x + conv1d(lrelu(conv1d(lrelu(x), w1, dilation=dil) + b1), w2) + b2; y = v | y + v | (y + v) / div (mode 0 | 1 | 2)."""
    lib = _require_gpu()
    x = x.contiguous().float()
    w1, w2, b1, b2 = (t.contiguous().float() for t in (w1, w2, b1, b2))
    B, Cc, Lx = x.shape
    k = w1.shape[2]
    assert tuple(w1.shape) == tuple(w2.shape) == (Cc, Cc, k)
    if y is None:
        assert mode == 0
        y = torch.zeros_like(x)
    if lens is not None:
        lens = lens.to(device=x.device, dtype=torch.int64).contiguous()
    n_packed = int(lib.ttsamd_resblock_pair_packed_floats(Cc, k, int(variant)))      # variants 4 / 5: + the convs' Winograd groups
    packed = torch.empty(max(n_packed, 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.ttsamd_resblock_pair(_ptr(x), _ptr(y), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), Cc, k, int(dil), _ptr(lens),
                                         int(len_mul), Lx, B, int(mode), float(div), float(slope), int(variant), _ptr(packed),
                                         n_packed, _stream()), 'resblock_pair')
    return y


def length_regulate(enc, reps, t_max):
    lib = _require_gpu()
    enc = enc.contiguous().float()
    reps = reps.contiguous().to(torch.int64)
    B, Cc, Lt = enc.shape
    out = torch.empty(B, Cc, t_max, dtype=torch.float32, device=enc.device)
    idx = torch.empty(B, t_max, dtype=torch.int32, device=enc.device)
    with torch.cuda.device(enc.device):
        L.check(lib.ttsamd_length_regulate(_ptr(enc), _ptr(reps), B, Lt, Cc, t_max, _ptr(out), _ptr(idx), _stream()),
                'length_regulate')
    return out, idx
