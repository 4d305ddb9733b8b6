"""ctypes binding of libttsamd.so — one Python function per symbol of include/ttsamd.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TTSAMD_LIB: another build of the SAME sources (tools/f2_exp.sh: a timing build in /tmp); the product path never sets it
LIB_PATH = os.environ.get('TTSAMD_LIB') or os.path.join(_HERE, 'lib', 'libttsamd.so')


class TtsAmdError(RuntimeError):
    pass


class Tensor(C.Structure):
    _fields_ = [('name', C.c_char_p), ('data', C.POINTER(C.c_float)), ('ndim', C.c_int32),
                ('shape', C.c_int64 * 4)]


class HifiGanCfg(C.Structure):
    _fields_ = [('num_mels', C.c_int32), ('upsample_initial_channel', C.c_int32), ('n_ups', C.c_int32),
                ('upsample_rates', C.c_int32 * 8), ('upsample_kernel_sizes', C.c_int32 * 8),
                ('n_kernels', C.c_int32), ('resblock_kernel_sizes', C.c_int32 * 8),
                ('n_dilations', C.c_int32), ('resblock_dilations', (C.c_int32 * 8) * 8)]


class FastPitchCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'n_mel_channels', 'n_symbols', 'padding_idx', 'd_model',
        'in_fft_n_layers', 'in_fft_n_heads', 'in_fft_d_head', 'in_fft_kernel', 'in_fft_filter',
        'out_fft_n_layers', 'out_fft_n_heads', 'out_fft_d_head', 'out_fft_kernel', 'out_fft_filter',
        'dur_kernel', 'dur_filter', 'dur_n_layers',
        'pitch_kernel', 'pitch_filter', 'pitch_n_layers', 'pitch_emb_kernel',
        'energy_conditioning', 'energy_kernel', 'energy_filter', 'energy_n_layers', 'energy_emb_kernel',
        'n_speakers')] + [('speaker_emb_weight', C.c_float)]


class Tacotron2Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'n_symbol', 'num_speakers', 'speaker_embedding_dim', 'symbol_embedding_dim', 'encoder_embedding_dim',
        'encoder_n_convolution', 'encoder_kernel_size', 'n_mels', 'prenet_dim', 'attention_rnn_dim',
        'decoder_rnn_dim', 'attention_hidden_dim', 'attention_location_n_filter',
        'attention_location_kernel_size', 'postnet_n_convolution', 'postnet_kernel_size',
        'postnet_embedding_dim')] + [('gate_threshold', C.c_float), ('decoder_early_stopping', C.c_int32)]


class TaggerCfg(C.Structure):
    _fields_ = [('n_vocab', C.c_int32), ('emb_dim', C.c_int32), ('n_lstm', C.c_int32), ('lstm_hidden', C.c_int32 * 4),
                ('hard_sigmoid', C.c_int32), ('bn_after_lstm0', C.c_int32), ('bn_eps', C.c_float),
                ('n_dense', C.c_int32), ('dense_dim', C.c_int32 * 4)]


# every symbol include/ttsamd.h declares: name -> (restype, argtypes)
_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    'ttsamd_last_error': (C.c_char_p, []),
    'ttsamd_version': (_I32, []),
    'ttsamd_set_option': (_I32, [C.c_char_p, C.c_char_p]),
    'ttsamd_get_option': (_I32, [C.c_char_p, C.c_char_p, _I32]),
    'ttsamd_option_name': (C.c_char_p, [_I32]),
    'ttsamd_options_check': (_I32, []),
    'ttsamd_device_ok': (_I32, []),
    'ttsamd_hifigan_create': (_I32, [C.POINTER(Tensor), _I32, C.POINTER(HifiGanCfg), C.POINTER(_P)]),
    'ttsamd_hifigan_destroy': (_I32, [_P]),
    'ttsamd_hifigan_workspace_bytes': (_I64, [_P, _I32, _I32]),
    'ttsamd_hifigan_forward': (_I32, [_P, _P, _P, _I32, _I32, _P, _P, _I64, _P]),
    'ttsamd_fastpitch_create': (_I32, [C.POINTER(Tensor), _I32, C.POINTER(FastPitchCfg), C.POINTER(_P)]),
    'ttsamd_fastpitch_destroy': (_I32, [_P]),
    'ttsamd_fastpitch_encode_workspace_bytes': (_I64, [_P, _I32, _I32]),
    'ttsamd_fastpitch_decode_workspace_bytes': (_I64, [_P, _I32, _I32]),
    'ttsamd_fastpitch_encode': (_I32, [_P, _P, _I32, _I32, _I32, _F, _P, _P, _P, _F, _F, _F,
                                       _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    'ttsamd_length_regulate': (_I32, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    'ttsamd_fastpitch_decode': (_I32, [_P, _P, _P, _I32, _I32, _P, _P, _I64, _P]),
    'ttsamd_fastpitch_set_batch_mode': (_I32, [_P, _I32]),
    'ttsamd_denoiser_create': (_I32, [C.POINTER(_P)]),
    'ttsamd_denoiser_destroy': (_I32, [_P]),
    'ttsamd_denoiser_workspace_bytes': (_I64, [_I32, _I32]),
    'ttsamd_denoiser_bias_spec': (_I32, [_P, _P, _P, _I32, _P, _P, _I64, _P]),
    'ttsamd_denoise': (_I32, [_P, _P, _I64, _P, _I32, _I32, _P, _F, _P, _I64, _P]),
    'ttsamd_vocos_create': (_I32, [C.POINTER(Tensor), _I32, _I32, _I32, _I32, _I32, C.POINTER(_P)]),
    'ttsamd_vocos_destroy': (_I32, [_P]),
    'ttsamd_vocos_workspace_bytes': (_I64, [_P, _I32, _I32]),
    'ttsamd_vocos_bias_vec': (_I32, [_P, _P, _P, _I64, _P]),
    'ttsamd_vocos_forward': (_I32, [_P, _P, _P, _I32, _I32, _F, _P, _P, _P, _I64, _P]),
    'ttsamd_tacotron2_create': (_I32, [C.POINTER(Tensor), _I32, C.POINTER(Tacotron2Cfg), C.POINTER(_P)]),
    'ttsamd_tacotron2_destroy': (_I32, [_P]),
    'ttsamd_tacotron2_workspace_bytes': (_I64, [_P, _I32, _I32, _I32]),
    'ttsamd_tacotron2_infer': (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I64, _P, _P, _P, _P,
                                      C.POINTER(_I32), _P, _I64, _P]),
    'ttsamd_tagger_create': (_I32, [C.POINTER(Tensor), _I32, C.POINTER(TaggerCfg), C.POINTER(_P)]),
    'ttsamd_tagger_destroy': (_I32, [_P]),
    'ttsamd_tagger_workspace_bytes': (_I64, [_P, _I32, _I32]),
    'ttsamd_tagger_forward': (_I32, [_P, _P, _I32, _I32, _P, _P, _I64, _P]),
    'ttsamd_conv1d_packed_floats': (_I64, [_I32, _I32, _I32]),
    'ttsamd_conv1d': (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _I32, _P, _P, _P]),
    'ttsamd_conv1d_ex': (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _I32, _I32, _F, _P, _P, _P]),
    'ttsamd_resblock_pair_packed_floats': (_I64, [_I32, _I32, _I32]),
    'ttsamd_resblock_pair': (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _I32, _I32, _I32, _I32, _F, _F, _I32, _P, _I64, _P]),
    'ttsamd_bfo_pack': (_I32, [_P, _I32, _I32, _I32, _F, _P, _P]),
    'ttsamd_bfo_unpack': (_I32, [_P, _I32, _I32, _I32, _F, _P, _P]),
    'ttsamd_bfo_weight_elems': (_I64, [_I32, _I32, _I32, _I32]),
    'ttsamd_bfo_pack_weight': (_I32, [_P, _I32, _I32, _I32, _I32, _P]),
    'ttsamd_bfo_conv1d': (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _F, _F, _P, _P, _P, _P]),
    'ttsamd_bfo_resblock_pair': (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _F, _F, _F, _P, _P]),
    'ttsamd_bfo_resblock_chain': (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _F, _F, _F, _P, _P, _I32]),
    'ttsamd_bfo_conv_post': (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _I64, _P]),
    'ttsamd_bfo3_pack': (_I32, [_P, _I32, _I32, _I32, _F, _P, _P]),
    'ttsamd_bfo3_unpack': (_I32, [_P, _I32, _I32, _I32, _F, _P, _P]),
    'ttsamd_bfo3_weight_elems': (_I64, [_I32, _I32, _I32, _I32]),
    'ttsamd_bfo3_pack_weight': (_I32, [_P, _I32, _I32, _I32, _I32, _P]),
    'ttsamd_bfo3_conv1d': (_I32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _F, _F, _P, _P, _P, _P]),
    'ttsamd_bfo3_resblock_pair': (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _F, _F, _F, _F, _P, _P]),
    'ttsamd_bfo3_resblock_chain': (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _F, _F, _F, _P, _P]),
    'ttsamd_bfo3_conv_post': (_I32, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _I64, _P]),
    'ttsamd_set_precision': (_I32, [_I32]),
    'ttsamd_get_precision': (_I32, []),
    'ttsamd_dp_unique_id': (_I32, [_P]),
    'ttsamd_dp_init': (_I32, [_I32, _I32, _P, C.POINTER(_P)]),
    'ttsamd_dp_destroy': (_I32, [_P]),
    'ttsamd_dp_rank': (_I32, [_P]),
    'ttsamd_dp_world': (_I32, [_P]),
    'ttsamd_dp_broadcast': (_I32, [_P, _P, _I64, _I32, _P]),
    'ttsamd_dp_broadcast_weights': (_I32, [_P, _I32, _P, _I32, _P]),
    'ttsamd_dp_allgather': (_I32, [_P, _P, _P, _I64, _P]),
    'ttsamd_dp_pack_audio': (_I32, [_P, _I64, _P, _I32, _I64, _P, _P]),
    'ttsamd_dp_gather_audio': (_I32, [_P, _P, _P, C.POINTER(_I64), C.POINTER(_I64), _I32, _P]),
    'ttsamd_profile_enable': (_I32, [_I32]),
    'ttsamd_profile_read': (_I32, [C.POINTER(C.c_double)]),
}

_lib = None
ABI_VERSION = 7            # == TTSAMD_ABI_VERSION of include/ttsamd.h (struct layouts and argument meanings of this binding)


def load():
    """dlopen libttsamd.so (built by `make -C tts-arabic-pytorch_amd/csrc` or
    __graft_entry__.build()).  Fails loudly — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TtsAmdError(f'{LIB_PATH} is missing: build it with `make -C tts-arabic-pytorch_amd/csrc` '
                          '(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback')
    # torch bundles its own libamdhip64.so.7; it must be the first HIP runtime in the process
    # (two runtimes = "No HIP GPUs are available"), so pull it in before dlopen-ing ours, which then
    # binds to the already-loaded SONAME.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    got = lib.ttsamd_version()
    if got != ABI_VERSION:
        raise TtsAmdError(f'{LIB_PATH} was built from another revision of include/ttsamd.h (library ABI {got}, this binding '
                          f'{ABI_VERSION}): rebuild it with `make -C tts-arabic-pytorch_amd/csrc`')
    if lib.ttsamd_options_check() != 0:             # a malformed TTSAMD_<NAME> in the environment: loud, not ignored
        msg = lib.ttsamd_last_error()
        raise TtsAmdError(f'bad routing option in the environment: {msg.decode() if msg else "?"}')
    _lib = lib
    return lib


def set_option(name, value):
    """Routing option of the library (include/ttsamd.h: ttsamd_set_option): `name` with or without the TTSAMD_ prefix, `value` an int / str
    as the environment variable would hold it, None = back to the default.  Raises on an unknown name or a value out of range."""
    v = None if value is None else str(value).encode()
    check(load().ttsamd_set_option(name.encode(), v), f'set_option({name}={value})')


def get_option(name):
    """Current text of a routing option, None when unset (the default applies)."""
    buf = C.create_string_buffer(64)
    check(load().ttsamd_get_option(name.encode(), buf, 64), f'get_option({name})')
    return buf.value.decode() or None


def option_names():
    lib, out, i = load(), [], 0
    while True:
        n = lib.ttsamd_option_name(i)
        if not n:
            return out
        out.append(n.decode())
        i += 1


class options:
    """Context manager: set routing options for a block and restore what was there before.  `with options(TTSAMD_WINO=0): ...`"""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kv}
        for k, v in self.kv.items():
            set_option(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            set_option(k, v)


def check(rc, what):
    if rc != 0:
        msg = load().ttsamd_last_error()
        raise TtsAmdError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')


def make_tensors(state_dict):
    """{name: np.float32 array or torch tensor} -> (ctypes array of Tensor, keep-alive list)."""
    import numpy as np
    keep, items = [], []
    for name, v in state_dict.items():
        if hasattr(v, 'detach'):
            if not v.is_floating_point():
                continue
            v = v.detach().cpu().float().numpy()
        a = np.ascontiguousarray(v, dtype=np.float32)
        if a.ndim > 4:
            continue
        t = Tensor()
        nb = name.encode()
        t.name = nb
        t.data = a.ctypes.data_as(C.POINTER(C.c_float))
        t.ndim = a.ndim
        for i, s in enumerate(a.shape):
            t.shape[i] = s
        keep += [a, nb]
        items.append(t)
    arr = (Tensor * len(items))(*items)
    return arr, keep
