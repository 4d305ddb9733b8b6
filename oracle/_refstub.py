"""Test infrastructure only (never imported by the product path).

Makes the *real* reference (/root/reference, read-only, present only in the build
container) importable in a separate process: stubs the two third-party modules the
image lacks (`numba`, `torchaudio`) exactly as SURVEY.md §8(c) describes.  Used by
oracle/gen_golden.py to produce tests/golden/*.npz.  Nothing here travels as code the
product runs; the GPU box never has /root/reference.
"""
import os
import sys
import types

REF = os.environ.get("TTS_REFERENCE", "/root/reference")


def install():
    import torch

    # numba: training-only JIT decorators (models/fastpitch/fastpitch/alignment.py:16)
    numba = types.ModuleType("numba")

    def _jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba.jit = _jit
    numba.njit = _jit
    numba.prange = range
    sys.modules["numba"] = numba

    # torchaudio: only Spectrogram / InverseSpectrogram are used on the path
    # (vocoder/hifigan/denoiser.py:43-48).  Stand-in = torch.stft/istft with the
    # documented torchaudio defaults (hann periodic window, center, reflect, onesided,
    # unnormalised).  Denoiser goldens are therefore labelled "torch.stft-based".
    ta = types.ModuleType("torchaudio")
    tr = types.ModuleType("torchaudio.transforms")
    fn = types.ModuleType("torchaudio.functional")
    fnfn = types.ModuleType("torchaudio.functional.functional")

    class Spectrogram(torch.nn.Module):
        def __init__(self, n_fft, hop_length=None, win_length=None, power=2.0, **kw):
            super().__init__()
            self.n_fft, self.hop, self.win = n_fft, hop_length, win_length or n_fft
            self.power = power
            self.register_buffer("window", torch.hann_window(self.win))

        def forward(self, x):
            shp = x.shape
            y = torch.stft(x.reshape(-1, shp[-1]), self.n_fft, self.hop, self.win,
                           self.window, center=True, pad_mode="reflect",
                           normalized=False, onesided=True, return_complex=True)
            y = y.reshape(shp[:-1] + y.shape[-2:])
            if self.power is None:
                return y
            return y.abs().pow(self.power)

    class InverseSpectrogram(torch.nn.Module):
        def __init__(self, n_fft, hop_length=None, win_length=None, **kw):
            super().__init__()
            self.n_fft, self.hop, self.win = n_fft, hop_length, win_length or n_fft
            self.register_buffer("window", torch.hann_window(self.win))

        def forward(self, y, length=None):
            shp = y.shape
            x = torch.istft(y.reshape(-1, shp[-2], shp[-1]), self.n_fft, self.hop,
                            self.win, self.window, center=True, normalized=False,
                            onesided=True, length=length)
            return x.reshape(shp[:-2] + x.shape[-1:])

    class MelSpectrogram(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    fnfn._hz_to_mel = lambda *a, **k: None      # imported by vocoder/vocos/heads.py:5, used only by IMDCT heads
    fnfn._mel_to_hz = lambda *a, **k: None
    tr.Spectrogram, tr.InverseSpectrogram, tr.MelSpectrogram = \
        Spectrogram, InverseSpectrogram, MelSpectrogram
    ta.transforms, ta.functional = tr, fn
    fn.functional = fnfn
    ta.save = lambda *a, **k: None
    ta.load = lambda *a, **k: None
    for name, mod in (("torchaudio", ta), ("torchaudio.transforms", tr),
                      ("torchaudio.functional", fn),
                      ("torchaudio.functional.functional", fnfn)):
        sys.modules[name] = mod

    os.chdir(REF)                      # utils/__init__.py:32 opens a relative path
    sys.path[:] = [REF] + [p for p in sys.path if p and "repo" not in p]
