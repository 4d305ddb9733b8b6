"""Drop-in for vocoder.hifigan.denoiser.Denoiser (reference :29-72): removes the vocoder's
bias spectrum.  STFT / spectral gain / ISTFT run in libttsamd.so (csrc/denoiser.hip)."""
import torch

from ttsamd.engine import DenoiserEngine
from vocoder.hifigan.models import _HipModule


class Denoiser(_HipModule):
    def __init__(self, hifigan, filter_length=1024, n_overlap=4, win_length=1024, mode='zeros', **infer_kw):
        super().__init__()
        assert filter_length == 1024 and n_overlap == 4 and win_length == 1024, 'only the shipped 1024/256 STFT is built'
        self._mel_init = {'zeros': torch.zeros, 'normal': torch.randn}[mode]     # denoiser.py:50-51
        self._hifigan = [hifigan]            # not a sub-module: the vocoder is owned by the caller
        self._bias = {}
        dev = hifigan.device if hasattr(hifigan, 'device') else torch.device('cpu')
        if dev.type == 'cuda':               # the reference computes bias_spec eagerly on the vocoder's device
            self.to(dev)
            self._bias_spec(dev)

    def _bias_spec(self, dev):
        key = str(dev)
        if key not in self._bias:
            eng = self._engine(lambda d: DenoiserEngine(device=d))
            voc = self._hifigan[0]
            moved = voc.device != dev
            if moved:
                voc.to(dev)
            bias_audio = voc(self._mel_init((1, 80, 88), device=dev))         # denoiser.py:50-54
            self._bias[key] = eng.bias_spec(bias_audio.reshape(-1))
        return self._bias[key]

    @property
    def bias_spec(self):
        return self._bias_spec(self.device)

    @torch.inference_mode()
    def forward(self, audio, strength=0.1):
        """audio [1, n] (or [1,1,n] / [n]) -> denoised audio of the same shape (denoiser.py:66-72)."""
        shape = audio.shape
        wave = audio.float().reshape(1, -1).contiguous().clone()
        n = torch.full((1,), wave.shape[1], dtype=torch.int64, device=wave.device)      # a fill, not a (stream-ordered, host-blocking) H2D copy
        out = self.forward_batch(wave, n, strength, nsamples_min=wave.shape[1])
        n_out = (wave.shape[1] // 256) * 256
        return out[:, :n_out].reshape(shape[:-1] + (n_out,))

    @torch.inference_mode()
    def forward_batch(self, wave, nsamples, strength, nsamples_min=None):
        """Ragged batch (extension): wave [B, n_max] float32 on the GPU, nsamples int64 [B].  `nsamples_min`: the smallest
        length if the caller has it on the host already (FastPitch synchronises on the lengths anyway) -- reading it from the
        device here would stall the host until the vocoder has finished, which is what the pipelined `tts` list path overlaps."""
        if nsamples_min is None and wave.shape[0]:
            nsamples_min = int(nsamples.min())
        if wave.shape[0] and nsamples_min <= 512:
            # the reference's Spectrogram(center=True, pad_mode='reflect') needs more than n_fft/2 samples and raises
            raise ValueError('Denoiser: every utterance needs more than 512 samples (reflect padding of n_fft/2); '
                             f'shortest has {nsamples_min}')
        eng = self._engine(lambda d: DenoiserEngine(device=d))
        return eng.denoise(wave.contiguous(), nsamples, self._bias_spec(wave.device), strength)
