"""Drop-in for the inference part of the reference's vendored Vocos
(vocoder/vocos/pretrained.py:34-97 `MelVocos`, config '22k' of vocoder/vocos/__init__.py:35-67):
80-band mel -> waveform through the HIP ConvNeXt backbone + ISTFT head.  The reference never
wires it to FastPitch; `FastPitch2Wave`-style use is: `wave = vocos(mel_batch, denoise=0.)`."""
import numpy as np
import torch

from ttsamd.config import VOCOS_22K_CONFIG
from ttsamd.engine import VocosEngine
from ttsamd.lib import TtsAmdError
from vocoder.hifigan.models import _HipModule

config_22k = dict(VOCOS_22K_CONFIG)


class MelVocos(_HipModule):
    def __init__(self, config_name='22k'):
        super().__init__()
        if config_name != '22k':
            raise TtsAmdError("only MelVocos('22k') (80 mel bands, 22.05 kHz) is built")
        self.n_mels = config_22k['input_channels']
        self._sd = None

    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: (v.detach().cpu().float().numpy() if hasattr(v, 'detach') else np.asarray(v, np.float32))
                    for k, v in state_dict.items() if k.startswith(('backbone.', 'head.out.'))}
        self._engines.clear()

    def state_dict(self, *a, **k):
        return {k_: torch.from_numpy(v) for k_, v in (self._sd or {}).items()}

    def engine(self):
        if self._sd is None:
            raise TtsAmdError('MelVocos has no weights: call load_state_dict first')
        return self._engine(lambda dev: VocosEngine(self._sd, config_22k, device=dev))

    @property
    def bias_vec(self):
        return self.engine().bias_vec()

    @torch.inference_mode()
    def forward(self, mel_spec, denoise=0., lens=None):
        """mel_spec [B, 80, frames] -> wave [B, 256*frames]  (pretrained.py:73-93)."""
        return self.engine().forward(mel_spec, lens, denoise)
