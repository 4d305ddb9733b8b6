"""Drop-in for the reference's models/tacotron2/tacotron2_ms.py:119-332 (Tacotron2MS): same constructor
keywords and `infer(tokens, speaker_ids, lengths)` contract, arithmetic in libttsamd (csrc/tacotron2.hip).

torchaudio's _Prenet applies dropout(p=0.5) at inference too, so the reference's output is a random
variable.  Here the mask comes from a counter-based hash of (seed, layer, step, utterance, unit):
`dropout_seed=None` (default) draws a new seed per call like the reference draws new masks,
an int >= 0 makes the call reproducible, -1 switches the dropout off.
"""
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from ttsamd.config import TACOTRON2_CONFIG
from ttsamd.engine import Tacotron2Engine
from vocoder.hifigan.models import _HipModule


class Tacotron2MS(_HipModule):
    def __init__(self, mask_padding: bool = False, n_mels: int = 80, n_symbol: int = 148, n_frames_per_step: int = 1,
                 num_speakers=40, speaker_embedding_dim=128, symbol_embedding_dim: int = 512,
                 encoder_embedding_dim: int = 512, encoder_n_convolution: int = 3, encoder_kernel_size: int = 5,
                 decoder_rnn_dim: int = 1024, decoder_max_step: int = 2000, decoder_dropout: float = 0.1,
                 decoder_early_stopping: bool = True, attention_rnn_dim: int = 1024, attention_hidden_dim: int = 128,
                 attention_location_n_filter: int = 32, attention_location_kernel_size: int = 31,
                 attention_dropout: float = 0.1, prenet_dim: int = 256, postnet_n_convolution: int = 5,
                 postnet_kernel_size: int = 5, postnet_embedding_dim: int = 512, gate_threshold: float = 0.5):
        super().__init__()
        if n_frames_per_step != 1:
            raise ValueError('Only n_frames_per_step=1 is supported')
        self.mask_padding, self.n_mels, self.n_frames_per_step = mask_padding, n_mels, n_frames_per_step
        self.taco_config = dict(TACOTRON2_CONFIG)
        self.taco_config.update(
            n_mels=n_mels, n_symbol=n_symbol, num_speakers=num_speakers, speaker_embedding_dim=speaker_embedding_dim,
            symbol_embedding_dim=symbol_embedding_dim, encoder_embedding_dim=encoder_embedding_dim,
            encoder_n_convolution=encoder_n_convolution, encoder_kernel_size=encoder_kernel_size,
            decoder_rnn_dim=decoder_rnn_dim, decoder_max_step=decoder_max_step, attention_rnn_dim=attention_rnn_dim,
            attention_hidden_dim=attention_hidden_dim, attention_location_n_filter=attention_location_n_filter,
            attention_location_kernel_size=attention_location_kernel_size, prenet_dim=prenet_dim,
            postnet_n_convolution=postnet_n_convolution, postnet_kernel_size=postnet_kernel_size,
            postnet_embedding_dim=postnet_embedding_dim, gate_threshold=gate_threshold,
            decoder_early_stopping=bool(decoder_early_stopping))
        self.decoder_max_step = decoder_max_step
        self.dropout_seed: Optional[int] = None
        self._sd = None

    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: (v.detach().cpu().float().numpy() if hasattr(v, 'detach') else np.asarray(v, np.float32))
                    for k, v in state_dict.items()
                    if not (hasattr(v, 'is_floating_point') and not v.is_floating_point())}
        self._engines.clear()

    def state_dict(self, *a, **k):
        return {k_: torch.from_numpy(v) for k_, v in (self._sd or {}).items()}

    def engine(self):
        if self._sd is None:
            from ttsamd.lib import TtsAmdError
            raise TtsAmdError('Tacotron2MS has no weights: call load_state_dict first')
        return self._engine(lambda dev: Tacotron2Engine(self._sd, self.taco_config, device=dev))

    @torch.inference_mode()
    def infer(self, tokens: Tensor, speaker_ids: Optional[Tensor] = None,
              lengths: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
        """tokens [B, L] zero-padded -> (mel_postnet [B, n_mels, T], mel_lengths [B],
        alignments [B, T, L]) with T = max(mel_lengths)   (tacotron2_ms.py:279-332)."""
        return self.engine().infer(tokens, speaker_ids, lengths, max_step=self.decoder_max_step,
                                   dropout_seed=self.dropout_seed)
