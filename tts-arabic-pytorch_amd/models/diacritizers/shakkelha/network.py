"""Drop-in for models/diacritizers/shakkelha/network.py:9-79: Embedding(91,25) -> 2 x BiLSTM(256) ->
Dense(512)+ReLU x2 -> Dense(19) -> softmax, evaluated by the HIP tagger engine."""
from typing import List, Union

import torch

from ttsamd.config import SHAKKELHA_CONFIG
from ttsamd.engine import TaggerEngine
from vocoder.hifigan.models import _HipModule

from . import decode, encode


class _Tagger(_HipModule):
    CONFIG = None

    def __init__(self, sd_path: str = None):
        super().__init__()
        self._sd = None
        if sd_path is not None:
            self.load_state_dict(torch.load(sd_path, map_location='cpu'))
        self.eval()

    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: v.detach().cpu().float().numpy() for k, v in state_dict.items()
                    if torch.is_tensor(v) and v.is_floating_point()}
        self._engines.clear()

    def state_dict(self, *a, **k):
        return {k_: torch.from_numpy(v) for k_, v in (self._sd or {}).items()}

    def engine(self):
        return self._engine(lambda dev: TaggerEngine(self._sd, self.CONFIG, device=dev))

    @torch.inference_mode()
    def forward(self, x: torch.Tensor):
        """ids int64 [B, T] -> class probabilities [B, T, n_classes]"""
        return self.engine().forward(x)

    infer = forward

    def predict(self, input: Union[str, List[str]], return_probs: bool = False):
        if isinstance(input, str):
            return self._predict_single(input, return_probs=return_probs)
        return self._predict_list(input, return_probs=return_probs)

    def _predict_list(self, input_list: List[str], return_probs: bool = False):
        results = [self._predict_single(t, return_probs=return_probs) for t in input_list]
        if return_probs:
            return [r[0] for r in results], [r[1] for r in results]
        return results


class Shakkelha(_Tagger):
    CONFIG = SHAKKELHA_CONFIG

    def __init__(self, dim_input: int = 91, dim_output: int = 19, sd_path: str = None):
        assert dim_input == 91 and dim_output == 19, 'only the shipped Shakkelha geometry is built'
        super().__init__(sd_path)

    def _predict_single(self, input_text: str, return_probs: bool = False):
        ids = torch.LongTensor(encode(input_text))[None]
        probs = self.infer(ids).cpu()
        output = decode(probs, input_text)
        return (output, probs) if return_probs else output
