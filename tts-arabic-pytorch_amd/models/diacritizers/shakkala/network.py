"""Drop-in for models/diacritizers/shakkala/network.py:8-77: Embedding(149,288) -> BiLSTM-hard-sigmoid(288) ->
BatchNorm1d(576, eps 1e-3) -> BiLSTM-hs(144) -> BiLSTM-hs(96) -> Dense(28) -> softmax on the HIP tagger."""
import torch

from ttsamd.config import SHAKKALA_CONFIG

from ..shakkelha.network import _Tagger
from . import decode, encode


class Shakkala(_Tagger):
    CONFIG = SHAKKALA_CONFIG

    def __init__(self, dim_input: int = 149, dim_output: int = 28, sd_path: str = None):
        assert dim_input == 149 and dim_output == 28, 'only the shipped Shakkala geometry is built'
        super().__init__(sd_path)
        self.max_sentence = None

    def _predict_single(self, input_text: str, return_probs: bool = False):
        ids_pad, ids = encode(input_text, self.max_sentence)
        probs = self.infer(torch.LongTensor(ids_pad)[None]).cpu()
        output = decode(probs, input_text, ids)
        return (output, probs) if return_probs else output
