"""Drop-in for the reference's models/tacotron2/networks.py: same classes, signatures, defaults and
return conventions, with the arithmetic routed to the HIP engines.

  Tacotron2      (reference :71-253)  checkpoint -> .infer/.ttmel/.ttmel_single/.ttmel_batch
  Tacotron2Wave  (reference :256-426) + HiFi-GAN + denoiser -> .tts/.tts_single/.tts_batch
  text_collate_fn (:16-35), needs_postprocessing (:38-40), truncate_mel (:43-48), resize_mel (:51-66)

Differences, host-side only: `tts_batch` vocodes the (truncated / resized, hence ragged) mels in ONE
ragged HiFi-GAN call instead of a per-mel loop (:340-346) — identical results, every layer pads at the
true utterance edge — `vowelizer=` runs the Shakkelha / Shakkala taggers of models/diacritizers on the HIP tagger engine.
"""
from typing import List, Optional, Union

import torch
import torch.nn as nn

import text
from text.symbols import EOS_TOKENS, SEPARATOR_TOKEN
from utils import get_basic_config
from models.diacritizers import load_vowelizer
from vocoder import load_hifigan
from vocoder.hifigan.denoiser import Denoiser

from .tacotron2_ms import Tacotron2MS


def text_collate_fn(batch: List[torch.Tensor]):
    """Sort by length (descending), zero-pad; returns (ids_pad, lens_sorted, reverse_ids)."""
    lens_sorted, sort_ids = torch.sort(torch.LongTensor([len(x) for x in batch]), descending=True)
    ids_pad = torch.zeros(len(batch), int(lens_sorted[0]), dtype=torch.long)
    for i, j in enumerate(sort_ids):
        ids_pad[i, :batch[j].size(0)] = batch[j]
    return ids_pad, lens_sorted, sort_ids.argsort()


_OPEN_ENDINGS = frozenset(['a', 'i', 'u', 'aa', 'ii', 'uu', 'n', 'm', 'h'])


def needs_postprocessing(token: str):
    """True when the utterance ends in a phoneme the model tends to drag out (everything but
    short/long vowels and n, m, h): a separator is inserted and the mel cut at its attention peak."""
    return token not in _OPEN_ENDINGS


def truncate_mel(mel_spec: torch.Tensor, ps_end):
    """Cut [F, T] at the first frame whose attention on the inserted separator reaches 80 % of its
    maximum, then repeat the last kept frame 3 times (reference :43-48)."""
    hit = (ps_end >= 0.8 * ps_end.max()).nonzero()
    n_end = int(hit[0])
    mel_cut = mel_spec[:, :n_end]
    return torch.nn.functional.pad(mel_cut, (0, 3), mode='replicate')


def resize_mel(mel: torch.Tensor, rate: Union[int, float] = 1.0, mode: str = 'bicubic'):
    """[F, T] -> [F, int(T / rate)] by image interpolation (reference :51-66)."""
    n_f, n_t = mel.shape[-2:]
    n_t_new = int(1 / rate * n_t)
    if n_t == n_t_new:
        return mel
    return torch.nn.functional.interpolate(mel[None, None, ...], (n_f, n_t_new), mode=mode)[0, 0]


class Tacotron2(Tacotron2MS):
    def __init__(self, checkpoint: str = None, n_symbol: int = 40, decoder_max_step: int = 3000,
                 arabic_in: bool = True, vowelizer: Optional[str] = None, **kwargs):
        super().__init__(n_symbol=n_symbol, decoder_max_step=decoder_max_step, **kwargs)
        self.n_eos = len(EOS_TOKENS)
        self.arabic_in = arabic_in
        state_dicts = None
        if checkpoint is not None:
            state_dicts = torch.load(checkpoint, map_location='cpu')
            self.load_state_dict(state_dicts['model'])
        self.config = get_basic_config()
        self.vowelizers = {}
        if vowelizer is not None:
            self.vowelizers[vowelizer] = load_vowelizer(vowelizer, self.config)
        self.default_vowelizer = vowelizer
        self.phon_to_id = None
        if state_dicts is not None and 'symbols' in state_dicts:
            self.phon_to_id = {phon: i for i, phon in enumerate(state_dicts['symbols'])}
        self.eval()

    def _vowelize(self, utterance: str, vowelizer=None):
        """Optional diacritization pre-step (reference :77-87): Buckwalter -> Arabic -> tagger.predict."""
        vowelizer = self.default_vowelizer if vowelizer is None else vowelizer
        if vowelizer is None:
            return utterance
        if vowelizer not in self.vowelizers:
            self.vowelizers[vowelizer] = load_vowelizer(vowelizer, self.config)
        tagger = self.vowelizers[vowelizer]
        if tagger.device != self.device:          # the dict is not a registered submodule, .to() does not reach it
            tagger.to(self.device)
        return tagger.predict(text.buckwalter_to_arabic(utterance))

    def _tokenize(self, utterance: str, vowelizer=None):
        utterance = self._vowelize(utterance, vowelizer)
        if self.arabic_in:
            return text.arabic_to_tokens(utterance)
        return text.buckwalter_to_tokens(utterance)

    def _tokens_for(self, utterance, vowelizer, postprocess_mel):
        """tokens (+ the extra separator before the EOS tokens when the ending needs the cut)"""
        tokens = self._tokenize(utterance, vowelizer=vowelizer)
        process_mel = False
        if postprocess_mel and needs_postprocessing(tokens[-self.n_eos - 1]):
            tokens.insert(-self.n_eos, SEPARATOR_TOKEN)
            process_mel = True
        return tokens, process_mel

    @torch.inference_mode()
    def ttmel_single(self, utterance: str, speaker_id: int = 0, speed: Union[int, float, None] = None,
                     vowelizer=None, postprocess_mel: bool = True):
        tokens, process_mel = self._tokens_for(utterance, vowelizer, postprocess_mel)
        ids_batch = torch.LongTensor(text.tokens_to_ids(tokens, self.phon_to_id)).unsqueeze(0)
        sid = torch.LongTensor([speaker_id])
        mel_spec, _, alignments = self.infer(ids_batch, sid)
        mel_spec = mel_spec[0]
        if process_mel:
            mel_spec = truncate_mel(mel_spec, alignments[0, :, -self.n_eos - 1])
        if speed is not None:
            mel_spec = resize_mel(mel_spec, rate=speed)
        return mel_spec                                                  # [80, T]

    @torch.inference_mode()
    def ttmel_batch(self, batch: List[str], speaker_id: int = 0, speed: Union[int, float, None] = None,
                    vowelizer=None, postprocess_mel: bool = True):
        prepared = [self._tokens_for(line, vowelizer, postprocess_mel) for line in batch]
        batch_ids = [torch.LongTensor(text.tokens_to_ids(tokens, self.phon_to_id)) for tokens, _ in prepared]
        ids_pad, lens_sorted, reverse_ids = text_collate_fn(batch_ids)
        sids = lens_sorted * 0 + speaker_id
        mel_post, mel_lens, alignments = self.infer(ids_pad, sids, lens_sorted)
        mel_lens, in_lens = mel_lens.tolist(), lens_sorted.tolist()
        mel_list = []
        for i, j in enumerate(reverse_ids.tolist()):
            mel = mel_post[j, :, :mel_lens[j]]
            if prepared[i][1]:
                mel = truncate_mel(mel, alignments[j, :mel_lens[j], in_lens[j] - self.n_eos - 1])
            if speed is not None:
                mel = resize_mel(mel, rate=speed)
            mel_list.append(mel)
        return mel_list

    def ttmel(self, text_input: Union[str, List[str]], speaker_id: int = 0, speed: Union[int, float, None] = None,
              batch_size: int = 8, vowelizer=None, postprocess_mel: bool = True):
        args = (speaker_id, speed, vowelizer, postprocess_mel)
        if isinstance(text_input, str):
            return self.ttmel_single(text_input, *args)
        assert isinstance(text_input, list)
        if batch_size == 1:
            return [self.ttmel_single(sample, *args) for sample in text_input]
        mel_list = []
        for k in range(0, len(text_input), batch_size):
            mel_list += self.ttmel_batch(text_input[k:k + batch_size], *args)
        return mel_list


class Tacotron2Wave(nn.Module):
    def __init__(self, model_sd_path: str, vocoder_sd: Optional[str] = None, vocoder_config: Optional[str] = None,
                 vowelizer: Optional[str] = None, arabic_in: bool = True, n_symbol: int = 40):
        super().__init__()
        self.model = Tacotron2(model_sd_path, n_symbol=n_symbol, arabic_in=arabic_in, vowelizer=vowelizer)
        if vocoder_sd is None or vocoder_config is None:
            config = get_basic_config()
            vocoder_sd, vocoder_config = config.vocoder_state_path, config.vocoder_config_path
        self.vocoder = load_hifigan(vocoder_sd, vocoder_config)
        self.denoiser = Denoiser(self.vocoder)
        self.eval()

    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, x):
        return x

    @torch.inference_mode()
    def tts_single(self, text_input: str, speed: Union[int, float, None] = None, speaker_id: int = 0,
                   denoise: float = 0, vowelizer=None, postprocess_mel: bool = True, return_mel: bool = False):
        mel_spec = self.model.ttmel_single(text_input, speaker_id, speed, vowelizer, postprocess_mel)
        wave = self.vocoder(mel_spec)
        if denoise > 0:
            wave = self.denoiser(wave, denoise)
        if return_mel:
            return wave[0].cpu(), mel_spec
        return wave[0].cpu()

    @torch.inference_mode()
    def tts_batch(self, batch: List[str], speed: Union[int, float, None] = None, denoise: float = 0,
                  speaker_id: int = 0, vowelizer=None, postprocess_mel: bool = True, return_mel: bool = False):
        mel_list = self.model.ttmel_batch(batch, speaker_id, speed, vowelizer, postprocess_mel)
        eng = self.vocoder.engine()
        lens_host = [m.shape[-1] for m in mel_list]
        lens = torch.tensor(lens_host, dtype=torch.int64, device=mel_list[0].device)
        mel = torch.zeros(len(mel_list), mel_list[0].shape[0], max(lens_host), device=lens.device)
        for i, m in enumerate(mel_list):
            mel[i, :, :m.shape[-1]] = m
        wave = eng.forward(mel, lens)                                    # one ragged batched launch sequence
        n = [t * eng.hop for t in lens_host]
        if denoise > 0:
            wave = self.denoiser.forward_batch(wave, lens * eng.hop, denoise, nsamples_min=min(n))
        # one exact-size D2H per utterance (a padded [B, n_max] copy + per-row clones touches every host page twice)
        # NB the reference silently ignores return_mel here (:348-351); so do we
        return [wave[i, :n[i]].cpu() for i in range(len(mel_list))]

    def tts(self, text_buckw: Union[str, List[str]], speed: Union[int, float, None] = None, denoise: float = 0.005,
            speaker_id: int = 0, batch_size: int = 8, vowelizer=None, postprocess_mel: bool = True,
            return_mel: bool = False) -> Union[torch.Tensor, List[torch.Tensor]]:
        """Same contract as the reference (:353-426): str -> Tensor[n_samples] (CPU); list -> list of
        tensors, chunked by `batch_size`."""
        kw = dict(speaker_id=speaker_id, speed=speed, denoise=denoise, vowelizer=vowelizer,
                  postprocess_mel=postprocess_mel, return_mel=return_mel)
        if isinstance(text_buckw, str):
            return self.tts_single(text_buckw, **kw)
        assert isinstance(text_buckw, list)
        if batch_size == 1:
            return [self.tts_single(sample, **kw) for sample in text_buckw]
        wav_list = []
        for k in range(0, len(text_buckw), batch_size):
            wav_list += self.tts_batch(text_buckw[k:k + batch_size], **kw)
        return wav_list
