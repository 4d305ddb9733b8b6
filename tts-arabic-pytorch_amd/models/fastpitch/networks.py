"""Drop-in for the reference's models/fastpitch/networks.py: same classes, signatures,
defaults and return conventions, with the arithmetic routed to the HIP engines.

  FastPitch       (reference :45-253)  checkpoint -> .infer/.ttmel/.ttmel_single/.ttmel_batch
  FastPitch2Wave  (reference :256-435) + vocoder + denoiser -> .tts/.tts_single/.tts_batch
  text_collate_fn (:16-35), pitch_trf (:38-42)

Differences, all host-side: the vocoder runs ONCE on the ragged batch instead of in a
per-utterance loop (:340-345) — results are identical because every layer pads at the true
utterance edge — `vowelizer=` runs the Shakkelha / Shakkala taggers of models/diacritizers on the HIP tagger engine.
"""
from typing import List, Optional, Union

import numpy as np
import os
import torch
import torch.nn as nn

import text
from ttsamd.engine import FastPitchEngine
from ttsamd.lib import TtsAmdError
from utils import get_basic_config
from models.diacritizers import load_vowelizer
from vocoder import load_hifigan
from vocoder.hifigan.denoiser import Denoiser
from vocoder.hifigan.models import _HipModule


def text_collate_fn(batch: List[torch.Tensor]):
    """Sort by length (descending), zero-pad; returns (ids_pad, lens_sorted, reverse_ids)."""
    lens_sorted, sort_ids = torch.sort(torch.LongTensor([len(x) for x in batch]), descending=True)
    ids_pad = torch.zeros(len(batch), int(lens_sorted[0]), dtype=torch.long)
    for i, j in enumerate(sort_ids):
        ids_pad[i, :batch[j].size(0)] = batch[j]
    return ids_pad, lens_sorted, sort_ids.argsort()


def pitch_trf(mul: float = 1, add: float = 0):
    """Affine transform of the *normalised* pitch prediction; mean/std are ignored, as in the
    reference (:38-42).  Tagged so `infer` can run it inside the HIP predictor head."""
    def _pitch_trf(pitch_pred, enc_mask_sum, mean, std):
        return mul * pitch_pred + add
    _pitch_trf.affine = (float(mul), float(add))
    return _pitch_trf


class FastPitch(_HipModule):
    def __init__(self, checkpoint: str, arabic_in: bool = True, vowelizer: Optional[str] = None, **kwargs):
        super().__init__()
        from models.fastpitch import net_config
        state_dicts = torch.load(checkpoint, map_location='cpu')
        self.net_config = dict(state_dicts['config']) if 'config' in state_dicts else dict(net_config)
        self.arabic_in = arabic_in
        self._sd = {k: v.detach().cpu().float().numpy() for k, v in state_dicts['model'].items()
                    if torch.is_tensor(v) and v.is_floating_point() and not k.startswith('attention.')}
        self.config = get_basic_config()
        self.vowelizers = {}
        if vowelizer is not None:
            self.vowelizers[vowelizer] = load_vowelizer(vowelizer, self.config)
        self.default_vowelizer = vowelizer
        self.phon_to_id = None
        if 'symbols' in state_dicts:
            self.phon_to_id = {phon: i for i, phon in enumerate(state_dicts['symbols'])}
        self.pitch_mean = float(self._sd.get('pitch_mean', np.zeros(1))[0])
        self.pitch_std = float(self._sd.get('pitch_std', np.zeros(1))[0])
        self.eval()

    def engine(self):
        return self._engine(lambda dev: FastPitchEngine(self._sd, self.net_config, device=dev))

    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: v.detach().cpu().float().numpy() for k, v in state_dict.items()
                    if torch.is_tensor(v) and v.is_floating_point() and not k.startswith('attention.')}
        self._engines.clear()

    def state_dict(self, *a, **k):
        return {k_: torch.from_numpy(v) for k_, v in self._sd.items()}

    # ---- FastPitch.infer (models/fastpitch/fastpitch/model.py:351-353) -------------------
    @torch.inference_mode()
    def infer(self, inputs, pace=1.0, dur_tgt=None, pitch_tgt=None, energy_tgt=None, pitch_transform=None,
              max_duration=75, speaker=0, alone=False):
        """`alone=True` (not in the reference): every row of the batch as if it were the only utterance of the call, i.e. row b ==
        infer(inputs[b:b+1, :len_b]) within fp32 summation order (FastPitchEngine.infer) -- the batch_size = 1 loop as one ragged call."""
        ids = torch.as_tensor(inputs).long()
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.net_config['n_symbols']):
            raise IndexError(f'token id out of range [0, {self.net_config["n_symbols"]}) (nn.Embedding raises here too)')
        if self.net_config['n_speakers'] > 1 and not 0 <= int(speaker) < self.net_config['n_speakers']:
            raise IndexError(f'speaker {speaker} out of range [0, {self.net_config["n_speakers"]})')
        nz = (ids != self.net_config['padding_idx'])
        lens = nz.sum(1)
        if not bool((nz == (torch.arange(ids.shape[1], device=ids.device)[None] < lens[:, None])).all()):
            raise ValueError('ids must be zero-padded at the end of each row (text_collate_fn layout)')
        eng = self.engine()
        mul, add = 1.0, 0.0
        if pitch_transform is not None:
            if hasattr(pitch_transform, 'affine'):
                mul, add = pitch_transform.affine
            else:
                # arbitrary callable: run the predictor, transform on the host side, feed back
                _, _, _, pp, _ = eng.infer(ids, pace=pace, dur_tgt=dur_tgt, max_duration=max_duration, speaker=speaker, alone=alone)
                mean, std = (218.14, 67.24) if self.pitch_std == 0.0 else (self.pitch_mean, self.pitch_std)
                pp = pitch_transform(pp, lens.to(pp.device), mean, std)
                out = eng.infer(ids, pace=pace, dur_tgt=dur_tgt, pitch_tgt=pp if pitch_tgt is None else pitch_tgt,
                                energy_tgt=energy_tgt, max_duration=max_duration, speaker=speaker, alone=alone)
                return out[0], out[1], out[2], pp, out[4]
        return eng.infer(ids, pace=pace, dur_tgt=dur_tgt, pitch_tgt=pitch_tgt, energy_tgt=energy_tgt,
                         pitch_mul=mul, pitch_add=add, max_duration=max_duration, speaker=speaker, alone=alone)

    # ---- text -> mel (reference :77-253) -----------------------------------------------
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
            return text.arabic_to_tokens(utterance, append_space=False)
        return text.buckwalter_to_tokens(utterance, append_space=False)

    @staticmethod
    def _ptrf(pitch_mul, pitch_add, pitch_transform):
        if (pitch_mul != 1. or pitch_add != 0.) and pitch_transform is None:
            return pitch_trf(pitch_mul, pitch_add)
        return pitch_transform

    @torch.inference_mode()
    def ttmel_single(self, utterance: str, speed: float = 1, speaker_id: int = 0, vowelizer=None,
                     pitch_mul: float = 1., pitch_add: float = 0., dur_tgt=None, pitch_tgt=None,
                     energy_tgt=None, pitch_transform=None, max_duration=75):
        tokens = self._tokenize(utterance, vowelizer=vowelizer)
        ids = torch.LongTensor(text.tokens_to_ids(tokens, self.phon_to_id)).unsqueeze(0)
        mel, *_ = self.infer(ids, pace=speed, speaker=speaker_id, dur_tgt=dur_tgt, pitch_tgt=pitch_tgt,
                             energy_tgt=energy_tgt, pitch_transform=self._ptrf(pitch_mul, pitch_add, pitch_transform),
                             max_duration=max_duration)
        return mel[0]                                                   # [80, T]

    @torch.inference_mode()
    def _ttmel_batch_padded(self, batch, speed, speaker_id, vowelizer, pitch_mul, pitch_add, dur_tgt=None,
                            pitch_tgt=None, energy_tgt=None, pitch_transform=None, max_duration=75):
        batch_ids = [torch.LongTensor(text.tokens_to_ids(self._tokenize(line, vowelizer), self.phon_to_id))
                     for line in batch]
        ids_pad, lens_sorted, reverse_ids = text_collate_fn(batch_ids)
        mel, dec_lens, *_ = self.infer(ids_pad, pace=speed, speaker=speaker_id, dur_tgt=dur_tgt, pitch_tgt=pitch_tgt,
                                       energy_tgt=energy_tgt,
                                       pitch_transform=self._ptrf(pitch_mul, pitch_add, pitch_transform),
                                       max_duration=max_duration)
        return mel, dec_lens, reverse_ids

    @torch.inference_mode()
    def ttmel_lines_alone(self, lines: List[str], speed: float = 1, speaker_id: int = 0, vowelizer=None,
                          pitch_mul: float = 1., pitch_add: float = 0., max_duration=75):
        """`[ttmel_single(l) for l in lines]` as ONE ragged FastPitch call (infer(..., alone=True)): (mel [B, 80, T_max], dec_lens int64
        [B]) in HBM, rows in the order of `lines`; mel[b, :, :dec_lens[b]] equals ttmel_single(lines[b]) within fp32 summation order
        (frames past dec_lens[b] are undefined).  What `FastPitch2Wave.tts(list, batch_size=1)` runs per group of lines."""
        batch_ids = [text.tokens_to_ids(self._tokenize(line, vowelizer), self.phon_to_id) for line in lines]
        ids = torch.zeros(len(batch_ids), max(len(i) for i in batch_ids), dtype=torch.int64)
        for b, i in enumerate(batch_ids):
            ids[b, :len(i)] = torch.as_tensor(i, dtype=torch.int64)
        mel, dec_lens, *_ = self.infer(ids, pace=speed, speaker=speaker_id, pitch_transform=self._ptrf(pitch_mul, pitch_add, None),
                                       max_duration=max_duration, alone=True)
        return mel, dec_lens

    @torch.inference_mode()
    def ttmel_batch(self, batch: List[str], speed: float = 1, speaker_id: int = 0, vowelizer=None,
                    pitch_mul: float = 1., pitch_add: float = 0., dur_tgt=None, pitch_tgt=None, energy_tgt=None,
                    pitch_transform=None, max_duration=75):
        mel, dec_lens, reverse_ids = self._ttmel_batch_padded(batch, speed, speaker_id, vowelizer, pitch_mul,
                                                              pitch_add, dur_tgt, pitch_tgt, energy_tgt,
                                                              pitch_transform, max_duration)
        lens = dec_lens.tolist()
        return [mel[j, :, :lens[j]] for j in reverse_ids.tolist()]      # original order

    def ttmel(self, text_input: Union[str, List[str]], speed: float = 1, speaker_id: int = 0, batch_size: int = 1,
              vowelizer=None, pitch_mul: float = 1., pitch_add: float = 0.):
        kw = dict(speed=speed, speaker_id=speaker_id, vowelizer=vowelizer, pitch_mul=pitch_mul, pitch_add=pitch_add)
        if isinstance(text_input, str):
            return self.ttmel_single(text_input, **kw)
        assert isinstance(text_input, list)
        if batch_size == 1:
            return [self.ttmel_single(sample, **kw) for sample in text_input]
        mel_list = []
        for k in range(0, len(text_input), batch_size):
            mel_list += self.ttmel_batch(text_input[k:k + batch_size], **kw)
        return mel_list


class FastPitch2Wave(nn.Module):
    def __init__(self, model_sd_path: str, vocoder_sd: Optional[str] = None, vocoder_config: Optional[str] = None,
                 vowelizer: Optional[str] = None, arabic_in: bool = True):
        super().__init__()
        self.model = FastPitch(model_sd_path, arabic_in=arabic_in, vowelizer=vowelizer)
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
    def tts_single(self, text_buckw: str, speed: float = 1, speaker_id: int = 0, denoise: float = 0,
                   vowelizer=None, pitch_mul: float = 1., pitch_add: float = 0., return_mel: bool = False):
        mel_spec = self.model.ttmel_single(text_buckw, speed, speaker_id, vowelizer, pitch_mul=pitch_mul,
                                           pitch_add=pitch_add)
        wave = self.vocoder(mel_spec)
        if denoise > 0:
            wave = self.denoiser(wave, denoise)
        if return_mel:
            return wave[0].cpu(), mel_spec
        return wave[0].cpu()

    @torch.inference_mode()
    def tts_batch_device(self, batch: List[str], speed: float = 1, speaker_id: int = 0, denoise: float = 0,
                         vowelizer=None, pitch_mul: float = 1., pitch_add: float = 0., return_mel: bool = False):
        """`tts_batch` without the device->host copy: (wave [B, n_max] float32, n_samples int64 [B]), both in HBM,
        rows in the order of `batch` (zeros past n_samples[b]).  What the data-parallel gather (ttsamd.dp) and
        any GPU-side consumer take; `tts_batch` is this plus one D2H."""
        mel, dec_lens, reverse_ids = self.model._ttmel_batch_padded(batch, speed, speaker_id, vowelizer, pitch_mul,
                                                                    pitch_add)
        wave = self.vocoder.engine().forward(mel, dec_lens)             # one ragged batched launch sequence
        n = (dec_lens * self.vocoder.engine().hop)
        if denoise > 0:
            wave = self.denoiser.forward_batch(wave, n, denoise)
        rev = reverse_ids.to(wave.device)
        return wave.index_select(0, rev), n.index_select(0, rev)

    @torch.inference_mode()
    def tts_batch(self, batch: List[str], speed: float = 1, speaker_id: int = 0, denoise: float = 0, vowelizer=None,
                  pitch_mul: float = 1., pitch_add: float = 0., return_mel: bool = False):
        mel, dec_lens, reverse_ids = self.model._ttmel_batch_padded(batch, speed, speaker_id, vowelizer, pitch_mul,
                                                                    pitch_add)
        wave = self.vocoder.engine().forward(mel, dec_lens)             # one ragged batched launch sequence
        n = (dec_lens * self.vocoder.engine().hop)
        if denoise > 0:
            wave = self.denoiser.forward_batch(wave, n, denoise)
        n = n.tolist()
        # one exact-size D2H per utterance (a padded [B, n_max] copy + per-row clones touches every host page twice)
        # NB the reference silently ignores return_mel here (:347-350); so do we
        return [wave[j, :n[j]].cpu() for j in reverse_ids.tolist()]

    def tts(self, text_input: Union[str, List[str]], speed: float = 1., denoise: float = 0.005, speaker_id: int = 0,
            batch_size: int = 2, vowelizer=None, pitch_mul: float = 1., pitch_add: float = 0.,
            return_mel: bool = False) -> Union[torch.Tensor, List[torch.Tensor]]:
        """Same contract as the reference (:352-435): str -> Tensor[n_samples] (CPU);
        list -> list of tensors, chunked by `batch_size`."""
        kw = dict(speaker_id=speaker_id, speed=speed, denoise=denoise, vowelizer=vowelizer, pitch_mul=pitch_mul,
                  pitch_add=pitch_add, return_mel=return_mel)
        if isinstance(text_input, str):
            return self.tts_single(text_input, **kw)
        assert isinstance(text_input, list)
        if (len(text_input) > batch_size and not return_mel and self.device.type == 'cuda'
                and os.environ.get('TTSAMD_TTS_PIPELINE', '1') != '0'):
            return self._tts_list_pipelined(text_input, batch_size, **kw)
        if batch_size == 1:
            return [self.tts_single(sample, **kw) for sample in text_input]
        wav_list = []
        for k in range(0, len(text_input), batch_size):
            wav_list += self.tts_batch(text_input[k:k + batch_size], **kw)
        return wav_list

    # utterances per vocoder call of the list pipeline: chunks of `batch_size` lines go through FastPitch one by one (a chunk is the
    # reference's padded batch: its results depend on the chunk's composition), their mels are vocoded together
    _VOCODER_GROUP = 16
    # lines per ragged FastPitch + vocoder call of the batch_size = 1 list path (rows computed as if alone: FastPitch.infer(alone=True))
    _ALONE_GROUP = 32
    _ALONE_CHARS = 12288         # ... and at most this many characters x lines per call (lines x the longest line: bounds the padded batch)

    @staticmethod
    def _alone_groups(lengths, group, budget):
        """Index groups over lines SORTED by length (ascending `lengths`): a group is filled until it has `group` lines or
        lines x longest line would pass `budget` -- a list of very long lines must not become one 32-row batch padded to the longest."""
        groups, fill = [], []
        for i, n in enumerate(lengths):
            if fill and (len(fill) >= group or (len(fill) + 1) * n > budget):
                groups.append(fill)
                fill = []
            fill.append(i)
        if fill:
            groups.append(fill)
        return groups

    @torch.inference_mode()
    def _tts_list_pipelined(self, text_input, batch_size, speed, denoise, speaker_id, vowelizer, pitch_mul, pitch_add,
                            return_mel=False):
        """The list path of `tts` over several chunks, as a three-stage pipeline on three HIP streams: tokenisation + FastPitch
        of the next chunks (host work and ~150 short launches that leave most CUs idle) run under the vocoder + denoiser of the
        previous ones, whose audio is copied to the host on a third stream.  FastPitch sees exactly the chunks of the one-stream
        loop (`ttmel_single` per line for batch_size 1, the padded batch otherwise).  The vocoder is batch-independent -- every
        layer pads each utterance at its own true edge, so a ragged batch equals the per-utterance loop of the reference
        (networks.py:340-345; tests: test_hifigan_ragged_batch_matches_unbatched, test_full_size_bench_workload_properties) -- and
        therefore takes the mels of up to _VOCODER_GROUP utterances in ONE ragged call: at batch_size 1 its launches fill the chip
        like a batch-16 call instead of 100 batch-1 calls (C1, 100 lines: 861 -> see DESIGN.md).  Waves equal the one-stream loop's
        within the vocoder's fp32 summation-order noise (1e-6; a larger batch picks other tiles), lengths exactly.
        batch_size 1 (TTSAMD_TTS_ALONE=0 restores the line-by-line FastPitch calls): FastPitch too is made batch-independent -- batch mode 1
        of the engine computes every row of a ragged batch as if it were alone (ttsamd_fastpitch_set_batch_mode; tests/test_gpu_alone.py) --
        so the lines are sorted by length and go through FastPitch AND the vocoder in balanced groups of up to _ALONE_GROUP similar
        lengths (little padding), waves handed back in input order: 100 launch-bound batch-1 calls of ~1.7 ms become 4 chip-filling ones
        (C1, 100 lines: 500 -> see DESIGN.md); each wave equals its line-by-line result within fp32 summation order, lengths exactly."""
        dev = self.device
        if getattr(self, '_pipe_streams', None) is None or self._pipe_streams[0].device != dev:
            self._pipe_streams = tuple(torch.cuda.Stream(dev) for _ in range(3))
        s_fp, s_hg, s_cp = self._pipe_streams
        cur = torch.cuda.current_stream(dev)
        for st in (s_fp, s_hg, s_cp):
            st.wait_stream(cur)
        eng = self.vocoder.engine()
        hop = eng.hop
        out, pending = [], None
        group = max(1, self._VOCODER_GROUP // batch_size)           # chunks per vocoder call
        n_in = len(text_input)
        order = list(range(n_in))
        alone_ok = batch_size == 1 and n_in > 1 and os.environ.get('TTSAMD_TTS_ALONE', '1') != '0'
        if alone_ok:
            order.sort(key=lambda i: len(text_input[i]))
            n_groups = (n_in + self._ALONE_GROUP - 1) // self._ALONE_GROUP
            group = (n_in + n_groups - 1) // n_groups
            text_input = [text_input[i] for i in order]

        def flush(item):
            wave, n, done = item
            s_cp.wait_event(done)
            with torch.cuda.stream(s_cp):
                wave.record_stream(s_cp)
                out.extend(wave[j, :n[j]].cpu() for j in range(len(n)))      # blocks the host on THIS group's audio only

        chunks = [text_input[k:k + batch_size] for k in range(0, len(text_input), batch_size)]
        groups = [chunks[g0:g0 + group] for g0 in range(0, len(chunks), group)]
        if alone_ok:
            groups = [[chunks[i] for i in g] for g in self._alone_groups([len(ch[0]) for ch in chunks], group, self._ALONE_CHARS)]
        for grp in groups:
            mels, lens = [], []                                     # this group's utterances in input order
            with torch.cuda.stream(s_fp):
                alone = alone_ok and len(grp) > 1
                if alone:
                    # the batch_size = 1 loop of this group's lines as ONE ragged FastPitch call whose rows are computed as if alone
                    # (ttsamd_fastpitch_set_batch_mode 1); the mel batch and its lengths go to the vocoder as they are
                    mel_b, lens_d = self.model.ttmel_lines_alone([c[0] for c in grp], speed, speaker_id, vowelizer,
                                                                 pitch_mul=pitch_mul, pitch_add=pitch_add)
                    lens = lens_d.cpu().tolist()
                for chunk in ([] if alone else grp):
                    if batch_size == 1:
                        mel = self.model.ttmel_single(chunk[0], speed, speaker_id, vowelizer, pitch_mul=pitch_mul, pitch_add=pitch_add)
                        mels.append(mel)
                        lens.append(int(mel.shape[-1]))
                    else:
                        mel, dec_lens, rev = self.model._ttmel_batch_padded(chunk, speed, speaker_id, vowelizer, pitch_mul, pitch_add)
                        dl = dec_lens.cpu().tolist()                # (FastPitch has synchronised on these lengths already)
                        for j in rev.tolist():
                            mels.append(mel[j, :, :dl[j]])
                            lens.append(int(dl[j]))
                if alone:
                    pass
                elif len(mels) == 1:
                    mel_b = mels[0][None]
                else:
                    mel_b = torch.zeros(len(mels), mels[0].shape[0], max(lens), dtype=mels[0].dtype, device=dev)
                    for i, m in enumerate(mels):
                        mel_b[i, :, :lens[i]] = m
                if not alone:
                    lens_d = torch.tensor(lens, dtype=torch.int64).to(dev, non_blocking=True)
            s_hg.wait_stream(s_fp)
            with torch.cuda.stream(s_hg):
                mel_b.record_stream(s_hg)
                lens_d.record_stream(s_hg)
                wave = eng.forward(mel_b, lens_d)
                n_host = [t * hop for t in lens]
                if denoise > 0:
                    wave = self.denoiser.forward_batch(wave, lens_d * hop, denoise, nsamples_min=min(n_host))
                done = torch.cuda.Event()
                done.record(s_hg)
            if pending is not None:
                flush(pending)
            pending = (wave, n_host, done)
        if pending is not None:
            flush(pending)
        cur.wait_stream(s_hg)
        cur.wait_stream(s_cp)
        if alone_ok:                                                # back to the order of the input
            res = [None] * n_in
            for pos, w in enumerate(out):
                res[order[pos]] = w
            return res
        return out
