"""CPU: the Tacotron2 wrapper logic of the drop-in (tts-arabic-pytorch_amd/models/tacotron2/networks.py) against goldens
produced by the REAL reference wrapper (/root/reference/models/tacotron2/networks.py:16-67,123-253) in oracle/gen_golden_taco.py.

Pinned here: text_collate_fn, needs_postprocessing, truncate_mel, resize_mel, the separator insertion, the ids / speaker ids /
lengths the wrapper hands to `infer`, and what it makes of infer's result (attention-peak cut, replicated frames, bicubic resize,
un-sorting, chunking).  NOT pinned (cannot be, in this image): `Tacotron2MS.infer` itself, whose arithmetic lives in the absent,
un-vendored `torchaudio.models.tacotron2` -- both sides run the same closed-form stand-in for it (`fake_infer`, the fixture's
input generator).  The HIP core is compared with oracle/taco_oracle.py in tests/test_gpu_tacotron2.py ("parity unpinned").
"""
import numpy as np
import pytest
import torch

import text
from models.tacotron2 import networks as N


def fake_infer(ids, sids, lens=None):
    """The stand-in core the golden run used (oracle/gen_golden_taco.py::fake_infer): a closed-form function of the ids."""
    ids = ids.cpu()
    B, L = ids.shape
    if lens is None:
        lens = torch.full((B,), L, dtype=torch.long)
    lens = lens.cpu()
    mel_lens = lens // 2 + (ids.sum(1) % 7) + 5
    T = int(mel_lens.max())
    t = torch.arange(T, dtype=torch.float32)
    f = torch.arange(80, dtype=torch.float32)
    mel = torch.zeros(B, 80, T)
    al = torch.zeros(B, T, L)
    for b in range(B):
        s = float(ids[b].sum() % 13)
        mel[b] = torch.sin(0.37 * f[:, None] + 0.11 * t[None, :] + s) - 0.01 * t[None, :]
        n = float(mel_lens[b])
        centre = (t[:, None] / n) * float(lens[b])
        l = torch.arange(L, dtype=torch.float32)[None, :]
        al[b] = torch.exp(-0.5 * ((l - centre) / 1.5) ** 2)
        al[b] = al[b] / al[b].sum(1, keepdim=True)
        mel[b, :, int(mel_lens[b]):] = 0
        al[b, int(mel_lens[b]):] = 0
    return mel, mel_lens, al


class _Taco(N.Tacotron2):
    """the product wrapper with the fake core (no weights, no GPU: the engine is only built by the real infer)"""

    def __init__(self):
        super().__init__(checkpoint=None, n_symbol=len(text.symbols))
        self.calls = []

    def infer(self, ids, sids, lens=None):
        self.calls.append((ids.clone(), sids.clone(), None if lens is None else lens.clone()))
        return fake_infer(ids, sids, lens)


@pytest.fixture(scope='module')
def g(golden):
    return golden('taco_wrapper')


def _lst(g, name):
    return [g[f'{name}_{i}'] for i in range(int(g[name + '_n']))]


def test_needs_postprocessing_every_symbol(g):
    assert list(g['npp_symbols']) == text.symbols
    assert [N.needs_postprocessing(s) for s in text.symbols] == g['npp'].tolist()


def test_text_collate_fn_reference_golden(g):
    ids_pad, lens_sorted, rev = N.text_collate_fn([torch.from_numpy(a) for a in _lst(g, 'collate_in')])
    assert np.array_equal(ids_pad.numpy(), g['collate_ids'])
    assert np.array_equal(lens_sorted.numpy(), g['collate_lens'])
    assert np.array_equal(rev.numpy(), g['collate_rev'])


def test_truncate_mel_reference_golden(g):
    mel, cols = torch.from_numpy(g['trunc_mel']), torch.from_numpy(g['trunc_cols'])
    for i in range(4):          # maximum at frame 1, plateau (first index wins), monotone, maximum at the last frame
        got = N.truncate_mel(mel, cols[i]).numpy()
        assert got.shape == g[f'trunc_out{i}'].shape and np.array_equal(got, g[f'trunc_out{i}']), i


def test_resize_mel_reference_golden(g):
    mel = torch.from_numpy(g['resize_mel'])
    for rate in (0.8, 1.0, 1.25, 2):
        got = N.resize_mel(mel, rate=rate).numpy()
        want = g[f'resize_out_{rate}']
        assert got.shape == want.shape and np.abs(got - want).max() <= 1e-6, rate


def test_wrapper_single_calls_reference_golden(g):
    """ttmel_single (:123-152): ids with the separator inserted before the EOS tokens, cut at the attention peak of that
    separator, 3 replicated frames; speed; postprocess_mel=False."""
    model = _Taco()
    lines = [str(s) for s in g['lines']]
    sep_ids, flags = _lst(g, 'sep_ids'), g['sep_flags']
    mels = _lst(g, 'single_mels')
    assert flags.any() and not flags.all()
    for i, ln in enumerate(lines):
        toks, flag = model._tokens_for(ln, None, True)
        assert bool(flag) == bool(flags[i])
        assert text.tokens_to_ids(toks, model.phon_to_id) == sep_ids[i].tolist(), i
        model.calls.clear()
        mel = model.ttmel_single(ln).numpy()
        assert np.array_equal(model.calls[0][0][0].numpy(), sep_ids[i])
        assert tuple(mel.shape) == tuple(g['single_shapes'][i])
        assert abs(mel.astype(np.float64).sum() - g['single_sums'][i]) <= 1e-3
        if i < len(mels):
            assert np.array_equal(mel, mels[i]), i
    for i, want in enumerate(_lst(g, 'single_nopost')):
        assert np.array_equal(model.ttmel_single(lines[i], postprocess_mel=False).numpy(), want)
    for i, want in enumerate(_lst(g, 'single_speed')):
        got = model.ttmel_single(lines[i], speed=1.25).numpy()
        assert got.shape == want.shape and np.abs(got - want).max() <= 1e-6


def test_wrapper_batch_reference_golden(g):
    """ttmel_batch / ttmel (:155-253): collation handed to infer, speaker ids, un-sorting, per-utterance cut, chunking."""
    model = _Taco()
    lines = [str(s) for s in g['lines']]
    mels = model.ttmel_batch(lines[:8], speaker_id=3)
    ids, sids, lens = model.calls[0]
    assert np.array_equal(ids.numpy(), g['batch_ids']) and np.array_equal(sids.numpy(), g['batch_sids'])
    assert np.array_equal(lens.numpy(), g['batch_lens'])
    for got, want in zip(mels, _lst(g, 'batch_mels')):
        assert np.array_equal(got.numpy(), want)
    for got, want in zip(model.ttmel_batch(lines[:8], speed=0.8), _lst(g, 'batch_speed')):
        assert got.shape == want.shape and np.abs(got.numpy() - want).max() <= 1e-6
    model.calls.clear()
    mels = model.ttmel(lines, batch_size=5)
    assert np.array_equal(np.array([c[0].shape for c in model.calls]), g['chunk_calls'])
    assert np.array_equal(np.array([m.shape for m in mels]), g['chunk_shapes'])
    assert np.abs(np.array([m.double().sum().item() for m in mels]) - g['chunk_sums']).max() <= 1e-3
    for got, want in zip(mels[-4:], _lst(g, 'chunk_mels_tail')):
        assert np.array_equal(got.numpy(), want)
    singles = model.ttmel(lines[:6], batch_size=1)
    for got, want in zip(singles, _lst(g, 'single_mels')):
        assert np.array_equal(got.numpy(), want)


def test_expected_cut_helper_is_the_reference(g):
    """tests/test_gpu_tacotron2.py checks the HIP path's post-processed mels against `expected_cut`; that helper is pinned
    to the reference's truncate_mel outputs here (so the GPU test no longer compares the product with itself)."""
    from test_gpu_tacotron2 import expected_cut
    mel, cols = torch.from_numpy(g['trunc_mel']), torch.from_numpy(g['trunc_cols'])
    for i in range(4):
        assert np.array_equal(expected_cut(mel, cols[i]).numpy(), g[f'trunc_out{i}']), i
