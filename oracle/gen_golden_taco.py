"""Generate tests/golden/taco_wrapper.npz by running the REAL reference wrapper code of
models/tacotron2/networks.py (/root/reference) in this container.  Test infrastructure; run manually:
    python oracle/gen_golden_taco.py

What can be pinned of config 4 without torchaudio: everything in models/tacotron2/networks.py is plain
torch / Python (text_collate_fn :16-35, needs_postprocessing :38-40, truncate_mel :43-48, resize_mel
:51-66, the separator insertion and un-sorting of Tacotron2.ttmel_single / ttmel_batch :123-208).  Only
`Tacotron2MS.infer` reaches into `torchaudio.models.tacotron2` (tacotron2_ms.py:113) -- absent from this
image.  So the reference classes are imported with EMPTY stand-ins for `_Encoder / _Decoder / _Postnet /
_get_mask_from_lengths`, and `infer` is replaced by a deterministic fake (a closed-form function of the
ids it is handed, defined below and restated in tests/test_taco_wrapper_golden.py).  The goldens hold what the
reference wrapper passes INTO infer (token ids with the inserted separator, collation order, lengths,
speaker ids) and what it makes OUT of infer's result (attention-peak cut, 3 replicated frames, bicubic
resize, un-sorting).  The torchaudio core itself stays "parity unpinned" (SURVEY 8c).
No reference source is copied; the GPU box only sees the .npz.
"""
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _refstub  # noqa: E402

_refstub.install()

import torch  # noqa: E402

# empty stand-ins: the constructor of Tacotron2MS builds them (tacotron2_ms.py:188-207), nothing here calls them
tm = types.ModuleType('torchaudio.models')
tt = types.ModuleType('torchaudio.models.tacotron2')


class _Empty(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


tt.Tacotron2 = tt._Encoder = tt._Decoder = tt._Postnet = _Empty
tt._get_mask_from_lengths = lambda lengths: None
tm.tacotron2 = tt
sys.modules['torchaudio'].models = tm
sys.modules['torchaudio.models'] = tm
sys.modules['torchaudio.models.tacotron2'] = tt

import text as ref_text  # noqa: E402
from text.symbols import EOS_TOKENS, SEPARATOR_TOKEN  # noqa: E402
from models.tacotron2 import networks as N  # noqa: E402

torch.set_grad_enabled(False)


def fake_infer(ids, sids, lens=None, max_step=None):
    """Deterministic stand-in for Tacotron2MS.infer (tacotron2_ms.py:279-332): same signature and return
    shapes.  mel[b, f, t] and alignments[b, t, l] are closed-form functions of (ids, b, f, t, l); frame counts
    depend on the token count.  Restated verbatim in tests/test_taco_wrapper_golden.py (this IS the fixture's input)."""
    ids = ids.cpu()
    B, L = ids.shape
    if lens is None:
        lens = torch.full((B,), L, dtype=torch.long)
    lens = lens.cpu()
    mel_lens = lens // 2 + (ids.sum(1) % 7) + 5                     # frames per utterance
    T = int(mel_lens.max())
    t = torch.arange(T, dtype=torch.float32)
    f = torch.arange(80, dtype=torch.float32)
    mel = torch.zeros(B, 80, T)
    al = torch.zeros(B, T, L)
    for b in range(B):
        s = float(ids[b].sum() % 13)
        mel[b] = torch.sin(0.37 * f[:, None] + 0.11 * t[None, :] + s) - 0.01 * t[None, :]
        # attention: a Gaussian ridge moving diagonally over the tokens, so that the column of the inserted separator
        # (tokens[-n_eos-1]) rises towards the end of the utterance, with a plateau -> the 80 % cut lands mid-way
        n = float(mel_lens[b])
        centre = (t[:, None] / n) * float(lens[b])                   # [T, 1]
        l = torch.arange(L, dtype=torch.float32)[None, :]
        al[b] = torch.exp(-0.5 * ((l - centre) / 1.5) ** 2)
        al[b] = al[b] / al[b].sum(1, keepdim=True)
        mel[b, :, int(mel_lens[b]):] = 0
        al[b, int(mel_lens[b]):] = 0
    return mel, mel_lens, al


class RefTaco(N.Tacotron2):
    """The reference wrapper class with infer swapped for the fake: records what the wrapper hands to infer."""

    def __init__(self):
        super().__init__(checkpoint=None, n_symbol=len(ref_text.symbols))
        self.calls = []
        self._dummy = torch.nn.Parameter(torch.zeros(1))             # .device reads next(self.parameters())

    def infer(self, ids, sids, lens=None):
        self.calls.append((ids.clone(), sids.clone(), None if lens is None else lens.clone()))
        return fake_infer(ids, sids, lens)


def put_list(out, name, arrs):
    """a ragged list as name_n + name_0 .. name_{n-1} (no pickled object arrays in the fixture)"""
    out[name + '_n'] = np.int64(len(arrs))
    for i, a in enumerate(arrs):
        out[f'{name}_{i}'] = np.asarray(a)


def main():
    with open(os.path.join(OUT, 'infer_text_lines.json')) as fh:
        lines = json.load(fh)
    lines = [ln for ln in lines if ln.strip()][:24]
    out = {}

    # ---- needs_postprocessing on every symbol of the table (:38-40)
    out['npp_symbols'] = np.array(list(ref_text.symbols))
    out['npp'] = np.array([N.needs_postprocessing(s) for s in ref_text.symbols], dtype=bool)

    # ---- text_collate_fn on ragged id lists incl. ties (torch.sort is not stable by default: ties pinned as data) (:16-35)
    g = torch.Generator().manual_seed(11)
    lens = [5, 9, 9, 1, 14, 9, 3]
    batch = [torch.randint(1, 40, (n,), generator=g) for n in lens]
    ids_pad, lens_sorted, rev = N.text_collate_fn(batch)
    put_list(out, 'collate_in', [b.numpy() for b in batch])
    out['collate_ids'] = ids_pad.numpy()
    out['collate_lens'] = lens_sorted.numpy()
    out['collate_rev'] = rev.numpy()

    # ---- truncate_mel (:43-48): random mel + attention columns; case 0 has its maximum at frame 1 (n_end small),
    # case 1 a plateau (first index wins), case 2 monotone rising, case 3 maximum at the last frame
    mel = torch.randn(80, 40, generator=g)
    cols = torch.rand(4, 40, generator=g)
    cols[0, 1] = 5.0
    cols[1, 10:20] = 3.0
    cols[2] = torch.linspace(0.01, 1, 40)
    cols[3, 39] = 9.0
    out['trunc_mel'] = mel.numpy()
    out['trunc_cols'] = cols.numpy()
    for i in range(4):
        out[f'trunc_out{i}'] = N.truncate_mel(mel, cols[i]).numpy()

    # ---- resize_mel (:51-66): bicubic image interpolation over [F, T]
    mel = torch.randn(80, 57, generator=g)
    out['resize_mel'] = mel.numpy()
    for rate in (0.8, 1.0, 1.25, 2):
        out[f'resize_out_{rate}'] = N.resize_mel(mel, rate=rate).numpy()

    # ---- the wrapper end to end with the fake core: ttmel_single / ttmel_batch / ttmel chunking (:123-253)
    model = RefTaco()
    out['lines'] = np.array(lines)
    tok_lists, flags = [], []
    for ln in lines:
        toks = ref_text.arabic_to_tokens(ln)
        flag = N.needs_postprocessing(toks[-len(EOS_TOKENS) - 1])
        if flag:
            toks.insert(-len(EOS_TOKENS), SEPARATOR_TOKEN)
        tok_lists.append(np.array(ref_text.tokens_to_ids(toks), dtype=np.int64))
        flags.append(flag)
    put_list(out, 'sep_ids', tok_lists)               # restated from :131-137 with the reference's own text module
    out['sep_flags'] = np.array(flags)
    # single calls: ids handed to infer and the mel that comes back out of the wrapper
    singles, single_ids = [], []
    for ln in lines:
        model.calls.clear()
        singles.append(model.ttmel_single(ln).numpy())
        single_ids.append(model.calls[0][0][0].numpy())
    put_list(out, 'single_ids', single_ids)
    put_list(out, 'single_mels', singles[:12])
    out['single_shapes'] = np.array([m.shape for m in singles], dtype=np.int64)
    out['single_sums'] = np.array([m.astype(np.float64).sum() for m in singles])
    assert all(np.array_equal(a, b) for a, b in zip(single_ids, tok_lists))
    put_list(out, 'single_nopost', [model.ttmel_single(ln, postprocess_mel=False).numpy() for ln in lines[:4]])
    put_list(out, 'single_speed', [model.ttmel_single(ln, speed=1.25).numpy() for ln in lines[:4]])
    # one batch of 8: what reaches infer + the un-sorted list
    model.calls.clear()
    mels = model.ttmel_batch(lines[:8], speaker_id=3)
    ids, sids, ln_s = model.calls[0]
    out['batch_ids'] = ids.numpy(); out['batch_sids'] = sids.numpy(); out['batch_lens'] = ln_s.numpy()
    put_list(out, 'batch_mels', [m.numpy() for m in mels])
    put_list(out, 'batch_speed', [m.numpy() for m in model.ttmel_batch(lines[:8], speed=0.8)])
    # ttmel over all lines in chunks of 5 (ragged last chunk), and batch_size=1
    model.calls.clear()
    mels = model.ttmel(lines, batch_size=5)
    out['chunk_calls'] = np.array([c[0].shape for c in model.calls], dtype=np.int64)
    out['chunk_shapes'] = np.array([m.shape for m in mels], dtype=np.int64)
    out['chunk_sums'] = np.array([m.double().sum().item() for m in mels])
    put_list(out, 'chunk_mels_tail', [m.numpy() for m in mels[-4:]])           # the ragged last chunk (lines 20..23)
    mels1 = model.ttmel(lines[:6], batch_size=1)
    assert all(np.array_equal(a.numpy(), b) for a, b in zip(mels1, singles[:6]))

    path = os.path.join(OUT, 'taco_wrapper.npz')
    np.savez_compressed(path, **out)
    print(f'taco_wrapper: {os.path.getsize(path) / 1024:.1f} kB; separator inserted on {int(sum(flags))} of {len(lines)} lines')


if __name__ == '__main__':
    main()
