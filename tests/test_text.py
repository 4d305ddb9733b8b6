"""CPU: own text front-end (tts-arabic-pytorch_amd/text) against token ids produced by the
reference's front-end (fixtures made by oracle/gen_golden.py and oracle/check_text_frontend.py)."""
import json
import os

import numpy as np
import pytest
import torch

import text
from conftest import GOLDEN


def test_symbols_and_infer_text_ids(golden):
    g = golden('infer_text_ids')
    assert list(g['symbols']) == text.symbols
    with open(os.path.join(GOLDEN, 'infer_text_lines.json'), encoding='utf-8') as f:
        lines = json.load(f)
    assert len(lines) == 100
    for i, ln in enumerate(lines):
        ids = text.tokens_to_ids(text.arabic_to_tokens(ln, append_space=False))
        assert ids == g['flat'][g['offsets'][i]:g['offsets'][i + 1]].tolist(), i


def test_fuzz_tokens(golden):
    g = golden('text_fuzz')
    vocab = list(g['vocab'])
    for i, s in enumerate(g['inputs']):
        want = [vocab[j] for j in g['flat'][g['offsets'][i]:g['offsets'][i + 1]]]
        assert text.arabic_to_tokens(str(s), append_space=False) == want, repr(s)


def test_oov_raises_keyerror():
    # the default symbol table has no punctuation (text/__init__.py:27 raises KeyError)
    with pytest.raises(KeyError):
        text.tokens_to_ids(text.buckwalter_to_tokens('marHabAF.'))


def test_roundtrip_and_append_space():
    bw = ">als~alAmu Ealaykum yA Sadiyqiy"
    assert text.arabic_to_buckwalter(text.buckwalter_to_arabic(bw)) == bw
    t0 = text.buckwalter_to_tokens(bw, append_space=False)
    t1 = text.buckwalter_to_tokens(bw)
    assert t1[:-2] == t0[:-1] and t1[-2:] == ['_+_', '_eos_'] and t0[-1] == '_eos_'


def test_text_collate_fn(golden):
    pytest.importorskip('ttsamd.lib')
    from ttsamd import lib
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip('libttsamd.so not built')
    from models.fastpitch.networks import text_collate_fn
    g = golden('infer_text_ids')
    seqs = [torch.from_numpy(g['flat'][g['offsets'][i]:g['offsets'][i + 1]]) for i in range(5)]
    padded, lens, rev = text_collate_fn(seqs)
    assert torch.equal(padded, torch.from_numpy(g['collate5_padded']))
    assert torch.equal(lens, torch.from_numpy(g['collate5_lens']))
    assert torch.equal(rev, torch.from_numpy(g['collate5_rev']))


def test_list_path_groups_of_the_batch_size_1_pipeline():
    """Host logic of `FastPitch2Wave.tts(list, batch_size=1)` (models/fastpitch/networks.py: _alone_groups): length-sorted lines go to the
    ragged FastPitch call in groups of at most `group` lines AND at most budget characters x lines (lines x the longest line of the group);
    every line lands in exactly one group, in order."""
    pytest.importorskip('ttsamd.lib')
    from ttsamd import lib
    if not os.path.exists(lib.LIB_PATH):
        pytest.skip('libttsamd.so not built')
    from models.fastpitch.networks import FastPitch2Wave
    f = FastPitch2Wave._alone_groups
    assert f([], 25, 1000) == []
    assert f([5], 25, 1000) == [[0]]
    lens = sorted([35 + 3 * i for i in range(100)])
    g = f(lens, 25, 12288)
    assert [len(x) for x in g] == [25, 25, 25, 25] and sum(g, []) == list(range(100))
    g = f([1000] * 40, 32, 12288)                                   # very long lines: 12 per call, not 32
    assert [len(x) for x in g] == [12, 12, 12, 4] and sum(g, []) == list(range(40))
    g = f([10, 10, 10, 5000, 20000], 32, 12288)                     # a line longer than the budget still goes through, alone
    assert g == [[0, 1, 2], [3], [4]]
    for x in f(lens, 7, 600):
        assert len(x) <= 7 and (len(x) == 1 or len(x) * lens[x[-1]] <= 600)
