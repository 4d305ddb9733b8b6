"""FastPitch batch mode 1 (`ttsamd_fastpitch_set_batch_mode`, FastPitchEngine.infer(alone=True)): a ragged batch whose rows are computed as
if each were the only utterance of the call -- the reference's batch_size = 1 loop (models/fastpitch/networks.py:402-411: `tts_single` per
line, FastPitch on ids[None] with no padding) as ONE call.  pytest -m gpu.

Checked here at the engine level against the one-by-one calls of the same engine on the real tokeniser's ids of the first committed
infer_text lines (tests/golden/infer_text_ids.npz); against the ORACLE (per line, batch of one) the mode is checked end to end by
tests/test_gpu_fullsize.py::test_config1_all_100_lines_batch_size_1, whose `tts(list, batch_size=1)` runs on it."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('precision,tol', [('f32', 2e-5), ('bf16x3', 5e-4)])
def test_alone_rows_equal_the_one_by_one_calls(synth_weights, precision, tol):
    """12 lines of 35-268 tokens, predicted durations: dec_lens, durations and pitch of every row EXACTLY / within fp32 noise those of
    infer(ids[b:b+1, :len_b]); mel within `tol` max-abs (fp32: other tiles, other summation order; bf16x3: a batch-1 call runs FastPitch on
    the fp32 kernels, the batch on the split-bf16 ones).  And the mode matters: in the default padded-batch mode at least one SHORT row's
    last frames differ from its one-by-one call by more than 100x that (the un-masked hidden activations)."""
    from ttsamd import engine as E
    from ttsamd.config import NET_CONFIG
    dev = torch.device('cuda:0')
    g = dict(np.load(os.path.join(GOLDEN, 'infer_text_ids.npz'), allow_pickle=False))
    rows = [np.asarray(g['flat'][g['offsets'][i]:g['offsets'][i + 1]], np.int64) for i in range(12)]
    ids = np.zeros((len(rows), max(len(r) for r in rows)), np.int64)
    for b, r in enumerate(rows):
        ids[b, :len(r)] = r
    E.set_precision(precision)
    try:
        eng = E.FastPitchEngine(synth_weights['fastpitch'], NET_CONFIG, device=dev)
        mel_a, lens_a, dur_a, pitch_a, _ = eng.infer(ids, alone=True)
        mel_p, lens_p, *_ = eng.infer(ids, alone=False)
        worst, worst_p = 0.0, 0.0
        for b, r in enumerate(rows):
            mel_1, lens_1, dur_1, pitch_1, _ = eng.infer(r[None])
            n, t = len(r), int(lens_1[0])
            assert int(lens_a[b]) == t and torch.equal(torch.round(dur_a[b, :n]), torch.round(dur_1[0, :n]))
            assert float((pitch_a[b, 0, :n] - pitch_1[0, 0, :n]).abs().max()) < tol
            worst = max(worst, float((mel_a[b, :, :t] - mel_1[0, :, :t]).abs().max()))
            if int(lens_p[b]) == t:
                worst_p = max(worst_p, float((mel_p[b, :, :t] - mel_1[0, :, :t]).abs().max()))
        print(f'{precision}: alone vs one-by-one mel max-abs {worst:.2e} (tol {tol}); padded-batch mode vs one-by-one {worst_p:.2e}')
        assert worst < tol
        assert worst_p > 100 * 2e-5, 'the padded-batch mode is expected to differ from the one-by-one calls at the end of short rows'
    finally:
        E.set_precision('f32')


def test_batch_mode_argument_is_validated(synth_weights):
    from ttsamd import engine as E, lib as L
    from ttsamd.config import NET_CONFIG
    eng = E.FastPitchEngine(synth_weights['fastpitch'], NET_CONFIG, device=torch.device('cuda:0'))
    lib = L.load()
    assert lib.ttsamd_fastpitch_set_batch_mode(eng.handle, 2) != 0 and b'mode 2' in lib.ttsamd_last_error()
    assert lib.ttsamd_fastpitch_set_batch_mode(None, 1) != 0
    assert lib.ttsamd_fastpitch_set_batch_mode(eng.handle, 1) == 0 and lib.ttsamd_fastpitch_set_batch_mode(eng.handle, 0) == 0
