"""GPU (pytest -m gpu): the fused c1 -> c2 ResBlock1 pair kernels of the fp32 engine (csrc/resblock_fused.hip, resblock_fused2.hip)
driven one launch at a time through the C ABI (ttsamd_resblock_pair) against the reference's ops (vocoder/hifigan/models.py:46-53)
in float64, and forced on inside the whole generator against the oracle.

Cases the domain offers: ragged batches whose utterances end inside a block's halo (every conv zero-pads at the TRUE edge,
SURVEY §3.4-5: the reference vocodes exact-length mels), an empty utterance, lengths that are not a multiple of the block's
output count, all three accumulate modes of the stage sum, every (C, k, dilation) of the generator."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import WAVE_TOL

pytestmark = pytest.mark.gpu


def _ref_pair(x, w1, b1, w2, b2, dil, lens, slope=0.1):
    """per utterance on its exact length, float64"""
    k = w1.shape[2]
    out = torch.zeros_like(x, dtype=torch.float64)
    for b, n in enumerate(lens):
        if n == 0:
            continue
        xb = x[b:b + 1, :, :n].double()
        t = F.conv1d(F.leaky_relu(xb, slope), w1.double(), b1.double(), dilation=dil, padding=(k * dil - dil) // 2)
        t = F.conv1d(F.leaky_relu(t, slope), w2.double(), b2.double(), padding=(k - 1) // 2)
        out[b, :, :n] = (xb + t)[0]
    return out


CASES = [(c, k, d, v) for c in (32, 64, 128) for k in (3, 7, 11) for d in (1, 3, 5) for v in (2, 3)
         if (d == 5 or (c, k) in ((64, 3), (128, 7)))]             # every (C, k) at the widest dilation, two at all three
CASES += [(32, k, 3, 1) for k in (3, 7, 11)] + [(64, 3, 5, 1)]      # the first-generation kernel through the same entry
# variant 4: the second generation with phase B (c2, dilation 1) as Winograd F(2,3) over the even / odd column arrays of the intermediate
CASES += [(c, k, d, 4) for c in (32, 64) for k in (3, 7, 11) for d in (1, 5)] + [(64, 7, 3, 4)]
# variant 5: phase A too (output pairs (n, n + d): 256 / 252 / 250 intermediate columns per block at d = 1 / 3 / 5), window staged activated,
# residual from memory in the row epilogue
CASES += [(c, k, d, 5) for c in (32, 64) for k in (3, 7, 11) for d in (1, 3, 5)]
CASES += [(128, 3, 1, 5), (128, 3, 3, 5), (128, 3, 5, 5), (128, 7, 3, 5), (128, 11, 5, 5)]      # C = 128: the 8-wave block (two row halves)


@pytest.mark.parametrize('C,k,dil,variant', CASES)
def test_resblock_pair_kernel_vs_float64(C, k, dil, variant):
    from ttsamd.engine import resblock_pair
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1000 * C + 10 * k + dil)
    # block outputs: 256 - (k-1) & ~3 (variant 1 / 2), 128 - (k-1) & ~3 (variant 3): lengths around one and two blocks, one
    # utterance ending inside the halo of a block edge, one shorter than the kernel, one empty
    ts = ((128 if variant == 3 else 256) - (k - 1)) & ~3
    if variant == 5:
        ts = (2 * dil * (128 // dil) - (k - 1)) & ~3
    lens = [2 * ts + 8, ts + 4, ts - 4, ts, 4, 0, 3 * ts - 12]
    Lx = max(lens)
    x = torch.randn(len(lens), C, Lx, generator=g)
    w1 = torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
    w2 = torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    ref = _ref_pair(x, w1, b1, w2, b2, dil, lens)
    xd, lens_d = x.to(dev), torch.tensor(lens, device=dev)
    args = (xd, w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), dil)
    y = resblock_pair(*args, lens=lens_d, variant=variant)
    torch.cuda.synchronize()
    y = y.cpu()
    for b, n in enumerate(lens):
        err = float((y[b, :, :n].double() - ref[b, :, :n]).abs().max()) if n else 0.0
        assert err < 2e-5, (C, k, dil, variant, b, n, err)
        assert n == Lx or float(y[b, :, n:].abs().max()) == 0.0                 # nothing is written past the utterance
    # accumulate modes of the stage sum (models.py:119-122): y_prev + v, (y_prev + v) / 3 -- bit-identical to the mode-0 result combined in fp32
    prev = torch.randn(len(lens), C, Lx, generator=g)
    for mode, div in ((1, 1.0), (2, 3.0)):
        ya = resblock_pair(*args, lens=lens_d, y=prev.to(dev).clone(), mode=mode, div=div, variant=variant).cpu()
        for b, n in enumerate(lens):
            want = (prev[b, :, :n].double() + ref[b, :, :n]) / div
            assert n == 0 or float((ya[b, :, :n].double() - want).abs().max()) < 2e-5, (mode, b)
            assert torch.equal(ya[b, :, n:], prev[b, :, n:])                     # untouched past the utterance
    # run-to-run bit determinism
    assert torch.equal(resblock_pair(*args, lens=lens_d, variant=variant).cpu(), y)


def test_resblock_pair_len_mul_and_full_batch():
    """lens given in mel frames with len_mul (how the generator calls it), and lens = NULL (every utterance is L long)."""
    from ttsamd.engine import resblock_pair
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    C, k, dil = 64, 7, 3
    frames, mul = [5, 2, 7], 64
    Lx = max(frames) * mul
    x = torch.randn(3, C, Lx, generator=g)
    w1, w2 = torch.randn(C, C, k, generator=g) / np.sqrt(C * k), torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    args = (x.to(dev), w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), dil)
    for variant in (2, 3):
        y = resblock_pair(*args, lens=torch.tensor(frames, device=dev), len_mul=mul, variant=variant).cpu()
        ref = _ref_pair(x, w1, b1, w2, b2, dil, [f * mul for f in frames])
        assert float((y.double() - ref).abs().max()) < 2e-5
        y = resblock_pair(*args, variant=variant).cpu()
        ref = _ref_pair(x, w1, b1, w2, b2, dil, [Lx] * 3)
        assert float((y.double() - ref).abs().max()) < 2e-5


@pytest.mark.parametrize('mask,mask_n1', [('1ff', '000'), ('1ff', '1ff')])
def test_hifigan_every_pair_on_the_second_generation_kernel_vs_oracle(synth_weights, monkeypatch, mask, mask_n1, ttsopt):
    """The whole generator with EVERY ResBlock pair of the C = 128 / 64 / 32 stages forced onto resblock_pair2 (256-column blocks, then
    128-column blocks) against the oracle on a ragged batch whose utterances end inside a block's halo: 1-, 2- and 4-frame utterances are
    64 ... 1024 positions in those stages, 4 ... 12 columns around a block edge (blocks hold 252 / 248 / 244 resp. 124 / 120 / 116 outputs)."""
    import tts_oracle as O
    from ttsamd.config import HIFIGAN_CONFIG
    from ttsamd.engine import HifiGanEngine
    dev = torch.device('cuda:0')
    ttsopt.set('TTSAMD_FUSED_PAIR', '1')
    ttsopt.set('TTSAMD_FUSED2', '1')
    ttsopt.set('TTSAMD_FUSED2_MASK', mask)
    ttsopt.set('TTSAMD_FUSED2_MASK_N1', mask_n1)
    w = O.fold_weight_norm(synth_weights['hifigan'])
    rng = np.random.default_rng(11)
    lens = [19, 1, 2, 4, 8]
    mel = (rng.standard_normal((5, 80, 19)) * 1.5 - 4.0).astype(np.float32)
    wave = HifiGanEngine(synth_weights['hifigan'], device=dev).forward(torch.from_numpy(mel).to(dev), torch.tensor(lens).to(dev)).cpu()
    for b, n in enumerate(lens):
        ref = O.hifigan_forward(w, mel[b, :, :n], HIFIGAN_CONFIG)[0]
        err = float((wave[b, :256 * n] - ref.reshape(-1)).abs().max())
        assert err < WAVE_TOL, (mask_n1, b, err)
        assert n == 19 or float(wave[b, 256 * n:].abs().max()) == 0.0


def test_resblock_pair_entry_rejects_what_it_cannot_run():
    """C-ABI error behaviour of the kernel-level entry: an unknown variant, a geometry no instantiation covers, x == y (the halo of
    a neighbouring block would read what this block wrote) and a length that is not a multiple of 4 return an error code with a
    message instead of launching anything."""
    from ttsamd.engine import resblock_pair
    from ttsamd.lib import TtsAmdError
    dev = torch.device('cuda:0')
    x = torch.randn(1, 64, 64, device=dev)
    w = torch.randn(64, 64, 3, device=dev)
    b = torch.zeros(64, device=dev)
    with pytest.raises(TtsAmdError, match='variant'):
        resblock_pair(x, w, b, w, b, 1, variant=9)
    with pytest.raises(TtsAmdError, match='unsupported geometry'):
        resblock_pair(x, torch.randn(64, 64, 5, device=dev), b, torch.randn(64, 64, 5, device=dev), b, 1, variant=2)      # k = 5
    with pytest.raises(TtsAmdError, match='unsupported geometry'):
        resblock_pair(x, w, b, w, b, 7, variant=2)                                                                          # dilation > 5
    with pytest.raises(TtsAmdError, match='unsupported geometry'):
        resblock_pair(x, w, b, w, b, 1, y=x, variant=3)                                                                     # in place
    with pytest.raises(TtsAmdError, match='unsupported geometry'):
        resblock_pair(x[:, :, :62].contiguous(), w, b, w, b, 1, variant=2)                                                  # L % 4 != 0
    with pytest.raises(TtsAmdError, match='unsupported geometry'):
        resblock_pair(torch.randn(1, 128, 64, device=dev), torch.randn(128, 128, 3, device=dev), torch.zeros(128, device=dev),
                      torch.randn(128, 128, 3, device=dev), torch.zeros(128, device=dev), 1, variant=1)                     # first generation: C <= 64
