"""GPU parity of the split-bf16 ("x3") mode of the octet engine, layer by layer through the C ABI (pytest -m gpu).

Checker: the reference's ops (torch conv1d / conv_transpose1d / leaky_relu, vocoder/hifigan/models.py:46-53, 96-99, 111-127) in
float64 on the host with the TRUE fp32 weights, fed the stored (hi + lo) activations.  The mode's claim is fp32-class results
on bf16 MFMAs: every operand carries 16 mantissa bits (2^-17 relative), the dropped Wl xl term is 2^-18, accumulation is fp32.
Stated tolerance per layer: |diff| <= 2^-13 |ref| + 5e-5 -- 60x tighter than the plain bf16 engine's (tests/test_gpu_bfo.py:
2^-7 |ref| + 1e-2), so a missing lo term (error 2^-9) fails; an indexing bug shows up as O(1).  The end-to-end north-star
tolerances (mel 1e-3, wave 1e-4) are in tests/test_gpu_fullsize.py / test_gpu_parity.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import MEL_TOL, WAVE_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from ttsamd import lib
    assert lib.load().ttsamd_device_ok() == 1
    return torch.device('cuda:0')


def lrelu(x, s):
    return torch.where(x > 0, x, x * s)


def close(got, ref, what, rel=2.0 ** -13, abs_=5e-5):
    got, ref = got.double(), ref.double()
    err = (got - ref).abs()
    tol = ref.abs() * rel + abs_
    bad = err > tol
    assert not bool(bad.any()), f'{what}: {int(bad.sum())} of {bad.numel()} off, worst {float(err.max()):.3e} ' \
                                f'(ref there {float(ref.flatten()[err.argmax()]):.4f})'
    return float((err / (ref.abs() + 1)).max())


def test_pack3_unpack3_roundtrip(dev):
    from ttsamd import bfo
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 80, 333, generator=g) * 3
    for slope in (1.0, 0.1, 0.01):
        t = bfo.pack3(x.to(dev), slope)
        assert t.shape == (2, 10, 333, 16)
        raw = bfo.unpack3(t, 1.0).cpu()
        a = lrelu(x, slope).float()
        hi = a.to(torch.bfloat16).float()
        lo = (a - hi).to(torch.bfloat16).float()
        assert torch.equal(raw, hi + lo)                                           # hi = RNE bf16, lo = RNE bf16 of the exact rest
        assert float(((raw - a).abs() / a.abs().clamp_min(1e-30)).max()) <= 2.0 ** -16
        back = bfo.unpack3(t, slope).cpu()
        assert float(((back - x).abs() / x.abs()).max()) <= 2.0 ** -15


def _pair_ref(a, w1, b1, w2, b2, k, dil, sum_raw, mode, div, in_slope, out_slope):
    """a: stored (activated) input [C, n] float64 of one utterance -> stored output [C, n]; fp32 weights in float64."""
    x_raw = torch.where(a >= 0, a, a / in_slope)
    t = F.conv1d(a[None], w1.double(), b1.double(), dilation=dil, padding=(k - 1) * dil // 2)[0]
    t = lrelu(t, 0.1)
    v = x_raw + F.conv1d(t[None], w2.double(), b2.double(), padding=(k - 1) // 2)[0]
    if mode != 0:
        v = v + sum_raw
    if mode == 2:
        v = v / div
    return lrelu(v, out_slope)


@pytest.mark.parametrize('C,k,dil,L,mode,out_slope', [
    (128, 3, 1, 700, 0, 0.1), (128, 7, 3, 515, 1, 1.0), (128, 11, 5, 300, 2, 0.01), (128, 11, 3, 400, 0, 0.1), (128, 11, 1, 260, 1, 1.0),
    (128, 3, 5, 300, 2, 0.1), (128, 7, 5, 300, 0, 0.1),
    (64, 3, 5, 1100, 2, 0.1), (64, 7, 1, 600, 0, 0.1), (64, 11, 3, 1000, 1, 1.0), (64, 11, 5, 600, 0, 0.1),
    (32, 3, 3, 2100, 1, 1.0), (32, 7, 5, 1030, 2, 0.01), (32, 11, 1, 2500, 0, 0.1), (32, 11, 5, 1200, 1, 0.1),
])
def test_resblock_pair3(dev, C, k, dil, L, mode, out_slope):
    """Fused c1 -> c2 pair vs the reference ops in float64; ragged batch incl. an utterance ending inside a tile halo, one ending
    before the first tile boundary, and the untouched tail past each length; every window geometry of the launcher (two 4-wave blocks
    per CU at C <= 64, the 8-wave block of C = 128)."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(C * 100 + k * 10 + dil)
    B = 3
    x = torch.randn(B, C, L, generator=g) * 1.5
    w1 = torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
    w2 = torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
    b1, b2 = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    s_raw = torch.randn(B, C, L, generator=g)
    ncols = {128: 256, 64: 256, 32: 512}[C]                                            # C = 128: one 8-wave block, 2 x 128 columns
    ts = ncols - (k - 1)                                                               # outputs per block (Bfo3PairGeo::TS)
    lens = torch.tensor([L, min(L, ts + 2), max(1, min(L, ts) - 5)], dtype=torch.int64)     # ends 2 columns into tile 1 / inside tile 0
    xo = bfo.pack3(x.to(dev), 0.1)
    so = bfo.pack3(s_raw.to(dev), 1.0)
    y = torch.full_like(xo, 0x4242)
    if mode != 0:
        y.copy_(so)                                                                    # sum_in may alias y
    bfo.resblock_pair3(xo, bfo.pack_weight3(w1, device=dev), b1.to(dev), bfo.pack_weight3(w2, device=dev), b2.to(dev), k, dil,
                       lens=lens.to(dev), sum_in=y if mode != 0 else None, mode=mode, div=3.0, out_slope=out_slope, y=y)
    got = bfo.unpack3(y, 1.0).cpu()
    a_all = bfo.unpack3(xo, 1.0).cpu().double()
    s_all = bfo.unpack3(so, 1.0).cpu().double()
    before = bfo.unpack3(so if mode != 0 else torch.full_like(xo, 0x4242), 1.0).cpu()
    worst = 0.0
    for i in range(B):
        n = int(lens[i])
        ref = _pair_ref(a_all[i, :, :n], w1, b1, w2, b2, k, dil, s_all[i, :, :n], mode, 3.0, 0.1, out_slope)
        worst = max(worst, close(got[i, :, :n], ref, f'utt {i} (len {n})'))
        assert torch.equal(got[i, :, n:], before[i, :, n:]), 'positions past the utterance must stay untouched'
    print(f'pair3 C={C} k={k} d={dil}: worst |err| / (|ref| + 1) = {worst:.2e}')


def test_resblock_pair3_is_deterministic_and_batch_independent(dev):
    """Same bits run to run, and an utterance's result does not depend on its neighbours in the batch (compact block order)."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(3)
    C, k, dil, L = 64, 7, 3, 900
    x = torch.randn(4, C, L, generator=g)
    w1, w2 = torch.randn(C, C, k, generator=g) / np.sqrt(C * k), torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
    b1, b2 = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    lens = torch.tensor([900, 333, 1, 640], dtype=torch.int64)
    xo = bfo.pack3(x.to(dev), 0.1)
    args = (bfo.pack_weight3(w1, device=dev), b1.to(dev), bfo.pack_weight3(w2, device=dev), b2.to(dev), k, dil)
    y1 = bfo.resblock_pair3(xo, *args, lens=lens.to(dev))
    y2 = bfo.resblock_pair3(xo, *args, lens=lens.to(dev))
    assert torch.equal(y1, y2)
    for i in range(4):
        n = int(lens[i])
        yi = bfo.resblock_pair3(xo[i:i + 1, :, :n].contiguous(), *args)
        assert torch.equal(yi[0], y1[i, :, :n]), i


@pytest.mark.parametrize('cin,cout,k,dil,L,mode,res', [
    (256, 256, 3, 1, 600, 0, False), (256, 256, 7, 3, 300, 1, True), (256, 256, 11, 5, 515, 2, True),
    (80, 512, 7, 1, 90, 0, False), (128, 64, 3, 1, 700, 0, False), (64, 32, 7, 1, 1500, 0, True), (512, 128, 1, 1, 100, 0, False),
    (384, 1536, 3, 1, 64, 0, False), (384, 192, 1, 1, 500, 0, False),
])
def test_conv1d3(dev, cin, cout, k, dil, L, mode, res):
    from ttsamd import bfo
    g = torch.Generator().manual_seed(cin + cout + k)
    B = 2
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g) * 0.3
    r_raw, s_raw = torch.randn(B, cout, L, generator=g), torch.randn(B, cout, L, generator=g)
    lens = torch.tensor([L, max(1, L - 37)], dtype=torch.int64)
    xo, ro, so = bfo.pack3(x.to(dev), 0.1), bfo.pack3(r_raw.to(dev), 0.1), bfo.pack3(s_raw.to(dev), 1.0)
    y = so.clone() if mode != 0 else torch.full_like(so, 0x4242)
    bfo.conv1d3(xo, bfo.pack_weight3(w, device=dev), b.to(dev), cout, k, dilation=dil, lens=lens.to(dev), res=ro if res else None,
                res_slope=0.1, sum_in=y if mode != 0 else None, mode=mode, div=3.0, out_slope=0.1, y=y)
    got = bfo.unpack3(y, 1.0).cpu()
    a_all, r_all, s_all = (bfo.unpack3(t, 1.0).cpu().double() for t in (xo, ro, so))
    for i in range(B):
        n = int(lens[i])
        v = F.conv1d(a_all[i:i + 1, :, :n], w.double(), b.double(), dilation=dil, padding=(k - 1) * dil // 2)[0]
        if res:
            rr = r_all[i, :, :n]
            v = v + torch.where(rr >= 0, rr, rr / 0.1)
        if mode != 0:
            v = v + s_all[i, :, :n]
        if mode == 2:
            v = v / 3.0
        close(got[i, :, :n], lrelu(v, 0.1), f'utt {i}')


@pytest.mark.parametrize('S', [64, 40, 300])
def test_conv_ff_pair3_fp32_stream(dev, S):
    """FastPitch's PositionwiseConvFF core (transformer.py:72-90) in this mode: Conv1d(384 -> 1536, k3) + ReLU with the 1536-channel
    intermediate as an x3 tensor, Conv1d(1536 -> 384, k3) + fp32 residual -> fp32 channel-first."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(S)
    B, d, di = 3, 384, 1536
    x = torch.randn(B, d, S, generator=g)
    w0 = torch.randn(di, d, 3, generator=g) / np.sqrt(d * 3)
    w2 = torch.randn(d, di, 3, generator=g) / np.sqrt(di * 3)
    b0, b2 = torch.randn(di, generator=g) * 0.3, torch.randn(d, generator=g) * 0.3
    xo = bfo.pack3(x.to(dev), 1.0)
    hid = bfo.conv1d3(xo, bfo.pack_weight3(w0, device=dev), b0.to(dev), di, 3, out_slope=0.0)
    y = bfo.conv1d3(hid, bfo.pack_weight3(w2, device=dev), b2.to(dev), d, 3, f32_out=True, res_f32=x.to(dev)).cpu()
    a = bfo.unpack3(xo, 1.0).cpu().double()
    h_ref = torch.relu(F.conv1d(a, w0.double(), b0.double(), padding=1))
    close(bfo.unpack3(hid, 1.0).cpu(), h_ref, 'intermediate')
    ref = F.conv1d(h_ref, w2.double(), b2.double(), padding=1) + x.double()
    err = float((y.double() - ref).abs().max())
    assert err < 5e-5, err


@pytest.mark.parametrize('cin,cout,u,L', [(512, 256, 8, 150), (256, 128, 8, 300), (128, 64, 2, 700), (64, 32, 2, 1100)])
def test_conv_transpose1d3(dev, cin, cout, u, L):
    """The four HiFi-GAN upsamplers (models.py:96-99): ConvTranspose1d(kernel 2u, stride u, padding u/2)."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(cin + u)
    B = 2
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cin, cout, 2 * u, generator=g) / np.sqrt(cin * 2)
    b = torch.randn(cout, generator=g) * 0.3
    lens = torch.tensor([L, max(1, L - 41)], dtype=torch.int64)
    xo = bfo.pack3(x.to(dev), 0.1)
    y = torch.full((B, cout // 8, L * u, 16), 0x4242, dtype=torch.int16, device=dev)
    bfo.conv1d3(xo, bfo.pack_weight3(w, up=u, device=dev), b.to(dev), cout, 2 * u, up=u, lens=lens.to(dev), out_slope=0.1, y=y)
    got = bfo.unpack3(y, 1.0).cpu()
    untouched = bfo.unpack3(torch.full((1, 1, 1, 16), 0x4242, dtype=torch.int16, device=dev), 1.0).cpu().flatten()[0]
    a_all = bfo.unpack3(xo, 1.0).cpu().double()
    for i in range(B):
        n = int(lens[i])
        v = F.conv_transpose1d(a_all[i:i + 1, :, :n], w.double(), b.double(), stride=u, padding=u // 2)[0]
        close(got[i, :, :n * u], lrelu(v, 0.1), f'utt {i}')
        if n < L:
            assert bool((got[i, :, n * u:] == untouched).all())


def test_conv_post3(dev):
    from ttsamd import bfo
    g = torch.Generator().manual_seed(5)
    B, C, L = 2, 32, 1500
    x = torch.randn(B, C, L, generator=g)
    w = torch.randn(1, C, 7, generator=g) / np.sqrt(C * 7)
    b = torch.randn(1, generator=g) * 0.1
    lens = torch.tensor([L, L - 300], dtype=torch.int64)
    xo = bfo.pack3(x.to(dev), 0.01)
    wave = bfo.conv_post3(xo, w.reshape(C, 7).contiguous().to(dev), b.to(dev), lens=lens.to(dev)).cpu()
    a_all = bfo.unpack3(xo, 1.0).cpu().double()
    for i in range(B):
        n = int(lens[i])
        ref = torch.tanh(F.conv1d(a_all[i:i + 1, :, :n], w.double(), b.double(), padding=3))[0, 0]
        assert float((wave[i, :n].double() - ref).abs().max()) < 1e-5
        assert float(wave[i, n:].abs().max()) == 0.0 if n < L else True


@pytest.mark.parametrize('C,L,mode,out_slope,B', [(128, 700, 0, 1.0, 3), (64, 900, 1, 1.0, 3), (32, 1500, 2, 0.1, 3), (128, 300, 2, 0.01, 70),
                                                  (64, 520, 0, 0.1, 5), (32, 1100, 1, 1.0, 4)])
def test_resblock_chain3_equals_three_pairs_bit_for_bit(dev, C, L, mode, out_slope, B):
    """The whole k = 3 ResBlock in one launch (bfo3_chain.hip) against three fused-pair launches of the same weights: identical bits
    (the chained kernel splits the tensor between two pairs exactly where the pair launch splits it for HBM) -- and the pairs are
    checked against float64 above.  Ragged batch: an utterance ending inside the halo of a tile (24 of 256 / 512 columns), one shorter
    than a tile, (one case) more than 64 utterances."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(C + L + mode)
    k, dils = 3, (1, 3, 5)
    x = torch.randn(B, C, L, generator=g) * 1.5
    ws = [[torch.randn(C, C, k, generator=g) / np.sqrt(C * k) for _ in range(3)] for _ in range(2)]
    bs = [[torch.randn(C, generator=g) * 0.3 for _ in range(3)] for _ in range(2)]
    s_raw = torch.randn(B, C, L, generator=g)
    ts = {128: 256, 64: 256, 32: 512}[C] - 24
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0], lens[1], lens[2] = L, min(L, ts + 5), max(1, min(L, ts) - 9)
    lens_d = lens.to(dev)
    xo, so = bfo.pack3(x.to(dev), 0.1), bfo.pack3(s_raw.to(dev), 1.0)
    w1p, w2p = [bfo.pack_weight3(w, device=dev) for w in ws[0]], [bfo.pack_weight3(w, device=dev) for w in ws[1]]
    b1d, b2d = [b.to(dev) for b in bs[0]], [b.to(dev) for b in bs[1]]
    t = xo
    for m in range(3):                      # three pair launches, as csrc/hifigan.hip issues them under TTSAMD_BFO_CHAIN=0
        last = m == 2
        y = (so.clone() if mode != 0 else torch.full_like(xo, 0x4242)) if last else torch.zeros_like(xo)
        bfo.resblock_pair3(t, w1p[m], b1d[m], w2p[m], b2d[m], k, dils[m], lens=lens_d, sum_in=y if (last and mode != 0) else None,
                           mode=mode if last else 0, div=3.0, out_slope=out_slope if last else 0.1, y=y)
        t = y
    ref = t
    y = so.clone() if mode != 0 else torch.full_like(xo, 0x4242)
    bfo.resblock_chain3(xo, w1p, b1d, w2p, b2d, dils, lens=lens_d, sum_in=y if mode != 0 else None, mode=mode, div=3.0,
                        out_slope=out_slope, y=y)
    torch.cuda.synchronize()
    assert torch.equal(y, ref)


def test_hifigan_x3_chained_resblocks_change_no_bit(dev, synth_weights, monkeypatch, ttsopt):
    """The generator's launch schedule in this mode is a routing choice, not a numeric one: k = 3 ResBlocks as one launch (default)
    or as three pair launches (TTSAMD_BFO_CHAIN=0) give the same wave bit for bit on a ragged batch."""
    from ttsamd.engine import HifiGanEngine, set_precision
    g = torch.Generator().manual_seed(9)
    mel = torch.randn(3, 80, 57, generator=g).to(dev)
    lens = torch.tensor([57, 31, 4], dtype=torch.int64, device=dev)
    set_precision('bf16x3')
    try:
        hg = HifiGanEngine(synth_weights['hifigan'], device=dev)
        wave = hg.forward(mel, lens).clone()
        ttsopt.set('TTSAMD_BFO_CHAIN', '0')
        wave_p = hg.forward(mel, lens).clone()
    finally:
        set_precision('f32')
    assert torch.equal(wave, wave_p)
