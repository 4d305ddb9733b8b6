"""GPU parity of the bf16 octet engine (BASELINE config 3), layer by layer through the C ABI (pytest -m gpu).

Checker: the reference's ops (torch conv1d / conv_transpose1d / leaky_relu, vocoder/hifigan/models.py:46-53, 96-99,
111-127) in float64 on the host, fed the SAME bf16-rounded operands the kernel sees (stored activations, bf16 weights, the
bf16 c1 -> c2 intermediate), so what is left is fp32-vs-fp64 accumulation plus one bf16 rounding of the output:
|diff| <= 2^-7 |ref| + 1e-2 (stated tolerance; an indexing bug shows up as O(1)).  The end-to-end bf16 tolerances
against the fp32 oracle are in tests/test_gpu_fullsize.py / test_gpu_parity.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import BF16_MEL_TOL, BF16_WAVE_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from ttsamd import lib
    assert lib.load().ttsamd_device_ok() == 1
    return torch.device('cuda:0')


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def lrelu(x, s):
    return torch.where(x > 0, x, x * s)


def close(got, ref, what):
    got, ref = got.double(), ref.double()
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -7 + 1e-2
    bad = err > tol
    assert not bool(bad.any()), f'{what}: {int(bad.sum())} of {bad.numel()} off, worst {float(err.max()):.4f} ' \
                                f'(ref there {float(ref.flatten()[err.argmax()]):.4f})'


def test_pack_unpack_roundtrip(dev):
    from ttsamd import bfo
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 80, 333, generator=g) * 3
    for slope in (1.0, 0.1, 0.01):
        t = bfo.pack(x.to(dev), slope)
        raw = bfo.unpack(t, 1.0).cpu()
        assert torch.equal(raw.double(), bf(lrelu(x, slope).float()))              # RNE, exactly torch's bf16 cast
        back = bfo.unpack(t, slope).cpu()
        assert float((back - x).abs().max()) <= float(x.abs().max()) * 2.0 ** -7


def _pair_ref(a, w1, b1, w2, b2, k, dil, n, sum_raw, mode, div, in_slope, out_slope):
    """a: stored (activated, bf16-valued) input [C, n] float64 of one utterance -> stored output [C, n]."""
    x_raw = torch.where(a >= 0, a, a / in_slope)
    t = F.conv1d(a[None], bf(w1), b1.double(), dilation=dil, padding=(k - 1) * dil // 2)[0]
    t = bf(lrelu(t, 0.1).float())
    v = x_raw + F.conv1d(t[None], bf(w2), b2.double(), padding=(k - 1) // 2)[0]
    if mode != 0:
        v = v + sum_raw
    if mode == 2:
        v = v / div
    return lrelu(v, out_slope)


@pytest.mark.parametrize('C,k,dil,L,mode,out_slope', [
    (128, 3, 1, 700, 0, 0.1), (128, 7, 3, 515, 1, 1.0), (128, 11, 5, 300, 2, 0.01),
    (64, 3, 5, 1100, 2, 0.1), (64, 7, 1, 600, 0, 0.1), (64, 11, 3, 1000, 1, 1.0),
    (32, 3, 3, 2100, 1, 1.0), (32, 7, 5, 1030, 2, 0.01), (32, 11, 1, 2500, 0, 0.1),
])
@pytest.mark.parametrize('small_tiles', ['0', '1'])
def test_resblock_pair(dev, monkeypatch, C, k, dil, L, mode, out_slope, small_tiles, ttsopt):
    """Fused c1 -> c2 pair vs the reference ops; ragged batch incl. an utterance ending inside a tile halo, one ending
    before the first tile boundary, and the untouched tail past each length."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(C * 100 + k * 10 + dil)
    B = 3
    x = torch.randn(B, C, L, generator=g) * 1.5
    w1 = torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
    w2 = torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
    b1, b2 = torch.randn(C, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    s_raw = torch.randn(B, C, L, generator=g)
    # both tile widths of the kernel: 256 columns per wave, or 128 (always for k = 3 at C <= 64; for small grids otherwise)
    ttsopt.set('TTSAMD_BFO_SMALL_TILES', small_tiles)
    half = small_tiles == '1' or (k == 3 and C <= 64)
    ts = {128: 256, 64: 512, 32: 1024}[C] // (2 if half else 1) - (k - 1)                     # outputs per block (BfoPairGeo::TS)
    lens = torch.tensor([L, min(L, ts + 2), max(1, min(L, ts) - 5)], dtype=torch.int64)     # ends 2 columns into tile 1 / inside tile 0
    xo = bfo.pack(x.to(dev), 0.1)
    so = bfo.pack(s_raw.to(dev), 1.0)
    y = torch.full_like(xo, 0x4242)
    if mode != 0:
        y.copy_(so)                                                                    # sum_in may alias y
    bfo.resblock_pair(xo, bfo.pack_weight(w1, device=dev), b1.to(dev), bfo.pack_weight(w2, device=dev), b2.to(dev), k, dil,
                      lens=lens.to(dev), sum_in=y if mode != 0 else None, mode=mode, div=3.0, out_slope=out_slope, y=y)
    got = bfo.unpack(y, 1.0).cpu()
    a_all = bfo.unpack(xo, 1.0).cpu().double()
    s_all = bfo.unpack(so, 1.0).cpu().double()
    before = bfo.unpack(so if mode != 0 else torch.full_like(xo, 0x4242), 1.0).cpu()
    for i in range(B):
        n = int(lens[i])
        ref = _pair_ref(a_all[i, :, :n], w1, b1, w2, b2, k, dil, n, s_all[i, :, :n], mode, 3.0, 0.1, out_slope)
        close(got[i, :, :n], ref, f'utt {i} (len {n})')
        assert torch.equal(got[i, :, n:], before[i, :, n:]), 'positions past the utterance must stay untouched'


@pytest.mark.parametrize('C,L,mode,out_slope,B,k', [(128, 700, 0, 1.0, 3, 3), (64, 900, 1, 1.0, 3, 3), (32, 1500, 2, 0.1, 3, 3), (128, 300, 2, 0.01, 70, 3),
                                                    (64, 900, 0, 1.0, 3, 7), (32, 1500, 1, 1.0, 3, 7), (64, 700, 2, 0.01, 70, 7), (32, 1300, 2, 0.1, 5, 7)])
def test_resblock_chain_equals_three_pairs_bit_for_bit(dev, C, L, mode, out_slope, B, k):
    """The whole ResBlock in one launch (bfo_chain.hip; k = 3, and k = 7 at C <= 64) against three fused-pair launches of the same
    weights: identical bits (the chained kernel rounds the tensor between two pairs exactly where the pair launch rounds it for HBM).
    Ragged batch: an utterance ending inside the halo of a tile (24 of 256 / 512 columns at k = 3, 72 at k = 7), one shorter than a
    tile, (one case each) more than 64 utterances."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(C + L + mode + k)
    dils = (1, 3, 5)
    x = torch.randn(B, C, L, generator=g) * 1.5
    ws = [[torch.randn(C, C, k, generator=g) / np.sqrt(C * k) for _ in range(3)] for _ in range(2)]
    bs = [[torch.randn(C, generator=g) * 0.3 for _ in range(3)] for _ in range(2)]
    s_raw = torch.randn(B, C, L, generator=g)
    ts = {128: 256, 64: 256, 32: 512}[C] - (k - 1) * 12
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0], lens[1], lens[2] = L, min(L, ts + 5), max(1, min(L, ts) - 9)
    lens_d = lens.to(dev)
    xo, so = bfo.pack(x.to(dev), 0.1), bfo.pack(s_raw.to(dev), 1.0)
    w1p, w2p = [bfo.pack_weight(w, device=dev) for w in ws[0]], [bfo.pack_weight(w, device=dev) for w in ws[1]]
    b1d, b2d = [b.to(dev) for b in bs[0]], [b.to(dev) for b in bs[1]]
    # three pair launches, as csrc/hifigan.hip issues them
    t = xo
    for m in range(3):
        last = m == 2
        y = (so.clone() if mode != 0 else torch.full_like(xo, 0x4242)) if last else torch.zeros_like(xo)
        bfo.resblock_pair(t, w1p[m], b1d[m], w2p[m], b2d[m], k, dils[m], lens=lens_d, sum_in=y if (last and mode != 0) else None,
                          mode=mode if last else 0, div=3.0, out_slope=out_slope if last else 0.1, y=y)
        t = y
    ref = t
    y = so.clone() if mode != 0 else torch.full_like(xo, 0x4242)
    bfo.resblock_chain(xo, w1p, b1d, w2p, b2d, dils, lens=lens_d, sum_in=y if mode != 0 else None, mode=mode, div=3.0,
                       out_slope=out_slope, y=y, k=k)
    torch.cuda.synchronize()
    assert torch.equal(y, ref)


@pytest.mark.parametrize('cin,cout,k,dil,L,mode,res', [
    (256, 256, 3, 1, 600, 0, False), (256, 256, 7, 3, 300, 1, True), (256, 256, 11, 5, 515, 2, True),
    (80, 512, 7, 1, 90, 0, False), (128, 64, 3, 1, 700, 0, False), (64, 32, 7, 1, 1500, 0, True), (512, 128, 1, 1, 100, 0, False),
])
def test_conv1d(dev, cin, cout, k, dil, L, mode, res):
    from ttsamd import bfo
    g = torch.Generator().manual_seed(cin + cout + k)
    B = 2
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g) * 0.3
    r_raw, s_raw = torch.randn(B, cout, L, generator=g), torch.randn(B, cout, L, generator=g)
    lens = torch.tensor([L, max(1, L - 37)], dtype=torch.int64)
    xo, ro, so = bfo.pack(x.to(dev), 0.1), bfo.pack(r_raw.to(dev), 0.1), bfo.pack(s_raw.to(dev), 1.0)
    y = so.clone() if mode != 0 else torch.full_like(so, 0x4242)
    bfo.conv1d(xo, bfo.pack_weight(w, device=dev), b.to(dev), cout, k, dilation=dil, lens=lens.to(dev), res=ro if res else None,
               res_slope=0.1, sum_in=y if mode != 0 else None, mode=mode, div=3.0, out_slope=0.1, y=y)
    got = bfo.unpack(y, 1.0).cpu()
    a_all, r_all, s_all = (bfo.unpack(t, 1.0).cpu().double() for t in (xo, ro, so))
    for i in range(B):
        n = int(lens[i])
        v = F.conv1d(a_all[i:i + 1, :, :n], bf(w), b.double(), dilation=dil, padding=(k - 1) * dil // 2)[0]
        if res:
            rr = r_all[i, :, :n]
            v = v + torch.where(rr >= 0, rr, rr / 0.1)
        if mode != 0:
            v = v + s_all[i, :, :n]
        if mode == 2:
            v = v / 3.0
        close(got[i, :, :n], lrelu(v, 0.1), f'utt {i}')


@pytest.mark.parametrize('S', [64, 40, 300])
def test_conv_ff_pair_fp32_stream(dev, S):
    """FastPitch's PositionwiseConvFF core (transformer.py:72-90) as config 3 runs it: fp32 LayerNorm output packed to bf16,
    Conv1d(384 -> 1536, k3) + ReLU with a bf16 intermediate, Conv1d(1536 -> 384, k3) + fp32 residual -> fp32 channel-first;
    64-column tiles for the encoder's short sequences (S <= 96), 256-column tiles otherwise."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(S)
    B, d, di = 3, 384, 1536
    x = torch.randn(B, d, S, generator=g)
    w0 = torch.randn(di, d, 3, generator=g) / np.sqrt(d * 3)
    w2 = torch.randn(d, di, 3, generator=g) / np.sqrt(di * 3)
    b0, b2 = torch.randn(di, generator=g) * 0.3, torch.randn(d, generator=g) * 0.3
    xo = bfo.pack(x.to(dev), 1.0)
    hid = bfo.conv1d(xo, bfo.pack_weight(w0, device=dev), b0.to(dev), di, 3, out_slope=0.0)
    y = bfo.conv1d(hid, bfo.pack_weight(w2, device=dev), b2.to(dev), d, 3, f32_out=True, res_f32=x.to(dev)).cpu()
    a = bfo.unpack(xo, 1.0).cpu().double()
    h_ref = bf(torch.relu(F.conv1d(a, bf(w0), b0.double(), padding=1)).float())
    close(bfo.unpack(hid, 1.0).cpu(), h_ref, 'intermediate')
    ref = F.conv1d(h_ref, bf(w2), b2.double(), padding=1) + x.double()
    err = float((y.double() - ref).abs().max())
    assert err < 2e-3, err                                  # fp32 output: only accumulation-order noise on O(1) values


@pytest.mark.parametrize('cin,cout,u,L', [(512, 256, 8, 150), (256, 128, 8, 300), (128, 64, 2, 700), (64, 32, 2, 1100)])
def test_conv_transpose1d(dev, cin, cout, u, L):
    """The four HiFi-GAN upsamplers (models.py:96-99): ConvTranspose1d(kernel 2u, stride u, padding u/2)."""
    from ttsamd import bfo
    g = torch.Generator().manual_seed(cin + u)
    B = 2
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cin, cout, 2 * u, generator=g) / np.sqrt(cin * 2)
    b = torch.randn(cout, generator=g) * 0.3
    lens = torch.tensor([L, max(1, L - 41)], dtype=torch.int64)
    xo = bfo.pack(x.to(dev), 0.1)
    y = torch.full((B, cout // 8, L * u, 8), 0x4242, dtype=torch.int16, device=dev)
    bfo.conv1d(xo, bfo.pack_weight(w, up=u, device=dev), b.to(dev), cout, 2 * u, up=u, lens=lens.to(dev), out_slope=0.1, y=y)
    got = bfo.unpack(y, 1.0).cpu()
    a_all = bfo.unpack(xo, 1.0).cpu().double()
    for i in range(B):
        n = int(lens[i])
        v = F.conv_transpose1d(a_all[i:i + 1, :, :n], bf(w), b.double(), stride=u, padding=u // 2)[0]
        close(got[i, :, :n * u], lrelu(v, 0.1), f'utt {i}')
        assert bool((got[i, :, n * u:] == got[i, 0, -1]).all()) if n < L else True


def test_conv_post(dev):
    from ttsamd import bfo
    g = torch.Generator().manual_seed(5)
    B, C, L = 2, 32, 1500
    x = torch.randn(B, C, L, generator=g)
    w = torch.randn(1, C, 7, generator=g) / np.sqrt(C * 7)
    b = torch.randn(1, generator=g) * 0.1
    lens = torch.tensor([L, L - 300], dtype=torch.int64)
    xo = bfo.pack(x.to(dev), 0.01)
    wave = bfo.conv_post(xo, w.reshape(C, 7).contiguous().to(dev), b.to(dev), lens=lens.to(dev)).cpu()
    a_all = bfo.unpack(xo, 1.0).cpu().double()
    for i in range(B):
        n = int(lens[i])
        ref = torch.tanh(F.conv1d(a_all[i:i + 1, :, :n], w.double(), b.double(), padding=3))[0, 0]
        assert float((wave[i, :n].double() - ref).abs().max()) < 1e-5
        assert float(wave[i, n:].abs().max()) == 0.0 if n < L else True


@pytest.mark.parametrize('T', [1, 7, 40])
def test_hifigan_bf16_octet_engine_golden(dev, golden, synth_weights, T, ttsopt):
    """The whole generator on the octet engine vs the real reference's fp32 golden (stated bf16 tolerance), and
    that the round-2 bf16 engine (TTSAMD_BFO=0) and the octet engine agree to the same tolerance."""
    import os
    from ttsamd.engine import HifiGanEngine, set_precision
    gd = golden(f'hifigan_T{T}')
    mel = torch.from_numpy(gd['mel']).to(dev)
    set_precision('bf16')
    try:
        hg = HifiGanEngine(synth_weights['hifigan'], device=dev)
        wave = hg.forward(mel[None] if mel.dim() == 2 else mel).cpu().reshape(-1)
        ttsopt.set('TTSAMD_BFO', '0')
        wave_old = hg.forward(mel[None] if mel.dim() == 2 else mel).cpu().reshape(-1)
    finally:
        ttsopt.set('TTSAMD_BFO', None)
        set_precision('f32')
    ref = torch.from_numpy(gd['wave']).reshape(-1)
    err, err_old = float((wave - ref).abs().max()), float((wave_old - ref).abs().max())
    print(f'T={T}: octet engine wave max-abs {err:.2e}, round-2 bf16 engine {err_old:.2e}')
    assert err < BF16_WAVE_TOL and err_old < BF16_WAVE_TOL


def test_bf16_fft_block_matches_the_fp32_kernels(dev, synth_weights, monkeypatch, ttsopt):
    """FastPitch under config 3 with the whole FFT block on the bf16 matrix cores (octet-engine qkv / o_net / conv-FF convs, LayerNorm
    that writes the octet copies, bf16 MFMA attention: 32-query tiles, the four waves of a block split the keys) against the same run
    on the round-2 bf16 conv engine with the fp32 attention kernel, and against the exact-fp32 engine: ragged batch, sequence lengths that are not
    multiples of the 32-key tiles (70 / 41 / 13 tokens -> ~490 / 290 / 90 frames), so key masking, partial tiles, empty key
    ranges of a wave and the four-way combine are all on the path (transformer.py:131-141)."""
    from ttsamd import synth
    from ttsamd.engine import FastPitchEngine, set_precision
    ids = synth.synth_ids(3, 70)
    ids[1, 41:] = 0
    ids[2, 13:] = 0
    dur = synth.synth_durations(3, 70) * (ids != 0)
    fp = FastPitchEngine(synth_weights['fastpitch'], device=dev)
    mel32, lens32, *_ = fp.infer(ids, dur_tgt=dur)
    set_precision('bf16')
    try:
        ttsopt.set('TTSAMD_BF16_ATTN', '1')                       # the whole FFT block on the bf16 matrix cores (the default)
        ttsopt.set('TTSAMD_BFO_FF', '1')
        mel_a, lens_a, *_ = fp.infer(ids, dur_tgt=dur)
        ttsopt.set('TTSAMD_BF16_ATTN', '0')                       # round-2 bf16 conv engine + the fp32 attention kernel
        ttsopt.set('TTSAMD_BFO_FF', '0')
        mel_b, lens_b, *_ = fp.infer(ids, dur_tgt=dur)
    finally:
        set_precision('f32')
    assert torch.equal(lens_a, lens32) and torch.equal(lens_b, lens32)
    worst_ab = worst_a = worst_b = 0.0
    for i in range(3):
        n = int(lens32[i])
        worst_ab = max(worst_ab, float((mel_a[i, :, :n] - mel_b[i, :, :n]).abs().max()))
        worst_a = max(worst_a, float((mel_a[i, :, :n] - mel32[i, :, :n]).abs().max()))
        worst_b = max(worst_b, float((mel_b[i, :, :n] - mel32[i, :, :n]).abs().max()))
    print(f'bf16 attention vs fp32 attention (both under bf16 GEMMs): {worst_ab:.2e}; vs the fp32 engine: {worst_a:.2e} (fp32 attention: {worst_b:.2e})')
    assert bool(torch.isfinite(mel_a).all())
    assert worst_a < BF16_MEL_TOL and worst_ab < 4e-2


def test_hifigan_bf16_chained_resblocks_change_no_bit(dev, synth_weights, monkeypatch, ttsopt):
    """The generator's launch schedule is a routing choice, not a numeric one: three pair launches per ResBlock, the k = 3 ResBlocks
    chained, the k = 7 ResBlocks of the C = 32 / 64 stages chained as well (the default below 1.5 M columns) -- the same waveform bits on
    a ragged batch."""
    from ttsamd.engine import HifiGanEngine, set_precision
    rng = np.random.default_rng(5)
    lens = torch.tensor([23, 1, 9, 16])
    mel = torch.from_numpy((rng.standard_normal((4, 80, 23)) * 1.5 - 4.0).astype(np.float32)).to(dev)
    set_precision('bf16')
    try:
        hg = HifiGanEngine(synth_weights['hifigan'], device=dev)
        waves = []
        for chain, chain7 in (('0', '0'), ('1', '0'), ('1', '1')):
            ttsopt.set('TTSAMD_BFO_CHAIN', chain)
            ttsopt.set('TTSAMD_BFO_CHAIN7', chain7)
            waves.append(hg.forward(mel, lens.to(dev)).cpu())
        ttsopt.set('TTSAMD_BFO_CHAIN', None)
        ttsopt.set('TTSAMD_BFO_CHAIN7', None)
        waves.append(hg.forward(mel, lens.to(dev)).cpu())
    finally:
        set_precision('f32')
    assert float(waves[0].abs().max()) > 1e-3
    for w in waves[1:]:
        assert torch.equal(w, waves[0])


def test_split_k_small_batch(dev, synth_weights, monkeypatch, ttsopt):
    """Batch 1: the octet engine's convs split K over their C-in slabs (partial sums in fp32, summed in slice order by
    bfo_splitk_reduce: deterministic) because a batch-1 launch is a handful of blocks.  Same results as the un-split launches up to
    the fp32 summation order, run-to-run bit-identical, and inside the bf16 tolerances against the exact-fp32 engines."""
    from ttsamd import synth
    from ttsamd.engine import FastPitchEngine, HifiGanEngine, set_precision
    ids = synth.synth_ids(1, 40)
    dur = synth.synth_durations(1, 40)
    fp, hg = FastPitchEngine(synth_weights['fastpitch'], device=dev), HifiGanEngine(synth_weights['hifigan'], device=dev)
    mel32, lens32, *_ = fp.infer(ids, dur_tgt=dur)
    wave32 = hg.forward(mel32, lens32)
    set_precision('bf16')
    try:
        ttsopt.set('TTSAMD_BFO_SPLITK', '1')
        mel_a, _, *_ = fp.infer(ids, dur_tgt=dur)
        wave_a = hg.forward(mel32, lens32)
        mel_a2, _, *_ = fp.infer(ids, dur_tgt=dur)
        wave_a2 = hg.forward(mel32, lens32)
        ttsopt.set('TTSAMD_BFO_SPLITK', '0')
        mel_b, _, *_ = fp.infer(ids, dur_tgt=dur)
        wave_b = hg.forward(mel32, lens32)
    finally:
        set_precision('f32')
    assert torch.equal(mel_a, mel_a2) and torch.equal(wave_a, wave_a2)
    e_ab, e_a = float((mel_a - mel_b).abs().max()), float((mel_a - mel32).abs().max())
    w_ab, w_a = float((wave_a - wave_b).abs().max()), float((wave_a - wave32).abs().max())
    print(f'split K vs un-split: mel {e_ab:.2e}, wave {w_ab:.2e}; vs fp32: mel {e_a:.2e}, wave {w_a:.2e}')
    assert e_a < BF16_MEL_TOL and w_a < BF16_WAVE_TOL and e_ab < 2e-2 and w_ab < 2e-2


def test_resblock_chain_entry_rejects_what_it_cannot_run(dev):
    """C-ABI error behaviour of the kernel-level chain entry: k = 7 exists for C <= 64 only, k = 11 not at all, x == y never; what it can run it runs
    whatever the generator's routing switches say (TTSAMD_BFO_CHAIN / _CHAIN7 steer hifigan.hip, not the entry)."""
    from ttsamd import bfo
    from ttsamd.lib import TtsAmdError
    x = bfo.pack(torch.randn(1, 128, 300, device=dev), 0.1)
    def ws(C, k):
        return ([bfo.pack_weight(torch.randn(C, C, k) / np.sqrt(C * k), device=dev) for _ in range(3)], [torch.zeros(C, device=dev) for _ in range(3)])
    w7, b7 = ws(128, 7)
    with pytest.raises(TtsAmdError, match='unsupported geometry'):
        bfo.resblock_chain(x, w7, b7, w7, b7, (1, 3, 5), k=7)                    # C = 128 at k = 7
    w11, b11 = ws(128, 11)
    with pytest.raises(TtsAmdError, match='kernel size'):
        bfo.resblock_chain(x, w11, b11, w11, b11, (1, 3, 5), k=11)
    w3, b3 = ws(128, 3)
    with pytest.raises(TtsAmdError, match='differ'):
        bfo.resblock_chain(x, w3, b3, w3, b3, (1, 3, 5), y=x, k=3)
    with pytest.raises(TtsAmdError, match='unsupported geometry'):
        bfo.resblock_chain(x, w3, b3, w3, b3, (1, 3, 9), k=3)                    # dilation beyond the built window
