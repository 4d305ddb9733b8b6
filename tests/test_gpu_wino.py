"""GPU parity of the Winograd F(2,3) conv kernels of the exact-fp32 engine (csrc/conv_wino.hip: k = 3; csrc/conv_wino2.hip: k = 3 / 7 / 11
as sums of three-tap sub-filters and single taps) through the C ABI (pytest -m gpu).

Checker: torch conv1d in float64 on the host (the reference's op for FastPitch's conv-FF convs, transformer.py:59-65, and HiFi-GAN's
k = 3 ResBlock convs, vocoder/hifigan/models.py:30-44).  Stated tolerance: |diff| <= 2e-5 max-abs on O(1) outputs with K = 3 * Cin up
to 4608 products each -- the same bound the direct fp32 kernel is held to in tests/test_gpu_parity.py (its measured error on these
shapes is 2-4e-6; Winograd adds one rounding per transformed operand).  The direct kernel (TTSAMD_WINO=0) runs beside it: the two
must agree to the same bound and -- being different arithmetic -- differ in their last bits, which proves the routing."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from ttsamd import lib
    assert lib.load().ttsamd_device_ok() == 1
    return torch.device('cuda:0')


@pytest.mark.parametrize('cin,cout,L,B,slope,relu', [
    (384, 1536, 496, 6, 1.0, True),          # FastPitch conv-FF, first conv (+ ReLU)
    (1536, 384, 500, 24, 1.0, False),        # ... second conv: 96 chunks
    (256, 256, 1032, 16, 0.1, False),        # HiFi-GAN stage 1 (leaky-relu on load); odd utterance lengths below
    (128, 128, 700, 48, 0.1, False),
    (16, 128, 300, 100, 1.0, False),         # two chunks: prologue = whole ring
])
def test_wino_conv_matches_float64_and_the_direct_kernel(dev, monkeypatch, cin, cout, L, B, slope, relu, ttsopt):
    from ttsamd.engine import conv1d
    g = torch.Generator().manual_seed(cin + cout + L)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, 3, generator=g) / np.sqrt(cin * 3)
    b = torch.randn(cout, generator=g) * 0.3
    # ragged: full length, odd lengths (a pair cut by the utterance end), one ending right after a tile edge, a 1-frame utterance
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0], lens[1], lens[2], lens[3] = L, L - 1 if L % 2 == 0 else L - 2, 129, 1
    lens_d = lens.to(dev)
    ttsopt.set('TTSAMD_WINO4', '0')           # these cases pin the F(2,3) kernels
    ttsopt.set('TTSAMD_WINO2', '6')           # k = 3 on conv_wino.hip (the decomposition kernel's k = 3 is tested below)
    ttsopt.set('TTSAMD_WINO', '1')
    y_w = conv1d(x.to(dev), w.to(dev), b.to(dev), lens=lens_d, in_slope=slope, relu_out=relu).cpu()
    ttsopt.set('TTSAMD_WINO', '0')
    y_d = conv1d(x.to(dev), w.to(dev), b.to(dev), lens=lens_d, in_slope=slope, relu_out=relu).cpu()
    worst_w = worst_d = 0.0
    for i in range(B):
        n = int(lens[i])
        xi = F.leaky_relu(x[i:i + 1, :, :n].double(), slope)
        ref = F.conv1d(xi, w.double(), b.double(), padding=1)[0]
        if relu:
            ref = torch.relu(ref)
        worst_w = max(worst_w, float((y_w[i, :, :n].double() - ref).abs().max()))
        worst_d = max(worst_d, float((y_d[i, :, :n].double() - ref).abs().max()))
        assert float(y_w[i, :, n:].abs().max() if n < L else 0.0) == 0.0, 'positions past the utterance must stay untouched'
    print(f'cin={cin} cout={cout}: Winograd max-abs {worst_w:.2e}, direct {worst_d:.2e}')
    assert worst_w < 2e-5 and worst_d < 2e-5
    assert not torch.equal(y_w, y_d), 'TTSAMD_WINO=1 must route these shapes to the Winograd kernel'


def test_wino_is_deterministic_and_skips_small_problems(dev, monkeypatch, ttsopt):
    from ttsamd.engine import conv1d
    g = torch.Generator().manual_seed(1)
    x, w = torch.randn(8, 384, 496, generator=g), torch.randn(1536, 384, 3, generator=g) / 34.0
    ttsopt.set('TTSAMD_WINO4', '0')           # these cases pin the F(2,3) kernels
    ttsopt.set('TTSAMD_WINO2', '6')
    ttsopt.set('TTSAMD_WINO', '1')
    a, b2 = conv1d(x.to(dev), w.to(dev)), conv1d(x.to(dev), w.to(dev))
    assert torch.equal(a, b2)
    # short sequences (the 64-token encoder) and grids under one block per CU keep the direct kernel: same bits either way
    xs = torch.randn(4, 384, 64, generator=g)
    ttsopt.set('TTSAMD_WINO', '0')
    d = conv1d(xs.to(dev), w.to(dev))
    ttsopt.set('TTSAMD_WINO', '1')
    assert torch.equal(conv1d(xs.to(dev), w.to(dev)), d)
    # rows that are not float4-aligned (L % 4 != 0), a slope outside [0, 1] (the kernels activate with max(x, slope x)) and a kernel size
    # without a decomposition keep the direct kernel too: same bits with the switch on and off
    xo = torch.randn(8, 384, 1031, generator=g)
    w5 = torch.randn(1536, 384, 5, generator=g) / 44.0
    for args, kw in (((xo, w), {}), ((x, w), {'in_slope': 1.5}), ((x, w5), {})):
        outs = []
        for flag in ('0', '1'):
            ttsopt.set('TTSAMD_WINO', flag)
            outs.append(conv1d(args[0].to(dev), args[1].to(dev), **kw))
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('cin,cout,L,B,mode', [(1536, 384, 496, 24, 0), (256, 256, 1032, 16, 1), (256, 256, 520, 40, 2)])
def test_wino_residual_preload_and_accumulate_modes(dev, monkeypatch, cin, cout, L, B, mode, ttsopt):
    """The epilogue that FastPitch's second conv-FF conv (conv + residual, transformer.py:83-86) and the c2 convs of HiFi-GAN's
    ResBlocks (x + conv, summed over the three branches and divided, vocoder/hifigan/models.py:46-53,116-122) use: the residual -- and
    in the accumulate modes the previous y -- enters through the accumulators (res[2j] -> M0, -res[2j + 1] -> M3).  Ragged, odd
    lengths; against float64 and against the direct kernel's own preload epilogue."""
    from ttsamd.engine import conv1d
    g = torch.Generator().manual_seed(cin + L + mode)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, 3, generator=g) / np.sqrt(cin * 3)
    b = torch.randn(cout, generator=g) * 0.3
    res = torch.randn(B, cout, L, generator=g)
    y0 = torch.randn(B, cout, L, generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0], lens[1], lens[2] = L, L - 1, 131
    outs = {}
    ttsopt.set('TTSAMD_WINO4', '0')           # these cases pin the F(2,3) kernels
    ttsopt.set('TTSAMD_WINO2', '6')
    for flag in ('1', '0'):
        ttsopt.set('TTSAMD_WINO', flag)
        y = y0.clone().to(dev)
        conv1d(x.to(dev), w.to(dev), b.to(dev), lens=lens.to(dev), in_slope=0.1, res=res.to(dev), mode=mode, div=3.0, y=y)
        outs[flag] = y.cpu()
    worst = {'1': 0.0, '0': 0.0}
    for i in range(B):
        n = int(lens[i])
        v = F.conv1d(F.leaky_relu(x[i:i + 1, :, :n].double(), 0.1), w.double(), b.double(), padding=1)[0] + res[i, :, :n].double()
        ref = v if mode == 0 else (y0[i, :, :n].double() + v if mode == 1 else (y0[i, :, :n].double() + v) / 3.0)
        for flag in ('1', '0'):
            worst[flag] = max(worst[flag], float((outs[flag][i, :, :n].double() - ref).abs().max()))
            assert torch.equal(outs[flag][i, :, n:], y0[i, :, n:]), 'positions past the utterance must stay untouched'
    print(f'cin={cin} mode={mode}: Winograd max-abs {worst["1"]:.2e}, direct {worst["0"]:.2e}')
    assert worst["1"] < 5e-5 and worst["0"] < 5e-5                      # K = 4608 products per output at cin = 1536
    assert not torch.equal(outs['1'], outs['0'])


@pytest.mark.parametrize('k,d,cin,cout,L,B,mode', [
    (3, 1, 256, 256, 1032, 16, None), (3, 1, 1536, 384, 496, 24, 0),
    (7, 1, 256, 256, 1032, 16, None), (7, 1, 256, 256, 1028, 16, 1), (7, 1, 128, 128, 2052, 16, 2),
    (11, 1, 256, 256, 1032, 16, None), (11, 1, 256, 256, 1028, 16, 2), (11, 1, 128, 128, 2052, 16, 1), (7, 1, 32, 128, 700, 48, None),
    (3, 3, 256, 256, 1032, 16, None), (3, 5, 128, 128, 2052, 16, None), (7, 3, 256, 256, 1028, 16, None), (7, 5, 128, 128, 2052, 16, 1),
    (11, 3, 128, 128, 2052, 16, None), (11, 5, 256, 256, 1032, 16, None), (11, 5, 256, 256, 1028, 16, 2),
    # Cout = 64: the 64-row x 128-pair variant (k = 11 in two phases of 8 groups); 252- / 240-output tiles at dilation 3 / 5
    (3, 1, 64, 64, 4100, 12, 1), (3, 3, 64, 64, 4100, 12, None), (7, 1, 64, 64, 4100, 12, 2), (7, 5, 64, 64, 4100, 12, None),
    (11, 1, 64, 64, 4100, 12, None), (11, 3, 64, 64, 4100, 12, 1), (11, 5, 64, 64, 4100, 12, 2), (11, 1, 128, 64, 4100, 12, 0),
])
@pytest.mark.parametrize('scheme', ['f23', 'f43'])
def test_wino_decomposition_k3_k7_k11(dev, monkeypatch, k, d, cin, cout, L, B, mode, scheme, ttsopt):
    """scheme f43 = conv_wino4.hip: the same filters as F(4,3) sub-filters -- 6 / 16 / 23 products per output QUAD (q, q + d, q + 2 d, q + 3 d),
    64 rows x 64 quads per block, k = 11 in two phases of 12 / 11 groups, the single tap of k = 7 through two extra planes; tiles of 256 /
    252 / 240 outputs at dilation 1 / 3 / 5.  Same float64 checker, same ragged odd lengths and epilogues; F(2,3) and F(4,3) must differ
    in bits (the routing took effect) and both stay within the bound.
    scheme f23 = conv_wino2.hip: a k-tap filter as k // 3 three-tap F(2,3) sub-filters + k % 3 single taps accumulating into the same four planes
    (HiFi-GAN's ResBlock convs, vocoder/hifigan/models.py:30-44); at dilation d = 3 / 5 the output pair is (q, q + d) and a tile holds
    120 outputs.  Float64 checker, ragged odd lengths, with and without the residual / accumulate epilogues; the direct kernel beside it."""
    from ttsamd.engine import conv1d
    g = torch.Generator().manual_seed(k * 1000 + cin + L + d)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / np.sqrt(cin * k)
    b = torch.randn(cout, generator=g) * 0.3
    res = torch.randn(B, cout, L, generator=g) if mode is not None else None
    y0 = torch.randn(B, cout, L, generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0], lens[1], lens[2], lens[3] = L, L - 1, 131, 1
    outs = {}
    ttsopt.set('TTSAMD_WINO2', '31')
    ttsopt.set('TTSAMD_WINO4', '15' if scheme == 'f43' else '0')
    for flag in ('1', '0'):
        ttsopt.set('TTSAMD_WINO', flag)
        y = y0.clone().to(dev)
        conv1d(x.to(dev), w.to(dev), b.to(dev), lens=lens.to(dev), dilation=d, in_slope=0.1, res=None if res is None else res.to(dev),
               mode=mode or 0, div=3.0, y=y)
        outs[flag] = y.cpu()
    worst = {'1': 0.0, '0': 0.0}
    for i in range(B):
        n = int(lens[i])
        v = F.conv1d(F.leaky_relu(x[i:i + 1, :, :n].double(), 0.1), w.double(), b.double(), padding=d * (k - 1) // 2, dilation=d)[0]
        if res is not None:
            v = v + res[i, :, :n].double()
        ref = v if not mode else (y0[i, :, :n].double() + v if mode == 1 else (y0[i, :, :n].double() + v) / 3.0)
        for flag in ('1', '0'):
            worst[flag] = max(worst[flag], float((outs[flag][i, :, :n].double() - ref).abs().max()))
            assert torch.equal(outs[flag][i, :, n:], y0[i, :, n:]), 'positions past the utterance must stay untouched'
    print(f'{scheme} k={k} d={d} cin={cin} mode={mode}: decomposition max-abs {worst["1"]:.2e}, direct {worst["0"]:.2e}')
    assert worst["1"] < 5e-5 and worst["0"] < 5e-5
    assert not torch.equal(outs['1'], outs['0'])
    if scheme == 'f43':                                   # ... and it is not the F(2,3) kernel that ran
        ttsopt.set('TTSAMD_WINO4', '0')
        ttsopt.set('TTSAMD_WINO', '1')
        y = y0.clone().to(dev)
        conv1d(x.to(dev), w.to(dev), b.to(dev), lens=lens.to(dev), dilation=d, in_slope=0.1, res=None if res is None else res.to(dev),
               mode=mode or 0, div=3.0, y=y)
        assert not torch.equal(outs['1'], y.cpu())


@pytest.mark.parametrize('cin,cout,L,B,act,mode', [
    # (every case is >= 512 blocks of 64 rows x 256 outputs: what wino_route asks of a k = 1 launch)
    (512, 1536, 496, 12, 2, None),       # Vocos pwconv1 + GELU
    (1536, 512, 500, 32, 0, 0),          # pwconv2 + residual
    (384, 192, 1030, 36, 0, None),       # FastPitch's qkv projection: three 64-row blocks
    (64, 384, 1030, 18, 0, 0),           # ... o_net: two 32-channel chunks
    (256, 256, 777, 32, 1, 0), (256, 256, 777, 32, 0, 1), (256, 128, 2052, 32, 0, 2), (512, 512, 300, 32, 3, None),
])
def test_k1_gemm_on_the_wino4_skeleton(dev, cin, cout, L, B, act, mode, ttsopt):
    """k = 1 (conv_wino4.hip, Wino4Geo::WSHARE -- TTSAMD_WINO4 bit 4): a pointwise conv as the four single-tap planes of the F(4,3) kernel,
    one weight fragment per octet for 16 MFMAs, the direct engine's packed weights as they are.  Same products in the same order as
    conv1d_mfma_f32<1> (bit-identical results where that kernel does not split K); float64 checker; GELU / ReLU / tanh, residual, accumulate modes, ragged
    odd lengths (reference ops: vocoder/vocos/modules.py ConvNeXtBlock pwconv1 / pwconv2, models/fastpitch/fastpitch/transformer.py:96-99)."""
    from ttsamd.engine import conv1d
    g = torch.Generator().manual_seed(cin + 7 * cout + L)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, 1, generator=g) / np.sqrt(cin)
    b = torch.randn(cout, generator=g) * 0.3
    res = torch.randn(B, cout, L, generator=g) if mode is not None else None
    y0 = torch.randn(B, cout, L, generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0], lens[1], lens[2], lens[3] = L, L - 1, 131, 1
    outs = {}
    for mask in ('31', '15'):
        ttsopt.set('TTSAMD_WINO4', mask)
        y = y0.clone().to(dev)
        conv1d(x.to(dev), w.to(dev), b.to(dev), lens=lens.to(dev), in_slope=0.1, relu_out=act, res=None if res is None else res.to(dev),
               mode=mode or 0, div=3.0, y=y)
        outs[mask] = y.cpu()
    worst = 0.0
    for i in range(B):
        n = int(lens[i])
        v = F.conv1d(F.leaky_relu(x[i:i + 1, :, :n].double(), 0.1), w.double(), b.double())[0]
        if act == 2:
            v = F.gelu(v)
        if res is not None:
            v = v + res[i, :, :n].double()
        if act == 1:
            v = F.relu(v)
        if act == 3:
            v = torch.tanh(v)
        ref = v if not mode else (y0[i, :, :n].double() + v if mode == 1 else (y0[i, :, :n].double() + v) / 3.0)
        worst = max(worst, float((outs['31'][i, :, :n].double() - ref).abs().max()))
        assert torch.equal(outs['31'][i, :, n:], y0[i, :, n:]), 'positions past the utterance must stay untouched'
    print(f'k=1 cin={cin} cout={cout} act={act} mode={mode}: max-abs {worst:.2e}')
    assert worst < 2e-5
    # same products in the same order: bit for bit the direct kernel's result unless that one splits K (under 320 blocks of its own tiles)
    # or preloads the residual into its accumulators (other summation order of the epilogue terms)
    same = torch.equal(outs['31'], outs['15'])
    print('   bit-identical to conv1d_mfma_f32<1>:', same)
    assert float((outs['31'] - outs['15']).abs().max()) < 2e-5
    if (cin, cout) == (512, 1536):
        assert same
