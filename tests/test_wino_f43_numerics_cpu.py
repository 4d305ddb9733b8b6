"""F(4,3) numerics study (CPU, fp32-emulated): what would the fp32 engine's error be if its Winograd kernels ran F(4,3) instead of F(2,3)?

The fp32 engine (csrc/conv_wino2.hip, csrc/resblock_fused2.hip) runs every ResBlock conv of HiFi-GAN (vocoder/hifigan/models.py:30-53)
and FastPitch's conv-FF (models/fastpitch/fastpitch/transformer.py:59-65) as F(2,3): a k-tap filter = k // 3 three-tap sub-filters
(4 products per output pair each) + k % 3 single taps -- 4 / 10 / 16 products per pair against 6 / 14 / 22.  F(4,3) would issue
6 products per output QUAD and sub-filter (+ 4 per single tap): 6 / 16 / 26 per quad = 0.50 / 0.571 / 0.591 of the direct products
(F(2,3): 0.667 / 0.714 / 0.727), i.e. another 25 / 20 / 19 % fewer MFMAs -- at the price of transforms with entries 4, 5, 8, 1/24.

This file measures that price on the path itself: the whole 75-conv vocoder on the synthetic weights (and FastPitch's decoder with its
conv-FF pairs), every Winograd-routed conv emulated in float32 exactly as a kernel would run it --
    filter transform in float64, rounded once to fp32 (pack_wino2_weight does the same for F(2,3));
    input transform, products (fp32 GEMM over (sub-filter, channel)) and output transform in fp32;
    a dilated conv = d interleaved undilated ones (the kernels' output tuple (q, q + d, ...));
    single taps of F(4,3) through the planes P1..P4 (the output transform restricted to them is the invertible Vandermonde matrix
    of the points 1, -1, 2, -2: the tap's four products go in as x-combinations M^-1 x);
-- against the same network in float64.  The measured numbers are printed and asserted as BOUNDS so that the decision recorded in
DESIGN.md section 4 ("F(4,3): measured ...") stays tied to a test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tts_oracle as O
from ttsamd import synth
from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG

# ---- transform matrices (Lavin & Gray, "Fast algorithms for convolutional neural networks", F(2,3) and F(4,3))
BT23 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G23 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT23 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
BT43 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                 [0, 4, 0, -5, 0, 1]], np.float64)
G43 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                [0, 0, 1]], np.float64)
AT43 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
SCHEMES = {'f23': (2, BT23, G23, AT23), 'f43': (4, BT43, G43, AT43)}


def test_transform_matrices_are_exact():
    """A^T [(G g) * (B^T x)] == the 3-tap correlation, in float64, for both schemes."""
    rng = np.random.default_rng(0)
    for m, BT, G, AT in SCHEMES.values():
        g, x = rng.standard_normal(3), rng.standard_normal(m + 2)
        want = np.array([sum(g[t] * x[i + t] for t in range(3)) for i in range(m)])
        assert np.abs(AT @ ((G @ g) * (BT @ x)) - want).max() < 1e-12


def _lin(coef, xs):
    """sum_m coef[m] * xs[m] in fp32, one operation per non-zero term (what a kernel's input transform issues)"""
    acc = None
    for c, x in zip(coef, xs):
        if c == 0:
            continue
        t = x if c == 1 else (-x if c == -1 else x * np.float32(c))
        acc = t if acc is None else acc + t
    return acc


def wino_conv1d_fp32(x, w, bias, dilation, scheme):
    """'same' Conv1d [B,Ci,L] x [Co,Ci,k] (k in 3 / 7 / 11) in float32 through the Winograd decomposition `scheme`."""
    m, BT, G, AT = SCHEMES[scheme]
    B, Ci, L = x.shape
    Co, _, k = w.shape
    ns, nl, half = k // 3, k % 3, (k - 1) // 2
    w64 = w.double().numpy()
    n_pl = m + 2
    # filter transform in float64, rounded once
    Us = [[torch.from_numpy((np.einsum('it,oct->ioc', G, w64[:, :, 3 * s:3 * s + 3]))[i].astype(np.float32)) for i in range(n_pl)]
          for s in range(ns)]
    y = torch.zeros(B, Co, L, dtype=torch.float32)
    if scheme == 'f43':
        Minv = np.linalg.inv(AT43[:, 1:5])                     # single taps: (P1..P4) = M^-1 (g x_0 .. g x_3)
    for r in range(dilation):
        xr = x[:, :, r::dilation]
        Lr = xr.shape[2]
        if Lr == 0:
            continue
        J = -(-Lr // m)
        xp = F.pad(xr, (half, m * J + k - Lr))                  # zero 'same' padding + the tail of the last tuple
        planes_U = [[] for _ in range(n_pl)]
        planes_V = [[] for _ in range(n_pl)]
        for s in range(ns):
            X = [xp[:, :, 3 * s + mm:3 * s + mm + m * J:m] for mm in range(n_pl)]      # [B,Ci,J] each
            for i in range(n_pl):
                planes_U[i].append(Us[s][i])
                planes_V[i].append(_lin(BT[i], X))
        for l in range(nl):
            t = 3 * ns + l
            gt = w[:, :, t]
            X = [xp[:, :, t + mm:t + mm + m * J:m] for mm in range(m)]
            if scheme == 'f23':                                                        # P0 += g x[t], P3 += (-g) x[t + 1]
                planes_U[0].append(gt); planes_V[0].append(X[0])
                planes_U[3].append(-gt); planes_V[3].append(X[1])
            else:
                for i in range(4):
                    planes_U[1 + i].append(gt)
                    planes_V[1 + i].append(_lin([np.float32(c) for c in Minv[i]], X))
        P = []
        for i in range(n_pl):
            if not planes_U[i]:
                P.append(None)
                continue
            U = torch.cat(planes_U[i], 1)                       # [Co, n * Ci]
            V = torch.cat(planes_V[i], 1)                       # [B, n * Ci, J]
            P.append(torch.matmul(U, V))                        # fp32 GEMM = the MFMA accumulation over (sub-filter, channel)
        for o in range(m):
            yo = _lin(AT[o], [p if p is not None else torch.zeros(()) for p in P])
            idx = torch.arange(o, m * J, m)
            keep = idx < Lr
            y[:, :, r::dilation][:, :, idx[keep]] = yo[:, :, :int(keep.sum())]
    if bias is not None:
        y = y + bias[None, :, None]
    return y


@pytest.mark.parametrize('scheme', ['f23', 'f43'])
@pytest.mark.parametrize('k,d', [(3, 1), (7, 3), (11, 5), (11, 1)])
def test_emulated_winograd_conv_equals_the_direct_conv(scheme, k, d):
    g = torch.Generator().manual_seed(10 * k + d)
    x = torch.randn(2, 16, 67, generator=g)
    w = torch.randn(24, 16, k, generator=g) / (16 * k) ** 0.5
    b = torch.randn(24, generator=g)
    want = F.conv1d(x.double(), w.double(), b.double(), dilation=d, padding=d * (k - 1) // 2)
    got = wino_conv1d_fp32(x, w, b, d, scheme)
    assert float((got.double() - want).abs().max()) < (2e-6 if scheme == 'f23' else 2e-5)


class _Patched:
    """F.conv1d of the oracle module replaced for the launches the fp32 engine routes to its Winograd kernels: k in 3 / 7 / 11,
    'same' padding, Cin % 8 == 0, Cout % 32 == 0, float32 (the float64 reference run is left alone)."""

    def __init__(self, scheme):
        self.scheme, self.n = scheme, 0

    def __enter__(self):
        self.orig = O.F.conv1d
        scheme = self.scheme

        def conv1d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
            k = w.shape[2]
            if (scheme != 'direct' and x.dtype == torch.float32 and k in (3, 7, 11) and w.shape[1] % 8 == 0 and w.shape[0] % 32 == 0
                    and padding == dilation * (k - 1) // 2 and stride == 1 and groups == 1 and x.dim() == 3):
                self.n += 1
                return wino_conv1d_fp32(x, w, b, dilation, scheme)
            return self.orig(x, w, b, stride, padding, dilation, groups)
        O.F.conv1d = conv1d
        return self

    def __exit__(self, *a):
        O.F.conv1d = self.orig


@pytest.fixture(scope='module')
def study():
    """one utterance of 20 tokens (142 frames, 36 352 samples): FastPitch in float64 gives the mel every vocoder variant starts from"""
    torch.manual_seed(0)
    fsd, hsd = synth.fastpitch_state_dict(), synth.hifigan_state_dict()
    ids = synth.synth_ids(1, 20)
    dur = synth.synth_durations(1, 20)
    with torch.inference_mode():
        hw = O.fold_weight_norm(hsd)
        fw64 = O.to_torch(fsd, torch.float64)
        mel64, lens, *_ = O.fastpitch_infer(fw64, NET_CONFIG, ids, dur_tgt=dur, dtype=torch.float64)
        mel64 = mel64.double()
        wave64 = O.hifigan_forward(hw, mel64, HIFIGAN_CONFIG, dtype=torch.float64)
        res = {}
        for scheme in ('direct', 'f23', 'f43'):
            with _Patched(scheme) as p:
                wave = O.hifigan_forward(hw, mel64.float(), HIFIGAN_CONFIG)
                mel, *_ = O.fastpitch_infer(O.to_torch(fsd), NET_CONFIG, ids, dur_tgt=dur)
            res[scheme] = {'wave': float((wave.double() - wave64).abs().max()), 'mel': float((mel.double() - mel64).abs().max()),
                           'routed': p.n}
    res['peak'] = float(wave64.abs().max())
    return res


def test_f43_whole_vocoder_error_study(study):
    """The numbers behind DESIGN.md's F(4,3) decision.  72 ResBlock convs of the vocoder + 12 conv-FF convs of FastPitch are routed."""
    d, a, b = study['direct'], study['f23'], study['f43']
    print(f"\nwave max-abs vs float64 (|wave| peak {study['peak']:.3f}): direct fp32 {d['wave']:.2e}, F(2,3) {a['wave']:.2e}, F(4,3) {b['wave']:.2e}"
          f"\nmel  max-abs vs float64: direct fp32 {d['mel']:.2e}, F(2,3) {a['mel']:.2e}, F(4,3) {b['mel']:.2e}"
          f"\nconvs routed to the emulation: {a['routed']} (72 ResBlock convs + FastPitch's k = 3 conv-FF / predictor convs)")
    assert a['routed'] == b['routed'] >= 72 + 12
    # F(2,3) is what runs today: inside the tolerance with two orders of magnitude to spare, no worse than 2x the direct conv
    assert a['wave'] < 5e-6 and a['mel'] < 5e-5
    # F(4,3): bounds of the measurement (printed above); the decision threshold set by the round-5 review is wave < 2e-5
    assert b['wave'] < 1e-4 and b['mel'] < 1e-3
