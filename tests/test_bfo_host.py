"""CPU: host-side pieces of the bf16 octet engine (config 3) -- the weight packers behind ttsamd_bfo_pack_weight (pure host
code, no GPU) against a numpy restatement of the layout DESIGN.md §3 states, and the algorithmic byte counts bench.py's bf16
roofline divides by against the figures quoted in DESIGN.md §4."""
import ctypes as C
import importlib.util
import os

import numpy as np

from conftest import REPO


def _bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7fff + ((u >> 16) & 1)                       # round to nearest even
    return ((u >> 16) & 0xffff).astype(np.uint16)


def _pack(w, up):
    from ttsamd import lib as L
    lib = L.load()
    w = np.ascontiguousarray(w, np.float32)
    cin, cout, k = (w.shape if up > 1 else (w.shape[1], w.shape[0], w.shape[2]))
    n = lib.ttsamd_bfo_weight_elems(cout, cin, k, up)
    out = np.zeros(n, np.uint16)
    L.check(lib.ttsamd_bfo_pack_weight(w.ctypes.data_as(C.c_void_p), cout, cin, k, up, out.ctypes.data_as(C.c_void_p)), 'pack')
    return out


def test_conv_weight_layout():
    """[Cin/16][K][2 (kk)][CoutP][8]: element e of lane (row co, kk) at step (h, tap) = w[co][16 h + 8 kk + e][tap]; Cin is
    zero-padded to a multiple of 16 (conv_pre: 80 channels = 5 groups), Cout to a multiple of 32."""
    rng = np.random.default_rng(0)
    for cout, cin, k in ((64, 32, 3), (40, 80, 7), (128, 24, 1)):
        w = rng.standard_normal((cout, cin, k)).astype(np.float32)
        got = _pack(w, 1)
        cp, nh = (cout + 31) // 32 * 32, (cin + 15) // 16
        assert got.size == nh * k * 2 * cp * 8
        got = got.reshape(nh, k, 2, cp, 8)
        want = np.zeros((nh, k, 2, cp, 8), np.uint16)
        wb = _bf16(w)
        for h in range(nh):
            for kk in range(2):
                for e in range(8):
                    ci = 16 * h + 8 * kk + e
                    if ci < cin:
                        want[h, :, kk, :cout, e] = wb[:, ci, :].T
        assert np.array_equal(got, want)


def test_conv_transpose_weight_layout():
    """ConvTranspose1d(stride u, kernel 2u, padding u/2) as u polyphase 2-tap filters: phase rho, tap t2 -> kernel index
    (rho + u/2) % u + t2 u (vocoder/hifigan/models.py:96-99); layout [u][Cin/16][2][2][CoutP][8]."""
    rng = np.random.default_rng(1)
    for cin, cout, u in ((32, 64, 2), (48, 32, 8)):
        w = rng.standard_normal((cin, cout, 2 * u)).astype(np.float32)
        got = _pack(w, u).reshape(u, cin // 16, 2, 2, (cout + 31) // 32 * 32, 8)
        wb = _bf16(w)
        for rho in range(u):
            ka = (rho + u // 2) % u
            for t2 in range(2):
                for h in range(cin // 16):
                    for kk in range(2):
                        for e in range(8):
                            assert np.array_equal(got[rho, h, t2, kk, :cout, e], wb[16 * h + 8 * kk + e, :, ka + t2 * u])
    # and what the phases mean: y[q u + rho] = sum_ci W[ka] x[q + dl] + W[ka + u] x[q + dl - 1], dl = (rho + u/2) // u
    import torch
    u, cin, cout, n = 8, 16, 32, 5
    x = torch.randn(1, cin, n, dtype=torch.float64)
    wt = torch.randn(cin, cout, 2 * u, dtype=torch.float64)
    ref = torch.nn.functional.conv_transpose1d(x, wt, stride=u, padding=u // 2)[0]
    xp = torch.nn.functional.pad(x[0], (1, 1))
    for rho in range(u):
        ka, dl = (rho + u // 2) % u, (rho + u // 2) // u
        for q in range(n):
            y = wt[:, :, ka].T @ xp[:, q + dl + 1] + wt[:, :, ka + u].T @ xp[:, q + dl]
            assert torch.allclose(y, ref[:, q * u + rho])


def test_bf16_roofline_byte_counts():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(REPO, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from ttsamd.config import HIFIGAN_CONFIG, NET_CONFIG
    by = mod.hifigan_octet_bytes_per_frame(HIFIGAN_CONFIG)
    # by hand: conv_pre 2 (80 + 512); stage i: upsampler in + out, 9 pairs x (2 passes, or 5 at C = 256) x 2 C m bytes -- the k = 3
    # ResBlock at C <= 128 is one chained launch: 7 "pairs" of traffic instead of 9 --, 2 sum re-reads
    want = 2 * (80 + 512)
    for cin, c, m_in, m in ((512, 256, 1, 8), (256, 128, 8, 64), (128, 64, 64, 128), (64, 32, 128, 256)):
        want += 2 * (cin * m_in + c * m) + (9 if c > 128 else 7) * 2 * (5 if c > 128 else 2) * c * m + 2 * 2 * c * m
    assert by == want
    assert abs(by / 1e6 - 1.071) < 0.001                   # 1.071 MB per mel frame (4.478 MB layer-wise in fp32)
    dec, enc = mod.fastpitch_conv_bytes_per_pos(NET_CONFIG)
    d, f = 384, 1536
    layer = 4 * ((d + 192) + (64 + 2 * d)) + 2 * (d + f) + (2 * f + 8 * d)
    assert dec == 6 * layer + 4 * (d + 80)
    assert enc == 6 * layer + 3 * 4 * ((d + 256) + 2 * 256)
