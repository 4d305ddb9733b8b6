"""CPU check of the arithmetic behind csrc/conv_wino2.hip and the Winograd phases of csrc/resblock_fused2.hip (no GPU, no library):
a k-tap 'same' conv at dilation d equals, output pair (q, q + d) by output pair, the F(2,3) decomposition the kernels run --

    k // 3 three-tap sub-filters s, each four products  P_i += U_{s,i} V_{s,i}
        U = g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2                      (pack_wino2_weight)
        V = x0 - x2, x1 + x2, x2 - x1, x1 - x3  on the positions 3 s + {0..3} of the pair's d-decimated window
    k % 3 single taps t:  P_0 += g_t x[t],  P_3 += (-g_t) x[t + 1]
    y[q] = P_0 + P_1 + P_2,   y[q + d] = P_1 - P_2 - P_3

-- in float64, against numpy's direct correlation (the reference's op: torch Conv1d, vocoder/hifigan/models.py:30-44).  The product
counts per pair (4 / 10 / 16 against 6 / 14 / 22) are what bench.py's `frac_issued` is computed from."""
import numpy as np
import pytest


def direct(x, g, d):
    k = len(g)
    pad = d * (k - 1) // 2
    xp = np.concatenate([np.zeros(pad), x, np.zeros(pad)])
    return np.array([sum(g[t] * xp[q + t * d] for t in range(k)) for q in range(len(x))])


def decomposed(x, g, d):
    k = len(g)
    ns, nl = k // 3, k % 3
    pad = d * (k - 1) // 2
    n = len(x)
    xp = np.concatenate([np.zeros(pad), x, np.zeros(pad + 2 * d)])
    y = np.zeros(n + 2 * d)
    products = 0
    # pairs (q, q + d): q runs over groups of 2 d, the first d positions of each
    for q0 in range(0, n, 2 * d):
        for r in range(d):
            q = q0 + r
            w = [xp[q + m * d] for m in range(k + 1)]          # the pair's d-decimated window: K + 1 positions
            P = [0.0, 0.0, 0.0, 0.0]
            for s in range(ns):
                g0, g1, g2 = g[3 * s:3 * s + 3]
                U = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)
                x0, x1, x2, x3 = w[3 * s:3 * s + 4]
                V = (x0 - x2, x1 + x2, x2 - x1, x1 - x3)
                for i in range(4):
                    P[i] += U[i] * V[i]
                    products += 1
            for l in range(nl):
                t = 3 * ns + l
                P[0] += g[t] * w[t]
                P[3] += -g[t] * w[t + 1]
                products += 2
            y[q] = P[0] + P[1] + P[2]
            y[q + d] = P[1] - P[2] - P[3]
    return y[:n], products


@pytest.mark.parametrize('k', [3, 7, 11])
@pytest.mark.parametrize('d', [1, 3, 5])
def test_winograd_decomposition_equals_the_direct_conv(k, d):
    rng = np.random.default_rng(100 * k + d)
    n = 2 * d * 9                                       # whole groups of 2 d (the kernels' tiles: 120 / 252 / 250 outputs)
    x, g = rng.standard_normal(n), rng.standard_normal(k)
    want = direct(x, g, d)
    got, products = decomposed(x, g, d)
    assert np.abs(got - want).max() < 1e-12
    ng = 4 * (k // 3) + 2 * (k % 3)
    assert products == ng * (n // 2)                    # 4 / 10 / 16 products per output pair ...
    assert (ng, 2 * k) in ((4, 6), (10, 14), (16, 22))  # ... against 6 / 14 / 22 of the direct conv


def test_residual_and_running_sum_enter_through_the_planes():
    """The kernels preload res[q] -> P_0 and -res[q + d] -> P_3 (and add the running ResBlock sum the same way): the output
    transform then carries them to y[q], y[q + d] unchanged."""
    rng = np.random.default_rng(7)
    p1, p2, r0, r1 = rng.standard_normal(4)
    P = [r0, p1, p2, -r1]
    assert np.isclose(P[0] + P[1] + P[2], r0 + p1 + p2)
    assert np.isclose(P[1] - P[2] - P[3], p1 - p2 + r1)
