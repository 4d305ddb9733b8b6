"""Per-layer timing of the bf16 octet engine at the bench workload's sizes (B = 32 utterances x 448 frames):
algorithmic TFLOP/s against the 2.5 PFLOP/s bf16 MFMA peak and algorithmic GB/s against 8 TB/s HBM.
  python tools/bfo_bench.py [--iters 20] [--json out.json]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tts-arabic-pytorch_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--frames', type=int, default=448)
    ap.add_argument('--json', default=None)
    ap.add_argument('--only', default='')
    ap.add_argument('--x3', action='store_true', help='the split-bf16 mode (ttsamd_bfo3_*): three MFMAs per product, 4-byte activations')
    args = ap.parse_args()
    from ttsamd import bfo as _bfo
    dev = torch.device('cuda:0')
    X3 = args.x3
    EB = 4.0 if X3 else 2.0                    # bytes per activation element
    EW = 16 if X3 else 8                       # int16 words per (octet, position)

    class bfo:                                 # the layer entries of the chosen mode under the plain names
        pack = staticmethod(_bfo.pack3 if X3 else _bfo.pack)
        pack_weight = staticmethod(_bfo.pack_weight3 if X3 else _bfo.pack_weight)
        conv1d = staticmethod(_bfo.conv1d3 if X3 else _bfo.conv1d)
        resblock_pair = staticmethod(_bfo.resblock_pair3 if X3 else _bfo.resblock_pair)
        conv_post = staticmethod(_bfo.conv_post3 if X3 else _bfo.conv_post)
        resblock_chain = staticmethod(_bfo.resblock_chain)
    B, T = args.batch, args.frames
    g = torch.Generator().manual_seed(0)
    rows = []

    def timeit(fn, flops, byts, name):
        if args.only and args.only not in name:
            return
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        # warm the clock (cold-clock artefacts)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_end = 0.15
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < t_end:
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.iters * 1e3
        mf = 3.0 if X3 else 1.0                # MFMA FLOPs per algorithmic FLOP
        row = {'layer': name, 'us': us, 'tflops': flops / us / 1e6, 'mfma_frac': mf * flops / us / 1e6 / 2500.0,
               'gbs': byts / us / 1e3, 'hbm_frac': byts / us / 1e3 / 8000.0}
        rows.append(row)
        print(f"{name:34s} {us:9.1f} us  {row['tflops']:7.1f} TF ({row['mfma_frac']:.3f})  {row['gbs']:7.0f} GB/s ({row['hbm_frac']:.3f})",
              flush=True)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)

    for C, mul in ((128, 64), (64, 128), (32, 256)):
        L = T * mul
        x = bfo.pack(rnd(B, C, L), 0.1)
        y = torch.zeros_like(x)
        for k, d in ((3, 1), (3, 5), (7, 3), (11, 5)):
            w1 = bfo.pack_weight(torch.randn(C, C, k, generator=g) / np.sqrt(C * k), device=dev)
            w2 = bfo.pack_weight(torch.randn(C, C, k, generator=g) / np.sqrt(C * k), device=dev)
            b1, b2 = rnd(C), rnd(C)
            timeit(lambda: bfo.resblock_pair(x, w1, b1, w2, b2, k, d, y=y), 2 * 2.0 * C * C * k * L * B, 2 * EB * C * L * B,
                   f'pair C={C} k={k} d={d} L={L}')
        # the whole k = 3 ResBlock (three pairs, dilations 1 / 3 / 5) as one launch: FLOPs and bytes of the ALGORITHM (three pairs' FLOPs,
        # one read + one write), so the line compares with three 'pair k=3' lines
        ws = [[bfo.pack_weight(torch.randn(C, C, 3, generator=g) / np.sqrt(C * 3), device=dev) for _ in range(3)] for _ in range(2)]
        bs = [[rnd(C) for _ in range(3)] for _ in range(2)]
        if not X3:
            timeit(lambda: bfo.resblock_chain(x, ws[0], bs[0], ws[1], bs[1], (1, 3, 5), y=y), 3 * 2 * 2.0 * C * C * 3 * L * B, 2 * 2.0 * C * L * B,
                   f'chain C={C} k=3 (3 pairs) L={L}')
        del x, y
    C, L = 256, T * 8
    x = bfo.pack(rnd(B, C, L), 0.1)
    y, r = torch.zeros_like(x), bfo.pack(rnd(B, C, L), 0.1)
    for k, d in ((3, 1), (7, 3), (11, 5)):
        w = bfo.pack_weight(torch.randn(C, C, k, generator=g) / np.sqrt(C * k), device=dev)
        b = rnd(C)
        timeit(lambda: bfo.conv1d(x, w, b, C, k, dilation=d, out_slope=0.1, y=y), 2.0 * C * C * k * L * B, 2 * EB * C * L * B,
               f'conv C=256 k={k} d={d} (c1)')
        timeit(lambda: bfo.conv1d(x, w, b, C, k, dilation=1, res=r, res_slope=0.1, out_slope=0.1, y=y), 2.0 * C * C * k * L * B,
               3 * EB * C * L * B, f'conv C=256 k={k} (c2 + res)')
    del x, y, r
    for cin, cout, u, mul in ((512, 256, 8, 1), (256, 128, 8, 8), (128, 64, 2, 64), (64, 32, 2, 128)):
        L = T * mul
        x = bfo.pack(rnd(B, cin, L), 0.1)
        y = torch.zeros(B, cout // 8, L * u, EW, dtype=torch.int16, device=dev)
        w = bfo.pack_weight(torch.randn(cin, cout, 2 * u, generator=g) / np.sqrt(cin * 2), up=u, device=dev)
        b = rnd(cout)
        timeit(lambda: bfo.conv1d(x, w, b, cout, 2 * u, up=u, out_slope=0.1, y=y), 2.0 * cin * cout * 2 * u * L * B,
               EB * (cin * L + cout * L * u) * B, f'convt {cin}->{cout} u={u} L={L}')
        del x, y
    x = bfo.pack(rnd(B, 80, T), 1.0)
    w = bfo.pack_weight(torch.randn(512, 80, 7, generator=g) / np.sqrt(80 * 7), device=dev)
    b = rnd(512)
    y = torch.zeros(B, 64, T, EW, dtype=torch.int16, device=dev)
    timeit(lambda: bfo.conv1d(x, w, b, 512, 7, out_slope=0.1, y=y), 2.0 * 512 * 80 * 7 * T * B, EB * (80 + 512) * T * B, 'conv_pre 80->512 k=7')
    L = T * 256
    x = bfo.pack(rnd(B, 32, L), 0.01)
    w, b = rnd(32, 7), rnd(1)
    timeit(lambda: bfo.conv_post(x, w, b), 2.0 * 32 * 7 * L * B, (32 * EB + 4.0) * L * B, 'conv_post 32->1 k=7')
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
