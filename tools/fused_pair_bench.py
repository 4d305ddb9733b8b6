"""Stand-alone timing of the fused ResBlock1 pair kernels of the fp32 engine on the bench workload's shapes
(B = 32 utterances x 64 tokens, forced durations: 14 341 frames; stage lengths = frames x 64 / 128 / 256 for C = 128 / 64 / 32).

  python tools/fused_pair_bench.py [--batch 32] [--reps 20] [--cases 64:3:1,128:3:5] [--variants 1,2,3]

variant 1 = resblock_pair (weights through an LDS ring), 2 / 3 = resblock_pair2 with 256- / 128-column blocks (weights from L2 into a
register queue).  Prints us per launch and algorithmic TFLOP/s (2 convs x 2 C C k per VALID position) against the 157.3 TFLOP/s
fp32 MFMA peak; the clock is warmed for 200 ms first (a cold launch runs at 1.7 GHz)."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tts-arabic-pytorch_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--cases', default='')
    ap.add_argument('--variants', default='1,2,3')
    args = ap.parse_args()
    from ttsamd import synth, lib as L
    from ttsamd.engine import resblock_pair, _ptr, _stream
    import ctypes as C
    dev = torch.device('cuda:0')
    frames = synth.synth_durations(args.batch, 64).sum(1).astype(np.int64)
    mul_of = {128: 64, 64: 128, 32: 256}
    cases = [(c, k, d) for c in (128, 64, 32) for k in (3, 7, 11) for d in (1, 5)]
    if args.cases:
        cases = [tuple(int(v) for v in s.split(':')) for s in args.cases.split(',')]
    variants = [int(v) for v in args.variants.split(',')]
    lib = L.load()
    # warm the clock
    a = torch.randn(4096, 4096, device=dev)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    for _ in range(30):
        a @ a
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(0)
    print(f'batch {args.batch}: {int(frames.sum())} frames; us per launch (TFLOP/s, fraction of 157.3)')
    for (c, k, d) in cases:
        mul = mul_of[c]
        Lx = int(frames.max()) * mul
        x = torch.randn(args.batch, c, Lx, generator=g).to(dev)
        y = torch.zeros_like(x)
        w1 = (torch.randn(c, c, k, generator=g) / np.sqrt(c * k)).to(dev)
        w2 = (torch.randn(c, c, k, generator=g) / np.sqrt(c * k)).to(dev)
        b1, b2 = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        lens = torch.from_numpy(frames).to(dev)
        n_packed = max(int(lib.ttsamd_resblock_pair_packed_floats(c, k, v)) for v in variants + [2])
        packed = torch.empty(n_packed, dtype=torch.float32, device=dev)
        flops = 2 * 2.0 * c * c * k * float(frames.sum()) * mul
        row = f'C={c:3d} k={k:2d} d={d}:'
        pack_us = 0.0
        for v in [-1] + variants:
            def run():
                return lib.ttsamd_resblock_pair(_ptr(x), _ptr(y), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), c, k, d, _ptr(lens), mul, Lx,
                                                args.batch, 0, C.c_float(1.0), C.c_float(0.1), v, _ptr(packed), n_packed, _stream())
            if run() != 0:
                row += f'   v{v}: n/a'
                continue
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            t0.record()
            for _ in range(args.reps):
                run()
            t1.record()
            torch.cuda.synchronize()
            us = t0.elapsed_time(t1) * 1e3 / args.reps
            if v == -1:
                pack_us = us                         # the two weight re-layout launches every call of the entry makes
                continue
            us -= pack_us
            tf = flops / us / 1e6
            row += f'   v{v}: {us:7.1f} us ({tf:5.1f} TF, {tf / 157.3:.2f})'
        print(row, flush=True)
        del x, y


if __name__ == '__main__':
    main()
