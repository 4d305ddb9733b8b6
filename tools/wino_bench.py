"""Kernel-level A/B of the fp32 conv kernels on HiFi-GAN's dilation-1 shapes: direct (TTSAMD_WINO=0), Winograd F(2,3) (k = 3) and the
decomposition kernels (F(2,3): TTSAMD_WINO2 mask, F(4,3): TTSAMD_WINO4 mask; WINO_BENCH_ONLY=wino2,wino4 picks the variants).  Run under rocprofv3 --kernel-trace --stats for per-kernel durations; prints wall-clock per call.
gpurun -- 'python3 tools/wino_bench.py'"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
from ttsamd import lib  # noqa: E402
from ttsamd.engine import _ptr, _stream  # noqa: E402

dev = torch.device('cuda:0')
L = lib.load()
B, T = 32, 640
only = [int(a) for a in sys.argv[1:]]
for C, mul in ((256, 8), (128, 64), (64, 128)):
    if only and C not in only:
        continue
    for k in (3, 7, 11):
        n = T * mul
        x = torch.randn(B, C, n, device=dev)
        res = torch.randn(B, C, n, device=dev)
        w = torch.randn(C, C, k, device=dev) / (C * k) ** 0.5
        b = torch.zeros(C, device=dev)
        y = torch.empty_like(x)
        packed = torch.empty(L.ttsamd_conv1d_packed_floats(C, C, k), dtype=torch.float32, device=dev)
        for name, env in (('direct', {'TTSAMD_WINO': '0'}), ('wino', {'TTSAMD_WINO': '1', 'TTSAMD_WINO2': '0', 'TTSAMD_WINO4': '0'}),
                          ('wino2', {'TTSAMD_WINO': '1', 'TTSAMD_WINO2': '31', 'TTSAMD_WINO4': '0'}),
                          ('wino4', {'TTSAMD_WINO': '1', 'TTSAMD_WINO2': '31', 'TTSAMD_WINO4': '15'})):
            if name == 'wino' and (k != 3 or C == 64):
                continue
            if os.environ.get('WINO_BENCH_ONLY') and name not in os.environ['WINO_BENCH_ONLY'].split(','):
                continue
            for k_, v_ in env.items():                   # routing options live in the library's table (ttsamd_set_option), not in the environment
                lib.set_option(k_, v_)
            def call():
                lib.check(L.ttsamd_conv1d_ex(_ptr(x), _ptr(w), _ptr(b), _ptr(res), None, B, C, C, k, 1, n, 0.1, 0, 0, 1.0, _ptr(y),
                                             _ptr(packed), _stream()), 'conv1d')
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                call()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 100
            fl = 2.0 * C * C * k * n * B
            print(f'C={C} k={k} L={n} {name:7s} {ms:7.3f} ms/call (incl. weight packing)  {fl / ms / 1e9:7.1f} TFLOP/s un-reduced')
