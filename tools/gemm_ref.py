"""Vendor-library calibration: fp32 torch.mm (rocBLAS / hipBLASLt) on the GEMMs that are arithmetically equal to
the HiFi-GAN ResBlock convs (M = C_out, K = C_in * taps, N = batch * positions) — no im2col, operands already in
GEMM layout.  Prints TFLOP/s per shape next to the conv kernel's figure from tools/conv_bench (PROD=1)."""
import time

import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda:0')
shapes = [(256, 3, 32 * 3584), (256, 7, 32 * 3584), (256, 11, 32 * 3584), (128, 3, 32 * 28672), (128, 7, 32 * 28672),
          (128, 11, 32 * 28672), (64, 3, 32 * 57344), (64, 7, 32 * 57344), (64, 11, 32 * 57344), (32, 3, 32 * 114688),
          (32, 7, 32 * 114688), (32, 11, 32 * 114688)]
for C, k, N in shapes:
    N = min(N, 1 << 20)                       # keep B within memory: K * N * 4 bytes
    a = torch.randn(C, C * k, device=dev)
    b = torch.randn(C * k, N, device=dev)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        c = a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'C={C:3d} k={k:2d}: M={C} K={C * k} N={N}: {dt * 1e3:.3f} ms  {2.0 * C * C * k * N / dt / 1e12:.1f} TFLOP/s')
