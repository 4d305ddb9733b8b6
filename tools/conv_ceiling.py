"""GPU probe: asymptotic rate of the MFMA conv main loop (deep K) vs the production shapes."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tts-arabic-pytorch_amd'))
import torch
from ttsamd.engine import conv1d, set_precision
import sys as _s
set_precision(_s.argv[1] if len(_s.argv) > 1 else 'f32')
dev = torch.device('cuda:0')
for (B, cin, cout, k, dil, L) in [(8, 1024, 128, 3, 1, 28672), (8, 1024, 128, 11, 5, 28672), (8, 2048, 64, 3, 1, 28672),
                                  (32, 128, 128, 3, 1, 28672), (32, 128, 128, 11, 5, 28672), (32, 32, 32, 3, 1, 114688),
                                  (8, 1024, 32, 3, 1, 57344)]:
    x = torch.randn(B, cin, L, device=dev)
    w = torch.randn(cout, cin, k, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    for _ in range(2): y = conv1d(x, w, b, dilation=dil, in_slope=0.1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): y = conv1d(x, w, b, dilation=dil, in_slope=0.1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    fl = 2.0 * cout * cin * k * B * L
    print(f'B{B} cin{cin} cout{cout} k{k} d{dil} L{L}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s')
    del x, w, y
