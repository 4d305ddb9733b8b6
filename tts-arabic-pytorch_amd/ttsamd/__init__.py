"""ttsamd — MI355X-native FastPitch -> HiFi-GAN hot path (host side of libttsamd.so).

The arithmetic lives in hand-written HIP kernels behind the C ABI of include/ttsamd.h;
this package only moves pointers: PyTorch-ROCm tensors supply device memory and the
current stream.  There is NO CPU / PyTorch fallback: importing `ttsamd.lib` without the
built shared object, or running an engine without a gfx950 device, raises.
"""
from .config import NET_CONFIG, HIFIGAN_CONFIG, SAMPLE_RATE, HOP  # noqa: F401
