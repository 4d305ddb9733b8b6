"""ttsamd — MI355X-native FastPitch -> HiFi-GAN hot path (host side of libttsamd.so).

The arithmetic lives in hand-written HIP kernels behind the C ABI of include/ttsamd.h;
this package only moves pointers: PyTorch-ROCm tensors supply device memory and the
current stream.  There is NO CPU / PyTorch fallback: importing `ttsamd.lib` without the
built shared object, or running an engine without a gfx950 device, raises.
"""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
# queue run one after the other.  The path uses up to seven at once (caller's stream, the two stages of ttsamd.pipeline, the two
# ResBlock branch streams of HiFi-GAN, the list pipeline's copy stream): with four queues the FastPitch stage of the pipeline
# lands on a vocoder queue whenever a one-stream call came first and nothing overlaps (bf16 B=32: 11.7 ms per step instead of
# 10.2, tools/pipe_debug.py).  Read by the runtime when it initialises, i.e. before the first GPU call of the process: set
# here unless the user chose a value; a process that touched the GPU before importing ttsamd keeps its own setting.
_user_set = 'GPU_MAX_HW_QUEUES' in _os.environ
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def _warn_if_hip_is_already_up():
    """The setting above is read once, when the HIP runtime initialises.  A host application that made a GPU call before
    importing this package keeps the default of 4 queues: the results are the same, the two-stream schedules overlap less
    (DESIGN.md §4).  Say so instead of silently measuring something else."""
    import sys as _sys
    import warnings as _warnings
    _torch = _sys.modules.get('torch')
    if _user_set or _torch is None:
        return
    try:
        up = _torch.cuda.is_initialized()
    except Exception:                                              # noqa: BLE001
        up = False
    if up:
        _warnings.warn('ttsamd: the HIP runtime was initialised before `import ttsamd`, so GPU_MAX_HW_QUEUES=8 comes too late '
                       '(the path uses up to seven streams; with the default four hardware queues the two-stream schedules '
                       'serialise).  Set GPU_MAX_HW_QUEUES=8 in the environment of the process, or import ttsamd first.',
                       RuntimeWarning, stacklevel=3)


_warn_if_hip_is_already_up()

from .config import NET_CONFIG, HIFIGAN_CONFIG, SAMPLE_RATE, HOP  # noqa: F401
