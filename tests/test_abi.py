"""CPU: the C-ABI library loads and exports every symbol include/ttsamd.h declares; no compute."""
import os
import re

from conftest import REPO


def test_header_symbols_exported():
    from ttsamd import lib
    with open(os.path.join(REPO, 'include', 'ttsamd.h')) as f:
        hdr = f.read()
    declared = set(re.findall(r'\b(ttsamd_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations found'
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    handle = lib.load()
    for name in declared:
        assert getattr(handle, name) is not None
    # one ABI revision in the header, the binding and the built library (a mismatch must fail at load, not corrupt a struct)
    assert int(re.search(r'#define TTSAMD_ABI_VERSION (\d+)', hdr).group(1)) == lib.ABI_VERSION == handle.ttsamd_version()


def test_no_gpu_fails_loudly():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from ttsamd.engine import HifiGanEngine
    from ttsamd.lib import TtsAmdError
    with pytest.raises(TtsAmdError):
        HifiGanEngine({})


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under tts-arabic-pytorch_amd/ may reference it."""
    root = os.path.join(REPO, 'tts-arabic-pytorch_amd')
    for dp_, _, files in os.walk(root):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.hpp', '.h')):
                with open(os.path.join(dp_, fn), encoding='utf-8') as f:
                    src = f.read()
                assert 'tts_oracle' not in src and 'import oracle' not in src, os.path.join(dp_, fn)


def test_package_import_asks_for_one_hardware_queue_per_stream():
    """ttsamd/__init__.py: GPU_MAX_HW_QUEUES=8 unless the user chose a value (read by the HIP runtime at its first GPU call)."""
    import subprocess
    import sys
    code = "import os, sys; sys.path.insert(0, %r); import ttsamd; print(os.environ['GPU_MAX_HW_QUEUES'])" % os.path.join(REPO, 'tts-arabic-pytorch_amd')
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    assert subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.strip() == '8'
    env['GPU_MAX_HW_QUEUES'] = '2'
    assert subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.strip() == '2'
