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


def test_routing_options_set_get_validate():
    """include/ttsamd.h: ttsamd_set_option / ttsamd_get_option / ttsamd_option_name -- the validated options table that replaced the per-call
    environment reads.  Unknown names and values outside an option's range are errors and change nothing; None restores the default."""
    import pytest
    from ttsamd import lib
    names = lib.option_names()
    assert 'TTSAMD_WINO' in names and 'TTSAMD_FUSED2_MASK' in names and 'TTSAMD_WINO4' in names and len(names) == len(set(names))
    assert lib.get_option('TTSAMD_WINO4') is None                       # unset: the documented default applies
    lib.set_option('WINO4', 6)                                          # with or without the prefix
    assert lib.get_option('TTSAMD_WINO4') == '6'
    for bad in ('99', '-1', 'x', '6 '):
        with pytest.raises(lib.TtsAmdError):
            lib.set_option('TTSAMD_WINO4', bad)
        assert lib.get_option('TTSAMD_WINO4') == '6'
    lib.set_option('TTSAMD_FUSED2_MASK', '1ff')                         # hex mask
    with pytest.raises(lib.TtsAmdError):
        lib.set_option('TTSAMD_FUSED2_MASK', '200')
    with pytest.raises(lib.TtsAmdError):
        lib.set_option('TTSAMD_NO_SUCH_SWITCH', '1')
    with lib.options(TTSAMD_WINO4=15, TTSAMD_WINO=0):
        assert lib.get_option('WINO4') == '15' and lib.get_option('WINO') == '0'
    assert lib.get_option('WINO4') == '6' and lib.get_option('WINO') is None
    lib.set_option('TTSAMD_WINO4', None)
    lib.set_option('TTSAMD_FUSED2_MASK', None)
    assert lib.get_option('TTSAMD_WINO4') is None


def test_options_are_seeded_from_the_environment_once_and_a_bad_value_is_loud():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from ttsamd import lib; lib.load(); print(lib.get_option('TTSAMD_WINO2'))"
            % os.path.join(REPO, 'tts-arabic-pytorch_amd'))
    ok = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, TTSAMD_WINO2='6'), capture_output=True, text=True)
    assert ok.returncode == 0 and ok.stdout.strip() == '6', ok.stderr[-800:]
    bad = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, TTSAMD_WINO2='banana'), capture_output=True, text=True)
    assert bad.returncode != 0 and 'TTSAMD_WINO2' in bad.stderr


def test_no_per_call_getenv_in_the_product_sources():
    """Routing switches are read through the options table (api.hip: one getenv per option and process); what may still call getenv:
    the table itself, the closed-experiment hook exp_env (common.hpp, -DTTS_EXPERIMENT builds) and the conv-log path (read once)."""
    root = os.path.join(REPO, 'tts-arabic-pytorch_amd', 'csrc')
    for fn in sorted(os.listdir(root)):
        if not fn.endswith(('.hip', '.hpp')):
            continue
        with open(os.path.join(root, fn), encoding='utf-8') as f:
            for ln, line in enumerate(f, 1):
                code = line.split('//')[0]
                if 'getenv(' in code:
                    assert (fn, 'TTSAMD_CONV_LOG' in code or 'kOpts[i].name' in code or fn == 'common.hpp') in (
                        ('conv_mfma.hip', True), ('api.hip', True), ('common.hpp', True)), f'{fn}:{ln}: {line.strip()}'
