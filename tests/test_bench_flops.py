"""The numerator of bench.py's roofline: algorithmic FLOPs per unit must equal SURVEY §8(d)'s figures
(HiFi-GAN 614.1 MFLOP per mel frame incl. 0.11 of conv_post; FastPitch 22.88 GFLOP per 64-token / 448-frame
utterance incl. the 0.26 GFLOP of attention scores that do not run in conv launches)."""
import importlib.util
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(REPO, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_hifigan_flops_per_frame():
    from ttsamd.config import HIFIGAN_CONFIG
    f = _bench().hifigan_flops_per_frame(HIFIGAN_CONFIG)
    conv_post = 2.0 * 32 * 7 * 256
    assert abs((f + conv_post) / 1e6 - 614.11) < 0.05
    assert abs(f / 256 / 1e6 - 2.3984) < 0.001            # MFLOP per output sample


def test_fastpitch_flops_per_utterance():
    from ttsamd.config import NET_CONFIG
    dec, enc = _bench().fastpitch_conv_flops_per_pos(NET_CONFIG)
    conv = dec * 448 + enc * 64                              # conv launches only
    attn = 6 * (2 * 2 * 64 * 448) * 448 + 6 * (2 * 2 * 64 * 64) * 64   # QK^T and PV, 1 head of 64, both stacks
    assert abs((conv + attn) / 1e9 - 22.88) < 0.05
    assert abs(dec / 1e6 - 43.71) < 0.02 and abs(enc / 1e6 - 46.60) < 0.02


def test_bf16_roofline_bound_follows_the_algorithmic_intensity():
    """`roofline.bound` of a bf16 launch sequence is whichever roof the algorithmic FLOP / HBM-byte ratio puts first (ridge =
    2.5 PFLOP/s / 8 TB/s = 312.5): the fused-pair step (562 FLOP/B) is MFMA-bound and `frac` is the MFMA fraction; a layer-wise
    step (137 FLOP/B) is HBM-bound and `frac` is the HBM fraction.  Both sides always ride along."""
    b = _bench()
    r = b.bf16_roofline(9.53e12, 16.95e9, 10e-3)
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - 953.0 / 2500.0) < 1e-6
    assert abs(r['hbm_frac'] - 1695.0 / 8000.0) < 1e-6 and abs(r['algorithmic_flop_per_byte'] - 562.2) < 0.5
    r = b.bf16_roofline(9.53e12, 9.53e12 / 137.0, 10e-3)
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and abs(r['frac'] - r['hbm_frac']) < 1e-12
    assert abs(r['mfma_frac'] - 953.0 / 2500.0) < 1e-6


def test_median_and_gpu_count_helpers(tmp_path, monkeypatch):
    b = _bench()
    assert b._median([3.0, 1.0, 2.0]) == 2.0 and b._median([4.0, 1.0, 2.0, 3.0]) == 2.5 and b._median([]) is None
    # the launcher parents count GPUs without the HIP runtime: the visibility variables win over the topology
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,3')
    assert b._count_gpus() == 2
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert b._count_gpus() == 0
