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
