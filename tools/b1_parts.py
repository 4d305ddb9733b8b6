"""Batch-1 (and small-batch) latency by part: FastPitch alone, HiFi-GAN alone (one / three streams), the whole call; fp32 and split bf16.
gpurun -- 'python3 tools/b1_parts.py [batch]'"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
from ttsamd import synth, lib  # noqa: E402
from ttsamd.engine import FastPitchEngine, HifiGanEngine, set_precision  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fp, hg = FastPitchEngine(synth.fastpitch_state_dict(), device=dev), HifiGanEngine(synth.hifigan_state_dict(), device=dev)
ids = torch.from_numpy(synth.synth_ids(32, 64)[:B]).to(dev)
dur = torch.from_numpy(synth.synth_durations(32, 64)[:B]).to(dev)


def timed(f, n=100):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for prec in ('f32', 'bf16x3'):
    set_precision(prec)
    mel, dl, *_ = fp.infer(ids, dur_tgt=dur)
    row = f'{prec} batch {B}: FastPitch {timed(lambda: fp.infer(ids, dur_tgt=dur)):.3f} ms'
    for s in ('0', '1', None):
        lib.set_option('TTSAMD_HIFIGAN_STREAMS', s)
        row += f', HiFi-GAN (streams={s}) {timed(lambda: hg.forward(mel, dl)):.3f}'
    row += f', whole call {timed(lambda: hg.forward(*fp.infer(ids, dur_tgt=dur)[:2])):.3f} ms'
    print(row)
set_precision('f32')
