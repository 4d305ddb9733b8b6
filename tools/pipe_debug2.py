"""Small-batch one-stream latency before / after the two-stream schedule has run in the process (debug helper)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
import torch
from ttsamd import synth
from ttsamd.engine import FastPitchEngine, HifiGanEngine, set_precision
from ttsamd.pipeline import FastPitchHifiGan

dev = torch.device('cuda:0')
set_precision('bf16')
fp, hg = FastPitchEngine(synth.fastpitch_state_dict()), HifiGanEngine(synth.hifigan_state_dict())
ids = torch.from_numpy(synth.synth_ids(32, 64)).to(dev)
dur = torch.from_numpy(synth.synth_durations(32, 64)).to(dev)
pipe = FastPitchHifiGan(fp, hg, dev)

def one(b):
    def f():
        mel, dl, *_ = fp.infer(ids[:b], dur_tgt=dur[:b])
        return hg.forward(mel, dl)
    return f

def two(b):
    return lambda: pipe.submit(ids[:b], dur_tgt=dur[:b])[2]

def t(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)

mode = sys.argv[1] if len(sys.argv) > 1 else ''
if 'w' in mode:      # a kernel on the default stream before anything else
    torch.zeros(16, device=dev).add_(1)
    torch.cuda.synchronize()
if 'h' in mode:      # one HiFi-GAN forward (three-stream schedule) on the default stream first
    hg.forward(torch.zeros(1, 80, 8, device=dev))
    torch.cuda.synchronize()
if 't' in mode:
    print('two-stream B=32 first', t(two(32)))
print('one-stream B=1', t(one(1)), 'B=8', t(one(8)), 'B=32', t(one(32)))
print('two-stream B=32', t(two(32)), 'B=8', t(two(8)), 'B=1', t(two(1)))
print('one-stream B=1', t(one(1)), 'B=8', t(one(8)), 'B=32', t(one(32)))
