"""Why does the two-stream schedule not overlap after an fp32 run in the same process?  (debug helper)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
import torch
from ttsamd import synth, lib as L
from ttsamd.engine import FastPitchEngine, HifiGanEngine, set_precision
from ttsamd.pipeline import FastPitchHifiGan

dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'a'
if 'b' in mode:
    set_precision('bf16')
fp, hg = FastPitchEngine(synth.fastpitch_state_dict()), HifiGanEngine(synth.hifigan_state_dict())
ids = torch.from_numpy(synth.synth_ids(32, 64)).to(dev)
dur = torch.from_numpy(synth.synth_durations(32, 64)).to(dev)
pipe = FastPitchHifiGan(fp, hg, dev)
lib = L.load()

def one():
    mel, dl, *_ = fp.infer(ids, dur_tgt=dur)
    return hg.forward(mel, dl)

def two():
    return pipe.submit(ids, dur_tgt=dur)[2]

def t(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

if 'f' in mode:
    print('fp32 one-stream', t(one, 5))
if 'p' in mode:
    lib.ttsamd_profile_enable(1); t(one, 3); lib.ttsamd_profile_enable(0)
set_precision('bf16')
if 't' in mode:
    print('bf16 two-stream (first)', t(two))
print('bf16 one-stream', t(one))
print('bf16 two-stream', t(two))
print('bf16 one-stream', t(one))
print('bf16 two-stream', t(two))
