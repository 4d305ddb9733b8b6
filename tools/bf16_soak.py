"""Determinism soak of the bf16 octet engine (chained ResBlock, fused pairs, split-K reductions, two-stream schedule): N back-to-back
calls of alternating shapes must reproduce their first result bit for bit.   python tools/bf16_soak.py [N = 200]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
import torch
from ttsamd import synth
from ttsamd.engine import FastPitchEngine, HifiGanEngine, set_precision
from ttsamd.pipeline import FastPitchHifiGan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
set_precision('bf16')
fp, hg = FastPitchEngine(synth.fastpitch_state_dict()), HifiGanEngine(synth.hifigan_state_dict())
pipe = FastPitchHifiGan(fp, hg, dev)
cases = []
for b, lt in ((32, 64), (1, 64), (3, 40), (8, 64)):
    ids = synth.synth_ids(b, lt)
    if b == 3:
        ids[1, 25:] = 0
    dur = synth.synth_durations(b, lt) * (ids != 0)
    cases.append((torch.from_numpy(ids).to(dev), torch.from_numpy(dur).to(dev)))
ref, bad = {}, 0
for it in range(n):
    k = it % len(cases)
    ids, dur = cases[k]
    if it % 2:
        wave = pipe.submit(ids, dur_tgt=dur)[2]
    else:
        mel, dl, *_ = fp.infer(ids, dur_tgt=dur)
        wave = hg.forward(mel, dl)
    torch.cuda.synchronize()
    if k not in ref:
        ref[k] = wave.clone()
    elif not torch.equal(ref[k], wave):
        bad += 1
print(f'{n} calls, {bad} differ from the first result of their shape')
sys.exit(1 if bad else 0)
