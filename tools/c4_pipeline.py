"""Config 4 (Tacotron2 + HiFi-GAN, batch 8 x 448 decoder steps) as a two-stage pipeline: the persistent decoder kernel of call i + 1 on one
stream UNDER the vocoder of call i on another.  Prints the one-stream step, the pipelined step, the decoder's step time under the vocoder and
whether the hand-offs timed out (fallback to the graph replay).  python tools/c4_pipeline.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
from ttsamd import synth, engine as E
from ttsamd.config import TACOTRON2_CONFIG, HIFIGAN_CONFIG
dev = torch.device('cuda:0')
frames_t, bt, n = 448, 8, 10
taco = E.Tacotron2Engine(synth.tacotron2_state_dict(TACOTRON2_CONFIG, seed=0, gate_bias=-30.0), TACOTRON2_CONFIG, device=dev)
hg = E.HifiGanEngine(synth.hifigan_state_dict(), device=dev)
tids = torch.from_numpy(synth.synth_ids(bt, 64)).to(dev)
tlens = torch.full((bt,), 64, dtype=torch.int64, device=dev)
sids = torch.zeros(bt, dtype=torch.int64, device=dev)
sync = torch.cuda.synchronize

def one(seed):
    mel, ml, _ = taco.infer(tids, sids, tlens, max_step=frames_t, dropout_seed=seed)
    return hg.forward(mel.contiguous(), ml.to(torch.int64)), mel

for i in range(2): w_ref, mel_ref = one(100 + i)
sync(); t0 = time.perf_counter()
for i in range(n): w_ref, mel_ref = one(100 + i)
sync(); print('one stream: %.2f ms per call' % ((time.perf_counter() - t0) / n * 1e3))
t0 = time.perf_counter()
for i in range(n): taco.infer(tids, sids, tlens, max_step=frames_t, dropout_seed=100 + i)
sync(); print('decoder alone: %.2f ms' % ((time.perf_counter() - t0) / n * 1e3))

s_t, s_v = torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)
for prio in ('vocoder high', 'decoder high'):
    if prio == 'decoder high':
        s_t, s_v = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)
    sync(); t0 = time.perf_counter()
    pend = None
    for i in range(n + 1):
        if pend is not None:
            mel, ml = pend
            s_v.wait_stream(s_t)
            with torch.cuda.stream(s_v):
                mel.record_stream(s_v); ml.record_stream(s_v)
                w = hg.forward(mel, ml)
        if i < n:
            with torch.cuda.stream(s_t):
                mel, ml, _ = taco.infer(tids, sids, tlens, max_step=frames_t, dropout_seed=100 + i)
                pend = (mel.contiguous(), ml.to(torch.int64))
        else:
            pend = None
    sync(); el = (time.perf_counter() - t0) / n * 1e3
    print('%s: pipelined %.2f ms per call; last wave equals the one-stream one: %s (max diff %.2e)' % (prio, el, torch.equal(w, w_ref), float((w - w_ref).abs().max())))
