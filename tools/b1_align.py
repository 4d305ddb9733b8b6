import os, sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo/tts-arabic-pytorch_amd')
from ttsamd import synth, engine as E
dev = torch.device('cuda:0')
fp = E.FastPitchEngine(synth.fastpitch_state_dict(), device=dev)
hg = E.HifiGanEngine(synth.hifigan_state_dict(), device=dev)
ids = torch.from_numpy(synth.synth_ids(1, 64)).to(dev)
for T in (448, 447, 449, 450):
    dur = np.full((1, 64), 7.0, np.float32); dur[0, 0] += T - 448
    dur = torch.from_numpy(dur).to(dev)
    for _ in range(5): mel, dl, *_ = fp.infer(ids, dur_tgt=dur); w = hg.forward(mel, dl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): mel, dl, *_ = fp.infer(ids, dur_tgt=dur)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(30): mel, dl, *_ = fp.infer(ids, dur_tgt=dur); w = hg.forward(mel, dl)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('T = %d: FastPitch %.3f ms, whole call %.3f ms' % (int(dl[0]), (t1 - t0) / 30 * 1e3, (t2 - t1) / 30 * 1e3))
