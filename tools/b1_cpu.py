"""Is a batch-1 call bound by the HOST's launch rate?  CPU time of the enqueue (call returns) vs GPU time (sync), per part."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
from ttsamd import synth
from ttsamd.engine import FastPitchEngine, HifiGanEngine
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fp, hg = FastPitchEngine(synth.fastpitch_state_dict(), device=dev), HifiGanEngine(synth.hifigan_state_dict(), device=dev)
ids = torch.from_numpy(synth.synth_ids(32, 64)[:B]).to(dev)
dur = torch.from_numpy(synth.synth_durations(32, 64)[:B]).to(dev)
mel, dl, *_ = fp.infer(ids, dur_tgt=dur)
for name, f in (('HiFi-GAN', lambda: hg.forward(mel, dl)), ('FastPitch', lambda: fp.infer(ids, dur_tgt=dur))):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    cpu, tot = [], []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        cpu.append(t1 - t0); tot.append(t2 - t0)
    cpu.sort(); tot.sort()
    print(f'batch {B} {name}: host enqueue {cpu[15] * 1e3:.3f} ms, until the GPU is done {tot[15] * 1e3:.3f} ms (medians, one call at a time from an idle GPU)')
