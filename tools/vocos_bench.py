"""MelVocos alone (config 5's second stage) at batch 32: ms per call, with the k = 1 GEMMs on conv_wino4.hip's skeleton (default)
and on the direct kernel (TTSAMD_WINO4 = 15).  python tools/vocos_bench.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
from ttsamd import synth, engine as E, lib as L
dev = torch.device('cuda:0')
voc = E.VocosEngine(synth.vocos_state_dict(), device=dev)
dur = synth.synth_durations(32, 64)
lens = torch.from_numpy(dur.sum(1)).to(dev).to(torch.int64)
mel = torch.randn(32, 80, int(lens.max()), device=dev)
outs = {}
for rnd in range(3):
    for mask in ('31', '15'):
        L.set_option('TTSAMD_WINO4', mask)
        for _ in range(3): y = voc.forward(mel, lens)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): y = voc.forward(mel, lens)
        torch.cuda.synchronize(); print('vocos B=32 WINO4=%s: %.3f ms' % (mask, (time.perf_counter() - t0) * 100))
        outs[mask] = y[0] if isinstance(y, (tuple, list)) else y
print('bit-identical:', torch.equal(outs['31'], outs['15']), 'max diff %.3e' % float((outs['31'] - outs['15']).abs().max()))
# a T that is not a multiple of 4 (3 of 4 real batches): VocosEngine pads the frame rows to 16 bytes
L.set_option('TTSAMD_WINO4', None)
mel1 = torch.randn(32, 80, int(lens.max()) + 1, device=dev); lens1 = lens + 1
for _ in range(3): voc.forward(mel1, lens1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): voc.forward(mel1, lens1)
torch.cuda.synchronize(); print('vocos B=32, T = %d (not a multiple of 4): %.3f ms' % (mel1.shape[2], (time.perf_counter() - t0) * 100))
