"""Would a hipGraph replay shorten FastPitch's dependent launch chain at batch 1?  Captures the encode call and the
(length_regulate + decode) call of one utterance into two graphs (torch.cuda.CUDAGraph around the C-ABI calls on static tensors) and
times eager vs replay, and the whole vocoder call likewise.   gpurun -- 'python3 tools/graph_probe.py'"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
from ttsamd import synth, lib as L  # noqa: E402
from ttsamd.engine import FastPitchEngine, HifiGanEngine, _ptr  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fp, hg = FastPitchEngine(synth.fastpitch_state_dict(), device=dev), HifiGanEngine(synth.hifigan_state_dict(), device=dev)
lib = fp.lib
ids = torch.from_numpy(synth.synth_ids(32, 64)[:B]).to(dev)
dur = torch.from_numpy(synth.synth_durations(32, 64)[:B]).to(dev)
Lt, d = 64, fp.d_model
enc = torch.empty(B, d, Lt, device=dev); dur_pred = torch.empty(B, Lt, device=dev); pitch_pred = torch.empty(B, 1, Lt, device=dev)
energy_pred = torch.empty(B, Lt, device=dev); reps = torch.empty(B, Lt, dtype=torch.int64, device=dev)
dec_lens = torch.empty(B, dtype=torch.int64, device=dev)
nb_e = lib.ttsamd_fastpitch_encode_workspace_bytes(fp.handle, B, Lt)
ws_e = torch.empty(nb_e, dtype=torch.uint8, device=dev)


def encode(stream):
    L.check(lib.ttsamd_fastpitch_encode(fp.handle, _ptr(ids), B, Lt, 0, 1.0, _ptr(dur), None, None, 1.0, 0.0, 75.0, _ptr(enc), _ptr(dur_pred),
                                        _ptr(pitch_pred), _ptr(energy_pred), _ptr(reps), _ptr(dec_lens), _ptr(ws_e), nb_e, C.c_void_p(stream)), 'enc')


encode(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
t_max = int(dec_lens.max())
x = torch.empty(B, d, t_max, device=dev); mel = torch.empty(B, 80, t_max, device=dev)
nb_d = lib.ttsamd_fastpitch_decode_workspace_bytes(fp.handle, B, t_max)
ws_d = torch.empty(nb_d, dtype=torch.uint8, device=dev)


def decode(stream):
    L.check(lib.ttsamd_length_regulate(_ptr(enc), _ptr(reps), B, Lt, d, t_max, _ptr(x), None, C.c_void_p(stream)), 'lr')
    L.check(lib.ttsamd_fastpitch_decode(fp.handle, _ptr(x), _ptr(dec_lens), B, t_max, _ptr(mel), _ptr(ws_d), nb_d, C.c_void_p(stream)), 'dec')


def timed(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


cur = lambda: torch.cuda.current_stream().cuda_stream
print(f'batch {B}: eager encode {timed(lambda: encode(cur())):.3f} ms, decode ({t_max} frames) {timed(lambda: decode(cur())):.3f} ms')
s = torch.cuda.Stream()
graphs = {}
for name, f in (('encode', encode), ('decode', decode)):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        f(s.cuda_stream)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            f(s.cuda_stream)
    graphs[name] = g
mel_eager = mel.clone()
graphs['encode'].replay(); graphs['decode'].replay()
torch.cuda.synchronize()
print('graph replay reproduces the eager mel:', torch.equal(mel, mel_eager))
print(f'graph  encode {timed(graphs["encode"].replay):.3f} ms, decode {timed(graphs["decode"].replay):.3f} ms')
