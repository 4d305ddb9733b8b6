"""Manual GPU bring-up script (not a test): prints per-stage errors."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (tools/..)
sys.path[:0] = [os.path.join(REPO, 'tts-arabic-pytorch_amd'), os.path.join(REPO, 'oracle')]
import numpy as np, torch
import tts_oracle as O
from ttsamd import synth
from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
from ttsamd.engine import HifiGanEngine, FastPitchEngine, conv1d

dev = torch.device('cuda:0')
torch.manual_seed(0)
x = torch.randn(2, 32, 300); w = torch.randn(32, 32, 3) / 10; b = torch.randn(32)
y = conv1d(x.to(dev), w.to(dev), b.to(dev)).cpu()
ref = torch.nn.functional.conv1d(x, w, b, padding=1)
print('conv1d err', float((y - ref).abs().max()))

hsd = synth.hifigan_state_dict()
eng = HifiGanEngine(hsd)
wf = O.fold_weight_norm(hsd)
rng = np.random.default_rng(7)
mel = (rng.standard_normal((80, 12)) * 1.5 - 4.0).astype(np.float32)
wave = eng.forward(torch.from_numpy(mel)[None].to(dev)).cpu()
refw = O.hifigan_forward(wf, mel, HIFIGAN_CONFIG)
print('hifigan wave err', float((wave - refw).abs().max()), 'amp', float(refw.abs().max()))

fsd = synth.fastpitch_state_dict()
fe = FastPitchEngine(fsd)
ids = np.zeros((3, 16), np.int64); r = np.random.default_rng(5)
for i, n in enumerate([16, 9, 5]): ids[i, :n] = 1 + r.integers(0, 39, n)
dur = (1 + r.integers(0, 5, ids.shape)).astype(np.float32) * (ids != 0)
trace = {}
rm = O.fastpitch_infer(O.to_torch(fsd), NET_CONFIG, ids, dur_tgt=dur, trace=trace)
gm = fe.infer(ids, dur_tgt=dur)
for name, a, b_ in zip(['mel', 'dec_lens', 'dur', 'pitch', 'energy'], gm, rm):
    print(name, float((a.cpu().double() - b_.double()).abs().max()))
