"""Config 1 (100 infer_text lines, batch_size 1) by part: the whole tts() call, the vocoder alone over the same groups of 16 mels (the floor of
the pipelined schedule), FastPitch alone over the 100 lines, the device -> host copies alone.  python tools/c1_parts.py"""
import json, os, sys, tempfile, time
import torch
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(REPO, 'tts-arabic-pytorch_amd'))
import text
from ttsamd import synth
from ttsamd.config import NET_CONFIG, HIFIGAN_CONFIG
from models.fastpitch import FastPitch2Wave
dev = torch.device('cuda:0')
lines = json.load(open(os.path.join(REPO, 'tests', 'golden', 'infer_text_lines.json'), encoding='utf-8'))
with tempfile.TemporaryDirectory() as d:
    fp_sd = {k: torch.from_numpy(v.copy()) for k, v in synth.fastpitch_state_dict().items()}
    torch.save({'model': fp_sd, 'config': dict(NET_CONFIG), 'symbols': list(text.symbols)}, os.path.join(d, 'fp.pth'))
    torch.save({'generator': {k: torch.from_numpy(v.copy()) for k, v in synth.hifigan_state_dict().items()}}, os.path.join(d, 'hg.pth'))
    json.dump(HIFIGAN_CONFIG, open(os.path.join(d, 'config.json'), 'w'))
    model = FastPitch2Wave(os.path.join(d, 'fp.pth'), vocoder_sd=os.path.join(d, 'hg.pth'), vocoder_config=os.path.join(d, 'config.json')).to(dev)

def timed(f, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, r

model.tts(lines[:4], batch_size=1, denoise=0.0)
for bs in (1, 32):
    ms, waves = timed(lambda: model.tts(lines, batch_size=bs, denoise=0.0))
    print(f'tts(100 lines, batch_size={bs}): {ms:.1f} ms')
# tokenisation alone
ms, _ = timed(lambda: [text.tokens_to_ids(model.model._tokenize(l), model.model.phon_to_id) for l in lines])
print(f'tokenisation of the 100 lines: {ms:.1f} ms')
# FastPitch alone, line by line
ms, mels = timed(lambda: [model.model.ttmel_single(l) for l in lines])
print(f'FastPitch, 100 x ttmel_single: {ms:.1f} ms')
eng = model.vocoder.engine()
lens = [int(m.shape[-1]) for m in mels]
print('frames', sum(lens), 'max', max(lens), 'min', min(lens))
def groups(G):
    out = []
    for g0 in range(0, len(mels), G):
        ms_ = mels[g0:g0 + G]; ls = lens[g0:g0 + G]
        mb = torch.zeros(len(ms_), 80, max(ls), device=dev)
        for i, m in enumerate(ms_): mb[i, :, :ls[i]] = m
        out.append((mb, torch.tensor(ls, dtype=torch.int64, device=dev)))
    return out
for G in (16, 32, 50):
    gs = groups(G)
    ms, ws = timed(lambda: [eng.forward(mb, ld) for mb, ld in gs])
    print(f'vocoder alone, groups of {G}: {ms:.1f} ms')
gs = groups(16); ws = [eng.forward(mb, ld) for mb, ld in gs]; torch.cuda.synchronize()
def d2h():
    out = []
    for (mb, ld), w in zip(gs, ws):
        n = (ld * eng.hop).tolist()
        out.extend(w[j, :n[j]].cpu() for j in range(len(n)))
    return out
ms, _ = timed(d2h)
print(f'D2H alone, per-utterance .cpu(): {ms:.1f} ms')
pin = torch.empty(sum(lens) * eng.hop, dtype=torch.float32).pin_memory()
def d2h_pinned():
    o = 0
    for (mb, ld), w in zip(gs, ws):
        n = (ld * eng.hop).tolist()
        for j in range(len(n)):
            pin[o:o + n[j]].copy_(w[j, :n[j]], non_blocking=True); o += n[j]
    torch.cuda.synchronize()
ms, _ = timed(d2h_pinned)
print(f'D2H alone, into one pinned buffer: {ms:.1f} ms')
# the batch_size = 1 list path as it runs now: lines sorted by length, balanced groups of 25, FastPitch rows computed as if alone
order = sorted(range(len(lines)), key=lambda i: len(lines[i]))
sl = [lines[i] for i in order]
grp = [sl[k:k + 25] for k in range(0, len(sl), 25)]
ms, res = timed(lambda: [model.model.ttmel_lines_alone(g) for g in grp])
print(f'FastPitch, 4 ragged calls of 25 sorted lines (alone mode): {ms:.1f} ms')
ms, _ = timed(lambda: [eng.forward(m, l) for m, l in res])
print(f'vocoder alone on those 4 groups: {ms:.1f} ms')
for m, l in res:
    print('   group: T_max', m.shape[-1], 'frames', int(l.sum()), 'padded', m.shape[-1] * m.shape[0])
