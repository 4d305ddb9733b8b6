import json, os, sys, tempfile, time
import torch
REPO = '/root/repo'
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
eng = model.vocoder.engine()
order = sorted(range(len(lines)), key=lambda i: len(lines[i]))
sl = [lines[i] for i in order]
for G in (10, 13, 17, 20, 25, 34, 50, 100):
    grp = [sl[k:k + G] for k in range(0, len(sl), G)]
    msf, res = timed(lambda: [model.model.ttmel_lines_alone(g) for g in grp])
    msv, _ = timed(lambda: [eng.forward(m, l) for m, l in res])
    model._ALONE_GROUP = G
    mst, _ = timed(lambda: model.tts(lines, batch_size=1, denoise=0.0))
    print(f'groups of {G}: FastPitch {msf:.1f} ms, vocoder {msv:.1f} ms, tts() {mst:.1f} ms')
