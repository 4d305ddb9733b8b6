import json, os, sys, tempfile, time
import torch
REPO = '/root/repo'
sys.path.insert(0, os.path.join(REPO, 'tts-arabic-pytorch_amd'))
import text
from ttsamd import synth, lib as L
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
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
model.tts(lines[:4], batch_size=1, denoise=0.0)
for name, prios, hs in (('default', (0, 0, 0), None), ('fp high', (-1, 0, 0), None), ('fp high, vocoder one stream', (-1, 0, 0), '0'),
                        ('default, vocoder one stream', (0, 0, 0), '0'), ('fp high, hg low(1)', (-1, 1, 0), None)):
    try:
        model._pipe_streams = tuple(torch.cuda.Stream(dev, priority=p) for p in prios)
    except Exception as e:
        print(name, 'stream creation failed', e); continue
    L.set_option('TTSAMD_HIFIGAN_STREAMS', hs)
    for bs in (1, 32):
        ms, waves = timed(lambda: model.tts(lines, batch_size=bs, denoise=0.0))
        print(f'{name}: tts(100 lines, batch_size={bs}): {ms:.1f} ms')
