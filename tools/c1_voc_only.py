"""The vocoder alone on config 1's mels as the batch_size = 1 list path hands them over (4 ragged calls of 25 length-sorted lines) -- for a
rocprofv3 kernel table:  rocprofv3 --kernel-trace --stats -- python3 tools/c1_voc_only.py"""
import json, os, sys, tempfile, time
import torch
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(REPO, 'tts-arabic-pytorch_amd'))
import text
from ttsamd import synth, engine as E
from ttsamd.config import NET_CONFIG
from models.fastpitch.networks import FastPitch
lines = json.load(open(os.path.join(REPO, 'tests', 'golden', 'infer_text_lines.json'), encoding='utf-8'))
with tempfile.TemporaryDirectory() as d:
    fp_sd = {k: torch.from_numpy(v.copy()) for k, v in synth.fastpitch_state_dict().items()}
    torch.save({'model': fp_sd, 'config': dict(NET_CONFIG), 'symbols': list(text.symbols)}, os.path.join(d, 'fp.pth'))
    model = FastPitch(os.path.join(d, 'fp.pth')).to('cuda:0')
hg = E.HifiGanEngine(synth.hifigan_state_dict(), device=torch.device('cuda:0'))
sl = sorted(lines, key=len)
res = [model.ttmel_lines_alone(sl[k:k + 25]) for k in range(0, 100, 25)]
res = [(m.contiguous(), l) for m, l in res]
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for m, l in res:
        hg.forward(m, l)
    torch.cuda.synchronize()
    print('vocoder, 4 ragged calls: %.1f ms (%d frames)' % ((time.perf_counter() - t0) * 1e3, sum(int(l.sum()) for _, l in res)))
