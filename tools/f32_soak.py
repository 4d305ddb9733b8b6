"""Determinism soak of the fp32 engine with the second-generation fused pairs in (resblock_pair2 at both block widths, three-stream ResBlock
schedule, two-stream pipeline): N back-to-back calls of alternating shapes and routings must reproduce their first result bit for bit,
and the fused routings must agree with the un-fused engine to fp32 rounding.   python tools/f32_soak.py [N = 120]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
import torch
from ttsamd import synth, lib
from ttsamd.engine import FastPitchEngine, HifiGanEngine
from ttsamd.pipeline import FastPitchHifiGan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device('cuda:0')
fp, hg = FastPitchEngine(synth.fastpitch_state_dict()), HifiGanEngine(synth.hifigan_state_dict())
pipe = FastPitchHifiGan(fp, hg, dev)
cases = []
for b, lt in ((32, 64), (1, 64), (3, 40), (8, 64)):
    ids = synth.synth_ids(b, lt)
    if b == 3:
        ids[1, 25:] = 0
    dur = synth.synth_durations(b, lt) * (ids != 0)
    cases.append((torch.from_numpy(ids).to(dev), torch.from_numpy(dur).to(dev)))
routes = [{}, {'TTSAMD_FUSED2_MASK': '1ff', 'TTSAMD_FUSED2_MASK_N1': '000'}, {'TTSAMD_FUSED2_MASK': '1ff', 'TTSAMD_FUSED2_MASK_N1': '1ff'},
          {'TTSAMD_FUSED2': '0'}, {'TTSAMD_WINO4': '0'}]
ref, bad, worst = {}, 0, 0.0
for it in range(n):
    k, r = it % len(cases), (it // len(cases)) % len(routes)
    for key in ('TTSAMD_FUSED2', 'TTSAMD_FUSED2_MASK', 'TTSAMD_FUSED2_MASK_N1', 'TTSAMD_WINO4'):      # routing options go through the C ABI
        lib.set_option(key, routes[r].get(key))
    ids, dur = cases[k]
    if it % 2:
        wave = pipe.submit(ids, dur_tgt=dur)[2]
    else:
        mel, dl, *_ = fp.infer(ids, dur_tgt=dur)
        wave = hg.forward(mel, dl)
    torch.cuda.synchronize()
    if (k, r) not in ref:
        ref[(k, r)] = wave.clone()
    elif not torch.equal(ref[(k, r)], wave):
        bad += 1
    if (k, 3) in ref and r != 3:
        worst = max(worst, float((wave - ref[(k, 3)]).abs().max()))
print(f'{n} calls, {bad} differ from the first result of their (shape, routing); fused routings vs the un-fused engine: max-abs {worst:.2e}')
sys.exit(1 if bad or worst > 1e-5 else 0)
