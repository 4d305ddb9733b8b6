"""SURVEY §8(d) reporting list on one MI355X, synthetic weights (seed 0), one JSON object per config:
  C1  FastPitch2Wave.tts-equivalent, batch 1, the 100 committed infer_text.txt id sequences, denoise 0.005
  C2  FastPitch+HiFi-GAN synthetic 64 tokens, B in {1, 8, 32}, forced durations, fp32 (+ denoise 0.005 at B=32)
  C3  the per-GPU share of B=256 / 8 GPUs (B=32) with bf16 MFMA operands, and split-bf16
  C4  see tools/taco_bench.py (separate: autoregressive)
  C5  FastPitch 4-speaker + MelVocos('22k'), B=32
    python tools/bench_configs.py [--steps 10] [--warmup 3] > profiles/r1/configs.jsonl
Times are medians over --steps calls, each bracketed by torch.cuda.synchronize(); audio stays in HBM."""
import argparse
import json
import os
import statistics
import sys
import time

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(REPO, 'tts-arabic-pytorch_amd'))

import torch  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    a = ap.parse_args()
    from ttsamd import engine as E, synth
    from ttsamd.config import NET_CONFIG, SAMPLE_RATE, HOP
    dev = torch.device('cuda:0')
    fp = E.FastPitchEngine(synth.fastpitch_state_dict(), device=dev)
    hg = E.HifiGanEngine(synth.hifigan_state_dict(), device=dev)
    den = E.DenoiserEngine(device=dev)
    bias = den.bias_spec(hg.forward(torch.zeros(1, 80, 88, device=dev))[0])

    def report(name, sec, samples, **kw):
        print(json.dumps(dict(config=name, ms_per_call=sec * 1e3, samples_per_s=samples / sec,
                              rtf=sec / (samples / SAMPLE_RATE), **kw)), flush=True)

    def run(ids, dur=None, denoise=0.0, speaker=0, fpe=fp, voc=hg):
        mel, dec_lens, *_ = fpe.infer(ids, dur_tgt=dur, speaker=speaker)
        wave = voc.forward(mel, dec_lens)
        if denoise > 0:
            wave = den.denoise(wave, dec_lens * HOP, bias, denoise)
        return wave, dec_lens

    # ---- C2: B in {1, 8, 32}
    for B in (1, 8, 32):
        ids = torch.from_numpy(synth.synth_ids(B, 64)).to(dev)
        dur = torch.from_numpy(synth.synth_durations(B, 64)).to(dev)
        sec, (_, dl) = timed(lambda: run(ids, dur), a.steps, a.warmup)
        report(f'C2 FastPitch+HiFi-GAN fp32 B={B} x 64 tokens', sec, int(dl.sum()) * HOP, frames=int(dl.sum()))
    sec, (_, dl) = timed(lambda: run(ids, dur, denoise=0.005), a.steps, a.warmup)
    report('C2 + denoise 0.005, B=32', sec, int(dl.sum()) * HOP)
    # predicted durations (no dur_tgt): the synthetic duration head is calibrated to ~8 frames/token
    sec, (_, dl) = timed(lambda: run(ids), a.steps, a.warmup)
    report('C2 predicted durations, B=32', sec, int(dl.sum()) * HOP, frames=int(dl.sum()))
    # ---- C3 per-GPU share
    for prec in ('bf16x3', 'bf16'):
        E.set_precision(prec)
        sec, (_, dl) = timed(lambda: run(ids, dur), a.steps, a.warmup)
        report(f'C3 per-GPU share (B=256/8 = 32), {prec} MFMA operands', sec, int(dl.sum()) * HOP)
    E.set_precision('f32')
    # ---- C1: the 100 id sequences of data/infer_text.txt, batch 1, denoise 0.005, wave to the host
    import numpy as np
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'infer_text_ids.npz'), allow_pickle=True)
    flat, off = g['flat'].astype(np.int64), g['offsets']
    seqs = [torch.from_numpy(flat[off[i]:off[i + 1]].copy())[None].to(dev) for i in range(len(off) - 1)]
    if seqs:
        def c1():
            n = 0
            for s in seqs:
                wave, dl = run(s, denoise=0.005)
                n += int(dl.sum()) * HOP
                wave[0, :int(dl[0]) * HOP].cpu()
            return n
        sec, n = timed(c1, max(1, a.steps // 3), 1)
        report(f'C1 batch 1 over the {len(seqs)} infer_text.txt lines (incl. D2H of each wave)', sec, n,
               tokens=sum(int(s.shape[1]) for s in seqs), ms_per_utterance=sec * 1e3 / len(seqs))
    # ---- C5: 4-speaker FastPitch + MelVocos('22k')
    cfg4 = dict(NET_CONFIG, n_speakers=4, speaker_emb_weight=1.0)
    fp4 = E.FastPitchEngine(synth.fastpitch_state_dict(cfg4), cfg4, device=dev)
    voc = E.VocosEngine(synth.vocos_state_dict(), device=dev)
    spk = [0]

    def c5():
        spk[0] = (spk[0] + 1) % 4
        mel, dec_lens, *_ = fp4.infer(ids, dur_tgt=dur, speaker=spk[0])
        return voc.forward(mel, dec_lens), dec_lens
    sec, (_, dl) = timed(c5, a.steps, a.warmup)
    report('C5 FastPitch 4-speaker + MelVocos(22k), B=32', sec, int(dl.sum()) * HOP)


if __name__ == '__main__':
    main()
