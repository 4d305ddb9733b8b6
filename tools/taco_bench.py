"""BASELINE config 4: Tacotron2 autoregressive decode + HiFi-GAN, batch 8, synthetic weights/tokens.
The gate is biased never to fire so every run decodes exactly --frames steps (fixed work).
    python tools/taco_bench.py [--batch 8] [--tokens 64] [--frames 448] [--steps 5]
Prints one JSON line: ms per utterance batch, us per decoder step, audio samples/s, RTF."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--tokens', type=int, default=64)
    ap.add_argument('--frames', type=int, default=448)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--precision', default='f32')
    a = ap.parse_args()
    from ttsamd import engine as E
    from ttsamd.config import TACOTRON2_CONFIG, HIFIGAN_CONFIG, SAMPLE_RATE, HOP
    from ttsamd.synth import tacotron2_state_dict, hifigan_state_dict, synth_ids
    dev = torch.device('cuda:0')
    E.set_precision(a.precision)
    taco = E.Tacotron2Engine(tacotron2_state_dict(TACOTRON2_CONFIG, seed=0, gate_bias=-30.0), TACOTRON2_CONFIG, device=dev)
    voc = E.HifiGanEngine(hifigan_state_dict(HIFIGAN_CONFIG, seed=0), HIFIGAN_CONFIG, device=dev)
    ids = torch.from_numpy(synth_ids(a.batch, a.tokens)).to(dev)
    lens = torch.full((a.batch,), a.tokens, dtype=torch.int64, device=dev)
    sids = torch.zeros(a.batch, dtype=torch.int64, device=dev)

    def step(seed):
        mel, mel_lens, _ = taco.infer(ids, sids, lens, max_step=a.frames, dropout_seed=seed)
        return mel, voc.forward(mel.contiguous(), mel_lens.to(torch.int64))

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    # (1) whole batches back to back, no intermediate synchronisation: the throughput number
    t0 = time.perf_counter()
    for i in range(a.steps):
        mel, wave = step(100 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # (2) the Tacotron2 call alone, synchronised on both sides
    t_dec = 0.0
    for i in range(a.steps):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        mel, mel_lens, _ = taco.infer(ids, sids, lens, max_step=a.frames, dropout_seed=200 + i)
        torch.cuda.synchronize()
        t_dec += time.perf_counter() - t1
    assert mel.shape[2] == a.frames and bool(torch.isfinite(wave).all())
    samples = a.batch * a.frames * HOP
    print(json.dumps({
        'workload': f'tacotron2+hifigan b{a.batch} x {a.tokens} tokens -> {a.frames} frames', 'precision': a.precision,
        'ms_per_batch': dt * 1e3, 'ms_tacotron2': t_dec / a.steps * 1e3,
        'us_per_decoder_step': t_dec / a.steps / a.frames * 1e6,
        'samples_per_s': samples / dt, 'rtf': dt / (samples / SAMPLE_RATE)}))


if __name__ == '__main__':
    main()
