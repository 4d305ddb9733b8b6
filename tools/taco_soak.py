"""Soak test of the persistent Tacotron2 decoder (default schedule): N back-to-back calls with a fixed dropout seed must return the
same tensors bit for bit (the hand-offs are asynchronous, the arithmetic is not), alternating two shapes so that the exchange arena
is re-used with different layouts.  python tools/taco_soak.py [calls]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
import torch  # noqa: E402


def main():
    from ttsamd.config import TACOTRON2_CONFIG
    from ttsamd.synth import tacotron2_state_dict, synth_ids
    from ttsamd.engine import Tacotron2Engine
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device('cuda:0')
    eng = Tacotron2Engine(tacotron2_state_dict(TACOTRON2_CONFIG, seed=0, gate_bias=-30.0), TACOTRON2_CONFIG, device=dev)
    shapes = [(8, 64, 96), (3, 41, 50)]
    inputs, ref = [], []
    for B, L, steps in shapes:
        ids = torch.from_numpy(synth_ids(B, L)).to(dev)
        lens = torch.full((B,), L, dtype=torch.int64, device=dev)
        lens[-1] = max(1, L - 7)
        sids = torch.arange(B, device=dev) % 40
        inputs.append((ids, sids, lens, steps))
        mel, ml, al = eng.infer(ids, sids, lens, max_step=steps, dropout_seed=11)
        assert bool(torch.isfinite(mel).all())
        ref.append((mel.clone(), ml.clone(), al.clone()))
    bad = 0
    for i in range(n):
        k = i % len(shapes)
        ids, sids, lens, steps = inputs[k]
        mel, ml, al = eng.infer(ids, sids, lens, max_step=steps, dropout_seed=11)
        if not (torch.equal(mel, ref[k][0]) and torch.equal(ml, ref[k][1]) and torch.equal(al, ref[k][2])):
            bad += 1
            print(f'call {i} (shape {shapes[k]}): max |d mel| {float((mel - ref[k][0]).abs().max()):.3e}')
    print(f'{n} calls, {bad} differ from the first call of their shape')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
