"""Persistent decoder vs the per-step graph path on the same engine and inputs (env TTSAMD_TACO_PERSISTENT is read per call):
max-abs differences of mel / alignments per max_step, for the multi-speaker (memory dim 640) and single-speaker (512) models."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tts-arabic-pytorch_amd'))
import torch  # noqa: E402


def main():
    from ttsamd.config import TACOTRON2_CONFIG
    from ttsamd.synth import tacotron2_state_dict
    from ttsamd.engine import Tacotron2Engine
    dev = torch.device('cuda:0')
    for nspk in (40, 1):
        cfg = dict(TACOTRON2_CONFIG, num_speakers=nspk)
        sd = tacotron2_state_dict(cfg, seed=0, gate_bias=-20.0)
        eng = Tacotron2Engine(sd, cfg, device=dev)
        for B, L in ((3, 23), (8, 64), (1, 7)):
            g = torch.Generator().manual_seed(B)
            lens = torch.sort(torch.randint(max(1, L // 2), L + 1, (B,), generator=g), descending=True).values
            lens[0] = L
            tok = torch.randint(1, 40, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
            sids = (torch.arange(B) % nspk) if nspk > 1 else None
            for steps in (1, 2, 3, 8, 30):
                for seed in (-1, 5):
                    out = {}
                    for mode in ('0', '1'):
                        __import__('ttsamd.lib', fromlist=['x']).set_option('TTSAMD_TACO_PERSISTENT', mode)
                        mel, ml, al = eng.infer(tok, sids, lens, max_step=steps, dropout_seed=seed)
                        out[mode] = (mel.cpu(), ml.cpu(), al.cpu())
                    dm = (out['0'][0] - out['1'][0]).abs().amax(dim=(0, 1))
                    da = (out['0'][2] - out['1'][2]).abs().amax(dim=(0, 2))
                    print(f'spk {nspk} B {B} L {L} steps {steps} seed {seed}: mel {float(dm.max()):.2e} (first bad step {int((dm > 1e-4).float().argmax()) if (dm > 1e-4).any() else -1}) '
                          f'align {float(da.max()):.2e} (first bad step {int((da > 1e-5).float().argmax()) if (da > 1e-5).any() else -1}) lens {out["0"][1].tolist()} {out["1"][1].tolist()}')


if __name__ == '__main__':
    main()
