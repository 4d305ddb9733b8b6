"""Per-phase wall-clock stamps (100 MHz) of the persistent Tacotron2 decoder's LAST step, all 256 blocks; needs a library
built with -DTP_TIMING (tools/taco_phase_timing.sh stamps) and TTSAMD_TACO_DUMP."""
import os
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'tts-arabic-pytorch_amd'))
import torch  # noqa: E402


def main():
    from ttsamd.config import TACOTRON2_CONFIG
    from ttsamd.synth import tacotron2_state_dict, synth_ids
    from ttsamd.engine import Tacotron2Engine
    dev = torch.device('cuda:0')
    B, L, steps = 8, 64, 200
    eng = Tacotron2Engine(tacotron2_state_dict(TACOTRON2_CONFIG, seed=0, gate_bias=-30.0), TACOTRON2_CONFIG, device=dev)
    ids = torch.from_numpy(synth_ids(B, L)).to(dev)
    lens = torch.full((B,), L, dtype=torch.int64, device=dev)
    sids = torch.zeros(B, dtype=torch.int64, device=dev)
    os.environ['TTSAMD_TACO_DUMP'] = '/tmp/taco_dump.bin'
    for _ in range(2):
        eng.infer(ids, sids, lens, max_step=steps, dropout_seed=1)
    raw = open('/tmp/taco_dump.bin', 'rb').read()
    x = np.frombuffer(raw[32:], dtype=np.uint32)
    st = x[26624:26624 + 8192].reshape(256, 32)[:, 8:21].astype(np.int64)
    cyc = x[26624:26624 + 8192].reshape(256, 32)[:, 21].astype(np.int64)
    names = ['S1 att-lstm', 'bar1', 'S2/3 energies', 'bar2', 'S4 softmax+ctx', 'bar3', 'S5 dec-lstm', 'bar4', 'S6 proj', 'bar5', 'S7 prenet2', 'bar6']
    d = np.diff(st, axis=1) * 0.01           # us
    t0 = st[:, 0].min()
    print('step length (block 0): %.2f us, %d shader cycles -> %.0f MHz' % ((st[0, 12] - st[0, 0]) * 0.01, cyc[0], cyc[0] / ((st[0, 12] - st[0, 0]) * 0.01)))
    for i, n in enumerate(names):
        print(f'{n:16s} mean {d[:, i].mean():6.2f}  min {d[:, i].min():6.2f}  max {d[:, i].max():6.2f} us   (phase end spread over blocks: {(st[:, i + 1].max() - st[:, i + 1].min()) * 0.01:.2f} us)')


if __name__ == '__main__':
    main()
