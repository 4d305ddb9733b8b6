"""Wall-clock stamps (100 MHz) of the LAST step of the barrier-free persistent decoder (TTSAMD_TACO_PERSISTENT=2, library built
with -DTP_TIMING), all 256 blocks: where a step's 62 us go."""
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
    __import__('ttsamd.lib', fromlist=['x']).set_option('TTSAMD_TACO_PERSISTENT', '2')
    os.environ['TTSAMD_TACO_DUMP'] = '/tmp/taco_dump.bin'
    for _ in range(2):
        eng.infer(ids, sids, lens, max_step=steps, dropout_seed=1)
    raw = open('/tmp/taco_dump.bin', 'rb').read()
    x = np.frombuffer(raw[32:], dtype=np.uint32)
    st = x[26624:26624 + 8192].reshape(256, 32)[:, 8:19].astype(np.int64)
    names = ['stop flags in', 'S1 late (poll pre, gates, store att_h)', 'poll att_h', 'query + energies + store', 'S2 tail (att_h super-steps of both cells)',
             'S4 (poll epart, softmax, ctx)', 'S5 late (poll ctx, gates, store dec_h, context tails)', 'S6 (poll dec_h, proj, store h0)',
             'S6 tail (dec_h super-steps, next step)', 'S7 (poll h0, prenet 2)']
    d = np.diff(st, axis=1) * 0.01
    print('step length (block 0): %.2f us' % ((st[0, 10] - st[0, 0]) * 0.01))
    ends = (st[:, 1:] - st[:, :1].min()) * 0.01     # absolute end time of every phase on every block, from the earliest block's step top
    for i, n in enumerate(names):
        print(f'   [end of the phase below over the blocks: earliest {ends[:, i].min():6.2f}  mean {ends[:, i].mean():6.2f}  latest {ends[:, i].max():6.2f} us, latest block {int(ends[:, i].argmax())}, earliest block {int(ends[:, i].argmin())}]')
        print(f'{n:42s} mean {d[:, i].mean():6.2f}  min {d[:, i].min():6.2f}  max {d[:, i].max():6.2f} us   blocks 0-63 {d[:64, i].mean():6.2f}  block 80 {d[80, i]:6.2f}')


if __name__ == '__main__':
    main()
