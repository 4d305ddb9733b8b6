"""Debugging aid: the persistent decoder's state after N steps (TTSAMD_TACO_DUMP) against oracle/taco_oracle.py's trace."""
import os
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'tts-arabic-pytorch_amd'))
sys.path.insert(0, os.path.join(R, 'oracle'))
import torch  # noqa: E402


def main():
    import taco_oracle as T
    from ttsamd.config import TACOTRON2_CONFIG
    from ttsamd.synth import tacotron2_state_dict
    from ttsamd.engine import Tacotron2Engine
    nspk = int(os.environ.get('NSPK', '1'))
    B, L = 3, 23
    cfg = dict(TACOTRON2_CONFIG, num_speakers=nspk)
    sd = tacotron2_state_dict(cfg, seed=0, gate_bias=-20.0)
    eng = Tacotron2Engine(sd, cfg, device=torch.device('cuda:0'))
    g = torch.Generator().manual_seed(B)
    lens = torch.sort(torch.randint(max(1, L // 2), L + 1, (B,), generator=g), descending=True).values
    lens[0] = L
    tok = torch.randint(1, 40, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    sids = (torch.arange(B) % nspk) if nspk > 1 else None
    for steps in (1, 2):
        tr = {}
        mel_ref, _, al_ref = T.tacotron2_infer(sd, cfg, tok, sids, lens, max_step=steps, seed=-1, trace=tr)
        __import__('ttsamd.lib', fromlist=['x']).set_option('TTSAMD_TACO_PERSISTENT', os.environ.get('PMODE', '1'))
        os.environ['TTSAMD_TACO_DUMP'] = '/tmp/taco_dump.bin'
        mel, ml, al = eng.infer(tok, sids, lens, max_step=steps, dropout_seed=-1)
        raw = open('/tmp/taco_dump.bin', 'rb').read()
        hdr = np.frombuffer(raw[:32], dtype=np.int32)
        x = np.frombuffer(raw[32:], dtype=np.float32)
        Bq, Lq, M, nst, Lp, PTp, stepf, _ = hdr.tolist()
        last = steps - 1
        att_h = x[0:8192].reshape(256, 8, 4).transpose(1, 0, 2).reshape(8, 1024)[:B]          # [block][utterance][unit]
        dec_h = x[8192:16384].reshape(256, 8, 4).transpose(1, 0, 2).reshape(8, 1024)[:B]
        mc = M // 32
        ctx = x[16384:24576].reshape(8, 32, 32)[:B, :, :mc].reshape(B, M)
        aw = x[34816:34816 + 8 * Lp].reshape(8, Lp)[:B, :L]
        hc = tr['hc'][last].numpy()
        print(f'steps {steps}: att_h {np.abs(att_h - tr["att_h"][last].numpy()).max():.2e}  dec_h {np.abs(dec_h - hc[:, :1024]).max():.2e} '
              f'ctx {np.abs(ctx - hc[:, 1024:]).max():.2e}  aw {np.abs(aw - al_ref[:, last].numpy()).max():.2e}  '
              f'mel_post {np.abs(mel.cpu().numpy() - mel_ref.numpy()).max():.2e}')
        if steps == 1:
            d = np.abs(dec_h - hc[:, :1024])
            print('   dec_h worst units', np.argsort(-d.max(0))[:12].tolist(), 'mean err', float(d.mean()), 'ref absmean', float(np.abs(hc[:, :1024]).mean()))
            d = np.abs(att_h - tr["att_h"][last].numpy())
            print('   att_h worst units', np.argsort(-d.max(0))[:12].tolist(), 'mean err', float(d.mean()))


if __name__ == '__main__':
    main()
