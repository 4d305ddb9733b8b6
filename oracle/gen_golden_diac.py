"""Generate tests/golden/diacritizers.npz by running the REAL reference diacritizers
(/root/reference/models/diacritizers: Shakkelha, Shakkala) in this container with the deterministic
synthetic weights of ttsamd.synth.  Test infrastructure; run manually:  python oracle/gen_golden_diac.py
No reference source is copied; only inputs/outputs are stored."""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, 'tts-arabic-pytorch_amd', *rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_pkg = types.ModuleType('ttsamd')
_pkg.__path__ = [os.path.join(REPO, 'tts-arabic-pytorch_amd', 'ttsamd')]
sys.modules['ttsamd'] = _pkg
_load('ttsamd.config', ('ttsamd', 'config.py'))
synth = _load('ttsamd.synth', ('ttsamd', 'synth.py'))

sys.path.insert(0, '/root/reference')
os.chdir('/root/reference')
import torch  # noqa: E402
from models.diacritizers.shakkelha.network import Shakkelha  # noqa: E402
from models.diacritizers.shakkala.network import Shakkala  # noqa: E402
from models.diacritizers.shakkelha import encode as enc_a  # noqa: E402
from models.diacritizers.shakkala import encode as enc_b  # noqa: E402

DIAC = 'ًٌٍَُِّْ'


def main():
    with open(os.path.join(OUT, 'infer_text_lines.json'), encoding='utf-8') as f:
        lines = json.load(f)
    import text as ref_text                              # the reference's own transliteration table
    plain = [''.join(ch for ch in ref_text.buckwalter_to_arabic(lines[i]) if ch not in DIAC) for i in (0, 3, 17, 42)]
    plain.append('abc 123 مرحبا!')                 # OOV + digits + punctuation
    plain.append('س')                                                   # single letter
    arrs = {'texts': np.array(plain)}
    a = Shakkelha()
    a.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.shakkelha_state_dict().items()})
    b = Shakkala()
    b.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.shakkala_state_dict().items()})
    a.eval(); b.eval()
    outs_a, outs_b = [], []
    for i, t in enumerate(plain):
        o, p = a.predict(t, return_probs=True)
        outs_a.append(o)
        arrs[f'shakkelha_ids_{i}'] = np.asarray(enc_a(t), np.int64)
        arrs[f'shakkelha_probs_{i}'] = p[0].numpy()
        o, p = b._predict_single(t, return_probs=True)
        outs_b.append(o)
        arrs[f'shakkala_ids_{i}'] = np.asarray(enc_b(t, None)[0], np.int64)
        arrs[f'shakkala_probs_{i}'] = p[0].numpy()
    arrs['shakkelha_out'] = np.array(outs_a)
    arrs['shakkala_out'] = np.array(outs_b)
    # a padded call as Shakkala(max_sentence=...) would make it (zeros after the text)
    b.max_sentence = 40
    o, p = b._predict_single(plain[5] + plain[4], return_probs=True)
    arrs['shakkala_padded_text'] = np.array([plain[5] + plain[4]])
    arrs['shakkala_padded_probs'] = p[0].numpy()
    arrs['shakkala_padded_out'] = np.array([o])
    np.savez_compressed(os.path.join(OUT, 'diacritizers.npz'), **arrs)
    n_diac = [sum(ch in DIAC for ch in s) for s in outs_a], [sum(ch in DIAC for ch in s) for s in outs_b]
    print('diacritics emitted', n_diac, 'bytes', os.path.getsize(os.path.join(OUT, 'diacritizers.npz')))
    print(outs_a[0][:60]); print(outs_b[0][:60])


if __name__ == '__main__':
    main()
