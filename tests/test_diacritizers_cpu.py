"""Diacritizer taggers on the CPU: the oracle against goldens produced by the real reference
modules (oracle/gen_golden_diac.py), the vocab tables, and the host-side encode/decode of the drop-in classes."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def diac(golden):
    return golden('diacritizers')


def test_oracle_matches_reference_probs(diac):
    import diac_oracle as D
    from ttsamd.synth import shakkelha_state_dict, shakkala_state_dict
    wa, wb = shakkelha_state_dict(), shakkala_state_dict()
    for i in range(len(diac['texts'])):
        pa = D.shakkelha_forward(wa, diac[f'shakkelha_ids_{i}'][None])[0].numpy()
        pb = D.shakkala_forward(wb, diac[f'shakkala_ids_{i}'][None])[0].numpy()
        assert np.max(np.abs(pa - diac[f'shakkelha_probs_{i}'])) < 1e-5
        assert np.max(np.abs(pb - diac[f'shakkala_probs_{i}'])) < 1e-5


def test_encode_decode_match_reference(diac):
    """models.diacritizers.{shakkelha,shakkala}: encode -> ids, decode(reference probs) -> reference strings."""
    from models.diacritizers import shakkelha as A, shakkala as B
    for i, t in enumerate(diac['texts']):
        assert A.encode(t) == diac[f'shakkelha_ids_{i}'].tolist()
        ids_pad, ids = B.encode(t, None)
        assert ids_pad == ids == diac[f'shakkala_ids_{i}'].tolist()
        assert A.decode(torch.from_numpy(diac[f'shakkelha_probs_{i}'])[None], t) == diac['shakkelha_out'][i]
        assert B.decode(torch.from_numpy(diac[f'shakkala_probs_{i}'])[None], t, ids) == diac['shakkala_out'][i]
    t = str(diac['shakkala_padded_text'][0])
    ids_pad, ids = B.encode(t, 40)
    assert len(ids_pad) == 40 and ids_pad[len(ids):] == [0] * (40 - len(ids))
    assert B.decode(torch.from_numpy(diac['shakkala_padded_probs'])[None], t, ids) == diac['shakkala_padded_out'][0]


def test_vocab_tables():
    from models.diacritizers.shakkelha.symbols import CHARACTERS_MAPPING, REV_CLASSES_MAPPING, DIACRITICS_LIST
    from models.diacritizers.shakkala.symbols import input_vocab_to_int, output_int_to_vocab
    assert len(CHARACTERS_MAPPING) == 91 and sorted(CHARACTERS_MAPPING.values()) == list(range(91))
    assert len(REV_CLASSES_MAPPING) == 19 and len(DIACRITICS_LIST) == 8
    assert len(input_vocab_to_int) == 120 and max(input_vocab_to_int.values()) == 120 and 4 not in input_vocab_to_int.values()
    assert len(output_int_to_vocab) == 28
