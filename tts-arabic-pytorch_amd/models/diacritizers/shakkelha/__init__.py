"""Host side of the Shakkelha tagger (reference models/diacritizers/shakkelha/__init__.py:14-44): text -> ids,
class probabilities -> diacritized text."""
import torch

from .symbols import ARABIC_LETTERS_LIST, CHARACTERS_MAPPING, DIACRITICS_LIST, REV_CLASSES_MAPPING

_STRIP = str.maketrans('', '', ''.join(DIACRITICS_LIST))


def remove_diacritics(data, diacritics=DIACRITICS_LIST):
    return data.translate(str.maketrans('', '', ''.join(diacritics)))


def encode(input_text: str):
    """<SOS> + one id per non-diacritic character (<UNK> for unknown ones) + <EOS>."""
    unk = CHARACTERS_MAPPING['<UNK>']
    body = [CHARACTERS_MAPPING.get(ch, unk) for ch in input_text.translate(_STRIP)]
    return [CHARACTERS_MAPPING['<SOS>']] + body + [CHARACTERS_MAPPING['<EOS>']]


def decode(probs, input_text: str):
    """probs [1, T, 19]: position 0 is <SOS>; Arabic letters get the arg-max class unless it is a special."""
    classes = torch.argmax(probs[0][1:], dim=-1).tolist()
    out = []
    for ch, k in zip(input_text.translate(_STRIP), classes):
        out.append(ch)
        if ch in ARABIC_LETTERS_LIST and '<' not in REV_CLASSES_MAPPING[k]:
            out.append(REV_CLASSES_MAPPING[k])
    return ''.join(out)
