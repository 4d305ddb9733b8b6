"""Host side of the Shakkala tagger (reference models/diacritizers/shakkala/__init__.py:5-31)."""
import torch

from .symbols import input_vocab_to_int, output_int_to_vocab


def combine_text_with_harakat(input_sent: str, output_sent):
    """character + its predicted haraka; '<UNK>' and tatweel predictions add nothing, missing ones neither."""
    harakat = list(output_sent) + [''] * max(0, len(input_sent) - len(output_sent))
    return ''.join(ch + ('' if h in ('<UNK>', 'ـ') else h) for ch, h in zip(input_sent, harakat))


def encode(input_text: str, max_sentence: int = 315):
    unk = input_vocab_to_int['<UNK>']
    ids = [input_vocab_to_int.get(ch, unk) for ch in input_text]
    padded = ids + [0] * (max_sentence - len(ids)) if max_sentence is not None else ids
    return padded, ids


def decode(probs, text_input: str, input_letters_ids):
    classes = torch.argmax(probs[0], dim=1).tolist()[:len(input_letters_ids)]
    return combine_text_with_harakat(text_input, [output_int_to_vocab[k] for k in classes])
