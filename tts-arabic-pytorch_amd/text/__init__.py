"""Text front-end with the reference's function names (text/__init__.py:24-78):
Arabic script / Buckwalter -> phonemes -> model tokens -> ids."""
from text.symbols import symbols, DOUBLING_TOKEN, EOS_TOKEN, SEPARATOR_TOKEN
from text.phonetise import arabic_to_buckwalter, buckwalter_to_arabic, process_utterance

# long vowels first so that prefix-sharing names are folded correctly by simplify_phonemes
vowel_map = {}
for _base, _names in (('aa', ('aa', 'AA')), ('uu', ('uu0', 'uu1', 'UU0', 'UU1')), ('ii', ('ii0', 'ii1', 'II0', 'II1')),
                      ('a', ('a', 'A')), ('u', ('u0', 'u1', 'U0', 'U1')), ('i', ('i0', 'i1', 'I0', 'I1'))):
    for _n in _names:
        vowel_map[_n] = _base
vowels = list(vowel_map)

phon_to_id_ = {phon: i for i, phon in enumerate(symbols)}


def tokens_to_ids(phonemes, phon_to_id=None):
    table = phon_to_id_ if phon_to_id is None else phon_to_id
    return [table[phon] for phon in phonemes]          # KeyError on OOV, as the reference


def ids_to_tokens(ids):
    return [symbols[i] for i in ids]


def arabic_to_phonemes(arabic):
    return process_utterance(arabic_to_buckwalter(arabic))


def buckwalter_to_phonemes(buckw):
    return process_utterance(buckw)


def phonemes_to_tokens(phonemes: str, append_space=True):
    """Geminated consonants 'bb' -> 'b', '_dbl_'; vowel variants folded to the 6 trained vowels."""
    tokens = []
    for phon in phonemes.replace('sil', '').replace('+', SEPARATOR_TOKEN).split():
        if len(phon) == 2 and phon not in vowel_map and phon[0] == phon[1]:
            first = phon[0]
            tokens += [vowel_map.get(first, first), DOUBLING_TOKEN]
        else:
            tokens.append(vowel_map.get(phon, phon))
    if append_space:
        tokens.append(SEPARATOR_TOKEN)
    tokens.append(EOS_TOKEN)
    return tokens


def buckwalter_to_tokens(buckw, append_space=True):
    return phonemes_to_tokens(buckwalter_to_phonemes(buckw), append_space=append_space)


def arabic_to_tokens(arabic, append_space=True):
    return buckwalter_to_tokens(arabic_to_buckwalter(arabic), append_space=append_space)


def simplify_phonemes(phonemes):
    for k, v in vowel_map.items():
        phonemes = phonemes.replace(k, v)
    return phonemes
