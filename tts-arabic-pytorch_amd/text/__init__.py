"""Text front-end of the drop-in surface: Arabic script / Buckwalter transliteration ->
phoneme string -> model tokens -> ids.  Function names follow the reference
(text/__init__.py:24-78) because callers import them; the implementation is table-driven.
"""
from text.symbols import symbols, DOUBLING_TOKEN, EOS_TOKEN, SEPARATOR_TOKEN
from text.phonetise import arabic_to_buckwalter, buckwalter_to_arabic, process_utterance

# phonetiser vowel variants (stress / emphasis / length marks) -> the six vowels the models know.
# Order matters for simplify_phonemes(): long vowels are replaced before their short prefixes.
_VOWEL_FAMILIES = (
    ('aa', 'aa AA'), ('uu', 'uu0 uu1 UU0 UU1'), ('ii', 'ii0 ii1 II0 II1'),
    ('a', 'a A'), ('u', 'u0 u1 U0 U1'), ('i', 'i0 i1 I0 I1'),
)
vowel_map = {variant: base for base, variants in _VOWEL_FAMILIES for variant in variants.split()}
vowels = list(vowel_map)

phon_to_id_ = {sym: idx for idx, sym in enumerate(symbols)}


def tokens_to_ids(phonemes, phon_to_id=None):
    """Token strings -> embedding rows; an unknown token raises KeyError like the reference."""
    lut = phon_to_id if phon_to_id is not None else phon_to_id_
    return [lut[tok] for tok in phonemes]


def ids_to_tokens(ids):
    return [symbols[i] for i in ids]


def _is_geminate(phon):
    return len(phon) == 2 and phon[0] == phon[1] and phon not in vowel_map


def phonemes_to_tokens(phonemes: str, append_space=True):
    """'b aa + dd a' -> ['b','aa','_+_','d','_dbl_','a', ('_+_',) '_eos_']: word separators become a
    token, doubled consonants become consonant + '_dbl_', vowel variants fold onto six vowels."""
    out = []
    for phon in phonemes.replace('sil', '').replace('+', SEPARATOR_TOKEN).split():
        if _is_geminate(phon):
            out.append(vowel_map.get(phon[0], phon[0]))
            out.append(DOUBLING_TOKEN)
        else:
            out.append(vowel_map.get(phon, phon))
    out.extend([SEPARATOR_TOKEN, EOS_TOKEN] if append_space else [EOS_TOKEN])
    return out


def buckwalter_to_phonemes(buckw):
    return process_utterance(buckw)


def arabic_to_phonemes(arabic):
    return buckwalter_to_phonemes(arabic_to_buckwalter(arabic))


def buckwalter_to_tokens(buckw, append_space=True):
    return phonemes_to_tokens(process_utterance(buckw), append_space=append_space)


def arabic_to_tokens(arabic, append_space=True):
    return buckwalter_to_tokens(arabic_to_buckwalter(arabic), append_space=append_space)


def simplify_phonemes(phonemes):
    for variant, base in vowel_map.items():
        phonemes = phonemes.replace(variant, base)
    return phonemes
