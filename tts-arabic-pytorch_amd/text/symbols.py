"""The 40-entry phoneme inventory the acoustic models were trained on.  This table is DATA
(ids index the embedding matrix, models/fastpitch/networks.py:69-71 may override it from the
checkpoint); order as in the reference's text/symbols.py:9-53."""

PADDING_TOKEN = '_pad_'
EOS_TOKEN = '_eos_'
DOUBLING_TOKEN = '_dbl_'
SEPARATOR_TOKEN = '_+_'
EOS_TOKENS = [SEPARATOR_TOKEN, EOS_TOKEN]

_SPECIAL = [PADDING_TOKEN, EOS_TOKEN, '_sil_', DOUBLING_TOKEN, SEPARATOR_TOKEN]
_CONSONANTS = list("<bt^jHxd*rzs$SDTZEgfqklmnhwyv")
_VOWELS = ['a', 'u', 'i', 'aa', 'uu', 'ii']

symbols = _SPECIAL + _CONSONANTS + _VOWELS
assert len(symbols) == 40
