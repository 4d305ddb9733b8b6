"""Vocabulary of the Shakkala diacritizer checkpoint (reference models/diacritizers/shakkala/symbols.py):
pure data, id-compatible with `shakkala_second_model6.pth`.  Input ids: 4 specials, id 4 unused, then the
characters of _CHARS from id 5 on (the order is the checkpoint's, not code-point order); 28 output classes."""

SPECIALS = ('<PAD>', '<UNK>', '<GO>', '<EOS>')
_CHARS = '\xb0\u0648\u03b5\t\u0640\u06f8\u0638\u03c8\ufedb\u03c7\u0622\ufe81\u061b\u0634\u062e\u03c5\ufef9\u062a\u2026\u063a\ufd3f\u03c1\u03c3 \u0644\xbb\u200d\ufe91\ufed3\u2018\u03ba\u03b9\u06d2\u0642\u0649\xad\u2019\u2013\ufee3\ufd3e\u0670\u0641\u03b1\u0645\u0647\u0624\u03b8\u200b\ufb90\u03bc\u201c\u0626\ufe87\ufe88\u062c\u200f\ufe84\u2022\u03bd\u05d5\u0631\ufee0\u0671\u0627\u03ad\u064a\u062b\u0643\u0623\xab\u0635\ufe94\u03cc\u03c4\ufefb\u03b3\u0646\u0633\ufef5\xa0\u201d\u062d\ufe83\ufef4\u03c9\ufe8c\u066a\u0632\u0637\u202b\u0639\ufe82\u0630\ufef7\ufedf\ufe8b\u061f\ufee7\u03ce\u062f\u06cc\u06f7\u0629\u202c\u06f5\xb4\u0636\u03af\ufe92\u03bf\u2030\u03c0\u200e\u0628\u0625\u0621'
input_vocab_to_int = {**{s: i for i, s in enumerate(SPECIALS)}, **{c: 5 + i for i, c in enumerate(_CHARS)}}

_OUT = ('\u0640', '\u064e', '\u064f\u0651', '\u064e\u0651', '\u0640', '\u0651\u0650', '\u0651', '\u0652\u0651', '\u0651\u064d', '\u0650\u0651', '\u064d\u0651', '\u064c\u0651', '\u0651\u064e', '\u064f', '\u0651\u064c', '\u0651\u064b', '\u0652', '\u064d', '\u0650', '\u0651\u064f', '\u064b\u0651', '\u064c', '\u064b', '\u0651\u0651')
output_int_to_vocab = {**{i: s for i, s in enumerate(SPECIALS)}, **{4 + i: d for i, d in enumerate(_OUT)}}
