"""Vocabulary of the Shakkelha diacritizer checkpoints (reference models/diacritizers/shakkelha/symbols.py):
pure data, kept id-compatible with `shakkelha_rnn_3_big_20.pth`.  The "big" input map is the four specials
followed by the character set in code-point order; classes 0-14 are the 8 diacritics and their shadda
combinations, 15-18 are specials that decode to nothing."""

ARABIC_LETTERS_LIST = '\u0621\u0622\u0623\u0624\u0625\u0626\u0627\u0628\u0629\u062a\u062b\u062c\u062d\u062e\u062f\u0630\u0631\u0632\u0633\u0634\u0635\u0636\u0637\u0638\u0639\u063a\u0641\u0642\u0643\u0644\u0645\u0646\u0647\u0648\u0649\u064a'
DIACRITICS_LIST = list('\u064e\u064b\u0650\u064d\u064f\u064c\u0652\u0651')

SPECIALS = ('<PAD>', '<SOS>', '<EOS>', '<UNK>')
_CHARS = '\n !"&\'()*+,-./0123456789:;=[]_`{}~\xab\xbb\u060c\u061b\u061f\u0621\u0622\u0623\u0624\u0625\u0626\u0627\u0628\u0629\u062a\u062b\u062c\u062d\u062e\u062f\u0630\u0631\u0632\u0633\u0634\u0635\u0636\u0637\u0638\u0639\u063a\u0641\u0642\u0643\u0644\u0645\u0646\u0647\u0648\u0649\u064a\u0660\u0661\u0662\u0664\u200d\u200f\u2013\u2019\u201c\u2026\ufd3e\ufd3f'
assert list(_CHARS) == sorted(_CHARS)
CHARACTERS_MAPPING = {**{s: i for i, s in enumerate(SPECIALS)}, **{c: 4 + i for i, c in enumerate(_CHARS)}}

_FATHA, _FATHATAN, _DAMMA, _DAMMATAN, _KASRA, _KASRATAN, _SUKUN, _SHADDA = (
    '\u064e', '\u064b', '\u064f', '\u064c', '\u0650', '\u064d', '\u0652', '\u0651')
_SIMPLE = ('', _FATHA, _FATHATAN, _DAMMA, _DAMMATAN, _KASRA, _KASRATAN, _SUKUN, _SHADDA)
REV_CLASSES_MAPPING = {**dict(enumerate(_SIMPLE)),
                       **{9 + i: _SHADDA + d for i, d in enumerate(_SIMPLE[1:7])},
                       15: '<PAD>', 16: '<SOS>', 17: '<EOS>', 18: '<N/A>'}
