"""Rule-based Modern-Standard-Arabic grapheme-to-phoneme conversion over Buckwalter
transliteration — own implementation of the behaviour of the reference's
text/phonetise_buckwalter.py (preprocess :164-194, per-word rules :197-378, utterance join
:381-400), restricted to what inference uses: the FIRST pronunciation of every word.
Pinned against the reference by tests/golden/infer_text_ids.npz and text_fuzz.npz."""
import re

# ---- script <-> Buckwalter (one table, inverted for the other direction) -----------------
_AR = ('بذطمترظنثزعهجسغحقف'
       'خصشدضكأءئؤإآاىةيلو'
       'ًٌٍَُِّْ')
_BW = "b*TmtrZn^zEhjsgHqfxS$dDk>'}&<|AYpylwFNKaui~o"
_AR2BW = dict(zip(_AR, _BW))
_BW2AR = dict(zip(_BW, _AR))


def arabic_to_buckwalter(word):
    return ''.join(_AR2BW.get(ch, ch) for ch in word)


def buckwalter_to_arabic(word):
    return ''.join(_BW2AR.get(ch, ch) for ch in word)


# ---- letter classes ----------------------------------------------------------------------
_HAMZAS = ">'}&<"
_PLAIN = {c: c for c in "b*TmtrZn^zEhjsgHqfxS$dDk"}
_PLAIN.update({c: '<' for c in _HAMZAS})           # every hamza seat is the glottal stop
_SHORT_NO_SHADDA = set('oauiFNK')
_DIACRITICS = _SHORT_NO_SHADDA | {'~'}
_VOWEL_LETTERS = set('AYwyaui')
_EMPHATIC = set('DSTZgxq')
_BACKWARD_EMPHATIC = _EMPHATIC - set('gx')
_CONSONANT_LETTERS = set(">< }&'bt^jHxd*rzs$SDTZEgfqklmnh|") - {' '}
_PUNCT = ('.', ',', '?', '!')
_LONG = {'w': ('uu0', 'UU0'), 'y': ('ii0', 'II0'), 'A': ('aa', 'AA'), 'Y': ('aa', 'AA')}
_SHORT = {'u': (('u0', 'u1'), ('U0', 'U1')), 'i': (('i0', 'i1'), ('I0', 'I1'))}

# irregular words, keyed by their letters out of "h*Ahn'>wl}kmyTtfd"; value = candidate
# pronunciations in priority order (a str = unconditional)
_IRREGULAR = {
    'h*A': ['h aa * aa', 'h aa * a'], 'h*h': ['h aa * i0 h i0', 'h aa * i1 h'],
    'h*An': ['h aa * aa n i0', 'h aa * aa n'], "h&lA'": ['h aa < u0 l aa < i0', 'h aa < u0 l aa <'],
    '*lk': ['* aa l i0 k a', '* aa l i0 k'], 'k*lk': ['k a * aa l i0 k a', 'k a * aa l i1 k'],
    '*lkm': '* aa l i0 k u1 m', '>wl}k': ['< u0 l aa < i0 k a', '< u0 l aa < i1 k'], 'Th': 'T aa h a',
    'lkn': ['l aa k i0 nn a', 'l aa k i1 n'], 'lknh': 'l aa k i0 nn a h u0', 'lknhm': 'l aa k i0 nn a h u1 m',
    'lknk': ['l aa k i0 nn a k a', 'l aa k i0 nn a k i0'], 'lknkm': 'l aa k i0 nn a k u1 m',
    'lknkmA': 'l aa k i0 nn a k u0 m aa', 'lknnA': 'l aa k i0 nn a n aa',
    'AlrHmn': ['rr a H m aa n i0', 'rr a H m aa n'],
    'Allh': ['ll aa h i0', 'll aa h', 'll AA h u0', 'll AA h a', 'll AA h', 'll A'],
    'h*yn': ['h aa * a y n i0', 'h aa * a y n'],
    'nt': 'n i1 t', 'fydyw': 'v i0 d y uu1', 'lndn': 'l A n d u1 n',
}
_IRREGULAR_KEEP = re.compile("[^h*Ahn'>wl}kmyTtfd]")

# ---- utterance normalisation: ordered rewrite rules (plain or regex) ---------------------
_REWRITES = [
    ('AF', 'F'), ('ـ', ''), ('o', ''), ('aA', 'A'), ('aY', 'Y'), (' A', ' '),
    ('F', 'an'), ('N', 'un'), ('K', 'in'), ('|', '>A'),
    ('i~', '~i'), ('a~', '~a'), ('u~', '~u'),
    (re.compile('Ai'), '<i'), (re.compile('Aa'), '>a'), (re.compile('Au'), '>u'),
    (re.compile('^>([^auAw])'), r'>a\1'), (re.compile(' >([^auAw ])'), r' >a\1'),
    (re.compile('<([^i])'), r'<i\1'),
    (re.compile(r'(\S)(\.|\?|,|!)'), r'\1 \2'),
]


def _normalise(utterance):
    for pat, rep in _REWRITES:
        utterance = utterance.replace(pat, rep) if isinstance(pat, str) else pat.sub(rep, utterance)
    return utterance.split(' ')


def _irregular(word):
    """First irregular pronunciation compatible with the word's final letter, or None."""
    cands = _IRREGULAR.get(_IRREGULAR_KEEP.sub('', word))
    if cands is None:
        return None
    if isinstance(cands, str):
        return cands.split(' ')
    last = word[-1] if word else ''
    if last == 'a':
        ok = lambda ph: ph in ('a', 'A')
    elif last == 'A':
        ok = lambda ph: ph == 'aa'
    elif last in 'ui' and last:
        ok = lambda ph: ph == last + '0'
    elif last in _PLAIN:
        ok = lambda ph: ph == _PLAIN[last]
    else:
        ok = lambda ph: ph in last          # substring test on the raw letter
    for c in cands:
        ph = c.split(' ')
        if ok(ph[-1]):
            return ph
    return None


def _first(x):
    """An entry of the phone list is a phone or a list of alternatives; inference keeps the first."""
    return x[0] if isinstance(x, list) else x


def _letter_phones(w, k, emph, out):
    """Append the phone entries letter w[k] contributes (w is padded with 'bb'..'ee')."""
    c, p1, p2, n1, n2 = w[k], w[k - 1], w[k - 2], w[k + 1], w[k + 2]
    if c in _PLAIN:
        out.append(_PLAIN[c])
    if c == 'l':                       # assimilated definite-article lam before a geminated sun letter
        out.append('' if (n1 not in _DIACRITICS and n1 not in _VOWEL_LETTERS and n2 == '~') else 'l')
    if c == '~' and p1 not in 'wy' and out:
        out[-1] = out[-1] + out[-1]    # gemination doubles the previous entry
    if c == '|':
        out.append(['<', 'AA' if emph else 'aa'])
    if c == 'p':                       # ta marbuta is pronounced only before a case vowel
        out.append('t' if n1 in _DIACRITICS else '')
    if c in 'wy':
        glide, (lng, lng_e) = c, _LONG[c]
        as_glide_ctx = (n1 in _SHORT_NO_SHADDA or n1 in 'AY'
                        or (n1 in 'wy' and n2 not in _DIACRITICS and n2 not in 'Awy')
                        or (p1 in _SHORT_NO_SHADDA and (n1 in _CONSONANT_LETTERS or n1 == 'e')))
        if as_glide_ctx:
            same_vowel = (c == 'w' and p1 == 'u' and n1 not in 'aiAY') or (c == 'y' and p1 == 'i' and n1 not in 'auAY')
            if same_vowel:
                out.append(lng_e if emph else lng)
            elif c == 'w' and n1 == 'A' and n2 == 'e':
                out.append([glide, lng])          # waw al-jama'a: alternatives, first one wins
            else:
                out.append(glide)
        elif n1 == '~':
            if p1 == 'a' or (c == 'w' and p1 in 'iy') or (c == 'y' and p1 in 'wu'):
                out.extend([glide, glide])
            else:
                out.extend([lng, glide])
        else:
            v = lng_e if emph else lng
            final_short = (p1 in _CONSONANT_LETTERS or p1 in 'ui') and n1 == 'e'
            out.append([v, v[1:]] if final_short else v)
    if c in 'ui':
        lax = (n1 in _PLAIN or n1 == 'l') and n2 == 'e' and len(w) > 7
        out.append(_SHORT[c][1 if emph else 0][1 if lax else 0])
    if c in 'aAY':
        if c == 'A' and p1 in 'wk' and p2 == 'b':
            out.append(['a', 'aa'])
        elif c == 'A' and p1 in 'ui':
            pass
        elif c == 'A' and p1 == 'w' and n1 == 'e':
            out.append(['aa', ''])
        elif c == 'a':
            out.append('A' if emph else 'a')
        elif n1 == 'e':
            out.append(['AA', 'A'] if emph else ['aa', 'a'])
        else:
            out.append('AA' if emph else 'aa')


def _tidy(ph):
    """Merge a short vowel into the identical long vowel that follows it, collapse repeated
    u0/i0 and fuse doubled glides — index bookkeeping as the reference does it (deletions are
    collected first and applied back to front)."""
    drop, prev = [], ''
    for i, cur in enumerate(ph):
        if cur in ('aa', 'uu0', 'ii0', 'AA', 'UU0', 'II0') and prev.lower() == cur[1:].lower():
            drop.append(i - 1)
            ph[i] = ph[i - 1][0] + ph[i - 1]
        if cur in ('u0', 'i0') and prev.lower() == cur.lower():
            drop.append(i - 1)
            ph[i] = ph[i - 1]
        if cur in ('y', 'w') and prev == cur:
            ph[i - 1] += ph[i - 1]
            drop.append(i)
        prev = cur
    for idx in reversed(drop):
        del ph[idx]
    return ph


def phonetise_word(word):
    if word in _PUNCT:
        return word
    fixed = _irregular(word)
    if fixed is not None:
        return _tidy(list(fixed))
    w = 'bb' + word + 'ee'
    entries, emph = [], False
    for k in range(2, len(w) - 2):
        c, n1 = w[k], w[k + 1]
        if (c in _CONSONANT_LETTERS or c in 'wy') and c not in _EMPHATIC:
            emph = False
        if c in _EMPHATIC or n1 in _BACKWARD_EMPHATIC:
            emph = True
        _letter_phones(w, k, emph, entries)
    return _tidy([p for p in (_first(e) for e in entries) if p != ''])


def process_utterance(utterance):
    words = []
    for word in _normalise(utterance):
        if word in ('-', 'sil'):
            words.append(['sil'])
            continue
        ph = phonetise_word(word)
        if isinstance(ph, str) and words:          # punctuation attaches to the previous word
            words[-1] = list(words[-1]) + list(ph)
        else:
            words.append(ph)
    return ' + '.join(' '.join(p for p in ph) for ph in words)
