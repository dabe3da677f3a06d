"""Query-side tokenizer: lowercase -> words of two or more [a-z0-9] -> drop stopwords -> Porter
stem -> counts in first-seen order.  Mirrors app/crawler/tokenizer.py:31-40 of the reference,
whose term strings are the keys of the `tokens` table, so a drop-in must produce the same
strings.

When nltk is importable (it is wherever the reference runs) its PorterStemmer and stopword list
are used, without ever downloading anything.  Otherwise the built-in stemmer below is used: the
Porter (1980) algorithm with the departures NLTK's default mode documents (irregular-form pool,
words of one or two letters left alone, -ies/-ied after a three-letter stem, the consonant test
for a final y, fulli/logi/bli in step 2).  nltk is not installed in the build container, so the
built-in stemmer is checked against known-answer pairs only (parity unpinned against nltk).
"""
from __future__ import annotations

import re
from collections import Counter

FALLBACK_STOPWORDS = frozenset(
    "a an and are as at be but by for if in into is it no not of on or such that the their then "
    "there these they this to was will with".split())

TOKEN_RE = re.compile(r"\b[a-zA-Z0-9]{2,}\b")

_VOWELS = "aeiou"
_POOL = {}
for _stem, _forms in {"sky": ["sky", "skies"], "die": ["dying"], "lie": ["lying"], "tie": ["tying"], "news": ["news"],
                      "inning": ["innings", "inning"], "outing": ["outings", "outing"],
                      "canning": ["cannings", "canning"], "howe": ["howe"], "proceed": ["proceed"],
                      "exceed": ["exceed"], "succeed": ["succeed"]}.items():
    for _f in _forms:
        _POOL[_f] = _stem


def _cons(w: str, i: int) -> bool:
    c = w[i]
    if c in _VOWELS:
        return False
    if c == "y":
        return i == 0 or not _cons(w, i - 1)
    return True


def _measure(stem: str) -> int:
    """Number of vowel-consonant sequences: the m of [C](VC)^m[V]."""
    m, prev_vowel = 0, False
    for i in range(len(stem)):
        v = not _cons(stem, i)
        if prev_vowel and not v:
            m += 1
        prev_vowel = v
    return m


def _has_vowel(stem: str) -> bool:
    return any(not _cons(stem, i) for i in range(len(stem)))


def _double_cons(w: str) -> bool:
    return len(w) >= 2 and w[-1] == w[-2] and _cons(w, len(w) - 1)


def _cvc(w: str) -> bool:
    n = len(w)
    if n >= 3 and _cons(w, n - 3) and not _cons(w, n - 2) and _cons(w, n - 1) and w[-1] not in "wxy":
        return True
    return n == 2 and not _cons(w, 0) and _cons(w, 1)


def _apply(word: str, rules) -> str:
    """First rule whose suffix matches decides: replaced if its condition holds, else unchanged."""
    for suffix, repl, cond in rules:
        if word.endswith(suffix):
            stem = word[: len(word) - len(suffix)] if suffix else word
            return stem + repl if cond is None or cond(stem) else word
    return word


def _m_gt0(s):
    return _measure(s) > 0


def _m_gt1(s):
    return _measure(s) > 1


def porter_stem(word: str) -> str:
    w = word.lower()
    if w in _POOL:
        return _POOL[w]
    if len(w) <= 2:
        return w
    # step 1a
    if w.endswith("ies") and len(w) == 4:
        w = w[:-3] + "ie"
    else:
        w = _apply(w, [("sses", "ss", None), ("ies", "i", None), ("ss", "ss", None), ("s", "", None)])
    # step 1b
    if w.endswith("ied"):
        w = w[:-3] + ("ie" if len(w) == 4 else "i")
    elif w.endswith("eed"):
        if _measure(w[:-3]) > 0:
            w = w[:-1]
    else:
        stem = None
        for suf in ("ed", "ing"):
            if w.endswith(suf):
                cand = w[: -len(suf)]
                if _has_vowel(cand):
                    stem = cand
                break
        if stem is not None:
            if stem.endswith(("at", "bl", "iz")):
                w = stem + "e"
            elif _double_cons(stem) and stem[-1] not in "lsz":
                w = stem[:-1]
            elif _measure(stem) == 1 and _cvc(stem):
                w = stem + "e"
            else:
                w = stem
    # step 1c
    if w.endswith("y") and len(w) > 2 and _cons(w, len(w) - 2):
        w = w[:-1] + "i"
    # step 2
    if w.endswith("alli") and _m_gt0(w[:-4]):
        w = w[:-4] + "al"
    w = _apply(w, [("ational", "ate", _m_gt0), ("tional", "tion", _m_gt0), ("enci", "ence", _m_gt0),
                   ("anci", "ance", _m_gt0), ("izer", "ize", _m_gt0), ("bli", "ble", _m_gt0), ("alli", "al", _m_gt0),
                   ("entli", "ent", _m_gt0), ("eli", "e", _m_gt0), ("ousli", "ous", _m_gt0),
                   ("ization", "ize", _m_gt0), ("ation", "ate", _m_gt0), ("ator", "ate", _m_gt0),
                   ("alism", "al", _m_gt0), ("iveness", "ive", _m_gt0), ("fulness", "ful", _m_gt0),
                   ("ousness", "ous", _m_gt0), ("aliti", "al", _m_gt0), ("iviti", "ive", _m_gt0),
                   ("biliti", "ble", _m_gt0), ("fulli", "ful", _m_gt0),
                   ("logi", "log", lambda s: _measure(s + "l") > 0)])
    # step 3
    w = _apply(w, [("icate", "ic", _m_gt0), ("ative", "", _m_gt0), ("alize", "al", _m_gt0), ("iciti", "ic", _m_gt0),
                   ("ical", "ic", _m_gt0), ("ful", "", _m_gt0), ("ness", "", _m_gt0)])
    # step 4
    w = _apply(w, [("al", "", _m_gt1), ("ance", "", _m_gt1), ("ence", "", _m_gt1), ("er", "", _m_gt1),
                   ("ic", "", _m_gt1), ("able", "", _m_gt1), ("ible", "", _m_gt1), ("ant", "", _m_gt1),
                   ("ement", "", _m_gt1), ("ment", "", _m_gt1), ("ent", "", _m_gt1),
                   ("ion", "", lambda s: _measure(s) > 1 and s[-1:] in ("s", "t")), ("ou", "", _m_gt1),
                   ("ism", "", _m_gt1), ("ate", "", _m_gt1), ("iti", "", _m_gt1), ("ous", "", _m_gt1),
                   ("ive", "", _m_gt1), ("ize", "", _m_gt1)])
    # step 5a
    if w.endswith("e"):
        stem = w[:-1]
        m = _measure(stem)
        if m > 1 or (m == 1 and not _cvc(stem)):
            w = stem
    # step 5b
    if w.endswith("ll") and _measure(w[:-1]) > 1:
        w = w[:-1]
    return w


def _load():
    """(stopwords, stem function): nltk's when importable and already installed, else built in."""
    stop, stem = FALLBACK_STOPWORDS, porter_stem
    try:
        from nltk.stem import PorterStemmer  # type: ignore

        stem = PorterStemmer().stem
        try:
            from nltk.corpus import stopwords  # type: ignore

            stop = frozenset(stopwords.words("english"))
        except LookupError:
            pass  # corpus not installed: the reference falls back to its short list as well
    except ImportError:
        pass
    return stop, stem


STOPWORDS, _stem = _load()


def tokenize(text: str) -> "Counter[str]":
    """term -> count, keys in first-seen order (what `list(tokenize(q).keys())` relies on)."""
    out: Counter[str] = Counter()
    for tok in TOKEN_RE.findall((text or "").lower()):
        if tok not in STOPWORDS:
            out[_stem(tok)] += 1
    return out
