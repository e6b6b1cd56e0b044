"""Synthetic text for the TEXT transform's parity cases: sentences over the static dictionary's words plus a Zipf vocabulary of made-up words,
capitalised sentence starts, punctuation, numbers, LF or CR+LF line ends, optional UTF-8 letters, markup and the codec's own escape bytes."""
import functools
import json
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


@functools.lru_cache(maxsize=None)
def static_words():
    letters = json.load(open(os.path.join(_HERE, "golden", "reference_constants.json")))["text_codec"]["static_dictionary_letters"]
    return [w.lower() for w in re.findall(r"[A-Z][a-z]*", letters)]


@functools.lru_cache(maxsize=256)
def make_text(n, seed=0, vocab=4000, crlf=False, utf8=0.0, markup=False, escapes=0.0, upper=0.08, max_word=14, static_share=0.5):
    rng = np.random.default_rng(seed)
    sw = static_words()
    alpha = "etaoinshrdlcumwfgypbvkjxqz"
    made = []
    for _ in range(vocab):
        ln = int(rng.integers(2, max_word + 1))
        made.append("".join(alpha[min(25, int(abs(rng.normal(0, 7))))] for _ in range(ln)))
    zipf = np.arange(1, vocab + 1, dtype=np.float64) ** -1.05
    zipf /= zipf.sum()
    out = []
    size = 0
    eol = "\r\n" if crlf else "\n"
    accents = ["é", "è", "ü", "ñ", "€", "中"]
    while size < n:
        nwords = int(rng.integers(3, 18))
        parts = []
        for k in range(nwords):
            if rng.random() < static_share:
                w = sw[int(rng.integers(0, len(sw)))] if rng.random() < 0.3 else sw[min(len(sw) - 1, int(abs(rng.normal(0, 120))))]
            else:
                w = made[int(rng.choice(vocab, p=zipf))]
            r = rng.random()
            if k == 0 or r < upper:
                w = w[0].upper() + w[1:]
            elif r < upper + 0.01:
                w = w.upper()
            if utf8 and rng.random() < utf8:
                w = w + accents[int(rng.integers(0, len(accents)))] + "s"
            if escapes and rng.random() < escapes:
                w = w + ("\x0f" if rng.random() < 0.5 else "\x0e")
            if rng.random() < 0.03:
                w = str(int(rng.integers(0, 100000)))
            if markup and rng.random() < 0.05:
                w = "<" + w + ">" if rng.random() < 0.7 else "&amp;" + w
            parts.append(w)
        seps = [" ", " ", " ", " ", ", ", "; ", " - ", "_", "  ", ": "]
        s = parts[0]
        for w in parts[1:]:
            s += seps[int(rng.integers(0, len(seps)))] + w
        s += [". ", ".", "!", "?", "." + eol, "." + eol + eol, eol][int(rng.integers(0, 7))]
        b = s.encode("utf-8")
        out.append(b)
        size += len(b)
    return b"".join(out)[:n]
