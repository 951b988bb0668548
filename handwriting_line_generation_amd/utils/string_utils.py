"""Label <-> string helpers (reference: utils/string_utils.py, utils/error_rates.py)."""
import numpy as np


def str2label_single(value, characterToIndex, unknown_index=None):
    return np.array([characterToIndex[v] for v in value if v in characterToIndex], np.uint32)


def label2str_single(label, indexToCharacter, asRaw, spaceChar="~"):
    out = ""
    for v in label:
        if v == 0:
            if not asRaw:
                break
            out += spaceChar
        else:
            out += indexToCharacter[v]
    return out


def naive_decode(output):
    """greedy CTC decode of [T, C] scores: collapse repeats, drop blanks"""
    raw = np.argmax(output, axis=1)
    pred = [raw[i] for i in range(len(raw)) if raw[i] != 0 and not (i > 0 and raw[i] == raw[i - 1])]
    return pred, list(raw)


def _levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def cer(r, h, casesensitive=True):
    r, h = " ".join(r.split()), " ".join(h.split())
    if not casesensitive:
        r, h = r.lower(), h.lower()
    if len(r) == 0:
        return len(h)
    return _levenshtein(r, h) / float(len(r))


def wer(r, h, casesensitive=True):
    if not casesensitive:
        r, h = r.lower(), h.lower()
    r, h = r.split(), h.split()
    if len(r) == 0:
        return len(h)
    return _levenshtein(r, h) / float(len(r))
