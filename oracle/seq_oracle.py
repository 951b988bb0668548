"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatements of the integer/sequence algorithms on the hot path, written from the reference's
behaviour:
  correct_pred   - model/hw_with_style.py:18-74 (banded DTW of 1-logp against the blank-interleaved label,
                   first-minimum tie break over (up, diag, left), backtrace, zero padding)
  gt_counts      - trainer/hw_with_style_trainer.py:670-697 (blank / duplicate run lengths per character)
  insert_spaces  - model/hw_with_style.py:302-328
  naive_decode   - utils/string_utils.py:51-57
Pinned against vectors produced by the reference itself: tests/golden/seq_kat.npz (tools/gen_golden.py).
"""
import math

import numpy as np
import torch


def correct_pred(pred, label):
    """pred [T,B,C] float log-probs, label [L,B] ints -> int64 [maxlen,B]"""
    p = pred.detach().cpu().float().numpy()
    lab = label.detach().cpu().long().numpy()
    T, B, _ = p.shape
    L = lab.shape[0]
    LL = 2 * L + 1
    ext = np.zeros((LL, B), dtype=np.int64)
    ext[1::2] = lab
    w = max(T // 2, abs(T - LL))
    paths = []
    one = np.float32(1.0)
    for b in range(B):
        cost = np.full((T + 1, LL + 1), np.inf, dtype=np.float32)
        cost[0, 0] = 0
        hist = np.zeros((T, LL), dtype=np.int8)
        for i in range(1, T + 1):
            lo, hi = max(1, i - w), min(LL, i + w)
            for j in range(lo, hi + 1):
                c = np.float32(one - p[i - 1, b, ext[j - 1, b]])
                cands = (cost[i - 1, j], cost[i - 1, j - 1], cost[i, j - 1])
                k = 0
                if cands[1] < cands[k]:
                    k = 1
                if cands[2] < cands[k]:
                    k = 2
                hist[i - 1, j - 1] = k
                cost[i, j] = np.float32(c + cands[k])
        i, j = T - 1, LL - 1
        path = [ext[j, b]]
        while i > 0 or j > 0:
            h = hist[i, j]
            if h == 0:
                i -= 1
            elif h == 1:
                i -= 1
                j -= 1
            else:
                j -= 1
            path.append(ext[j, b])
        paths.append(path[::-1])
    maxlen = max(len(q) for q in paths)
    out = torch.zeros((maxlen, B), dtype=torch.int64)
    for b, q in enumerate(paths):
        out[: len(q), b] = torch.tensor(q, dtype=torch.int64)
    return out


def gt_counts(index_spaced, label):
    """index_spaced int [T',B], label [L,B] -> (float [L,B,2], min over b of the final `pos`)"""
    idx = index_spaced.cpu().numpy()
    lab = label.cpu().numpy()
    Tp, B = idx.shape
    L = lab.shape[0]
    gt = torch.zeros((L, B, 2), dtype=torch.float32)
    minpos = None
    for b in range(B):
        c = d = pos = last = 0
        for i in range(Tp):
            v = int(idx[i, b])
            if v == 0 and last == 0:
                c += 1
            elif last == 0 or last == v:
                d += 1
                last = v
            else:
                assert int(lab[pos, b]) == last
                gt[pos, b, 0] = c
                gt[pos, b, 1] = d
                c, d = (1, 0) if v == 0 else (0, 1)
                pos += 1
                last = v
        minpos = pos if minpos is None else min(minpos, pos)
    return gt, minpos


def insert_spaces(label, label_lengths, counts, num_class, count_std, dup_std, count_duplicates=True, rng=np.random):
    """label [L,B], counts float [L,B,1|2] -> (one-hot [T,B,num_class], padded fractions)"""
    cn = counts.detach().cpu().numpy()
    lab = label.cpu().numpy()
    B = lab.shape[1]
    max_count = max(math.ceil(float(cn.max())), 3)
    lines = []
    for b in range(B):
        line = []
        for i in range(int(label_lengths[b])):
            n_blank = round(rng.normal(cn[i, b, 0].item(), count_std))
            n_dup = round(rng.normal(cn[i, b, 1].item(), dup_std)) if count_duplicates else 1
            line += [0] * n_blank + [int(lab[i, b])] * n_dup
        lines.append(line)
    T = max(len(l) for l in lines) + max_count
    spaced = torch.zeros((T, B, num_class))
    padded = []
    for b, line in enumerate(lines):
        for i, c in enumerate(line):
            spaced[i, b, c] = 1
        spaced[len(line):, b, 0] = 1
        padded.append((T - len(line)) / T)
    return spaced, padded


def naive_decode(logits):
    raw = np.argmax(logits, axis=1)
    out = [int(raw[i]) for i in range(len(raw)) if raw[i] != 0 and not (i > 0 and raw[i] == raw[i - 1])]
    return out, list(raw)


# ---- the same two algorithms through the C restatement (oracle/seq_oracle.c), for full-size checks ----
def _clib():
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libseq_oracle.so")
    if not os.path.exists(path):
        raise RuntimeError("oracle/libseq_oracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(path)
    lib.oracle_correct_pred.restype = ctypes.c_int
    lib.oracle_gt_counts.restype = ctypes.c_int
    return lib


def correct_pred_c(pred, label):
    import ctypes
    lib = _clib()
    p = np.ascontiguousarray(pred.detach().cpu().float().numpy())
    lab = np.ascontiguousarray(label.cpu().numpy().astype(np.int32))
    T, B, C = p.shape
    L = lab.shape[0]
    out = np.zeros((T + 2 * L + 1, B), dtype=np.int64)
    lens = np.zeros(B, dtype=np.int32)
    vp = ctypes.c_void_p
    n = lib.oracle_correct_pred(vp(p.ctypes.data), vp(lab.ctypes.data), T, B, C, L, vp(out.ctypes.data), vp(lens.ctypes.data))
    return torch.from_numpy(out[:n].copy())


def gt_counts_c(index_spaced, label):
    import ctypes
    lib = _clib()
    idx = np.ascontiguousarray(index_spaced.cpu().numpy().astype(np.int64))
    lab = np.ascontiguousarray(label.cpu().numpy().astype(np.int32))
    gt = np.zeros((lab.shape[0], lab.shape[1], 2), dtype=np.float32)
    vp = ctypes.c_void_p
    pos = lib.oracle_gt_counts(vp(idx.ctypes.data), vp(lab.ctypes.data), idx.shape[0], idx.shape[1], lab.shape[0], vp(gt.ctypes.data))
    return torch.from_numpy(gt), pos
