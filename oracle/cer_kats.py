"""Test infrastructure: inputs of the getCER known-answer cases (reference: trainer/hw_with_style_trainer.py:894-914,
utils/error_rates.py:2-26). tools/gen_golden_pretrain.py feeds them to the reference trainer's getCER and stores the answers
(tests/golden/valid_gan.json); tests/test_host_cpu.py feeds the same inputs to this package's getCER."""
import numpy as np

TEXTS = ["the quick brown fox", "Hello,  World", "a", "", "same same", "UPPER lower"]


def cases(idx_to_char, num_class):
    """-> [(pred [T,B,C] float32 scores, texts, casesensitive)]: trial 0 writes each text into the arg-max path, trial 1 drops every fifth
    character, trial 2 replaces every fourth one by another class; random scores elsewhere"""
    rs = np.random.RandomState(5)
    char_to_idx = {v: k for k, v in idx_to_char.items()}
    out = []
    for trial in range(3):
        T, B, C = 40, len(TEXTS), num_class
        pred = rs.randn(T, B, C).astype(np.float32)
        for b, txt in enumerate(TEXTS):
            t = 1
            for j, ch in enumerate(txt):
                if t >= T - 1 or ch not in char_to_idx:
                    continue
                if trial == 1 and j % 5 == 2:
                    continue
                ci = char_to_idx[ch]
                if trial == 2 and j % 4 == 1:
                    ci = ci % (C - 1) + 1
                pred[t, b, ci] += 12.0
                t += 2                      # a gap, so that doubled letters survive the repeat collapse
        for casesens in (True, False):
            out.append((pred, TEXTS, casesens))
    return out
