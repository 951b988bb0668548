"""Text-only instances for the "gen" lessons (reference: datasets/text_data.py:6-110, non-word / non-balanced mode).

Draws `batch_size` substrings of 'max_len-3 .. max_len' characters from a corpus with the same python/numpy RNG calls
as the reference (random.randint for the length, np.random.randint for the offset), so seeded runs pick the same text.
"""
import json
import random
import re

import numpy as np
import torch

from ..utils import string_utils


class TextData:
    def __init__(self, textfile="data/lotr.txt", char_set_path="", batch_size=1, max_len=20, words=False, characterBalance=False,
                 hardsplit_newline=False):
        if words or characterBalance or hardsplit_newline:
            raise NotImplementedError("word / character-balanced text sampling is not used by the shipped GAN configs")
        with open(textfile) as f:
            self.text = re.sub(r"\s+", " ", f.read())
        self.char_to_idx = None
        if len(char_set_path) > 0:
            with open(char_set_path) as f:
                self.char_to_idx = json.load(f)["char_to_idx"]
        self.batch_size = batch_size
        self.max_len = max_len
        self.min_len = max(max_len - 3, 1)

    def getInstance(self):
        labels, lengths, gt = [], [], []
        for _ in range(self.batch_size):
            length = random.randint(self.min_len, self.max_len)
            idx = np.random.randint(0, len(self.text) - length)
            text = self.text[idx:idx + length]
            assert len(text) > 0
            if text == " ":
                text = self.text[idx + 1]
            gt.append(text)
            if self.char_to_idx is not None:
                l = string_utils.str2label_single(text, self.char_to_idx)
                labels.append(l)
                lengths.append(len(l))
        if self.char_to_idx is None:
            return {"gt": gt, "image": None}
        lengths = torch.IntTensor(lengths)
        width = int(lengths.max())
        mat = np.stack([np.pad(l, (0, width - l.shape[0]), "constant") for l in labels], axis=1)
        return {"label": torch.from_numpy(mat.astype(np.int32)), "label_lengths": lengths, "gt": gt, "image": None}
