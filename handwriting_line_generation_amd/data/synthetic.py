"""Synthetic author-grouped batches with the schema of the reference's collate
(datasets/author_hw_dataset.py:27-112): image [B,1,64,W] in [-1,1] (pad value -1), label [L,B] int32 (pad 0),
label_lengths IntTensor[B], gt list[str], spaced_label None, a_batch_size, author, name.
Used by bench.py, the smoke test and the parity tests; there is no network / dataset here.
"""
import json

import numpy as np
import torch


class SyntheticAuthorDataset:
    def __init__(self, char_file, batch_size, a_batch_size, width=512, label_len=30, num_batches=1000, seed=100, min_width=None, maxlen=40):
        with open(char_file) as f:
            cs = json.load(f)
        self.idx_to_char = {int(k): v for k, v in cs["idx_to_char"].items()}
        self.char_to_idx = cs["char_to_idx"]
        self.num_class = len(self.idx_to_char) + 1
        self.batch_size, self.a_batch_size = batch_size, a_batch_size
        self.width, self.min_width, self.label_len = width, min_width, label_len
        self.num_batches, self.seed, self._maxlen = num_batches, seed, maxlen

    def max_len(self):
        return self._maxlen

    def __len__(self):
        return self.num_batches

    def batch(self, step):
        g = torch.Generator().manual_seed(self.seed + step)
        B = self.batch_size * self.a_batch_size
        if self.min_width is None:
            W = self.width
            image = torch.rand(B, 1, 64, W, generator=g) * 2 - 1
        else:   # variable widths (multiples of 8), padded with -1 to the widest line of the batch
            ws = (torch.randint(self.min_width // 8, self.width // 8 + 1, (B,), generator=g) * 8).tolist()
            W = max(ws)
            image = torch.full((B, 1, 64, W), -1.0)
            for b, w in enumerate(ws):
                image[b, :, :, :w] = torch.rand(1, 64, w, generator=g) * 2 - 1
        label = torch.randint(1, self.num_class, (self.label_len, B), generator=g, dtype=torch.int32)
        gt = ["".join(self.idx_to_char[int(c)] for c in label[:, b]) for b in range(B)]
        return {"image": image, "label": label, "label_lengths": torch.IntTensor([self.label_len] * B), "gt": gt, "spaced_label": None,
                "a_batch_size": self.a_batch_size, "author": ["a%d" % (i // self.a_batch_size) for i in range(B)],
                "name": ["syn%d_%d" % (step, i) for i in range(B)]}


class SyntheticLoader:
    """stands in for torch DataLoader: `.batch_size`, `.dataset`, iterable over instance dicts"""

    def __init__(self, dataset, rank=0, world=1):
        self.dataset = dataset
        self.batch_size = dataset.batch_size
        self.rank, self.world = rank, world
        self._resident = None

    def __len__(self):
        return len(self.dataset)

    def make_resident(self, n, device):
        """Build the first `n` batches now and keep their image/label tensors in HBM (what DataLoader workers + pinned prefetch
        deliver in a real run); iteration then hands out those instances, wrapping around after `n`."""
        from .. import ops
        self._resident = []
        for step in range(n):
            inst = self.dataset.batch(step * self.world + self.rank)
            inst["image"] = ops.h2d(inst["image"], device)
            inst["label"] = ops.h2d(inst["label"], device)
            self._resident.append(inst)

    def __iter__(self):
        if self._resident is not None:
            for inst in self._resident:
                yield inst
            return
        for step in range(len(self.dataset)):
            yield self.dataset.batch(step * self.world + self.rank)   # disjoint author shards per rank


def write_synthetic_corpus(path, char_file, n_chars=200000, seed=7):
    """a text file over the char set with word-like spacing, for TextData when no corpus ships"""
    with open(char_file) as f:
        chars = [c for c in json.load(f)["char_to_idx"] if c.strip() and c.isprintable()]
    rs = np.random.RandomState(seed)
    out = []
    n = 0
    while n < n_chars:
        w = "".join(rs.choice(chars, size=rs.randint(2, 9)))
        out.append(w)
        n += len(w) + 1
    with open(path, "w") as f:
        f.write(" ".join(out))
