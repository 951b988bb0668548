"""Author-grouped line batches from IAM-style data on disk (reference: datasets/author_hw_dataset.py:27-591, utils/parseIAM.py:11-70,
utils/augmentation.py:61-72), without OpenCV: decoding, cropping, height normalisation and the affine augmentation go through PIL / numpy.

Directory layout (the reference's): `<data_dir>/forms/<page>.png`, `<data_dir>/xmls/<page>.xml` (IAM form XML: writer id, lines with
word/cmp boxes), and `data/sets.json` = {"train": [pages], "valid": [...], "test": [...]} (looked up next to the char set file or in the
working directory). One dataset item = `a_batch_size` lines of one author; `collate` pads `batch_size` such items to a common width
(-1) and label length (0) - the instance dict `HWWithStyleTrainer.run_gen` consumes.

The interpolation ARITHMETIC is PIL's, not OpenCV's, and is NOT pinned to the reference (cv2 is absent from the build image, so no reference
pixel exists): PIL bicubic (a = -0.5, antialiased) for the height normalisation where cv2.INTER_CUBIC uses a = -0.75 without antialiasing,
PIL bilinear for the affine warp. What IS pinned to the reference (tests/golden/getitem_calls.json): which pixels are decoded and cropped,
the resize / warp geometry handed to the interpolator, output sizes, normalisation 1 - p/128, padding, label encoding, grouping of an
author's lines and the draw order of the augmentation parameters. `fg_mask` (Otsu threshold + dilation through cv2 when `fg_masks_dir` is
set, reference :407-420) is not produced: the key is absent from the items, as it is in the reference without `fg_masks_dir`; nothing on the
loss path reads it (INTEGRATION.md lists the difference).
"""
import json
import math
import os
import xml.etree.ElementTree as ET
from collections import defaultdict
from xml.sax.saxutils import unescape as _unescape

import numpy as np
import torch

from ..utils import string_utils

PADDING_CONSTANT = -1


def parse_iam_xml(path):
    """-> ([([y0, y1, x0, x1], text), ...], writer id): line boxes from the component boxes, padded to the page's mean line height
    (parseIAM.getWordAndLineBoundaries / getLineBoundaries)"""
    root = ET.parse(path).getroot()
    writer = root.attrib["writer-id"]
    lines, all_h = [], 0
    for line in root.findall("./handwritten-part/line"):
        text = _unescape(line.attrib["text"]).replace("&quot;", '"')
        min_x = min_y = 99999999
        max_x = max_y = -1
        for word in line.findall("word"):
            for cmp in word.findall("cmp"):
                x, y, w, h = (int(cmp.attrib[k]) for k in ("x", "y", "width", "height"))
                max_x, min_x = max(max_x, x + w), min(min_x, x)
                max_y, min_y = max(max_y, y + h), min(min_y, y)
        lines.append(([min_y, max_y + 1, min_x, max_x + 1], text))
        all_h += 1 + max_y - min_y
    mean_h = all_h / max(len(lines), 1)
    out = []
    for b, text in lines:
        diff = mean_h - (b[1] - b[0])
        if diff > 0:
            b[0] -= diff / 2
            b[1] += diff / 2
        b[2] -= mean_h / 4
        b[3] += mean_h / 4
        out.append(([round(v) for v in b], text))
    return out, writer


def _read_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"))


def _resize(img, percent):
    from PIL import Image
    w, h = max(int(round(img.shape[1] * percent)), 1), max(int(round(img.shape[0] * percent)), 1)
    return np.asarray(Image.fromarray(img).resize((w, h), Image.BICUBIC))


def affine_trans(img, skew, strech):
    """horizontal stretch + shear around the mid-line (augmentation.affine_trans): x' = strech*x + tan(skew)*(y - h/2), white border.
    cv2.warpAffine maps pixel INDICES (dst(x', y') = src(M^-1 (x', y'))); PIL evaluates its destination -> source map at pixel CENTRES
    (x' + 0.5, y' + 0.5) and takes 0.5 off the result, so the constant term is moved by 0.5 (1 - A - B) to sample the same source positions."""
    from PIL import Image
    m = math.tan(skew)
    h = img.shape[0] / 2
    size = (int(img.shape[1] * strech), img.shape[0])
    A, B, C = 1.0 / strech, -m / strech, h * m / strech      # PIL wants the destination -> source map (index convention so far)
    coeffs = (A, B, C + 0.5 * (1.0 - A - B), 0.0, 1.0, 0.0)
    return np.asarray(Image.fromarray(img).transform(size, Image.AFFINE, coeffs, resample=Image.BILINEAR, fillcolor=255))


class AuthorHWDataset(torch.utils.data.Dataset):
    def __init__(self, dirPath, split, config):
        split = config.get("split", split)
        self.img_height = config["img_height"]
        self.batch_size = config["a_batch_size"]          # lines per item (the reference's naming)
        self.no_spaces = config.get("no_spaces", False)
        self.max_width = config.get("max_width", 3000)
        for key in ("triplet", "style_loc", "spaced_loc", "fg_masks_dir", "include_stroke_aug", "remove_bg"):
            if key == "fg_masks_dir":
                continue     # only read by the (unreachable) no_bg_loss branch
            if config.get(key):
                raise NotImplementedError("data option %r is not used by the shipped GAN configs" % key)
        sets = None
        for cand in (os.path.join("data", "sets.json"), os.path.join(os.path.dirname(config["char_file"]), "sets.json"), os.path.join(dirPath, "sets.json")):
            if os.path.exists(cand):
                sets = json.load(open(cand))
                break
        if sets is None:
            raise FileNotFoundError("sets.json (train/valid/test page lists) not found in data/, next to the char set or in %s" % dirPath)
        pages = []
        for s in (split if isinstance(split, (list, tuple)) else [split]):
            pages += sets[s]
        self.authors = defaultdict(list)
        self.max_char_len = 0
        for name in pages:
            lines, author = parse_iam_xml(os.path.join(dirPath, "xmls", name + ".xml"))
            self.max_char_len = max([self.max_char_len] + [len(t) for _, t in lines])
            self.authors[author] += [(os.path.join(dirPath, "forms", name + ".png"),) + l for l in lines]
        self.author_list = sorted(self.authors)
        # items: consecutive groups of a_batch_size lines of one author; the left-over lines are topped up with the author's first ones
        self.lineIndex = []
        short = config.get("short", False)
        # `short` (debug option): the reference tests its loop variable `i` AFTER the loop (author_hw_dataset.py:176-180), where it still holds
        # whatever the last loop that ran left in it - this author's group loop, or, for an author with fewer than a_batch_size lines, the
        # previous author's top-up loops. Kept: the item index is part of the drop-in surface (tests/golden/iam_index.json).
        i = None
        for author, lines in self.authors.items():
            for i in range(len(lines) // self.batch_size):
                self.lineIndex.append((author, [self.batch_size * i + n for n in range(self.batch_size)]))
                if short and i >= short:
                    break
            if short:
                if i is None:
                    raise NameError("name 'i' is not defined (the reference fails the same way: `short` with a first author of fewer than a_batch_size lines)")
                if i >= short:
                    continue
            leftover = len(lines) % self.batch_size
            fill = self.batch_size - leftover
            last = []
            for i in range(fill):
                last.append(i)
            for i in range(leftover):
                last.append(len(lines) - (1 + i))
            self.lineIndex.append((author, last))
        if config.get("overfit"):
            self.lineIndex = self.lineIndex[:10]
        with open(config["char_file"]) as f:
            self.char_to_idx = json.load(f)["char_to_idx"]
        self.augmentation = config.get("augmentation")
        # the reference applies `affine_trans` when the string contains "affine" and, for ANY other non-None value, Tensmeyer brightness +
        # grid-distortion warping (datasets/author_hw_dataset.py:427-433, author_rimeslines_dataset.py:428-434) - OpenCV code outside this
        # package's scope: refuse instead of silently training un-augmented
        if self.augmentation is not None and not (isinstance(self.augmentation, str) and "affine" in self.augmentation and "normalization" not in self.augmentation):
            raise NotImplementedError("data option augmentation=%r: only None and 'affine' are implemented (the reference's 'warp' / brightness / "
                                      "'normalization' augmentations are OpenCV code outside the hot-path scope)" % (self.augmentation,))
        self.max_strech = 0.4
        self.max_rot_rad = 45 / 180 * math.pi
        self._pages = {}

    def __len__(self):
        return len(self.lineIndex)

    def max_len(self):
        return self.max_char_len

    def estimated_width(self, idx):
        """width (px) of item `idx` after height normalisation, from the line boxes alone (no image is read): what width bucketing sorts by"""
        author, lines = self.lineIndex[idx]
        w = 0
        for line in lines:
            if line >= len(self.authors[author]):
                line = (line + 37) % len(self.authors[author])
            lb = self.authors[author][line][1]
            w = max(w, min((lb[3] - lb[2]) * self.img_height / max(lb[1] - lb[0], 1), self.max_width))
        return w

    def _page(self, path):
        if path not in self._pages:
            if len(self._pages) > 8:
                self._pages.clear()
            self._pages[path] = _read_gray(path)
        return self._pages[path]

    def _fit(self, img):
        """height -> img_height (never wider than max_width; short results are centred on white), as author_hw_dataset.py:386-404"""
        if img.shape[0] != self.img_height:
            percent = float(self.img_height) / img.shape[0]
            if img.shape[1] * percent > self.max_width:
                percent = self.max_width / img.shape[1]
            img = _resize(img, percent)
        elif img.shape[1] > self.max_width:
            img = _resize(img, self.max_width / img.shape[1])
        if img.shape[0] < self.img_height:
            diff = self.img_height - img.shape[0]
            img = np.pad(img, ((diff // 2, diff // 2 + diff % 2), (0, 0)), "constant", constant_values=255)
        elif img.shape[0] > self.img_height:
            img = img[:self.img_height]
        return img

    def __getitem__(self, idx):
        if self.augmentation == "affine":     # one draw per item, shared by the author's lines (the reference's order: strech, then skew)
            strech = (self.max_strech * 2) * np.random.random() - self.max_strech + 1
            skew = (self.max_rot_rad * 2) * np.random.random() - self.max_rot_rad
        author, lines = self.lineIndex[idx]
        images = []
        for line in lines:
            if line >= len(self.authors[author]):
                line = (line + 37) % len(self.authors[author])
            path, lb, gt = self.authors[author][line]
            if self.no_spaces:
                gt = gt.replace(" ", "")
            page = self._page(path)
            img = self._fit(page[max(lb[0], 0):lb[1], max(lb[2], 0):lb[3]])
            if self.augmentation == "affine" and img.shape[1] * strech > self.max_width:
                strech = self.max_width / img.shape[1]
            images.append((line, gt, img))
        batch = []
        for line, gt, img in images:
            if isinstance(self.augmentation, str) and "affine" in self.augmentation:
                img = affine_trans(img, skew, strech)
            if len(gt) == 0:
                return None
            batch.append({"image": 1.0 - img.astype(np.float32)[..., None] / 128.0, "gt": gt,
                          "gt_label": string_utils.str2label_single(gt, self.char_to_idx), "name": "%s_%d" % (author, line)})
        dim1 = max(b["image"].shape[1] for b in batch)
        images_t = np.full((len(batch), self.img_height, dim1, 1), PADDING_CONSTANT, dtype=np.float32)
        max_label = max(b["gt_label"].shape[0] for b in batch)
        labels = np.zeros((max_label, len(batch)), dtype=np.int32)
        for i, b in enumerate(batch):
            images_t[i, :, :b["image"].shape[1], :] = b["image"]
            labels[:b["gt_label"].shape[0], i] = b["gt_label"]
        images_t = torch.from_numpy(images_t.transpose(0, 3, 1, 2))
        return {"image": images_t, "mask": None, "top_and_bottom": None, "center_line": None, "label": torch.from_numpy(labels), "style": None,
                "label_lengths": torch.IntTensor([b["gt_label"].shape[0] for b in batch]), "gt": [b["gt"] for b in batch], "spaced_label": None,
                "author": [author] * len(batch), "author_idx": [self.author_list.index(author)] * len(batch), "name": [b["name"] for b in batch]}


def collate(batch):
    """items -> instance dict (author_hw_dataset.py:27-112; the RIMES copy author_rimeslines_dataset.py:27-112 is the same function):
    images / masks padded with -1 to the widest item, labels with 0 to the longest, foreground masks and per-column line geometry
    (top_and_bottom 0, center_line H/2) padded likewise when the items carry them. One item is handed through as the batch."""
    if len(batch) == 1:
        batch[0]["a_batch_size"] = batch[0]["image"].size(0)
        return batch[0]
    batch = [b for b in batch if b is not None]
    A = len(batch[0]["gt"])
    n = len(batch) * A
    dim1, dim2 = batch[0]["image"].shape[1], batch[0]["image"].shape[2]
    dim3 = max(b["image"].shape[3] for b in batch)
    max_label = max(b["label"].size(0) for b in batch)
    images = torch.full((n, dim1, dim2, dim3), float(PADDING_CONSTANT))
    labels = torch.zeros((max_label, n), dtype=torch.int32)
    has = lambda key: batch[0].get(key) is not None          # noqa: E731
    masks = torch.full((n, dim1, dim2, dim3), float(PADDING_CONSTANT)) if has("mask") else None
    tab = torch.zeros((n, 2, dim3)) if has("top_and_bottom") else None
    centre = torch.full((n, dim3), dim2 / 2) if has("center_line") else None
    fg = torch.zeros((n, 1, dim2, dim3)) if "fg_mask" in batch[0] else None
    changed = torch.full((n, dim1, dim2, dim3), float(PADDING_CONSTANT)) if "changed_image" in batch[0] else None
    spaced = None
    if has("spaced_label"):
        spaced = torch.zeros((max(b["spaced_label"].size(0) for b in batch), n), dtype=torch.int32)
    for i, b in enumerate(batch):
        rows, w = slice(i * A, (i + 1) * A), b["image"].shape[3]
        images[rows, :, :, :w] = b["image"]
        labels[:b["label"].size(0), rows] = b["label"]
        if masks is not None:
            masks[rows, :, :, :w] = b["mask"]
        if tab is not None:
            tab[rows, :, :w] = b["top_and_bottom"]
        if centre is not None:
            centre[rows, :w] = b["center_line"]
        if fg is not None:
            fg[rows, :, :, :w] = b["fg_mask"]
        if changed is not None:
            changed[rows, :, :, :w] = b["changed_image"]
        if spaced is not None:
            spaced[:b["spaced_label"].size(0), rows] = b["spaced_label"]
    out = {"image": images, "mask": masks, "top_and_bottom": tab, "center_line": centre, "label": labels,
           "style": None if batch[0].get("style") is None else torch.cat([b["style"] for b in batch], dim=0),
           "label_lengths": torch.cat([b["label_lengths"] for b in batch], dim=0), "gt": [l for b in batch for l in b["gt"]], "spaced_label": spaced,
           "author": [l for b in batch for l in b["author"]], "author_idx": [l for b in batch for l in b["author_idx"]],
           "name": [l for b in batch for l in b["name"]], "a_batch_size": A}
    if fg is not None:
        out["fg_mask"] = fg
    if changed is not None:
        out["changed_image"] = changed
    return out


def pad_width(instance, multiple):
    """Width bucketing (not in the reference; off unless data_loader.width_bucket is set): widen the batch image with padding columns (-1,
    what collate pads with anyway) to the next multiple of `multiple`. Real line widths take hundreds of distinct values; every distinct
    padded width is a new set of layer geometries (conv plans, workspaces, Winograd / direct choice) for the kernels, so a training run
    would keep planning. With buckets the number of distinct widths is max_width / multiple. The extra columns are seen by the networks
    exactly like the reference's own padding columns (they take part in GroupNorm / InstanceNorm statistics, as padding does there)."""
    img = instance["image"]
    w = img.shape[3]
    target = -(-w // multiple) * multiple
    if target != w:
        for key, fill in (("image", float(PADDING_CONSTANT)), ("mask", float(PADDING_CONSTANT)), ("fg_mask", 0.0), ("changed_image", float(PADDING_CONSTANT))):
            t = instance.get(key)
            if t is not None:
                instance[key] = torch.nn.functional.pad(t, (0, target - w), value=fill)
    return instance


class ShardedLoader:
    """DataLoader over the items of one data-parallel rank: the epoch's (seeded) item order is cut into batches of `batch_size` items and
    rank r takes batches r, r + N, ... - every rank sees different authors, all ranks take the same number of steps per epoch. Batches are
    prefetched by `num_workers` torch DataLoader workers; `.batch_size` / `.dataset` as the trainer expects.

    `width_bucket` (pixels, 0 = off = the reference's behaviour): items are grouped by estimated line width before they are cut into
    batches - inside windows of `bucket_window` batches of the shuffled order, so the epoch stays shuffled - and the collated image is
    padded to a multiple of `width_bucket` (pad_width). Less padding per batch (an item of 300 px no longer shares a batch with one of
    1200 px) and a bounded number of distinct widths for the kernels' plan caches."""

    def __init__(self, dataset, batch_size, shuffle, num_workers, rank=0, world=1, seed=0, width_bucket=0, bucket_window=16):
        self.dataset, self.batch_size, self.shuffle, self.num_workers = dataset, batch_size, shuffle, num_workers
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        self.width_bucket, self.bucket_window = int(width_bucket or 0), bucket_window

    def _batches(self):
        n = len(self.dataset)
        order = np.random.RandomState(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)
        if self.width_bucket and hasattr(self.dataset, "estimated_width"):
            span = self.batch_size * self.bucket_window
            widths = np.array([self.dataset.estimated_width(int(i)) for i in order])
            order = np.concatenate([order[a:a + span][np.argsort(widths[a:a + span], kind="stable")] for a in range(0, n, span)])
        full = [order[i:i + self.batch_size].tolist() for i in range(0, n - self.batch_size + 1, self.batch_size)] or [order.tolist()]
        if self.width_bucket and self.shuffle:      # the batches of a window come out narrow-to-wide: shuffle the batches, not their contents
            full = [full[i] for i in np.random.RandomState(self.seed + self.epoch + 7919).permutation(len(full))]
        usable = len(full) // self.world * self.world or len(full)
        return full[:usable][self.rank::self.world] or full[:1]

    def __len__(self):
        return len(self._batches())

    def _collate(self, items):
        inst = collate(items)
        return pad_width(inst, self.width_bucket) if self.width_bucket else inst

    def __iter__(self):
        batches = self._batches()
        self.epoch += 1
        return iter(torch.utils.data.DataLoader(self.dataset, batch_sampler=batches, num_workers=self.num_workers, collate_fn=self._collate))


DATASETS = {}


def getDataLoader(config, split, rank=0, world=1):
    """data_loader.getDataLoader of the reference (data_loader/data_loaders.py:11-75) for the author-grouped line datasets -> (train, valid).
    `data_loader.width_bucket` (pixels, optional, not in the reference) switches width bucketing on, see ShardedLoader."""
    dl = config["data_loader"]
    if not DATASETS:
        from .author_rimeslines_dataset import AuthorRIMESLinesDataset
        DATASETS.update(AuthorHWDataset=AuthorHWDataset, AuthorRIMESLinesDataset=AuthorRIMESLinesDataset)
    if dl["data_set_name"] not in DATASETS:
        raise NotImplementedError("dataset %r: only the author-grouped IAM / RIMES line datasets have a loader here" % dl["data_set_name"])
    cls = DATASETS[dl["data_set_name"]]
    val = dict(config.get("validation", {}))
    for k, v in dl.items():
        val.setdefault(k, v)
    wb = dl.get("width_bucket", 0)
    if split == "train":
        train = cls(dl["data_dir"], "train", dl)
        valid = cls(dl["data_dir"], "valid", val)
        tl = ShardedLoader(train, dl["batch_size"], dl.get("shuffle", True), dl.get("num_workers", 1), rank, world, width_bucket=wb)
        vl = ShardedLoader(valid, val.get("batch_size", dl["batch_size"]), val.get("shuffle", False), val.get("num_workers", 1), width_bucket=wb) if len(valid) else None
        return tl, vl
    test = cls(dl["data_dir"], split, val)
    return ShardedLoader(test, val.get("batch_size", dl["batch_size"]), False, val.get("num_workers", 1), width_bucket=wb), None
