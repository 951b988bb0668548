"""Author-grouped RIMES line batches (reference: datasets/author_rimeslines_dataset.py:115-577, utils/parseRIMESlines.py:14-47), without OpenCV.

RIMES has no writer ids: an "author" is a page (a letter), its lines are the `Paragraph/Line` boxes of the annotation file
(`lines_training_2011.xml` for training, `lines_eval_2011_annotated.xml` for validation AND test - the reference validates on the test set and
says so), images under `<data_dir>/images_gray/`. For training with a_batch_size 2 every PAIR of a page's lines is an item
(itertools.combinations, so a page of n lines contributes n(n-1)/2 items); otherwise consecutive groups plus the topped-up remainder, as the
IAM dataset. Decoding, height normalisation, affine augmentation and item assembly are the IAM dataset's (data/author_hw_dataset.py)."""
import itertools
import json
import math
import os
import xml.etree.ElementTree as ET
from collections import defaultdict
from xml.sax.saxutils import unescape as _unescape

from .author_hw_dataset import AuthorHWDataset


def parse_rimes_xml(path):
    """-> {page image name: [(image name, [y0, y1, x0, x1], text), ...]}: line boxes padded to the page's mean line height (short lines only)
    and by a quarter of it on both sides (parseRIMESlines.getLineBoundaries)"""
    root = ET.parse(path).getroot()
    pages = defaultdict(list)
    for page in root.findall("SinglePage"):
        image = page.attrib["FileName"]
        image = image[image.index("/") + 1:]
        lines, all_h = [], 0
        for line in page.findall("Paragraph/Line"):
            text = _unescape(line.attrib["Value"]).replace("&quot;", '"').replace("&apos;", "'")
            top, bot, left, right = (int(line.attrib[k]) for k in ("Top", "Bottom", "Left", "Right"))
            lines.append(([top, bot + 1, left, right + 1], text))
            all_h += 1 + bot - top
        mean_h = all_h / len(lines)
        for b, text in lines:
            diff = mean_h - (b[1] - b[0])
            if diff > 0:
                b[0] -= diff / 2
                b[1] += diff / 2
            b[2] -= mean_h / 4
            b[3] += mean_h / 4
            pages[image].append((image, [round(v) for v in b], text))
    return pages


class AuthorRIMESLinesDataset(AuthorHWDataset):
    def __init__(self, dirPath, split, config):   # noqa: super().__init__ is the IAM constructor (different annotation format): not called
        split = config.get("split", split)
        xml = "lines_eval_2011_annotated.xml" if split in ("test", "valid") else "lines_training_2011.xml"
        self.img_height = config["img_height"]
        self.batch_size = config["a_batch_size"]
        self.no_spaces = config.get("no_spaces", False)
        self.max_width = config.get("max_width", 3000)
        for key in ("triplet", "style_loc", "spaced_loc", "include_stroke_aug", "remove_bg", "only_author", "skip_author"):
            if config.get(key):
                raise NotImplementedError("data option %r is not used by the shipped GAN configs" % key)
        pages = parse_rimes_xml(os.path.join(dirPath, xml))
        self.authors = {a: [(os.path.join(dirPath, "images_gray", img), lb, gt) for img, lb, gt in lines] for a, lines in pages.items()}
        self.author_list = sorted(self.authors)
        short = config.get("short", False)
        self.lineIndex = []
        self.max_char_len = 0
        i = None
        for author, lines in self.authors.items():
            self.max_char_len = max(self.max_char_len, max(len(l[2]) for l in lines))
            if split == "train" and self.batch_size == 2:
                combs = list(itertools.combinations(range(len(lines)), 2))
                if short:
                    combs = combs[:short]
                self.lineIndex += [(author, list(c)) for c in combs]
                continue
            # `short`: the reference tests its loop variable after the loop, where it may be stale (see AuthorHWDataset)
            for i in range(len(lines) // self.batch_size):
                self.lineIndex.append((author, [self.batch_size * i + n for n in range(self.batch_size)]))
                if short and i >= short:
                    break
            if short:
                if i is None:
                    raise NameError("name 'i' is not defined (the reference fails the same way)")
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
