"""Known answers from the reference's own pure-Python data code (build container only):
  * collate of author items (datasets/author_hw_dataset.py:27-112 and datasets/author_rimeslines_dataset.py:27-112) on the fabricated items
    of oracle/collate_items.py. The RIMES copy builds its buffers with torch.full(size, -1), which current torch types int64 (the image
    would be truncated to integers); under the torch of the reference's time an integer fill made a float tensor, so it is called with
    torch.full pinned to float32.
  * utils/parseRIMESlines.getLineBoundaries on a fabricated annotation file, and the item index (`lineIndex`, `max_char_len`, `author_list`)
    the unmodified AuthorRIMESLinesDataset constructor builds from it (train: all pairs of an author's lines at a_batch_size 2, otherwise
    consecutive groups + a topped-up remainder).
  * utils/parseIAM.getLineBoundaries (:88-135) on every form XML of the fabricated IAM directory of oracle/collate_items.fake_iam, and the
    item index (`lineIndex`, `max_char_len`, `author_list`, per-author line lists) the unmodified AuthorHWDataset constructor
    (datasets/author_hw_dataset.py:115-297) builds from it for train / valid / test at a_batch_size 2 and 3 and with `short`.
  * (round 5) what `__getitem__` of both dataset classes hands to OpenCV: the unmodified classes run with a RECORDING `cv2` stand-in
    (oracle/cv2_recorder.py: cv2 is not installed, so pixel vectors cannot exist) on the fabricated IAM / RIMES directories, with and
    without the affine augmentation, at a max_width that exercises both resize branches: per item the ordered call list - imread path +
    flag, resize source shape / fx / fy / interpolation / resulting size, warpAffine matrix / dsize / flags / border - the item's image
    shape, labels and names, and the state of numpy's global RNG after the item (the augmentation's draws).
-> tests/golden/collate.npz, tests/golden/rimes_index.json, tests/golden/iam_index.json, tests/golden/getitem_calls.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    import ref_bootstrap
    ref_bootstrap.bootstrap()
    import torch
    from oracle import collate_items
    from datasets import author_hw_dataset as ref_iam
    from datasets import author_rimeslines_dataset as ref_rimes
    out = {}
    cases = {"iam_plain": (ref_iam.collate, {}), "iam_extras": (ref_iam.collate, dict(extras=True, spaced=True, widths=(41, 33))),
             "rimes_plain": (ref_rimes.collate, dict(seed=4, widths=(29, 64, 30, 47))), "single": (ref_iam.collate, dict(widths=(25,)))}
    for name, (fn, kw) in cases.items():
        full = torch.full
        if fn is ref_rimes.collate:
            torch.full = lambda size, fill, **k: full(size, float(fill), **k)
        try:
            res = fn(collate_items.items(**kw))
        finally:
            torch.full = full
        for k, v in res.items():
            if torch.is_tensor(v):
                out["%s/%s" % (name, k)] = v.numpy()
            else:
                out["%s/%s" % (name, k)] = np.array(json.dumps(v))
    np.savez_compressed(os.path.join(GOLD, "collate.npz"), **out)
    print("collate.npz", os.path.getsize(os.path.join(GOLD, "collate.npz")) // 1024, "KiB", sorted(out)[:6])

    work = "/tmp/hwg_golden_rimes"
    os.makedirs(work, exist_ok=True)
    xml_text = collate_items.rimes_xml()
    for fn in ("lines_training_2011.xml", "lines_eval_2011_annotated.xml"):
        with open(os.path.join(work, fn), "w") as f:
            f.write(xml_text)
    from utils.parseRIMESlines import getLineBoundaries
    pages = getLineBoundaries(os.path.join(work, "lines_training_2011.xml"))
    idx = {"pages": {k: [[img, list(b), t] for img, b, t in v] for k, v in pages.items()}, "index": {}}
    char_file = os.path.join(ROOT, "handwriting_line_generation_amd", "data", "RIMES_characterset_lines.json")
    for split in ("train", "valid"):
        for A, extra in ((2, {}), (3, {}), (2, {"short": 1}), (3, {"short": 1})):
            ds = ref_rimes.AuthorRIMESLinesDataset(work, split, dict({"img_height": 64, "a_batch_size": A, "char_file": char_file, "max_width": 1300}, **extra))
            idx["index"]["%s_a%d%s" % (split, A, "_short" if extra else "")] = {"lineIndex": [[a, list(l)] for a, l in ds.lineIndex], "max_char_len": ds.max_char_len,
                                                    "author_list": ds.author_list, "len": len(ds)}
    with open(os.path.join(GOLD, "rimes_index.json"), "w") as f:
        json.dump(idx, f, separators=(",", ":"))
    print("rimes_index.json", {k: v["len"] for k, v in idx["index"].items()})

    # IAM: the reference opens data/sets.json relative to the working directory, so it is run from a scratch directory holding one
    import shutil
    work = "/tmp/hwg_golden_iam"
    shutil.rmtree(work, ignore_errors=True)
    root = os.path.join(work, "iam")
    os.makedirs(os.path.join(work, "data"))
    pages, sets = collate_items.fake_iam(root, with_images=False)
    shutil.copy(os.path.join(root, "sets.json"), os.path.join(work, "data", "sets.json"))
    from utils import parseIAM
    idx = {"sets": sets, "pages": {}, "index": {}}
    for name in pages:
        lines, writer = parseIAM.getLineBoundaries(os.path.join(root, "xmls", name + ".xml"))
        idx["pages"][name] = {"writer": writer, "lines": [[list(b), t] for b, t in lines]}
    char_file = os.path.join(ROOT, "handwriting_line_generation_amd", "data", "IAM_char_set.json")
    cwd = os.getcwd()
    os.chdir(work)
    try:
        for split in ("train", "valid", "test"):
            for A, extra in ((2, {}), (3, {}), (2, {"short": 1})):
                cfg = dict({"img_height": 64, "a_batch_size": A, "char_file": char_file, "max_width": 1400}, **extra)
                ds = ref_iam.AuthorHWDataset(root, split, cfg)
                key = "%s_a%d%s" % (split, A, "_short" if extra else "")
                idx["index"][key] = {"lineIndex": [[a, list(l)] for a, l in ds.lineIndex], "max_char_len": ds.max_char_len, "author_list": ds.author_list,
                                     "len": len(ds), "authors": {a: [[os.path.relpath(pth, root), list(b), t] for pth, b, t in v] for a, v in ds.authors.items()}}
    finally:
        os.chdir(cwd)
    with open(os.path.join(GOLD, "iam_index.json"), "w") as f:
        json.dump(idx, f, separators=(",", ":"))
    print("iam_index.json", {k: v["len"] for k, v in idx["index"].items()})


GETITEM_CASES = {
    # name: (dataset, split, config overrides)  - max_width 300 makes some lines hit the width limit (second resize branch, strech clamp)
    "iam_train_affine": ("iam", "train", {"augmentation": "affine", "max_width": 300}),
    "iam_train_plain": ("iam", "train", {"max_width": 1400}),
    "iam_valid_a3": ("iam", "valid", {"a_batch_size": 3, "max_width": 260}),
    "rimes_train_affine": ("rimes", "train", {"augmentation": "affine", "max_width": 700}),
    "rimes_valid_plain": ("rimes", "valid", {"max_width": 500}),
}


def getitem_calls():
    import shutil
    import ref_bootstrap
    ref_bootstrap.bootstrap()
    from oracle import collate_items, cv2_recorder
    from datasets import author_hw_dataset as ref_iam
    from datasets import author_rimeslines_dataset as ref_rimes
    from utils import augmentation as ref_aug
    work = "/tmp/hwg_golden_getitem"
    shutil.rmtree(work, ignore_errors=True)
    roots = {"iam": os.path.join(work, "iam"), "rimes": os.path.join(work, "rimes")}
    os.makedirs(os.path.join(work, "data"))
    collate_items.fake_iam(roots["iam"], with_images=True)
    shutil.copy(os.path.join(roots["iam"], "sets.json"), os.path.join(work, "data", "sets.json"))
    collate_items.fake_rimes(roots["rimes"])
    out = {}
    cwd = os.getcwd()
    os.chdir(work)
    try:
        for name, (which, split, over) in GETITEM_CASES.items():
            rec = cv2_recorder.RecordingCv2(roots[which])
            ref_iam.cv2 = ref_rimes.cv2 = ref_aug.cv2 = rec
            char_file = os.path.join(ROOT, "handwriting_line_generation_amd", "data", "IAM_char_set.json" if which == "iam" else "RIMES_characterset_lines.json")
            cfg = dict({"img_height": 64, "a_batch_size": 2, "char_file": char_file}, **over)
            ds = (ref_iam.AuthorHWDataset if which == "iam" else ref_rimes.AuthorRIMESLinesDataset)(roots[which], split, cfg)
            items = []
            for idx in range(min(len(ds), 10)):
                np.random.seed(4000 + idx)
                del rec.calls[:]
                it = ds[idx]
                items.append({"idx": idx, "calls": [list(c) for c in rec.calls], "image_shape": list(it["image"].shape), "gt": it["gt"],
                              "label": it["label"].tolist(), "label_lengths": it["label_lengths"].tolist(), "name": it["name"], "author": it["author"],
                              "rng_after": collate_items.rng_fingerprint()})
            out[name] = {"dataset": which, "split": split, "config": over, "len": len(ds), "items": items}
            print(name, len(ds), "items; calls of item 0:", [c[0] for c in items[0]["calls"]])
    finally:
        os.chdir(cwd)
    with open(os.path.join(GOLD, "getitem_calls.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("getitem_calls.json", os.path.getsize(os.path.join(GOLD, "getitem_calls.json")) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "getitem":
        getitem_calls()
    else:
        main()
        getitem_calls()
