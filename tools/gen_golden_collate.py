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
-> tests/golden/collate.npz, tests/golden/rimes_index.json, tests/golden/iam_index.json
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


if __name__ == "__main__":
    main()
