"""Test infrastructure: fabricated dataset items (what AuthorHWDataset / AuthorRIMESLinesDataset.__getitem__ return, reference:
datasets/author_hw_dataset.py:546-571) for the collate known-answer test, and a fabricated RIMES annotation file for the parser / item
index known-answer test. tools/gen_golden_collate.py feeds them to the reference's own collate / parser / dataset constructor."""
import numpy as np
import torch


def items(seed=3, widths=(37, 52, 44), A=2, H=8, extras=False, spaced=False):
    """one item per entry of `widths`: A lines of one author, image [A,1,H,W] in [-1,1], labels of ragged length (0-padded per item)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, W in enumerate(widths):
        L = 3 + 2 * i
        lens = [L - (a % 2) for a in range(A)]
        label = torch.zeros(L, A, dtype=torch.int32)
        for a, n in enumerate(lens):
            label[:n, a] = torch.randint(1, 80, (n,), generator=g, dtype=torch.int32)
        it = {"image": torch.rand(A, 1, H, W, generator=g) * 2 - 1,
              "mask": torch.rand(A, 1, H, W, generator=g).round() * 2 - 1,
              "top_and_bottom": torch.rand(A, 2, W, generator=g) * H,
              "center_line": torch.rand(A, W, generator=g) * H,
              "label": label, "style": None, "label_lengths": torch.IntTensor(lens),
              "gt": ["item%d line%d" % (i, a) for a in range(A)], "spaced_label": None,
              "author": ["w%02d" % i] * A, "author_idx": [i] * A, "name": ["w%02d_%d" % (i, a) for a in range(A)]}
        if spaced:
            S = L + 4
            it["spaced_label"] = torch.randint(0, 80, (S, A), generator=g, dtype=torch.int32)
        if extras:
            it["fg_mask"] = torch.rand(A, 1, H, W, generator=g).round()
            it["changed_image"] = torch.rand(A, 1, H, W, generator=g) * 2 - 1
            it["style"] = torch.randn(A, 16, generator=g)
        out.append(it)
    return out


RIMES_XML = """<?xml version="1.0" encoding="UTF-8"?>
<DocumentList>
%s
</DocumentList>
"""


def rimes_xml(n_pages=4, seed=9):
    """annotation file in the layout of RIMES' lines_training_2011.xml: pages with 1-6 lines of ragged heights and widths; includes
    escaped characters and a page with a single line"""
    rs = np.random.RandomState(seed)
    pages = []
    for p in range(n_pages):
        n_lines = [5, 1, 3, 6, 2, 4][p % 6]
        lines = []
        top = 40
        for l in range(n_lines):
            h = int(rs.randint(28, 60))
            left = int(rs.randint(10, 60))
            w = int(rs.randint(200, 900))
            text = ["Bonjour Monsieur,", "je vous &amp; &quot;prie&quot;", "d&apos;agr&#233;er", "mes salutations", "distingu&#233;es.", "A+"][(p + l) % 6]
            lines.append('      <Line Value="%s" Top="%d" Bottom="%d" Left="%d" Right="%d"/>' % (text, top, top + h, left, left + w))
            top += h + int(rs.randint(5, 25))
        pages.append('  <SinglePage FileName="images_gray/page%03d.png">\n    <Paragraph>\n%s\n    </Paragraph>\n  </SinglePage>' % (p, "\n".join(lines)))
    return RIMES_XML % "\n".join(pages)


def rng_fingerprint():
    """numpy's global RNG state as [position, sum of the key words, the next uniform draw a copy of the state would give]"""
    st = np.random.get_state()
    rs = np.random.RandomState()
    rs.set_state(st)
    return [int(st[2]), int(np.asarray(st[1], dtype=np.uint64).sum()), float(rs.random_sample())]


def fake_rimes(root, n_pages=4, seed=1):
    """fabricated RIMES directory in the reference's layout: `images_gray/page%03d.png` (dark strokes everywhere on white paper, so every line
    crop has ink) + the two annotation files of rimes_xml()"""
    import os
    from PIL import Image, ImageDraw
    os.makedirs(os.path.join(root, "images_gray"), exist_ok=True)
    text = rimes_xml(n_pages)
    for fn in ("lines_training_2011.xml", "lines_eval_2011_annotated.xml"):
        with open(os.path.join(root, fn), "w") as f:
            f.write(text)
    rs = np.random.RandomState(seed)
    for p in range(n_pages):
        img = Image.new("L", (1000, 420), 255)
        dr = ImageDraw.Draw(img)
        for _ in range(300):
            x, y = int(rs.randint(0, 980)), int(rs.randint(0, 400))
            dr.rectangle([x, y, x + int(rs.randint(3, 18)), y + int(rs.randint(6, 30))], fill=int(rs.randint(10, 120)))
        img.save(os.path.join(root, "images_gray", "page%03d.png" % p))


IAM_TEXTS = ["the quick brown", "fox jumps", "over a lazy dog", "pack my box", "with five dozen", "liquor jugs", "sphinx of black quartz",
             "he said &quot;no&quot;", "Tom &amp; Co.", "a", "judge my vow", "it&apos;s 4 o&apos;clock"]


def fake_iam(root, n_pages=6, seed=5, with_images=True):
    """fabricated IAM directory in the reference's layout (`forms/<page>.png`, `xmls/<page>.xml` with writer-id / line / word / cmp boxes,
    `sets.json`): pages of 1-6 lines with ragged line heights (so that both branches of the mean-height growth run), several cmp boxes per
    word, escaped characters, three writers of which one has a single line (the left-over-item quirk), pages shared between splits"""
    import json
    import os
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "forms"), exist_ok=True); os.makedirs(os.path.join(root, "xmls"), exist_ok=True)
    lines_per_page = [3, 1, 5, 4, 6, 2][:n_pages] + [3] * max(0, n_pages - 6)
    writers = ["000", "017", "000", "230", "017", "000"][:n_pages] + ["230"] * max(0, n_pages - 6)
    pages, t = [], 0
    for p in range(n_pages):
        name = "p%02d-%03d" % (p, 7 * p)
        pages.append(name)
        boxes = []
        xml = ['<form id="%s" writer-id="%s"><machine-printed-part/><handwritten-part>' % (name, writers[p])]
        y0 = 30
        for l in range(lines_per_page[p]):
            text = IAM_TEXTS[t % len(IAM_TEXTS)]; t += 1
            h = int(rs.randint(34, 78))
            xml.append('<line id="%s-%02d" text="%s">' % (name, l, text))
            x = int(rs.randint(20, 70))
            plain = text.replace("&quot;", '"').replace("&amp;", "&").replace("&apos;", "'")
            for w in plain.split(" "):
                xml.append('<word id="w" text="%s">' % w.replace("&", "&amp;").replace('"', "&quot;"))
                for c in range(1 + len(w) // 4):                 # several component boxes per word, ragged tops / heights
                    cw = int(rs.randint(14, 40)); dy = int(rs.randint(0, 9)); ch = h - dy - int(rs.randint(0, 7))
                    xml.append('<cmp x="%d" y="%d" width="%d" height="%d"/>' % (x, y0 + dy, cw, ch))
                    boxes.append((x, y0 + dy, cw, ch))
                    x += cw + int(rs.randint(1, 5))
                xml.append("</word>")
                x += int(rs.randint(12, 30))
            xml.append("</line>")
            y0 += h + int(rs.randint(18, 50))
        xml.append("</handwritten-part></form>")
        with open(os.path.join(root, "xmls", name + ".xml"), "w") as f:
            f.write("".join(xml))
        if with_images:
            from PIL import Image, ImageDraw
            W = max(b[0] + b[2] for b in boxes) + 80
            img = Image.new("L", (W, y0 + 40), 255)
            dr = ImageDraw.Draw(img)
            for (x, y, w, h) in boxes:
                dr.rectangle([x, y, x + w, y + h], fill=int(rs.randint(20, 110)))
            img.save(os.path.join(root, "forms", name + ".png"))
    sets = {"train": pages[:4], "valid": pages[4:], "test": pages[3:]}
    with open(os.path.join(root, "sets.json"), "w") as f:
        json.dump(sets, f)
    return pages, sets
