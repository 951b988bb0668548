"""GPU: the command-line entry point end to end - `train.py -c <config> --synthetic` runs BaseTrainer.train() (curriculum lessons, logging,
minor / major checkpoints through the atomic rank-0 writer), and `train.py -r <checkpoint>` resumes it (reference: train.py:21-93,
base/base_trainer.py:141-214, 401-479)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = "cf_IAMslant_noMask_charSpecSingleAppend_GANMedMT_autoAEMoPrcp2tightNewCTCUseGen_balB_hCF0.75_sMG.json"


def _run(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py")] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "train.py %s failed:\n%s\n%s" % (" ".join(args), r.stdout[-2000:], r.stderr[-3000:])
    return r.stdout + r.stderr


def test_train_cli_runs_saves_and_resumes(cuda, tmp_path):
    from handwriting_line_generation_amd.logger import load_checkpoint
    cfg = json.load(open(os.path.join(ROOT, "configs", CFG)))
    tr = cfg["trainer"]
    save_dir = str(tmp_path / "saved")
    tr.update(save_dir=save_dir, save_step=7, save_step_minor=3, log_step=2, val_step=0, print_dir=str(tmp_path / "out"),
              encoder_weights=str(tmp_path / "enc" / "encoder.pth"), text_data=str(tmp_path / "no_such_corpus.txt"))
    cfg["model"]["pretrained_hwr"] = None
    cfg["seed"] = 123
    path = str(tmp_path / CFG)
    json.dump(cfg, open(path, "w"))
    out = _run(["-c", path, "--synthetic", "--iterations", "8"], str(tmp_path))
    ckdir = os.path.join(save_dir, cfg["name"])
    assert os.path.exists(os.path.join(ckdir, "checkpoint-iteration7.pth")) and os.path.exists(os.path.join(ckdir, "checkpoint-latest.pth")), out[-1500:]
    ck = load_checkpoint(os.path.join(ckdir, "checkpoint-iteration7.pth"))
    assert ck["iteration"] == 7 and ck["config"]["name"] == cfg["name"] and "optimizer_discriminator" in ck and ck["rng"]["mode"] == "device"
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.dtype.is_floating_point)
    before = {k: v.clone() for k, v in ck["state_dict"].items()}
    # resume from the major checkpoint: continues at iteration 8 and writes the next major save at 14
    out2 = _run(["-r", os.path.join(ckdir, "checkpoint-iteration7.pth"), "--synthetic", "--iterations", "14"], str(tmp_path))
    assert os.path.exists(os.path.join(ckdir, "checkpoint-iteration14.pth")), out2[-1500:]
    ck2 = load_checkpoint(os.path.join(ckdir, "checkpoint-iteration14.pth"))
    assert ck2["iteration"] == 14
    assert ck2["rng"]["offset"] > ck["rng"]["offset"]                       # the Philox stream continued instead of being replayed
    moved = sum(1 for k, v in ck2["state_dict"].items() if v.dtype.is_floating_point and not torch.equal(v, before[k]))
    assert moved > 100


# ---- on-disk datasets through the HIP trainer -----------------------------------------------------------------------------
RIMES_CFG = "cf_RIMESLinesslant_noMask_charSpecSingleAppend_GANMedMT_autoAEMoPrcp2tightNewCTCUseGen_balB_hCF0.75_sMG.json"


def _fabricate(which, root):
    """a dataset directory in the reference's layout: IAM (forms/ xmls/ sets.json) or RIMES (images_gray/ lines_*_2011*.xml)"""
    import numpy as np
    from oracle import collate_items
    if which == "iam":
        collate_items.fake_iam(root, n_pages=6, with_images=True)
        return
    from PIL import Image, ImageDraw
    os.makedirs(os.path.join(root, "images_gray"))
    text = collate_items.rimes_xml()
    for fn in ("lines_training_2011.xml", "lines_eval_2011_annotated.xml"):
        open(os.path.join(root, fn), "w").write(text)
    rs = np.random.RandomState(1)
    for p in range(4):
        img = Image.new("L", (1000, 420), 255)
        dr = ImageDraw.Draw(img)
        for _ in range(300):                      # dark strokes on white paper, everywhere (every line crop has ink)
            x, y = int(rs.randint(0, 980)), int(rs.randint(0, 400))
            dr.rectangle([x, y, x + int(rs.randint(3, 18)), y + int(rs.randint(6, 30))], fill=int(rs.randint(10, 120)))
        img.save(os.path.join(root, "images_gray", "page%03d.png" % p))


def _disk_config(which, root, tmp_path, bucket):
    from handwriting_line_generation_amd.harness import CHAR_FILES
    name = CFG if which == "iam" else RIMES_CFG
    cfg = json.load(open(os.path.join(ROOT, "configs", name)))
    dl = cfg["data_loader"]
    dl.update(data_dir=root, batch_size=2, a_batch_size=2, num_workers=0, shuffle=True, char_file=CHAR_FILES[which], max_width=640)
    if bucket:
        dl["width_bucket"] = bucket
    cfg["validation"] = dict(cfg.get("validation", {}), batch_size=2, a_batch_size=2, shuffle=False, num_workers=0, augmentation=None)
    tr = cfg["trainer"]
    tr.update(save_dir=str(tmp_path / "saved"), save_step=10 ** 6, save_step_minor=10 ** 6, log_step=10 ** 6, val_step=7, print_dir=None,
              encoder_weights=str(tmp_path / "enc" / "encoder.pth"), text_data=str(tmp_path / "no_such_corpus.txt"), iterations=7)
    cfg["model"]["pretrained_hwr"] = None
    cfg["seed"] = 5
    cfg["cuda"], cfg["gpu"] = True, 0
    return cfg, name


@pytest.mark.parametrize("which,bucket", [("iam", 0), ("iam", 128), ("rimes", 0), ("rimes", 128)])
def test_on_disk_dataset_feeds_the_hip_trainer(cuda, tmp_path, which, bucket):
    """fabricated IAM / RIMES directory -> data.getDataLoader -> HWWithStyleTrainer on HIP kernels: one curriculum cycle + one _valid_epoch.
    Every batch the trainer consumed is `collate` (+ bucket padding) of the dataset items drawn for it; losses are finite; with width
    bucketing the consumed widths are multiples of the bucket and few."""
    import random

    import numpy as np
    from handwriting_line_generation_amd import model as M, ops, rng
    from handwriting_line_generation_amd.data import author_hw_dataset as D
    from handwriting_line_generation_amd.data.synthetic import write_synthetic_corpus
    from handwriting_line_generation_amd.model import loss as loss_fns
    from handwriting_line_generation_amd.trainer import HWWithStyleTrainer
    root = str(tmp_path / which)
    os.makedirs(root)
    _fabricate(which, root)
    cfg, _ = _disk_config(which, root, tmp_path, bucket)
    tr, dl = cfg["trainer"], cfg["data_loader"]
    os.makedirs(os.path.dirname(tr["encoder_weights"]))
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    rng.set_mode("device", seed=3)
    torch.save({"state_dict": M.Autoencoder({"type": tr.get("encoder_type", "2tight"), "hwr": cfg["model"]["num_class"]}).state_dict()}, tr["encoder_weights"])
    tr["text_data"] = str(tmp_path / "corpus.txt")
    write_synthetic_corpus(tr["text_data"], dl["char_file"])
    loader, vloader = D.getDataLoader(cfg, "train")
    assert vloader is not None and len(loader) > 0

    # record what the datasets hand out and what the trainer takes in
    drawn, consumed = [], []
    for ds in (loader.dataset, vloader.dataset):
        orig = ds.__class__.__getitem__

        def getitem(self, idx, _orig=orig):
            item = _orig(self, idx)
            drawn.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in item.items()})
            return item
        ds.__class__ = type(ds.__class__.__name__ + "Rec", (ds.__class__,), {"__getitem__": getitem})
    model = M.HWWithStyle(cfg["model"])
    losses = {k: getattr(loss_fns, v) for k, v in cfg["loss"].items()}
    trainer = HWWithStyleTrainer(model, losses, [], None, cfg, loader, vloader, None)
    to_tensor = trainer._to_tensor

    def spy(instance):
        if instance["image"] is not None:
            consumed.append(instance)
        return to_tensor(instance)
    trainer._to_tensor = spy
    for it in range(7):
        log = trainer._train_iteration(it)
        assert log and all(np.isfinite(v) for v in log.values()), (it, log)
    val = trainer._valid_epoch()
    assert val and all(np.isfinite(v) for v in val.values()), val
    torch.cuda.synchronize()
    assert len(consumed) >= 4 and len(drawn) == 2 * len(consumed)
    widths = set()
    for k, inst in enumerate(consumed):
        want = D.collate([dict(drawn[2 * k]), dict(drawn[2 * k + 1])])
        if bucket:
            want = D.pad_width(want, bucket)
        assert torch.equal(inst["image"], want["image"]) and torch.equal(inst["label"], want["label"]), k
        assert inst["gt"] == want["gt"] and inst["a_batch_size"] == 2 and inst["label_lengths"].tolist() == want["label_lengths"].tolist()
        assert inst["image"].shape[:3] == (4, 1, 64) and float(inst["image"].min()) >= -1 and float(inst["image"].max()) <= 1
        widths.add(inst["image"].shape[3])
    if bucket:
        assert all(w % bucket == 0 for w in widths) and len(widths) <= 4, sorted(widths)
        # the kernels' per-geometry plan cache planned every consumed (bucketed) width for the first layers on real lines (64 rows, one channel)
        planned = {key[2] for key in ops._conv_plans if key[1] == 64 and key[3] == 1}
        assert widths <= planned, (sorted(widths), sorted(planned))
    rng.set_mode("device")


def test_train_cli_on_a_dataset_directory(cuda, tmp_path):
    """`train.py -c <cfg>` WITHOUT --synthetic on a fabricated IAM directory: the CLI resolves the dataset on disk, runs a curriculum cycle
    with a validation epoch and writes a checkpoint (reference train.py:21-93)"""
    from handwriting_line_generation_amd.logger import load_checkpoint
    root = str(tmp_path / "iam")
    os.makedirs(root)
    _fabricate("iam", root)
    cfg, name = _disk_config("iam", root, tmp_path, 128)
    cfg["trainer"].update(save_step=7, iterations=8)
    path = str(tmp_path / name)
    json.dump(cfg, open(path, "w"))
    out = _run(["-c", path, "--random-init-aux", "--iterations", "8"], str(tmp_path))
    ckdir = os.path.join(cfg["trainer"]["save_dir"], cfg["name"])
    assert os.path.exists(os.path.join(ckdir, "checkpoint-iteration7.pth")), out[-2000:]
    ck = load_checkpoint(os.path.join(ckdir, "checkpoint-iteration7.pth"))
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.dtype.is_floating_point)
    assert "validation:" in out and "val_loss" in out, out[-2000:]
