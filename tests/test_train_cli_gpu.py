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
