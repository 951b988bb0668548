"""CPU: host-side behaviour of train.py and the trainer base that needs no device - warm-start files must exist (the reference fails
there: model/hw_with_style.py:166-178, trainer/hw_with_style_trainer.py:136-160), the SIGINT stop flag, and the optimizer state-dict layout
the reference's torch.optim.Adam (two parameter groups, base/base_trainer.py:95-97) accepts."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GAN_CFG = "cf_IAMslant_noMask_charSpecSingleAppend_GANMedMT_autoAEMoPrcp2tightNewCTCUseGen_balB_hCF0.75_sMG.json"


def _run_train(tmp_path, extra, mutate=None):
    cfg = json.load(open(os.path.join(ROOT, "configs", GAN_CFG)))
    cfg["data_loader"]["data_dir"] = str(tmp_path / "no_such_dataset")
    cfg["trainer"]["save_dir"] = str(tmp_path / "saved")
    if mutate:
        mutate(cfg)
    path = tmp_path / GAN_CFG
    path.write_text(json.dumps(cfg))
    return subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "-c", str(path)] + extra, capture_output=True, text=True, cwd=str(tmp_path),
                          timeout=600)


def test_missing_encoder_weights_is_an_error_without_an_explicit_flag(tmp_path):
    r = _run_train(tmp_path, [], lambda c: c["trainer"].__setitem__("encoder_weights", str(tmp_path / "absent_encoder.pth")))
    assert r.returncode != 0 and "encoder_weights" in r.stderr and "does not exist" in r.stderr, r.stderr[-400:]
    assert not os.path.exists(tmp_path / "absent_encoder.pth")      # nothing was silently written


def test_missing_pretrained_hwr_is_an_error_without_an_explicit_flag(tmp_path):
    def mutate(c):
        torch.save({"state_dict": {}}, str(tmp_path / "enc.pth"))
        c["trainer"]["encoder_weights"] = str(tmp_path / "enc.pth")
        c["model"]["pretrained_hwr"] = str(tmp_path / "absent_hwr.pth")
    r = _run_train(tmp_path, [], mutate)
    assert r.returncode != 0 and "pretrained_hwr" in r.stderr and "does not exist" in r.stderr, r.stderr[-400:]


def test_random_init_aux_flag_warns_loudly_and_proceeds_to_the_dataset_check(tmp_path):
    def mutate(c):
        c["trainer"]["encoder_weights"] = str(tmp_path / "aux" / "encoder.pth")
        c["model"]["pretrained_hwr"] = str(tmp_path / "absent_hwr.pth")
    r = _run_train(tmp_path, ["--random-init-aux"], mutate)
    assert "RANDOM-INIT Encoder2" in r.stderr and "recogniser stays RANDOM-INIT" in r.stderr, r.stderr[-600:]
    assert os.path.exists(tmp_path / "aux" / "encoder.pth")
    assert r.returncode != 0 and "no dataset" in r.stderr          # the next check (no IAM tree here) is what stops it


def test_stop_flag_is_only_raised_by_the_handler_and_agreed_in_train():
    from handwriting_line_generation_amd.base.base_trainer import BaseTrainer
    t = BaseTrainer.__new__(BaseTrainer)
    t._stop = False
    assert t._stop_agreed() is False
    t.request_stop()
    assert t._stop_agreed() is True
    saved = []
    t.start_iteration, t.iterations, t.lr_lambda = 5, 9, None
    t.save = lambda: saved.append(t.iteration)
    t._train_iteration = lambda it: pytest.fail("no iteration may start after a stop request")
    t.train()
    assert saved == [4]       # checkpoint of the last completed iteration


def test_main_optimizer_state_dict_loads_into_the_references_two_group_adam():
    """the reference builds Adam([{'params': main}, {'params': slow, 'lr': lr*0.1}]) (base/base_trainer.py:95-97); torch refuses a state dict
    with a different number of groups, so files written here must carry the empty second group"""
    from handwriting_line_generation_amd.trainer.flat_params import FlatParams, HipAdam
    ps = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    flat = FlatParams(ps, {"main": ps[:2], "disc": ps[2:]})
    opt = HipAdam(flat, "main", lr=2e-4, betas=(0.5, 0.999))
    opt.steps[:2] = 3
    opt.exp_avg.copy_(torch.arange(flat.total, dtype=torch.float32))
    opt.exp_avg_sq.fill_(0.25)
    sd = opt.state_dict()
    assert len(sd["param_groups"]) == 2 and sd["param_groups"][1]["params"] == [] and sd["param_groups"][1]["lr"] == pytest.approx(2e-5)
    ref = torch.optim.Adam([{"params": ps[:2]}, {"params": [], "lr": 2e-5}], lr=2e-4, betas=(0.5, 0.999))
    ref.load_state_dict(sd)
    st = ref.state[ps[1]]
    assert float(st["step"]) == 3 and torch.equal(st["exp_avg"], torch.arange(16, 23, dtype=torch.float32)) and float(st["exp_avg_sq"][0]) == 0.25
    # and back: a reference-written two-group state loads here
    opt2 = HipAdam(flat, "main", lr=1e-3, betas=(0.5, 0.999))
    opt2.load_state_dict(ref.state_dict())
    assert opt2.steps[:2].tolist() == [3, 3] and torch.equal(opt2.exp_avg[:15], opt.exp_avg[:15]) and opt2.param_groups[0]["lr"] == 2e-4
    dsd = HipAdam(flat, "disc", lr=1e-4).state_dict()
    assert len(dsd["param_groups"]) == 1          # the discriminator's Adam is built from a plain list in the reference
