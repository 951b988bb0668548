"""GPU: checkpoints written by the reference load into the HIP-backed trainer / model through the three paths the reference has
(SURVEY 8f-1): `-r` resume (base/base_trainer.py:401-479), `model.pretrained_hwr` with its 'hwr.' prefix strip (model/hw_with_style.py:166-178) and
`trainer.encoder_weights` with its 'encoder.' prefix (trainer/hw_with_style_trainer.py:136-160) - and the loaded networks compute what the
oracle computes from the very tensors stored in the file."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref
from test_reference_checkpoint_cpu import GOLD, unpack

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def _reduced_trainer(tmp_path, **kw):
    from handwriting_line_generation_amd import harness
    from handwriting_line_generation_amd.data.synthetic import SyntheticAuthorDataset, SyntheticLoader
    from handwriting_line_generation_amd.model import HWWithStyle
    from handwriting_line_generation_amd.model import loss as loss_fns
    from handwriting_line_generation_amd.trainer import HWWithStyleTrainer
    cfg, _ = harness.synthetic_gan_config("iam_gan", 2, 2, workdir=str(tmp_path))
    cfg["model"].update(json.load(open(os.path.join(GOLD, "ref_ckpt_reduced_model.json"))))
    cfg["model"]["pretrained_hwr"] = kw.get("pretrained_hwr")
    cfg["trainer"]["encoder_weights"] = kw["encoder_weights"]
    model = HWWithStyle(cfg["model"])
    dl = cfg["data_loader"]
    ds = SyntheticAuthorDataset(dl["char_file"], 2, 2, width=128, label_len=6)
    losses = {name: getattr(loss_fns, fn) for name, fn in cfg["loss"].items()}
    return HWWithStyleTrainer(model, losses, [], kw.get("resume"), cfg, SyntheticLoader(ds), None, None), cfg


def test_resume_from_reference_written_checkpoint(cuda, tmp_path):
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.logger import Logger, load_checkpoint
    gan, auto = unpack("gan", tmp_path), unpack("auto", tmp_path)
    ck = load_checkpoint(gan)
    torch.manual_seed(3)
    tr, cfg = _reduced_trainer(tmp_path, resume=gan, encoder_weights=auto)
    assert tr.start_iteration == 25001 and tr.monitor_best == 0.5
    assert isinstance(tr.train_logger, Logger) and tr.train_logger.entries[1]["loss"] == 1.25
    sd = tr.model.state_dict()
    for k, v in ck["state_dict"].items():
        assert torch.equal(sd[k].cpu(), v), "resumed %s differs from the file" % k
    # Adam state: keyed by the parameter's index within the main optimizer's group, exactly as torch.optim.Adam stores it
    back = tr.optimizer.state_dict()["state"]
    assert set(back) == set(ck["optimizer"]["state"])
    for j, st in ck["optimizer"]["state"].items():
        assert torch.equal(back[j]["exp_avg"].cpu(), st["exp_avg"]) and torch.equal(back[j]["exp_avg_sq"].cpu(), st["exp_avg_sq"]), j
        assert float(back[j]["step"]) == float(st["step"])
    # the loaded networks compute what the oracle computes from the stored tensors
    rng.set_mode("host")
    try:
        g = torch.Generator().manual_seed(1)
        sub = lambda p: {k[len(p):]: v for k, v in ck["state_dict"].items() if k.startswith(p)}   # noqa: E731
        content = F.one_hot(torch.randint(0, 80, (14, 2), generator=g), 80).float()
        style = torch.randn(2, 128, generator=g)
        torch.manual_seed(9)
        y = tr.model.generator(content.to(cuda), style.to(cuda))
        torch.manual_seed(9)
        assert _rel(y, torch_ref.generator(sub("generator."), content, style)) < TOL
        img = torch.rand(2, 1, 64, 96, generator=g) * 2 - 1
        tr.model.train()
        torch.manual_seed(9)
        outs = tr.model.discriminator(img.to(cuda))
        torch.manual_seed(9)
        for a, b in zip(outs, torch_ref.discriminator(sub("discriminator."), img)):
            assert _rel(a, b) < TOL
        # the recogniser in eval mode (running statistics from the file; batch statistics of 2 x 18 columns under periodic weights
        # would make some channels nearly constant and the comparison ill-conditioned)
        tr.model.hwr.eval()
        with torch.no_grad():
            pred = tr.model.hwr(img.to(cuda), None)
        tr.model.hwr.train()
        assert _rel(pred, torch_ref.hwr(sub("hwr."), img, training=False)) < TOL
        # ... and training continues from there
        np.random.seed(0)
        log = tr._train_iteration(25004)      # a `count` lesson (25004 % 7 == 0): with these periodic weights a generated line is only a few
        # columns wide, too short for the recogniser's unpadded 1-D convolutions (the reference fails there as well)
        assert all(np.isfinite(v) for v in log.values())
    finally:
        rng.set_mode("device")


def test_pretrained_hwr_and_encoder_weights_from_reference_files(cuda, tmp_path):
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.logger import load_checkpoint
    hwr, auto = unpack("hwr", tmp_path), unpack("auto", tmp_path)
    torch.manual_seed(4)
    tr, cfg = _reduced_trainer(tmp_path, pretrained_hwr=hwr, encoder_weights=auto)
    ckh, cka = load_checkpoint(hwr), load_checkpoint(auto)
    got = tr.model.hwr.state_dict()
    assert all(k.startswith("hwr.") for k in ckh["state_dict"])           # the reference saves the whole HWWithStyle: prefix strip on load
    for k, v in ckh["state_dict"].items():
        assert torch.equal(got[k[4:]].cpu(), v), k
    enc = tr.encoder.state_dict()
    n = 0
    for k, v in cka["state_dict"].items():
        if k.startswith("encoder."):
            assert torch.equal(enc[k[8:]].cpu(), v), k
            n += 1
    assert n == len(enc) and n > 10
    g = torch.Generator().manual_seed(2)
    img = torch.rand(2, 1, 64, 128, generator=g) * 2 - 1
    rng.set_mode("host")
    try:
        torch.manual_seed(5)
        feats = tr.encoder(img.to(cuda))
        torch.manual_seed(5)
        ref = torch_ref.encoder2({k[8:]: v for k, v in cka["state_dict"].items() if k.startswith("encoder.")}, img)
        for a, b in zip(feats, ref):
            assert _rel(a, b) < TOL
    finally:
        rng.set_mode("device")
