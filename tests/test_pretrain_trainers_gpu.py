"""GPU: one optimisation step of the two pre-training configs against the oracle networks + torch's CPU Adam:
  * cf_IAM_hwr_cnnOnly_batchnorm_aug (BASELINE configs[0]: CTC recogniser, HWWithStyleTrainer without curriculum)
  * cf_IAM_auto_2tight_newCTC       (BASELINE configs[1]: Autoencoder, AutoTrainer: L1 + CTC, clip 2, Adam)"""
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref

pytestmark = pytest.mark.gpu


def _leafify(sd, names):
    out = {k: v.clone() for k, v in sd.items()}
    for k in names:
        out[k].requires_grad_(True)
    return out


def _ctc(pred, label, L):
    T, B = pred.shape[0], pred.shape[1]
    l = F.ctc_loss(pred, label.t().long(), torch.full((B,), T, dtype=torch.long), torch.full((B,), L, dtype=torch.long))
    return torch.where(torch.isinf(l), torch.zeros_like(l), l)


def _check_params(model, ref_sd, before, names, lr, what):
    """The first Adam step moves every weight by ~lr*sign(grad): an element whose (tiny) gradient rounds to the other sign moves the
    other way. The recogniser's backward is only accurate to ~1e-2 in fp32 for EITHER implementation (tests/test_pipeline_gpu.py), so
    elements whose gradient is below ~1 % of the typical magnitude can legitimately flip. Compare update DIRECTIONS: at most 6 % of
    the elements of any tensor may disagree by more than lr/2 and the mean absolute difference must stay below 10 % of lr (a wrong
    gradient or optimizer gives ~50 % / ~100 %)."""
    sd = model.state_dict()
    # conv biases feeding a batch-statistics BatchNorm have an analytically zero gradient: rounding noise decides their Adam step
    dead = {"hwr.cnn.conv2.bias", "hwr.cnn.conv4.bias", "hwr.cnn.conv6.bias", "hwr.cnn1d.0.bias", "hwr.cnn1d.3.bias", "hwr.cnn1d.6.bias",
            "hwr.cnn1d.9.bias"}
    for k in names:
        if k in dead:
            continue
        mine = sd[k].cpu() - before[k]
        ref = ref_sd[k].detach() - before[k]
        diff = (mine - ref).abs()
        frac = float((diff > 0.5 * lr).float().mean())
        assert frac < 6e-2 and float(diff.mean()) < 1e-1 * lr, "%s: %s update differs (%.2e of elements, mean %.2e)" % (what, k, frac, float(diff.mean()))


def test_hwr_pretrain_step(cuda, tmp_path):
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_simple_trainer
    from handwriting_line_generation_amd.model import HWWithStyle
    rng.set_mode("host")
    cfgm = {"num_class": 80, "hwr": "CNNOnly batchnorm", "generator": "none", "style": "none"}
    msd = torch_ref.seeded_state_dict(HWWithStyle(cfgm), 41)
    trainer, cfg = build_simple_trainer("iam_hwr", batch_size=4, width=128, label_len=5, workdir=str(tmp_path), model_state=msd)
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    batch = trainer.data_loader.dataset.batch(0)
    log = trainer._train_iteration(0)
    # oracle step
    names = [k for k, p in trainer.model.named_parameters()]
    sd = _leafify(msd, names)
    pred = torch_ref.hwr(sd, batch["image"], prefix="hwr.")
    loss = _ctc(pred, batch["label"], 5)
    loss.backward()
    opt = torch.optim.Adam([sd[k] for k in names], lr=cfg["optimizer"]["lr"])
    opt.step()
    assert abs(log["recogLoss"] - float(loss)) < 1e-5 * max(abs(float(loss)), 1.0)
    _check_params(trainer.model, sd, msd, names, cfg["optimizer"]["lr"], "hwr pretrain")
    rng.set_mode("device")


def test_autoencoder_step(cuda, tmp_path):
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_simple_trainer
    from handwriting_line_generation_amd.model import Autoencoder
    rng.set_mode("host")
    msd = torch_ref.seeded_state_dict(Autoencoder({"type": "2tight", "hwr": 80}), 42)
    trainer, cfg = build_simple_trainer("iam_auto", batch_size=3, width=132, label_len=5, workdir=str(tmp_path), model_state=msd)
    batch = trainer.data_loader.dataset.batch(0)
    torch.manual_seed(7)
    log = trainer._train_iteration(0)
    names = [k for k, p in trainer.model.named_parameters()]
    sd = _leafify(msd, names)
    torch.manual_seed(7)
    img = F.pad(batch["image"], (2, 2), value=-1.0)          # 132 -> 136 (multiple of 8), centred
    code, mid = torch_ref.encoder2(sd, img, prefix="encoder.")
    recon = torch_ref.decoder_noskip(sd, code, prefix="decoder.")
    pred = torch_ref.e_hwr(sd, code, prefix="hwr.")
    assert recon.shape[3] == img.shape[3]
    l1 = F.l1_loss(recon, img)
    ctc = _ctc(pred, batch["label"], 5)
    (l1 + ctc).backward()
    torch.nn.utils.clip_grad_value_([sd[k] for k in names], 2)
    torch.optim.Adam([sd[k] for k in names], lr=2e-4, betas=(0.5, 0.999)).step()
    assert abs(log["autoLoss"] - float(l1)) < 1e-5 and abs(log["recogLoss"] - float(ctc)) < 1e-4 * max(float(ctc), 1.0)
    _check_params(trainer.model, sd, msd, names, 2e-4, "autoencoder")
    rng.set_mode("device")
