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


DEAD = {"hwr.cnn.conv2.bias", "hwr.cnn.conv4.bias", "hwr.cnn.conv6.bias", "hwr.cnn1d.0.bias", "hwr.cnn1d.3.bias", "hwr.cnn1d.6.bias",
        "hwr.cnn1d.9.bias"}     # conv biases feeding a batch-statistics BatchNorm: analytically zero gradient, rounding noise on every side


def _check_grads(got, sd32, sd64, names, what, cond=None):
    """The flat gradients the trainer is about to clip / hand to Adam, per parameter, against the oracle's - triangulated with the same
    oracle in fp64: the HIP gradient must be within 1e-4 (relative L2) of the fp64 value, or within twice the error the oracle's own fp32
    arithmetic has there (the recogniser's backward is ~1e-2 from fp64 in fp32 for either implementation, tests/test_pipeline_gpu.py).
    Comparing gradients instead of the first Adam update keeps sign flips of near-zero elements out of the picture. `cond`: how far the fp64
    gradients move under 1e-6 relative perturbations of weights and inputs (worst of 8 draws) - one ReLU / max-pool gate flipping near the top
    of the recogniser shifts every gradient below it by ~1e-2, and no fp32 implementation can be closer to another than that."""
    def l2(a, b):
        a = a.detach().double().cpu(); b = b.detach().double()
        return float((a - b).norm() / max(float(b.norm()), 1e-300))
    gmax = max(float(sd64[k].grad.abs().max()) for k in names if sd64[k].grad is not None)
    bad = []
    for k in names:
        d = sd64[k].grad
        if k in DEAD or d is None or float(d.abs().max()) < 1e-6 * gmax:
            continue
        assert got.get(k) is not None, "%s: no gradient for %s" % (what, k)
        eh, eo = l2(got[k], d), l2(sd32[k].grad, d)
        if eh > max(1e-4, 2 * eo, 3 * (cond or {}).get(k, 0.0)):
            bad.append("%s: error vs fp64 %.2e (fp32 oracle %.2e, sensitivity %.2e)" % (k, eh, eo, (cond or {}).get(k, 0.0)))
    assert not bad, "%s: %d gradients off: %s" % (what, len(bad), "; ".join(bad[:8]))


def _hook_grads(trainer):
    got = {}
    trainer.pre_clip_hook = lambda it: got.update({k: p.grad.detach().clone() for k, p in trainer.model.named_parameters() if p.grad is not None})
    return got


def _double(sd, names, perturb=None):
    out = {k: (v.detach().double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    if perturb is not None:
        g = torch.Generator().manual_seed(perturb)
        for k in names:
            out[k].mul_(1 + 1e-6 * torch.randn(out[k].shape, generator=g, dtype=torch.float64))
    for k in names:
        out[k].requires_grad_(True)
    return out


def _conditioning(run64, ref64, names):
    cond = {k: 0.0 for k in names}
    for trial in range(1, 9):
        w = run64(trial)
        for k in names:
            if ref64[k].grad is not None and w[k].grad is not None:
                cond[k] = max(cond[k], float((w[k].grad - ref64[k].grad).norm() / ref64[k].grad.norm().clamp_min(1e-300)))
    return cond


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
    got = _hook_grads(trainer)
    log = trainer._train_iteration(0)
    # oracle step, in fp32 and in fp64
    names = [k for k, p in trainer.model.named_parameters()]
    sd = _leafify(msd, names)
    pred = torch_ref.hwr(sd, batch["image"], prefix="hwr.")
    loss = _ctc(pred, batch["label"], 5)
    loss.backward()
    def run64(perturb=None):
        w = _double(msd, names, perturb)
        x = batch["image"].double()
        if perturb is not None:
            x = x * (1 + 1e-6 * torch.randn(x.shape, generator=torch.Generator().manual_seed(100 + perturb), dtype=torch.float64))
        _ctc(torch_ref.hwr(w, x, prefix="hwr."), batch["label"], 5).backward()
        return w
    sd64 = run64()
    assert abs(log["recogLoss"] - float(loss)) < 1e-5 * max(abs(float(loss)), 1.0)
    _check_grads(got, sd, sd64, names, "hwr pretrain", _conditioning(run64, sd64, names))
    rng.set_mode("device")


def test_autoencoder_step(cuda, tmp_path):
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_simple_trainer
    from handwriting_line_generation_amd.model import Autoencoder
    rng.set_mode("host")
    msd = torch_ref.seeded_state_dict(Autoencoder({"type": "2tight", "hwr": 80}), 42)
    trainer, cfg = build_simple_trainer("iam_auto", batch_size=3, width=132, label_len=5, workdir=str(tmp_path), model_state=msd)
    batch = trainer.data_loader.dataset.batch(0)
    got = _hook_grads(trainer)
    torch.manual_seed(7)
    log = trainer._train_iteration(0)
    names = [k for k, p in trainer.model.named_parameters()]
    sd, sd64 = _leafify(msd, names), _double(msd, names)
    img = F.pad(batch["image"], (2, 2), value=-1.0)          # 132 -> 136 (multiple of 8), centred

    def step(w, x):
        torch.manual_seed(7)
        code, mid = torch_ref.encoder2(w, x, prefix="encoder.")
        recon = torch_ref.decoder_noskip(w, code, prefix="decoder.")
        pred = torch_ref.e_hwr(w, code, prefix="hwr.")
        assert recon.shape[3] == x.shape[3]
        l1 = F.l1_loss(recon, x)
        ctc = _ctc(pred, batch["label"], 5)
        (l1 + ctc).backward()
        return l1, ctc
    l1, ctc = step(sd, img)

    def run64(perturb=None):
        w = _double(msd, names, perturb)
        x = img.double()
        if perturb is not None:
            x = x * (1 + 1e-6 * torch.randn(x.shape, generator=torch.Generator().manual_seed(100 + perturb), dtype=torch.float64))
        step(w, x)
        return w
    sd64 = run64()
    assert abs(log["autoLoss"] - float(l1)) < 1e-5 and abs(log["recogLoss"] - float(ctc)) < 1e-4 * max(float(ctc), 1.0)
    _check_grads(got, sd, sd64, names, "autoencoder", _conditioning(run64, sd64, names))
    rng.set_mode("device")


# =====================================================================================================================================
# Against goldens recorded from the UNMODIFIED reference trainers (tools/gen_golden_pretrain.py)
# =====================================================================================================================================
import json
import math
import os

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _fp(t, k):
    d = t.detach().double().flatten()
    r = torch.cos(torch.arange(d.numel(), dtype=torch.float64, device=d.device) * 0.37 + 1.3 * k)
    return torch.stack([d.sum(), d.abs().sum(), (d * d).sum(), (d * r).sum()]).cpu().tolist()


def _compare_sets(what, names, got, ref32, ref64, bad, lines, flip_floor=3e-3, skip=()):
    """pooled per sub-network, as tests/test_trainer_lessons_gpu.py: error vs the reference's fp64 value <= max(1e-4, 2 x the reference's own
    fp32 error) or below the gate-flip floor documented there; None-vs-present exact"""
    groups = {}
    dead = set()
    scale = max((b[2] for b in ref64 if b is not None), default=0.0)
    for k, (n, g, a, b) in enumerate(zip(names, got, ref32, ref64)):
        if (g is None) != (a is None):
            bad.append("%s %s: %s here, %s in the reference" % (what, n, "None" if g is None else "present", "None" if a is None else "present"))
            continue
        if g is None or n in skip:
            continue
        if b[2] < 1e-24 * scale:
            dead.add(n)
            # analytically zero (conv biases in front of a batch-statistics BatchNorm): the fp64 value is rounding noise 1e-12 of the
            # set's largest tensor, a relative error has no meaning; the HIP value must be fp32 rounding noise as well
            if g[2] > 1e-9 * scale:
                bad.append("%s %s: analytically zero gradient, here %.2e of the largest tensor's norm" % (what, n, math.sqrt(g[2] / scale)))
            continue
        nrm, l1 = math.sqrt(b[2]), max(b[1], 1e-300)
        eh = max(abs(g[3] - b[3]) / nrm, abs(g[1] - b[1]) / l1)
        er = max(abs(a[3] - b[3]) / nrm, abs(a[1] - b[1]) / l1)
        groups.setdefault(n.split(".")[0], []).append((eh, er, n))
    for top, items in sorted(groups.items()):
        rh = math.sqrt(sum(e[0] ** 2 for e in items) / len(items))
        rr = math.sqrt(sum(e[1] ** 2 for e in items) / len(items))
        bound = min(max(1e-4, 2 * rr), 1e-2)
        lines.append("   %-28s %-12s %4d tensors  HIP %.2e  reference fp32 %.2e  bound %.2e%s" % (what, top, len(items), rh, rr, bound,
                                                                                                  "  flip" if bound < rh <= flip_floor else ""))
        if rh > max(bound, flip_floor):
            worst = sorted(items, reverse=True)[:3]
            bad.append("%s %s: pooled error %.2e > %.2e (reference %.2e); worst %s" % (what, top, rh, bound, rr, ["%s %.1e" % (w[2], w[0]) for w in worst]))
    return dead


def _run_pretrain_golden(cuda, tmp_path, which, gold_name, ctor, seed_model):
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_simple_trainer
    from oracle import tf_state
    gold = json.load(open(os.path.join(GOLD, gold_name)))
    rng.set_mode("host")
    try:
        msd = torch_ref.seeded_state_dict(ctor(), gold["seed"])
        trainer, cfg = build_simple_trainer(which, batch_size=gold["B"], width=gold["W"], label_len=gold["L"], workdir=str(tmp_path), model_state=msd)
        names = gold["names"]
        assert [n for n, _ in trainer.model.named_parameters()] == names
        params = dict(trainer.model.named_parameters())
        seen = {}
        state = {"rms": None, "it": 0}
        f = trainer.flat

        def hook(it):
            # (parameter k of named_parameters() sits at position flat.pos[k] of the flat gradient buffer; untouched = the reference's None)
            seen[it] = [(_fp(params[n].grad, k) if f.touched[f.pos[k]] else None) for k, n in enumerate(names)]
            # Adam moments := seeded draws scaled by the REFERENCE's gradient RMS of this iteration (oracle/tf_state.py), as the golden tool did
            m_host, v_host = torch.zeros(f.total), torch.zeros(f.total)
            for k, n in enumerate(names):
                pos = f.pos[k]
                if not f.touched[pos]:
                    continue
                m, v = tf_state.seeded_moments(params[n].shape, state["rms"][k], tf_state.moment_key(state["it"], k))
                a = int(f.offsets[pos])
                m_host[a:a + int(f.numel[pos])] = m.flatten()
                v_host[a:a + int(f.numel[pos])] = v.flatten()
                trainer.optimizer.steps[pos] = tf_state.ADAM_STEP
            trainer.optimizer.exp_avg.copy_(m_host)
            trainer.optimizer.exp_avg_sq.copy_(v_host)
        trainer.pre_clip_hook = hook
        bad, lines = [], []
        for it, ref in enumerate(gold["iterations"]):
            # every iteration starts from the seeded weights and an empty optimizer state (teacher forcing, as in the golden tool)
            trainer.model.load_state_dict(msd)
            trainer.optimizer.reset_state()
            f.flat_grad.zero_(); f.touched[:] = False
            state["rms"], state["it"] = ref["rms"], it
            torch.manual_seed(7 + it); np.random.seed(7 + it); random.seed(7 + it)
            snap = [params[n].detach().clone() for n in names]
            log = trainer._train_iteration(it)
            for k, rv in ref["log"].items():
                r64 = ref["log64"][k]
                assert k in log, "iteration %d: %s missing from the log %s" % (it, k, sorted(log))
                if k in ("CER", "WER"):
                    # greedy decode of an untrained recogniser: equal unless an arg-max near-tie resolves differently (a character or two)
                    assert abs(log[k] - rv) <= 2.0 / (gold["B"] * gold["L"]) + 1e-9, "iteration %d %s: %r vs reference %r" % (it, k, log[k], rv)
                elif abs(log[k] - r64) > max(1e-5 * max(abs(r64), 1e-3), 4 * abs(rv - r64)):
                    bad.append("iteration %d %s: %.8g vs reference fp64 %.8g (fp32 %.8g)" % (it, k, log[k], r64, rv))
            dead = _compare_sets("it%d gradient" % it, names, seen[it], ref["grads"], ref["grads64"], bad, lines)
            upd = [_fp(params[n].detach() - s, k) for k, (n, s) in enumerate(zip(names, snap))]
            upd = [u if u[1] != 0.0 else None for u in upd]
            # (tensors whose gradient is analytically zero move by Adam's response to rounding noise: no comparison)
            _compare_sets("it%d update" % it, names, upd, ref["update"], ref["update64"], bad, lines, skip=dead)
        print("\n[%s vs reference trainer]\n%s" % (gold_name, "\n".join(lines)))
        if os.environ.get("HWG_PARITY_SUMMARY"):
            with open(os.environ["HWG_PARITY_SUMMARY"], "a") as fh:
                fh.write("[%s vs reference trainer]\n%s\n\n" % (gold_name, "\n".join(lines)))
        assert not bad, "; ".join(bad[:10])
        return trainer, gold
    finally:
        rng.set_mode("device")


def test_hwr_pretrain_vs_reference_trainer(cuda, tmp_path):
    """cf_IAM_hwr_cnnOnly_batchnorm_aug through HWWithStyleTrainer.run_hwr: 2 iterations recorded from the reference (losses, CER / WER,
    per-tensor gradients handed to Adam, updates)"""
    from handwriting_line_generation_amd.model import HWWithStyle
    cfgm = {"num_class": 80, "hwr": "CNNOnly batchnorm", "generator": "none", "style": "none"}
    _run_pretrain_golden(cuda, tmp_path, "iam_hwr", "pretrain_hwr.json", lambda: HWWithStyle(cfgm), 41)


def test_autoencoder_vs_reference_trainer_and_validation(cuda, tmp_path):
    """cf_IAM_auto_2tight_newCTC through AutoTrainer: 2 training iterations and one _valid_epoch over 3 batches, recorded from the reference"""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.model import Autoencoder
    trainer, gold = _run_pretrain_golden(cuda, tmp_path, "iam_auto", "pretrain_auto.json", lambda: Autoencoder({"type": "2tight", "hwr": 80}), 42)
    rng.set_mode("host")
    try:
        trainer.model.load_state_dict(torch_ref.seeded_state_dict(Autoencoder({"type": "2tight", "hwr": 80}), gold["seed"]))
        trainer.valid_data_loader = [trainer.data_loader.dataset.batch(i) for i in (50, 51, 52)]
        torch.manual_seed(77); np.random.seed(77); random.seed(77)
        val = trainer._valid_epoch()
        _check_valid(val, gold)
    finally:
        rng.set_mode("device")


def _check_valid(val, gold):
    assert set(val) == set(gold["valid"]), "validation log keys %s vs reference %s" % (sorted(val), sorted(gold["valid"]))
    for k, rv in gold["valid"].items():
        r64 = gold["valid64"][k]
        if k in ("val_CER", "val_WER"):
            # decoded strings come from an arg-max: equal unless a near-tie resolves differently (one character of one line at most)
            assert abs(val[k] - rv) <= 0.1, "%s: %r vs reference %r" % (k, val[k], rv)     # (the reference's own fp32 and fp64 runs differ by 0.044 here)
        else:
            tol = max(2e-5 * max(abs(r64), 1e-3), 4 * abs(rv - r64))
            assert abs(val[k] - r64) <= tol, "%s: %.8g vs reference fp64 %.8g (fp32 %.8g)" % (k, val[k], r64, rv)


def test_gan_validation_epoch_vs_reference(cuda, tmp_path):
    """HWWithStyleTrainer._valid_epoch (trainer/hw_with_style_trainer.py:437-486) on the GAN config: curriculum.getValid() lesson over three
    synthetic batches under no_grad in eval mode, every val_* value against the reference's"""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer, load_config
    from handwriting_line_generation_amd.model import Autoencoder, HWWithStyle
    gold = json.load(open(os.path.join(GOLD, "valid_gan.json")))
    cfg_model = dict(load_config("iam_gan")["model"], pretrained_hwr=None)
    msd = torch_ref.seeded_state_dict(HWWithStyle(cfg_model), gold["seed"])
    esd = torch_ref.seeded_state_dict(Autoencoder({"type": "2tight", "hwr": cfg_model["num_class"]}), 22)
    rng.set_mode("host")
    try:
        trainer, cfg = build_gan_trainer("iam_gan", gold["B"], gold["A"], width=gold["W"], label_len=gold["L"], workdir=str(tmp_path),
                                         model_state=msd, encoder_state=esd)
        assert sorted(trainer.curriculum.getValid()) == gold["valid_lesson"]
        trainer.valid_data_loader = [trainer.data_loader.dataset.batch(i) for i in (50, 51, 52)]
        trainer.valid = True
        torch.manual_seed(77); np.random.seed(77); random.seed(77)
        val = trainer._valid_epoch()
        _check_valid(val, gold)
    finally:
        rng.set_mode("device")


def test_train_loop_runs_validation_for_both_trainer_classes(cuda, tmp_path):
    """BaseTrainer.train() past val_step with a small validation loader, for AutoTrainer and HWWithStyleTrainer (a trainer without
    _valid_epoch used to die with AttributeError hours into a run)"""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer, build_simple_trainer
    rng.set_mode("device", seed=3)
    tr = build_simple_trainer("iam_auto", batch_size=2, width=128, label_len=5, workdir=str(tmp_path / "a"))
    tr = tr[0]
    tr.iterations, tr.val_step, tr.log_step, tr.save_step, tr.save_step_minor = 2, 2, 10 ** 6, 10 ** 6, None
    tr.valid_data_loader, tr.valid = [tr.data_loader.dataset.batch(9)], True
    seen = []
    tr.logger.info = lambda msg, *a: seen.append(msg % a if a else msg)
    tr.train()
    assert any("validation" in m and "val_loss" in m for m in seen), seen
    g, _ = build_gan_trainer("iam_gan", 1, 2, width=128, label_len=6, workdir=str(tmp_path / "g"))
    g.iterations, g.val_step, g.log_step, g.save_step, g.save_step_minor = 1, 1, 10 ** 6, 10 ** 6, None
    g.valid_data_loader, g.valid = [g.data_loader.dataset.batch(9)], True
    seen = []
    g.logger.info = lambda msg, *a: seen.append(msg % a if a else msg)
    g.train()
    assert any("validation" in m and "val_loss" in m for m in seen), seen
