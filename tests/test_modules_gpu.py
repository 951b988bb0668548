"""GPU: each HIP-backed module (through the C-ABI) against (a) the oracle restatement on the same seeded weights, inputs,
noise and dropout masks and (b) the golden vectors recorded from the reference. fp32 tolerance 1e-4 (north_star).

Gradients are triangulated against the same oracle evaluated in fp64: the HIP gradient of every tensor must be within 1e-4 (relative
L2) of the fp64 value, or - where the oracle's own fp32 arithmetic is further away than that (deep stacks of ReLU / max-pool gates,
small differences of large sums) - within twice the fp32 oracle's own error."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, torch_ref
from test_oracle_golden import GOLD, GRAD_INPUTS, ORACLE_FWD, PERTURB, _oracle_grads64, product_module, rel

pytestmark = pytest.mark.gpu
TOL = 1e-4


def hip_forward(name, m, i):
    name = cases.kind(name)
    if name == "generator":
        return [m(i["content"], i["style"])]
    if name == "discriminator":
        return m(i["x"])
    if name == "hwr":
        return [m(i["image"], None)]
    if name == "spacer":
        return [m(i["onehot"], i["style"])]
    if name == "style_extractor":
        return [m(i["x"], i["recog"])]
    if name == "encoder2":
        return list(m(i["x"]))
    if name == "decoder":
        return [m(i["x"], None)]
    if name == "e_hwr":
        return [m(i["x"])]


@pytest.mark.parametrize("name", list(cases.CASES))
def test_module_parity(cuda, name):
    from handwriting_line_generation_amd import rng
    rng.set_mode("host")  # draw noise / masks from torch's CPU generator in the reference's order
    gold = np.load(os.path.join(GOLD, "module_%s.npz" % name))
    m = product_module(name)
    sd = torch_ref.seeded_state_dict(m, cases.CASES[name]["wseed"])
    m.load_state_dict(sd)
    m.train().to(cuda)
    inp = {k: v.to(cuda) for k, v in cases.inputs(name).items()}
    for k in GRAD_INPUTS[name]:
        inp[k].requires_grad_(True)
    torch.manual_seed(cases.FWD_SEED)
    outs = hip_forward(name, m, inp)
    ws = cases.probe_weights([o.detach().cpu() for o in outs])
    sum((o * w.to(cuda)).sum() for o, w in zip(outs, ws)).backward()

    # oracle on the CPU with the same weights
    sd2 = {k: v.clone() for k, v in sd.items()}
    pnames = [k for k, p in m.named_parameters() if p.requires_grad]
    for k in pnames:
        sd2[k].requires_grad_(True)
    oin = cases.inputs(name)
    for k in GRAD_INPUTS[name]:
        oin[k] = oin[k].clone().requires_grad_(True)
    torch.manual_seed(cases.FWD_SEED)
    oouts = ORACLE_FWD[name](sd2, oin)
    sum((o * w).sum() for o, w in zip(oouts, ws)).backward()

    # the same oracle in fp64 (identical noise / dropout draws): the yardstick for the gradients ...
    g64, ig64 = _oracle_grads64(name, sd, pnames, ws)
    # ... and its conditioning: how far the fp64 gradients move when inputs and weights change by 1e-6 relative (eight draws, worst taken). Stacks of
    # ReLU / max-pool gates make single gradients jump by 1e-2 when one gate near the top flips; no fp32 implementation can be closer
    # to another one than that, so the bar below is max(1e-4, 2 x the fp32 oracle's own error, 3 x this sensitivity).
    cond, icond = {k: 0.0 for k in pnames}, {k: 0.0 for k in GRAD_INPUTS[name]}
    for trial in range(1, 9):
        gp, igp = _oracle_grads64(name, sd, pnames, ws, perturb=trial)
        for k in pnames:
            if g64[k] is not None and gp[k] is not None:
                cond[k] = max(cond[k], float((gp[k] - g64[k]).norm() / g64[k].norm().clamp_min(1e-300)))
        for k in GRAD_INPUTS[name]:
            icond[k] = max(icond[k], float((igp[k] - ig64[k]).norm() / ig64[k].norm().clamp_min(1e-300)))

    bad = []

    def l2(a, b):
        a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double()
        return float((a - b).norm() / max(float(b.norm()), 1e-300))

    for i, (o, oo) in enumerate(zip(outs, oouts)):
        assert o.shape == oo.shape, "%s out%d shape %s vs %s" % (name, i, tuple(o.shape), tuple(oo.shape))
        for ref, tag in ((oo.detach(), "oracle"), (gold["out%d" % i], "golden")):
            e = rel(o.detach().cpu(), ref)
            if e >= TOL:
                bad.append("out%d vs %s max-rel %.2e" % (i, tag, e))
    for k in GRAD_INPUTS[name]:
        eh, eo = l2(inp[k].grad, ig64[k]), l2(oin[k].grad, ig64[k])
        if eh > max(TOL, 2 * eo, 3 * icond[k]):
            bad.append("d%s: error vs fp64 %.2e (fp32 oracle %.2e, sensitivity %.2e)" % (k, eh, eo, icond[k]))
        # recorded on the build container's CPU by the reference itself: the fp32 reference there must be as close to fp64 as here
        eg = l2(gold["igrad_" + k], ig64[k])
        if eh > max(TOL, 2 * max(eo, eg), 3 * icond[k]):
            bad.append("d%s vs golden: HIP %.2e, reference %.2e" % (k, eh, eg))
    params = dict(m.named_parameters())
    gmax = max(float(v.abs().max()) for v in g64.values() if v is not None)
    realised = []          # (parameter, HIP error vs fp64, fp32 oracle's error, fp64 sensitivity, bound) - written to the parity summary
    for k in pnames:
        g = params[k].grad
        og, od = sd2[k].grad, g64[k]
        if od is not None and float(od.abs().max()) < 1e-6 * gmax:
            # analytically zero gradient (e.g. a conv bias in front of a batch-stat BatchNorm): every fp32 side holds rounding noise
            if g is not None and float(g.abs().max()) > 1e-4 * gmax:
                bad.append("grad %s should be ~0, got %.2e" % (k, float(g.abs().max())))
            continue
        if od is None:
            if not (g is None or float(g.abs().max()) == 0.0):
                bad.append("%s should have no gradient" % k)
            continue
        if g is None:
            bad.append("missing gradient for %s" % k)
            continue
        eh, eo = l2(g, od), l2(og, od)
        realised.append((k, eh, eo, cond[k], max(TOL, 2 * eo, 3 * cond[k])))
        if eh > max(TOL, 2 * eo, 3 * cond[k]):
            bad.append("grad %s: error vs fp64 %.2e (fp32 oracle %.2e, sensitivity %.2e)" % (k, eh, eo, cond[k]))
    if cases.kind(name) == "discriminator":  # spectral-norm u vectors mutate identically
        for k, v in m.state_dict().items():
            if k.endswith("weight_u") and rel(v.cpu(), gold["post_" + k.replace(".", "__")]) >= TOL:
                bad.append("post-forward %s" % k)
    if cases.kind(name) == "hwr":  # BatchNorm running statistics
        for k, v in m.state_dict().items():
            if "running_mean" in k and rel(v.cpu(), gold["post_" + k.replace(".", "__")]) >= TOL:
                bad.append("post-forward %s: %.2e" % (k, rel(v.cpu(), gold["post_" + k.replace(".", "__")])))
    rng.set_mode("device")
    if realised:
        import math
        n_tol = sum(1 for r in realised if r[4] == TOL)
        worst = max(realised, key=lambda r: r[4])
        rms = lambda j: math.sqrt(sum(r[j] ** 2 for r in realised) / len(realised))      # noqa: E731
        line = ("[module parity, %s] %d parameter gradients vs fp64: HIP rms error %.2e (worst %.2e), fp32 oracle rms %.2e; %d held at %.0e, largest realised "
                "bound %.2e (%s: HIP %.2e, oracle %.2e, fp64 sensitivity to 1e-6 perturbations %.2e)" % (
                    name, len(realised), rms(1), max(r[1] for r in realised), rms(2), n_tol, TOL, worst[4], worst[0], worst[1], worst[2], worst[3]))
        print("\n" + line)
        if os.environ.get("HWG_PARITY_SUMMARY"):
            os.makedirs(os.path.dirname(os.path.abspath(os.environ["HWG_PARITY_SUMMARY"])), exist_ok=True)
            with open(os.environ["HWG_PARITY_SUMMARY"], "a") as fh:
                fh.write(line + "\n")
    assert not bad, "%s: %d mismatches: %s" % (name, len(bad), "; ".join(bad[:12]))


def test_discriminator_hinge_step_gradients_vs_fp64(cuda):
    """The `disc` lesson's loss on the full-width discriminator: real lines and generated lines in one batch, hinge loss per head
    (trainer/hw_with_style_trainer.py:797-806). Early in training nearly every hinge term is active, so every bias gradient is the
    difference of two large, almost cancelling sums - the case that punishes sloppy accumulation. HIP vs the fp32 oracle vs fp64."""
    import torch.nn.functional as F
    from handwriting_line_generation_amd import model as M, ops, rng
    rng.set_mode("host")
    try:
        m = M.DiscriminatorAP(64, use_low=True)
        sd = torch_ref.seeded_state_dict(m, 33)
        m.load_state_dict(sd)
        m.train().to(cuda)
        pnames = [k for k, p in m.named_parameters() if p.requires_grad]
        g = torch.Generator().manual_seed(8)
        n_real = 4
        x = torch.rand(2 * n_real, 1, 64, 256, generator=g) * 2 - 1
        torch.manual_seed(cases.FWD_SEED)
        preds = m(x.to(cuda))
        loss = 0
        for p in preds:
            term = ops.add(ops.mean_loss(p[:n_real], ops.LOSS_HINGE_REAL), ops.mean_loss(p[n_real:], ops.LOSS_HINGE_FAKE))
            loss = term if isinstance(loss, int) else ops.add(loss, term)
        ops.scale(loss, 1.0 / len(preds)).backward()

        def hinge(outs):
            return sum(F.relu(1.0 - o[:n_real]).mean() + F.relu(1.0 + o[n_real:]).mean() for o in outs) / len(outs)
        sd2 = {k: v.clone() for k, v in sd.items()}
        for k in pnames:
            sd2[k].requires_grad_(True)
        torch.manual_seed(cases.FWD_SEED)
        l32 = hinge(torch_ref.discriminator(sd2, x))
        l32.backward()
        g64, _ = _oracle_grads64("discriminator", sd, pnames, None, loss_fn=hinge, inputs={"x": x})
        assert abs(float(loss.detach() / len(preds)) - float(l32)) < 1e-5 * abs(float(l32))
        bad, params = [], dict(m.named_parameters())
        for k in pnames:
            if g64[k] is None:
                continue
            nrm = max(float(g64[k].norm()), 1e-300)
            eh = float((params[k].grad.double().cpu() - g64[k]).norm()) / nrm
            eo = float((sd2[k].grad.double() - g64[k]).norm()) / nrm
            if eh > max(TOL, 2 * eo):
                bad.append("%s: HIP %.2e, fp32 oracle %.2e" % (k, eh, eo))
        assert not bad, "; ".join(bad)
    finally:
        rng.set_mode("device")


def test_spectral_weight_images_from_the_scaled_multi_pack_are_bit_identical(cuda):
    """ops.SpectralBank writes every image (direct / mirrored tap order, Winograd domain) of every spectral-norm layer's W_bar / sigma with ONE
    launch per forward pass (hwg_conv_pack_weight_multi_scaled), from the second pass on (the first pass records what the convolutions ask
    for). Against the per-layer scale + pack launches: outputs, input gradient and every parameter gradient of three consecutive
    forward / backward passes bit-identical, u / v updated alike, and the multi-pack really in use."""
    from handwriting_line_generation_amd import model as M, ops, rng
    rng.set_mode("host")
    try:
        runs = []
        for prepack in (False, True):
            m = M.DiscriminatorAP(64, use_low=True)
            m.load_state_dict(torch_ref.seeded_state_dict(m, 33))
            m.train().to(cuda)
            g = torch.Generator().manual_seed(8)
            rec = []
            for it in range(3):
                x = (torch.rand(4, 1, 64, 256 if it < 2 else 192, generator=g) * 2 - 1).to(cuda).requires_grad_(True)
                torch.manual_seed(cases.FWD_SEED + it)
                for p in m.parameters():
                    p.grad = None
                if it == 0:
                    m(x.detach())                              # builds the bank
                    for b in m._sn_banks.values():
                        b.prepack = prepack
                        b.requests.clear(); b._ptab = None
                preds = m(x)
                sum((p * p).sum() for p in preds).backward()
                torch.cuda.synchronize()
                rec.append(([p.detach().clone() for p in preds], x.grad.clone(),
                            {k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None},
                            {k: v.detach().clone() for k, v in m.state_dict().items() if k.endswith(("weight_u", "weight_v"))}))
            used = sum(len(b.requests) for b in m._sn_banks.values())
            runs.append((rec, used))
        (ra, ua), (rb, ub) = runs
        assert ub >= 20, ub                                        # ten layers x (forward + mirrored [+ Winograd]) images
        for it in range(3):
            for a, b in zip(ra[it][0], rb[it][0]):
                assert torch.equal(a, b), "pass %d: outputs differ" % it
            assert torch.equal(ra[it][1], rb[it][1]), "pass %d: input gradient differs" % it
            for k in ra[it][2]:
                assert torch.equal(ra[it][2][k], rb[it][2][k]), "pass %d: gradient of %s differs" % (it, k)
            for k in ra[it][3]:
                assert torch.equal(ra[it][3][k], rb[it][3][k]), "pass %d: %s differs" % (it, k)
    finally:
        rng.set_mode("device")
