"""GPU diagnostic: discriminator hinge-step gradients, HIP vs fp32 oracle vs fp64 oracle, for several kinds of 'fake' images."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cases, torch_ref  # noqa: E402
from handwriting_line_generation_amd import model as M, ops, rng  # noqa: E402

dev = torch.device("cuda:0")


def run(kind, dim=64, W=256, n_real=4):
    rng.set_mode("host")
    m = M.DiscriminatorAP(dim, use_low=True)
    sd = torch_ref.seeded_state_dict(m, 33)
    m.load_state_dict(sd)
    m.train().to(dev)
    pnames = [k for k, p in m.named_parameters() if p.requires_grad]
    g = torch.Generator().manual_seed(8)
    real = torch.rand(n_real, 1, 64, W, generator=g) * 2 - 1
    if kind == "noise":
        fake = torch.rand(n_real, 1, 64, W, generator=g) * 2 - 1
    elif kind == "smooth":
        z = torch.randn(n_real, 1, 8, W // 8, generator=g)
        fake = torch.tanh(F.interpolate(z, size=(64, W), mode="bilinear"))
    elif kind == "padded":     # what an untrained spacer/generator produce: a narrow image, replicate-padded to the width of the real lines
        z = torch.randn(n_real, 1, 8, 8, generator=g)
        narrow = torch.tanh(F.interpolate(z, size=(64, 60), mode="bilinear"))
        fake = F.pad(narrow, (0, W - 60, 0, 0), mode="replicate")
    elif kind == "flat":
        fake = torch.tanh(0.05 * torch.randn(n_real, 1, 64, W, generator=g) + 0.3)
    x = torch.cat([real, fake], 0)
    torch.manual_seed(cases.FWD_SEED)
    preds = m(x.to(dev))
    loss = 0
    for p in preds:
        term = ops.add(ops.mean_loss(p[:n_real], ops.LOSS_HINGE_REAL), ops.mean_loss(p[n_real:], ops.LOSS_HINGE_FAKE))
        loss = term if isinstance(loss, int) else ops.add(loss, term)
    ops.scale(loss, 1.0 / len(preds)).backward()

    def hinge(outs):
        return sum(F.relu(1.0 - o[:n_real]).mean() + F.relu(1.0 + o[n_real:]).mean() for o in outs) / len(outs)
    res = {}
    for dt in (torch.float32, torch.float64):
        sd2 = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        for k in pnames:
            sd2[k].requires_grad_(True)
        torch.manual_seed(cases.FWD_SEED)
        l = hinge(torch_ref.discriminator(sd2, x.to(dt)))
        l.backward()
        res[dt] = (float(l), {k: sd2[k].grad for k in pnames})
    print("== %s: loss HIP %.8f  fp32 %.8f  fp64 %.8f" % (kind, float(loss.detach()) / len(preds), res[torch.float32][0], res[torch.float64][0]))
    params = dict(m.named_parameters())
    for k in pnames:
        g64 = res[torch.float64][1][k]
        if g64 is None:
            continue
        nrm = max(float(g64.norm()), 1e-300)
        eh = float((params[k].grad.double().cpu() - g64).norm()) / nrm
        eo = float((res[torch.float32][1][k].double() - g64).norm()) / nrm
        flag = " <<<" if eh > max(1e-4, 2 * eo) else ""
        print("   %-36s HIP %.2e  fp32 oracle %.2e  |g| %.3e%s" % (k, eh, eo, nrm, flag))
    rng.set_mode("device")


for kind in sys.argv[1:] or ["noise", "smooth", "flat"]:
    run(kind)
