"""GPU diagnostic: the discriminator inside the trainer's `disc` lesson vs the oracle on the very same input / masks."""
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_ref  # noqa: E402
from handwriting_line_generation_amd import rng  # noqa: E402
from handwriting_line_generation_amd.harness import build_gan_trainer, load_config  # noqa: E402
from handwriting_line_generation_amd.model import Autoencoder, HWWithStyle  # noqa: E402

cfg_model = dict(load_config("iam_gan")["model"], pretrained_hwr=None)
msd = torch_ref.seeded_state_dict(HWWithStyle(cfg_model), 21)
esd = torch_ref.seeded_state_dict(Autoencoder({"type": "2tight", "hwr": 80}), 22)
rng.set_mode("host")
trainer, cfg = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=tempfile.mkdtemp(), model_state=msd, encoder_state=esd,
                                 curriculum=[["disc"]])
torch.manual_seed(0); np.random.seed(0); random.seed(0)
cap = {}


def hook(mod, args):
    cap["x"] = args[0].detach().cpu().clone()
    cap["rng"] = torch.get_rng_state()


trainer.model.discriminator.register_forward_pre_hook(hook)
grads = {}
trainer.pre_clip_hook = lambda it: grads.update({k: p.grad.detach().cpu().double().clone() for k, p in trainer.model.discriminator.named_parameters() if p.requires_grad})
dsd = {k[len("discriminator."):]: v.clone() for k, v in msd.items() if k.startswith("discriminator.")}
log = trainer._train_iteration(0)
print("loss", log, "D input", tuple(cap["x"].shape), "fake part std over width of row 10:", float(cap["x"][4, 0, 10].std()))
x = cap["x"]
n_real = x.shape[0] // 2


def hinge(outs):
    return sum(F.relu(1.0 - o[:n_real]).mean() + F.relu(1.0 + o[n_real:]).mean() for o in outs) / len(outs)


pnames = list(grads)
res = {}
for dt in (torch.float32, torch.float64):
    sd2 = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in dsd.items()}
    for k in pnames:
        sd2[k].requires_grad_(True)
    torch.set_rng_state(cap["rng"])
    l = hinge(torch_ref.discriminator(sd2, x.to(dt)))
    l.backward()
    res[dt] = (float(l.detach()), {k: sd2[k].grad for k in pnames})
print("oracle loss fp32 %.8f fp64 %.8f" % (res[torch.float32][0], res[torch.float64][0]))
for k in pnames:
    g64 = res[torch.float64][1][k]
    nrm = max(float(g64.norm()), 1e-300)
    eh = float((grads[k] - g64).norm()) / nrm
    eo = float((res[torch.float32][1][k].double() - g64).norm()) / nrm
    print("   %-36s HIP %.2e  fp32 oracle %.2e  |g| %.3e%s" % (k, eh, eo, nrm, " <<<" if eh > max(1e-4, 2 * eo) else ""))

dbg = torch.load(os.path.join(ROOT, "tests", "golden", "_dbg_disc.pt"))
xr = dbg["x"]
d = (x - xr).abs()
print("x_hip vs x_ref: max abs diff %.3e, rel l2 %.3e, per-sample max %s" % (float(d.max()), float((x - xr).norm() / xr.norm()), [float(d[i].max()) for i in range(x.shape[0])]))
col = d.amax(dim=(0, 1, 2))
print("columns with diff > 1e-5:", (col > 1e-5).nonzero().flatten().tolist()[:40])
print("rng states equal:", bool((cap["rng"] == dbg["rng"]).all()))
for k in pnames:
    gr = dbg["grads"][k].double()
    print("   %-36s HIP-vs-reftrainer %.2e" % (k, float((grads[k] - gr).norm() / gr.norm().clamp_min(1e-300))))
