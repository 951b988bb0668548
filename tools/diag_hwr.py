"""GPU: recogniser module, Winograd on vs off: output and per-parameter gradient differences."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cases, torch_ref  # noqa: E402
from handwriting_line_generation_amd import model as M, ops, rng  # noqa: E402

dev = torch.device("cuda:0")
m = M.CNNOnlyHWR(**cases.CASES["hwr"]["ctor"])
sd = torch_ref.seeded_state_dict(m, cases.CASES["hwr"]["wseed"])
m.load_state_dict(sd)
m.train().to(dev)
img = cases.inputs("hwr")["image"].to(dev)
res = []
for flag in (False, True):
    ops.WINOGRAD = flag
    m.load_state_dict(sd)
    m.zero_grad()
    x = img.clone().requires_grad_(True)
    y = m(x, None)
    w = cases.probe_weights([y.detach().cpu()])[0].to(dev)
    (y * w).sum().backward()
    res.append((y.detach().double(), x.grad.double(), {k: p.grad.double().clone() for k, p in m.named_parameters()}))
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-300))
print("y: %.2e   dimage: %.2e" % (rel(res[1][0], res[0][0]), rel(res[1][1], res[0][1])))
for k in res[0][2]:
    print("  %-28s %.2e   |g| %.3e" % (k, rel(res[1][2][k], res[0][2][k]), float(res[0][2][k].norm())))
