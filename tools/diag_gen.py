"""GPU diagnostic: generator forward, HIP vs fp32 oracle vs fp64 oracle (same noise)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, torch_ref  # noqa: E402
from handwriting_line_generation_amd import model as M, ops, rng  # noqa: E402

dev = torch.device("cuda:0")
rng.set_mode("host")
G = M.SpacedGenerator(80, 128, 256, n_style_trans=6, append_style=True)
sd = torch_ref.seeded_state_dict(G, 21)
G.load_state_dict(sd)
G.train().to(dev)
g = torch.Generator().manual_seed(5)
T, B = 61, 4
idx = torch.randint(0, 80, (T, B), generator=g)
content = F.one_hot(idx, 80).float()
style = torch.randn(B, 128, generator=g)
for wino in (True, False):
    ops.WINOGRAD = wino
    torch.manual_seed(77)
    y = G(content.to(dev), style.to(dev)).detach().cpu().double()
    outs = {}
    for dt in (torch.float32, torch.float64):
        sd2 = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        rl = torch.randn_like
        if dt == torch.float64:
            torch.randn_like = lambda t, **k: rl(t.float(), **k).double()
        torch.manual_seed(77)
        outs[dt] = torch_ref.generator(sd2, content.to(dt), style.to(dt)).double()
        torch.randn_like = rl
    r = outs[torch.float64]
    e = lambda a: (float((a - r).norm() / r.norm()), float((a - r).abs().max() / r.abs().max()))
    print("winograd=%s  HIP vs fp64: l2 %.2e max %.2e   fp32 oracle vs fp64: l2 %.2e max %.2e   (|y| rms %.3f, std over pixels %.3e)" %
          ((wino,) + e(y) + e(outs[torch.float32]) + (float(r.pow(2).mean().sqrt()), float(r.std()))))
