"""GPU: Winograd vs direct engine on many 3x3 geometries (forward, data gradient), relative L2 error; prints the worst cases."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from handwriting_line_generation_amd import ops  # noqa: E402

os.environ["HWG_WINO"] = "2"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
shapes = [(2, 32, 32, 64, 128, 1, 1), (2, 16, 16, 128, 256, 1, 1), (2, 16, 16, 256, 256, 1, 1), (2, 8, 17, 256, 512, 1, 1), (2, 8, 17, 512, 512, 0, 0),
          (2, 3, 16, 512, 512, 0, 0), (2, 64, 64, 64, 64, 1, 1), (4, 6, 15, 512, 512, 1, 1)]
for _ in range(60):
    N = int(torch.randint(1, 5, (1,), generator=g)); H = int(torch.randint(3, 40, (1,), generator=g)); W = int(torch.randint(3, 70, (1,), generator=g))
    C = 16 * int(torch.randint(1, 33, (1,), generator=g)); K = 16 * int(torch.randint(1, 33, (1,), generator=g))
    ph = int(torch.randint(0, 3, (1,), generator=g)); pw = int(torch.randint(0, 3, (1,), generator=g))
    shapes.append((N, H, W, C, K, ph, pw))
rows = []
for (N, H, W, C, K, ph, pw) in shapes:
    if H + 2 * ph < 3 or W + 2 * pw < 3:
        continue
    x = torch.randn(N, H, W, C, generator=g).to(dev)
    w = (torch.randn(K, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev)
    outs = []
    for flag in (True, False):
        ops.WINOGRAD = flag
        xg = x.clone().requires_grad_(True)
        y = ops.conv2d(xg, w, None, 1, (ph, pw))
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        y.backward(gy)
        outs.append((y.detach().double(), xg.grad.double()))
    ops.WINOGRAD = True
    ey = float((outs[0][0] - outs[1][0]).norm() / outs[1][0].norm())
    ex = float((outs[0][1] - outs[1][1]).norm() / outs[1][1].norm())
    rows.append((max(ey, ex), ey, ex, (N, H, W, C, K, ph, pw)))
rows.sort(reverse=True)
for r in rows[:12]:
    print("worst %.2e  y %.2e  dx %.2e  %s" % r)
print("cases:", len(rows), " all below 1e-5:", all(r[0] < 1e-5 for r in rows))
