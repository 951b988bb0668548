"""List the conv / wgrad calls with K<=2 or C<=2 during one curriculum cycle (which layers hit the direct kernels)."""
import sys, collections, torch, numpy as np, random
sys.path.insert(0, '.')
from handwriting_line_generation_amd import _lib as L, ops, rng
from handwriting_line_generation_amd.harness import build_gan_trainer
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
for it in range(7): tr._train_iteration(it)
seen = collections.Counter()
orig = L.call
def call(name, *a):
    if name in ("hwg_conv_fwd", "hwg_conv_wgrad"):
        d = a[0]._obj
        if d.K <= 2 or d.C <= 2:
            seen[(name, d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.P, d.Q, d.transposed)] += 1
    return orig(name, *a)
L.call = call; ops.L.call = call
for it in range(7, 14): tr._train_iteration(it)
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]): print(v, k)
