"""Run one curriculum cycle of the GAN trainer on unusual batch geometries (robustness check, not a benchmark)."""
import sys, torch, numpy as np, random, tempfile, math
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=1)
combos = [("iam_gan", 1, 1, 200, 7, None), ("iam_gan", 2, 1, 344, 23, None), ("iam_gan", 3, 2, 472, 31, None), ("iam_gan", 1, 3, 128, 3, None),
          ("rimes_gan", 2, 2, 1000, 37, 264), ("rimes_gan", 1, 1, 256, 5, None), ("iam_gan", 5, 1, 600, 40, 304)]
for which, b, a, w, ll, mw in combos:
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    tr, cfg = build_gan_trainer(which, b, a, width=w, label_len=ll, min_width=mw, workdir=tempfile.mkdtemp())
    bad = None
    for it in range(7):
        log = tr._train_iteration(it)
        for k, v in log.items():
            if isinstance(v, float) and not math.isfinite(v): bad = (it, k, v)
    torch.cuda.synchronize()
    print(which, b, a, w, ll, mw, "OK" if bad is None else "NONFINITE %s" % (bad,), flush=True)
