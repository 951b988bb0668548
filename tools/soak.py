"""Soak: N training iterations, report loss finiteness, memory high-water marks and the packed-weight / pointer-table cache sizes."""
import sys, time, math, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(64, tr.gpu); tr.data_loader_iter = iter(tr.data_loader)
bad = 0; t0 = time.time(); marks = []
for it in range(n):
    log = tr._train_iteration(it)
    bad += sum(1 for v in log.values() if isinstance(v, float) and not math.isfinite(v))
    if it % 100 == 99:
        torch.cuda.synchronize()
        marks.append((it + 1, torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, len(ops._pack_cache)))
print("iterations", n, "non-finite log values", bad, "wall %.1f s" % (time.time() - t0))
for m in marks: print("it %d: allocated %d MiB, peak %d MiB, reserved %d MiB, pack cache entries %d" % m)
print("last log", {k: round(v, 4) for k, v in log.items() if isinstance(v, float)})
