"""Does anything accumulate over a long run? torch's allocated bytes (not the peak) at the same point of the curriculum cycle every 700 steps,
plus the sizes of the host-side caches of ops.py."""
import sys, torch, numpy as np, random, gc
sys.path.insert(0, '.')
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import ops, rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(160, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4200
for it in range(N + 1):
    tr._train_iteration(it)
    if it % 700 == 0:
        tr.flush_log(); torch.cuda.synchronize(); gc.collect()
        caches = {k: len(getattr(ops, k)) for k in ("_conv_plans", "_wgrad_plans", "_set_views", "_ws_cache", "_wgrad_sets_ws") if hasattr(ops, k)}
        print("step %5d  allocated %8.1f MB  reserved %8.1f MB  peak %8.1f MB  caches %s" % (it, torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6,
                                                                                      torch.cuda.max_memory_allocated() / 1e6, caches), flush=True)
