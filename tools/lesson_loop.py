"""Run only some lessons of the 7-lesson GAN curriculum in a loop (argument: positions in the cycle, e.g. "2,5" = the auto lessons), for a
per-lesson-kind kernel view: `rocprofv3 --kernel-trace --stats -- python3 tools/lesson_loop.py 2,5 [cycles]` (tools/launch_census.py reads the
stats CSV). Prints wall time per lesson."""
import sys, time, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng, replay
replay.enable()
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(80, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = 2
which = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,5").split(",")]
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for it in range(14): tr._train_iteration(it)        # two whole cycles first: every network has stepped, caches are warm
for c in range(4):
    for w in which: tr._train_iteration(7 * (2 + c) + w)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for c in range(cycles):
    for w in which:
        tr._train_iteration(7 * (6 + c) + w); n += 1
tr.flush_log(); torch.cuda.synchronize()
print("lessons %s: %.3f ms per lesson over %d lessons" % (which, (time.perf_counter() - t0) / n * 1e3, n))
