"""Per-lesson host enqueue time vs GPU span (events) - which lessons are host-bound."""
import sys, time, torch, numpy as np, random, collections
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(64, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = True
for it in range(14): tr._train_iteration(it)
torch.cuda.synchronize()
names = "count,gen,auto,disc,gen,auto,disc".split(",")
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for it in range(14, 14 + 42):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); t0 = time.perf_counter()
    tr._train_iteration(it)
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
    a = acc[(it % 7, names[it % 7])]; a[0] += (t1 - t0) * 1e3; a[1] += e0.elapsed_time(e1); a[2] += 1
tot_h = tot_g = 0
for k in sorted(acc):
    h, g, n = acc[k]; tot_h += h / n; tot_g += g / n
    print("lesson %d %-6s host enqueue %.2f ms   gpu span %.2f ms" % (k[0], k[1], h / n, g / n))
print("per-step mean: host %.2f ms, gpu span %.2f ms" % (tot_h / 7, tot_g / 7))
