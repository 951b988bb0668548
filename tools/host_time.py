"""wall vs process CPU time per step (GIL-bound Python: CPU time ~ host critical path), and the step time with the GPU work removed
from the critical path is approximated by the CPU time."""
import sys, time, torch, numpy as np, random
sys.path.insert(0, '.')
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(64, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = True
for it in range(14): tr._train_iteration(it)
torch.cuda.synchronize()
w0, c0 = time.perf_counter(), time.process_time()
N = 42
for it in range(14, 14 + N): tr._train_iteration(it)
tr.flush_log(); torch.cuda.synchronize()
w1, c1 = time.perf_counter(), time.process_time()
print("wall %.2f ms/step, process CPU %.2f ms/step" % ((w1 - w0) / N * 1e3, (c1 - c0) / N * 1e3))
