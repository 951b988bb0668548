"""cProfile of the host side of the training step (tools/host_time.py says how much of the step is host enqueue time; this says where it goes):
top functions by own time over 6 curriculum cycles, autograd-engine thread excluded (cProfile sees the calling thread only - the backward
passes of taped sub-networks run on it, the autograd engine's do not)."""
import cProfile, pstats, sys, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
b, a = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 2)
tr, cfg = build_gan_trainer('iam_gan', b, a, width=512, label_len=30)
tr.data_loader.make_resident(80, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = 2
for it in range(14): tr._train_iteration(it)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(14, 56): tr._train_iteration(it)
pr.disable()
tr.flush_log(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(140)
st.sort_stats("cumulative").print_stats(60)
