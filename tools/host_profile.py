"""cProfile of the host side of the step (enqueue path only; the GPU runs behind): where the Python time of 21 iterations goes."""
import cProfile, pstats, sys, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(64, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = True
for it in range(14): tr._train_iteration(it)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(14, 35): tr._train_iteration(it)
pr.disable()
tr.flush_log(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(70); st.sort_stats("cumulative").print_stats(90)
