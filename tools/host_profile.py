import sys, time, torch, numpy as np, random, cProfile, pstats
sys.path.insert(0, '.')
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=3)
torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
for it in range(7): tr._train_iteration(it)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)   # run backward on this thread so that cProfile sees it
pr = cProfile.Profile()
t = time.time()
pr.enable()
for it in range(7, 21): tr._train_iteration(it)
torch.cuda.synchronize()
pr.disable()
print('wall per step ms', (time.time() - t) / 14 * 1e3)
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(45)
