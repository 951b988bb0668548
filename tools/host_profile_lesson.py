"""cProfile of the host side of ONE lesson kind (argument: the lesson's position(s) in the 7-lesson cycle, e.g. "1,4" = the gen lessons):
where the main thread's time goes in the lessons that are host-bound (tools/host_time.py)."""
import cProfile, pstats, sys, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(80, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = 2
which = set(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,4").split(","))
for it in range(14): tr._train_iteration(it)
torch.cuda.synchronize()
pr = cProfile.Profile()
n = 0
for it in range(14, 14 + 70):
    if it % 7 in which:
        torch.cuda.synchronize()      # (the GPU idle at the start: the profile shows pure enqueue time)
        pr.enable(); tr._train_iteration(it); pr.disable(); n += 1
    else:
        tr._train_iteration(it)
tr.flush_log(); torch.cuda.synchronize()
print("lessons profiled:", n)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
