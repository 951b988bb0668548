import sys, time, torch, numpy as np, random
sys.path.insert(0, '.')
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
B, A, W, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng.set_mode('device', seed=3)
torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', B, A, width=W, label_len=L)
print('built', flush=True)
for it in range(0, 14):
    torch.cuda.synchronize(); t = time.time()
    log = tr._train_iteration(it)
    torch.cuda.synchronize()
    print(it, tr.curriculum.current_lessons[it % 7], '%.1f ms' % ((time.time() - t) * 1e3), {k: round(v, 4) for k, v in log.items() if k not in ('CER', 'WER')}, flush=True)
