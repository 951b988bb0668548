import sys, time, torch, numpy as np, random
sys.path.insert(0, '.')
from handwriting_line_generation_amd.harness import build_gan_trainer
torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer(width=256, label_len=12)
for it in range(0, 14):
    torch.cuda.synchronize(); t=time.time()
    log = tr._train_iteration(it)
    torch.cuda.synchronize()
    print(it, tr.curriculum.current_lessons[it % 7], '%.1f ms' % ((time.time()-t)*1e3), {k: round(v, 5) for k, v in log.items()})
