"""Which ATen ops still run on the GPU during a curriculum cycle (everything else is libhwg_hip.so), grouped by op and input shapes."""
import sys, torch, numpy as np, random
sys.path.insert(0, '.')
from torch.profiler import profile, ProfilerActivity
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
for it in range(7): tr._train_iteration(it)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    for it in range(7, 14): tr._train_iteration(it)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="count", row_limit=45, max_name_column_width=40, max_shapes_column_width=60))
