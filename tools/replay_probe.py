import sys, time, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng, replay, ops
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
hwr = tr.model.hwr
x = torch.rand(8, 1, 64, 512, device=tr.gpu) * 2 - 1
for side in (False, True):
  for defer in (False, True):
    ops.SIDE_WGRAD = side; ops.DEFER_REDUCE = defer
    for mode in ("eager", "replay", "eager", "replay"):
        replay.ENABLED = mode == "replay"
        def step():
            xi = x.clone().requires_grad_(True)
            y = hwr(xi)
            y.backward(torch.ones_like(y) * 1e-3)
            ops.join_side_stream()
        for _ in range(6): step()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(20): step()
        e1.record(); th = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("side %d defer %d %-6s GPU %.3f ms/pass, host enqueue %.3f ms/pass  %s" % (side, defer, mode, e0.elapsed_time(e1) / 20, th / 20 * 1e3, dict(replay.STATS)), flush=True)
