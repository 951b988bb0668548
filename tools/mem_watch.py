"""Does anything accumulate over a long run? torch's allocated bytes (not the peak) at the same point of the curriculum cycle every 700 steps, the throughput of each 700-step window,
plus the sizes of the host-side caches of ops.py."""
import sys, time, torch, numpy as np, random, gc
sys.path.insert(0, '.')
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import ops, rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(160, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4200
TRIM = len(sys.argv) > 2 and sys.argv[2] == "trim"      # release the allocator's cached blocks at every mark (fragmentation experiment)
widths = [0, 0]
_gen_forward = tr.model.generator.forward


def _gen_fwd(content, style, *a, **k):      # generated line width (4 x the spaced text length): drifts as the spacer trains
    widths[0] += content.shape[0] if content.dim() == 3 else content.shape[2]; widths[1] += 1
    return _gen_forward(content, style, *a, **k)


tr.model.generator.forward = _gen_fwd
blocked = [0.0]
_sync = torch.cuda.Event.synchronize


def _timed_sync(self):
    t = time.perf_counter()
    try:
        return _sync(self)
    finally:
        blocked[0] += time.perf_counter() - t


torch.cuda.Event.synchronize = _timed_sync      # host time blocked on the GPU (log read-back, insert_spaces) is reported per window
t0 = time.perf_counter()
for it in range(N + 1):
    tr._train_iteration(it)
    if it % 700 == 0:
        tr.flush_log(); torch.cuda.synchronize()
        dt, t0 = time.perf_counter() - t0, time.perf_counter()
        gc.collect()
        blk, blocked[0] = blocked[0], 0.0
        wsum = list(widths); widths[0] = widths[1] = 0
        if TRIM:
            torch.cuda.empty_cache()
        caches = {k: len(getattr(ops, k)) for k in ("_conv_plans", "_wgrad_plans", "_set_views", "_ws_cache", "_wgrad_sets_ws") if hasattr(ops, k)}
        print("step %5d  %6.1f steps/s  host blocked %5.2f ms/step  mean generated T %6.1f  allocated %8.1f MB  reserved %8.1f MB  peak %8.1f MB  caches %s" % (it, (700 if it else 1) / dt, blk / (700 if it else 1) * 1e3, wsum[0] / max(wsum[1], 1), torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6,
                                                                                      torch.cuda.max_memory_allocated() / 1e6, caches), flush=True)
