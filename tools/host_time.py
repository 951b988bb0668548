"""Host side of the step: wall time per step, process CPU time per step and, per lesson, the host time spent ENQUEUEING (time until
`_train_iteration` returns minus the time the host spent blocked on the GPU inside it: event waits and device->host reads) against the GPU
time of the lesson (HIP events). The step is host-bound where enqueue time exceeds GPU time."""
import os, sys, time, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng, replay
import os
if os.environ.get('HOST_TIME_REPLAY', '1') != '0':
    replay.enable()          # (what train.py and bench.py run with)
WARM = int(os.environ.get('HOST_TIME_WARM', '84'))
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(80, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = int(os.environ.get('HWG_LOG_LAG', '2'))      # (bench.py's default)
blocked = [0.0]


where = {}      # HOST_TIME_WHERE=1: blocked time by call site (the innermost frames inside the package)


def timed(fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            blocked[0] += dt
            if os.environ.get('HOST_TIME_WHERE'):
                import traceback
                fr = [f for f in traceback.extract_stack()[:-1] if 'handwriting_line_generation_amd' in f.filename][-3:]
                key = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(fr))
                where[key] = where.get(key, 0.0) + dt
    return wrapper


torch.cuda.Event.synchronize = timed(torch.cuda.Event.synchronize)
_cpu = torch.Tensor.cpu
torch.Tensor.cpu = lambda self, *a, **k: (timed(_cpu)(self, *a, **k) if self.is_cuda else _cpu(self, *a, **k))
torch.Tensor.item = timed(torch.Tensor.item)
for it in range(WARM): tr._train_iteration(it)
torch.cuda.synchronize()
w0, c0 = time.perf_counter(), time.process_time()
N = 42
host = [0.0] * 7
wait = [0.0] * 7
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
evs[0].record()
for i, it in enumerate(range(WARM, WARM + N)):
    t0 = time.perf_counter(); blocked[0] = 0.0
    tr._train_iteration(it)
    host[it % 7] += time.perf_counter() - t0 - blocked[0]
    wait[it % 7] += blocked[0]
    evs[i + 1].record()
tr.flush_log(); torch.cuda.synchronize()
w1, c1 = time.perf_counter(), time.process_time()
print("wall %.2f ms/step, process CPU %.2f ms/step" % ((w1 - w0) / N * 1e3, (c1 - c0) / N * 1e3))
gpu = [0.0] * 7
for i in range(N):
    gpu[(WARM + i) % 7] += evs[i].elapsed_time(evs[i + 1])
for l in range(7):
    print("lesson %d: host enqueue %.2f ms (+ %.2f ms blocked on the GPU), GPU (event to event) %.2f ms" % (l, host[l] / (N / 7) * 1e3, wait[l] / (N / 7) * 1e3, gpu[l] / (N / 7)))
print("host enqueue %.2f ms/step, blocked %.2f ms/step" % (sum(host) / N * 1e3, sum(wait) / N * 1e3))
if where:
    for k, v in sorted(where.items(), key=lambda kv: -kv[1])[:8]:
        print("blocked %.2f ms/step at %s" % (v / (WARM + N) * 1e3, k))
