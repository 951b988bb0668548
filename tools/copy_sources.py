"""Where do a lesson's device copies come from? Tensor.copy_/clone/to/contiguous/cat are wrapped to count calls on CUDA tensors by calling frame:
python tools/copy_sources.py <lesson index>"""
import sys, collections, traceback, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
les = int(sys.argv[1])
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(64, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = True
for it in range(14): tr._train_iteration(it)
torch.cuda.synchronize()
agg = collections.Counter()
on = [False]


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if on[0]:
            t = a[0] if a and isinstance(a[0], torch.Tensor) else (a[0][0] if a and isinstance(a[0], (list, tuple)) and a[0] else None)
            if isinstance(t, torch.Tensor) and (t.is_cuda or name == "to"):
                fr = [x for x in traceback.extract_stack()[:-1] if "handwriting_line_generation_amd" in x.filename]
                w = fr[-1] if fr else None
                agg[(name, "%s:%d %s" % (w.filename.split("handwriting_line_generation_amd/")[-1], w.lineno, (w.line or "")[:90]) if w else "?")] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for nm in ("copy_", "clone", "to", "contiguous", "cpu", "float", "zero_", "fill_"):
    wrap(torch.Tensor, nm)
wrap(torch, "cat"); wrap(torch, "clone"); wrap(torch, "zeros_like"); wrap(torch, "empty_like")
on[0] = True
tr._train_iteration(14 + les)
on[0] = False
torch.cuda.synchronize()
for (n, where), c in agg.most_common(45):
    print("%4d  %-10s %s" % (c, n, where))
