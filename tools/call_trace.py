"""Ordered list of the C-ABI calls (and the torch-side device ops between them) of ONE curriculum cycle of the bench workload, per lesson:
what runs next to what - the input of every fusion decision (a launch census says how often, not in which order).

  python tools/call_trace.py [out.txt] [workload-batch "4,2"]         (GPU box)

Every line: lesson, index, entry point (or `aten::op` for a torch-level op that launches device work), shapes of the tensor arguments."""
import sys, random
import numpy as np, torch
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd import _lib as L, ops, rng, replay
from handwriting_line_generation_amd.harness import build_gan_trainer

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/call_trace.txt"
bs, abs_ = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4,2").split(",")]
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', bs, abs_, width=512, label_len=30)
tr.data_loader.make_resident(40, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = 2
for it in range(14):
    tr._train_iteration(it)
torch.cuda.synchronize()

log = []
orig = L.call


def traced(name, *args):
    log.append((name, [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]))
    return orig(name, *args)


from torch.utils._python_dispatch import TorchDispatchMode


class Aten(TorchDispatchMode):
    SKIP = ("aten::view", "aten::_unsafe_view", "aten::empty", "aten::as_strided", "aten::detach", "aten::slice", "aten::select", "aten::reshape", "aten::t",
            "aten::permute", "aten::expand", "aten::alias", "aten::unsqueeze", "aten::squeeze", "aten::transpose", "aten::empty_like", "aten::empty_strided",
            "aten::_local_scalar_dense", "aten::is_pinned", "aten::lift_fresh", "aten::new_empty", "aten::split", "aten::unbind", "aten::view_as")

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        r = func(*args, **(kwargs or {}))
        if not name.startswith(self.SKIP):
            ts = [a for a in list(args) + [r] if isinstance(a, torch.Tensor)]
            if any(t.is_cuda for t in ts):
                log.append((name, [tuple(t.shape) for t in ts if t.is_cuda]))
        return r


L.call = traced
names = ("count", "gen", "auto", "disc", "gen", "auto", "disc")
with open(out, "w") as fh, Aten():
    for k in range(7):
        del log[:]
        tr._train_iteration(14 + k)
        fh.write("==== lesson %d:%s  (%d calls)\n" % (k, names[k], len(log)))
        for i, (n, shapes) in enumerate(log):
            fh.write("%d %4d %-34s %s\n" % (k, i, n, " ".join(str(s) for s in shapes)))
L.call = orig
torch.cuda.synchronize()
print("wrote", out)
