"""Which framework (torch) calls still launch device work in the training step, and from which line of this package: the tensor methods and
factory functions that launch copy / fill / add kernels are wrapped with a call-site counter (host-side; what the autograd engine launches
itself - AccumulateGrad's add_, gradient buffers' copies - is the remainder against the kernel trace's at::native / copyBuffer counts)."""
import sys, collections, traceback, torch, numpy as np, random
sys.path.insert(0, '.')
torch.set_num_threads(1)
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd import rng
rng.set_mode('device', seed=3); torch.manual_seed(0); np.random.seed(0); random.seed(0)
tr, cfg = build_gan_trainer('iam_gan', 4, 2, width=512, label_len=30)
tr.data_loader.make_resident(80, tr.gpu); tr.data_loader_iter = iter(tr.data_loader); tr.async_log = 2
for it in range(14): tr._train_iteration(it)
torch.cuda.synchronize()
agg = collections.Counter()
on = [False]


def site():
    fr = [f for f in traceback.extract_stack()[:-2] if "handwriting_line_generation_amd/" in f.filename]
    return " <- ".join("%s:%d" % (f.filename.split("handwriting_line_generation_amd/")[-1], f.lineno) for f in reversed(fr[-3:]))


def wrap_method(name):
    orig = getattr(torch.Tensor, name)

    def w(self, *a, **k):
        if on[0] and (self.is_cuda or any(torch.is_tensor(x) and x.is_cuda for x in a)):
            agg[(name, site())] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, w)


def wrap_fn(name):
    orig = getattr(torch, name)

    def w(*a, **k):
        r = orig(*a, **k)
        if on[0] and torch.is_tensor(r) and r.is_cuda:
            agg[("torch." + name, site())] += 1
        return r
    setattr(torch, name, w)


for m in ("copy_", "fill_", "zero_", "add_", "add", "__add__", "__mul__", "mul", "mul_", "clone", "to", "cuda", "float", "sum", "mean", "__setitem__", "index_select", "repeat"):
    wrap_method(m)
_contig = torch.Tensor.contiguous


def contiguous(self, *a, **k):
    if on[0] and self.is_cuda and not self.is_contiguous():
        agg[("contiguous(copy)", site())] += 1
    return _contig(self, *a, **k)


torch.Tensor.contiguous = contiguous
# gradients RETURNED to autograd for leaf parameters (each costs the engine an add_ / copy_ into param.grad)
from handwriting_line_generation_amd import ops


def all_subclasses(c):
    for s_ in c.__subclasses__():
        yield s_
        yield from all_subclasses(s_)


for cls in set(all_subclasses(ops.Function)):
    if "forward" not in cls.__dict__ or "backward" not in cls.__dict__:
        continue
    f0, b0 = cls.__dict__["forward"].__func__, cls.__dict__["backward"].__func__

    def fwd(ctx, *a, _f=f0):
        ctx._leaf_pos = [i for i, x in enumerate(a) if isinstance(x, torch.nn.Parameter)]
        return _f(ctx, *a)

    def bwd(ctx, *g, _b=b0, _n=cls.__name__):
        r = _b(ctx, *g)
        if on[0]:
            rr = r if isinstance(r, tuple) else (r,)
            for i in getattr(ctx, "_leaf_pos", ()):
                if i < len(rr) and rr[i] is not None:
                    agg[("returned param grad", "%s arg %d" % (_n, i))] += 1
        return r
    cls.forward = staticmethod(fwd); cls.backward = staticmethod(bwd)
for f in ("cat", "zeros", "zeros_like", "ones", "ones_like", "full", "stack", "arange", "tensor", "as_tensor", "randn", "rand"):
    wrap_fn(f)
STEPS = 14
on[0] = True
for it in range(14, 14 + STEPS): tr._train_iteration(it)
on[0] = False
tr.flush_log(); torch.cuda.synchronize()
print("%-16s %7s  %s" % ("call", "/step", "call site (innermost first)"))
for key, n in sorted(agg.items(), key=lambda kv: -kv[1])[:80]:
    print("%-16s %7.2f  %s" % (key[0], n / STEPS, key[1]))
