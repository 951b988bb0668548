"""Which round-6 change moved the count-lesson recogniser gradients (tests/test_pipeline_gpu.py::test_count_lesson_recogniser_gradients_and_gate_flips)?
Per-tensor error vs fp64 of the planner's schedule, with the Python-level changes toggled one at a time."""
import sys, torch
sys.path.insert(0, '.')
from oracle import torch_ref
from handwriting_line_generation_amd import ops, rng
from handwriting_line_generation_amd.harness import load_config
from handwriting_line_generation_amd.model import HWWithStyle
dev = torch.device("cuda:0")
cfg = dict(load_config("iam_gan")["model"], pretrained_hwr=None)
model = HWWithStyle(cfg)
sd = torch_ref.seeded_state_dict(model, 21)
model.load_state_dict(sd); model.to(dev); model.train()
pnames = {k for k, _ in model.named_parameters()}
rng.set_mode("host")
g = torch.Generator().manual_seed(3)
B, A, W = 4, 2, 256
image = torch.rand(B, 1, 64, W, generator=g) * 2 - 1
wsty = torch.randn(B // A, 128, generator=g)
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
def oracle(dt):
    s = {}
    for k, v in sd.items():
        t = v.detach().clone().to(dt) if v.dtype.is_floating_point else v.clone()
        if k.startswith("hwr.") and v.dtype.is_floating_point and k in pnames: t.requires_grad_(True)
        s[k] = t
    img = image.to(dt)
    pred = torch_ref.hwr(s, img, prefix="hwr.")
    T = pred.shape[0]
    ci = img.reshape(B // A, A, 64, W).permute(0, 2, 1, 3).reshape(B // A, 1, 64, A * W)
    cr = pred.permute(1, 2, 0).reshape(B // A, A, pred.shape[2], T).permute(0, 2, 1, 3).reshape(B // A, pred.shape[2], A * T)
    style = torch_ref.style_extractor(s, ci, cr, prefix="style_extractor.")
    (style * wsty.to(dt)).sum().backward()
    return {k: s[k].grad for k in s if k.startswith("hwr.") and s[k].grad is not None and float(s[k].grad.norm()) > 1e-12}
g64 = oracle(torch.float64)
def hip(tag):
    for p in model.parameters(): p.grad = None
    model.pred = None
    style = model.extract_style(image.to(dev), None, A)
    (style.view(B // A, A, 128)[:, 0] * wsty.to(dev)).sum().backward()
    torch.cuda.synchronize()
    errs = {k: rel(p.grad, g64[k]) for k, p in model.named_parameters() if k in g64 and p.grad is not None}
    pooled = (sum(v * v for v in errs.values()) / len(errs)) ** 0.5
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print("%-40s pooled %.3e  worst %s" % (tag, pooled, ["%s %.2e" % (k[-32:], v) for k, v in worst]), flush=True)
hip("as shipped")
ops.ONEROW_TWIN_DEAD_ROWS = False
hip("one-row twin only for H == R")
ops.ONEROW_TWIN_DEAD_ROWS = True
orig_mp = ops.max_pool2d
ops.max_pool2d = lambda x, kernel, stride=None, padding=0, relu=False: (ops.bias_act(orig_mp(x, kernel, stride, padding), None, None, ops.ACT_RELU) if relu else orig_mp(x, kernel, stride, padding))
hip("ReLU as a pass of its own after the pools")
ops.max_pool2d = orig_mp
for env in ({"HWG_C1_ROWS": "0"}, {"HWG_WGRAD_C1_ROWS": "0"}, {"HWG_NORM_BIG": "0"}, {"HWG_WINO": "0"}):
    with ops.tuning(**env):
        hip(str(env))
