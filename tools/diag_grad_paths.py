"""Diagnostic (GPU box): two gradient paths the teacher-forced trainer parity flagged, each isolated against the oracle in fp32 and fp64.
 (A) recogniser parameter gradients when the loss reaches the recogniser only through the style extractor's `recog` input (count lesson)
 (B) the discriminator's gradient with respect to its input image (adversarial gradient that the generator receives)"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref
from handwriting_line_generation_amd import rng, ops
from handwriting_line_generation_amd.harness import load_config
from handwriting_line_generation_amd.model import HWWithStyle

torch.set_num_threads(16)
dev = torch.device('cuda:0')
cfg = dict(load_config("iam_gan")["model"], pretrained_hwr=None)
model = HWWithStyle(cfg)
sd = torch_ref.seeded_state_dict(model, 21)
model.load_state_dict(sd); model.to(dev); model.train()
rng.set_mode("host")


def rel(a, b):
    return float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def leaf(sd, dtype, prefix):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone().to(dtype) if v.dtype.is_floating_point else v.clone()
        if k.startswith(prefix) and v.dtype.is_floating_point and k in PNAMES:
            t.requires_grad_(True)
        out[k] = t
    return out


PNAMES = {k for k, _ in model.named_parameters()}
g = torch.Generator().manual_seed(3)
B, A, W = 4, 2, 256
image = torch.rand(B, 1, 64, W, generator=g) * 2 - 1

# ---------------- (A) ----------------
def path_a(sdx, img, use_style=True):
    pred = torch_ref.hwr(sdx, img, prefix="hwr.")                       # [T,B,C]
    T = pred.shape[0]
    ci = img.reshape(B // A, A, 64, W).permute(0, 2, 1, 3).reshape(B // A, 1, 64, A * W)
    cr = pred.permute(1, 2, 0).reshape(B // A, A, pred.shape[2], T).permute(0, 2, 1, 3).reshape(B // A, pred.shape[2], A * T)
    if use_style:
        style = torch_ref.style_extractor(sdx, ci, cr, prefix="style_extractor.")
        return pred, (style * wsty.to(style.dtype)).sum()
    return pred, (pred * wpred.to(pred.dtype)).sum()


wsty = torch.randn(B // A, 128, generator=g)
hip = {}
for tag, use_style in (("via style extractor", True), ("direct random dL/dpred", False)):
    for p in model.parameters():
        p.grad = None
    model.pred = None
    x = image.to(dev)
    if use_style:
        style = model.extract_style(x, None, A)          # repeated A times: [B,128]
        pred_h = model.pred
        pred_h.retain_grad()
        loss = (style.view(B // A, A, 128)[:, 0] * wsty.to(dev)).sum()
    else:
        pred_h = model.hwr(x, None)
        pred_h.retain_grad()
        wpred = torch.randn(pred_h.shape, generator=g)
        loss = (pred_h * wpred.to(dev)).sum()
    loss.backward()
    got = {k: p.grad.detach().clone() for k, p in model.named_parameters() if k.startswith("hwr.") and p.grad is not None}
    dpred_h = pred_h.grad.detach().clone()
    res = {}
    for dt in (torch.float32, torch.float64):
        s = leaf(sd, dt, "hwr.")
        pred, l = path_a(s, image.to(dt), use_style)
        pred.retain_grad()
        l.backward()
        res[dt] = (s, pred.grad)
    s32, s64 = res[torch.float32][0], res[torch.float64][0]
    print("(A) %s: loss HIP %.6g fp64 %.6g;  dL/dpred: HIP %.2e, fp32 oracle %.2e vs fp64" % (
        tag, float(loss), float(path_a(leaf(sd, torch.float64, "hwr."), image.double(), use_style)[1]), rel(dpred_h, res[torch.float64][1]), rel(res[torch.float32][1], res[torch.float64][1])))
    rows = []
    for k in sorted(got):
        if s64[k].grad is None or float(s64[k].grad.norm()) < 1e-12:
            continue
        rows.append((rel(got[k], s64[k].grad), rel(s32[k].grad, s64[k].grad), k))
    rows.sort(reverse=True)
    print("    recogniser parameter gradients: pooled HIP %.2e, fp32 oracle %.2e; worst:" % (
        (sum(r[0] ** 2 for r in rows) / len(rows)) ** 0.5, (sum(r[1] ** 2 for r in rows) / len(rows)) ** 0.5))
    for r in rows[:4]:
        print("      %-28s HIP %.2e  fp32 oracle %.2e" % (r[2], r[0], r[1]))
    # how far do the fp64 gradients themselves move when every weight changes by 3e-7 relative (one fp32 rounding)? A smooth function
    # would move by ~3e-7; ReLU / max-pool gates that flip move them by ~1/sqrt(#activations) each
    moves = []
    for trial in range(6):
        s = leaf(sd, torch.float64, "hwr.")
        gp = torch.Generator().manual_seed(100 + trial)
        with torch.no_grad():
            for k in s:
                if k.startswith("hwr.") and s[k].dtype == torch.float64 and s[k].dim() > 0:
                    s[k].mul_(1 + 3e-7 * torch.randn(s[k].shape, generator=gp, dtype=torch.float64))
        _, l = path_a(s, image.double(), use_style)
        l.backward()
        es = [rel(s[r[2]].grad, s64[r[2]].grad) for r in rows]
        moves.append((sum(e * e for e in es) / len(es)) ** 0.5)
    print("    fp64 oracle under 3e-7 relative weight perturbations, pooled gradient change per trial: %s" % " ".join("%.1e" % m for m in moves))

# ---------------- (B) ----------------
for Wd in (456, 384, 256):
    xin = torch.rand(4, 1, 64, Wd, generator=g) * 2 - 1
    torch.manual_seed(11)
    xh = xin.to(dev).requires_grad_(True)
    outs = model.discriminator(xh)
    lh = sum(ops.mean_loss(o, ops.LOSS_MEAN, -1.0) for o in outs)
    lh.backward()
    res = {}
    for dt in (torch.float32, torch.float64):
        s = {k[14:]: (v.detach().clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items() if k.startswith("discriminator.")}
        xr = xin.to(dt).clone().detach().requires_grad_(True)
        torch.manual_seed(11)
        o = torch_ref.discriminator(s, xr)
        (-sum(t.mean() for t in o)).backward()
        res[dt] = xr.grad
    moves = []
    for trial in range(6):
        s = {k[14:]: (v.detach().clone().double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items() if k.startswith("discriminator.")}
        gp = torch.Generator().manual_seed(200 + trial)
        xr = (xin.double() * (1 + 3e-7 * torch.randn(xin.shape, generator=gp, dtype=torch.float64))).requires_grad_(True)
        torch.manual_seed(11)
        o = torch_ref.discriminator(s, xr)
        (-sum(t.mean() for t in o)).backward()
        moves.append(rel(xr.grad, res[torch.float64]))
    print("(B) discriminator dL/dx at width %d: HIP %.2e, fp32 oracle %.2e vs fp64; fp64 oracle under 3e-7 input perturbations: %s" % (
        Wd, rel(xh.grad, res[torch.float64]), rel(res[torch.float32], res[torch.float64]), " ".join("%.1e" % m for m in moves)))
    model.load_state_dict(sd)      # the power iteration moved u / v: start every width from the same state
