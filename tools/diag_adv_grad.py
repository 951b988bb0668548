"""GPU-side diagnosis: gradients of the ADVERSARIAL generator loss -mean(D(G(content, style))) through the full-width
generator (dim 256) and discriminator (dim 64), HIP vs the oracle in fp32 vs the oracle in fp64 on the same seeded weights, noise and
dropout draws - the path of the `no-step+gen` lesson's second gradient set, where round 3 recorded a HIP error 100x the reference's own.
Prints, per generator tensor, the relative L2 error of the HIP gradient and of the fp32 oracle's against fp64; per-activation gradient
errors with --acts (where in the backward pass a deviation enters).   python tools/diag_adv_grad.py [B] [T] [seed]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, torch_ref  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 122
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 31
    from handwriting_line_generation_amd import model as M, ops, rng
    import torch.nn.functional as F
    dev = torch.device("cuda:0")
    rng.set_mode("host")
    g = M.SpacedGenerator(80, 128, dim=256, n_style_trans=6, append_style=True)
    d = M.DiscriminatorAP(64, use_low=True, use_med=True)
    gsd = torch_ref.seeded_state_dict(g, seed)
    dsd = torch_ref.seeded_state_dict(d, seed + 1)
    g.load_state_dict(gsd); d.load_state_dict(dsd)
    g.train().to(dev); d.train().to(dev)
    gen = torch.Generator().manual_seed(seed + 2)
    idx = torch.randint(0, 80, (T, B), generator=gen)
    content = F.one_hot(idx, 80).float()
    style = torch.randn(B, 128, generator=gen)
    gnames = [k for k, p in g.named_parameters() if p.requires_grad]

    def adv(outs):
        return -sum(o.mean() for o in outs) / len(outs)

    torch.manual_seed(cases.FWD_SEED)
    img = g(content.to(dev), style.to(dev))
    preds = d(img)
    loss = 0
    for p in preds:
        t = ops.mean_loss(p, ops.LOSS_MEAN, -1.0)
        loss = t if isinstance(loss, int) else ops.add(loss, t)
    ops.scale(loss, 1.0 / len(preds)).backward()
    torch.cuda.synchronize()
    hip = {k: p.grad.detach().double().cpu() for k, p in g.named_parameters() if p.grad is not None}

    def oracle(dtype):
        gs = {k: (v.detach().to(dtype).clone() if v.dtype.is_floating_point else v.clone()) for k, v in gsd.items()}
        ds = {k: (v.detach().to(dtype).clone() if v.dtype.is_floating_point else v.clone()) for k, v in dsd.items()}
        for k in gnames:
            gs[k].requires_grad_(True)
        rl = torch.randn_like
        torch.randn_like = lambda t, **kw: rl(t.to(torch.float32), **kw).to(dtype)
        torch.manual_seed(cases.FWD_SEED)
        try:
            im = torch_ref.generator(gs, content.to(dtype), style.to(dtype))
            outs = torch_ref.discriminator(ds, im)
        finally:
            torch.randn_like = rl
        adv(outs).backward()
        return {k: gs[k].grad.double() for k in gnames if gs[k].grad is not None}, im.detach().double()

    o32, im32 = oracle(torch.float32)
    o64, im64 = oracle(torch.float64)
    print("image: HIP vs fp64 max-rel %.2e, fp32 oracle vs fp64 %.2e" % (float((img.detach().double().cpu() - im64).abs().max() / im64.abs().max()),
                                                                           float((im32 - im64).abs().max() / im64.abs().max())))
    rows = []
    for k in gnames:
        if k not in o64:
            continue
        n = max(float(o64[k].norm()), 1e-300)
        rows.append((float((hip[k] - o64[k]).norm()) / n, float((o32[k] - o64[k]).norm()) / n, k))
    import math
    print("generator gradient sets (%d tensors): pooled rms HIP %.2e, fp32 oracle %.2e" % (
        len(rows), math.sqrt(sum(r[0] ** 2 for r in rows) / len(rows)), math.sqrt(sum(r[1] ** 2 for r in rows) / len(rows))))
    for eh, eo, k in sorted(rows, reverse=True)[:14]:
        print("   %-44s HIP %.2e   fp32 oracle %.2e" % (k, eh, eo))
    rng.set_mode("device")


if __name__ == "__main__":
    main()
