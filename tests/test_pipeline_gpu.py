"""GPU: gradients of the four loss paths of the 'auto' lesson (L1, perceptual, CTC through the recogniser, adversarial) with
respect to the generator's style input and parameters - HIP path vs the fp32 oracle vs the same oracle evaluated in fp64.

The recogniser/CTC path is ill-conditioned in fp32: the reference's own CPU arithmetic is ~1e-2 away from the fp64 result
there (1e-4 for the perceptual / adversarial paths, 1e-6 for L1). The HIP path is therefore required to be as accurate as the
reference's arithmetic is (error vs fp64 within 4x of the fp32 oracle's own error), and forward losses to agree to 1e-5."""
import pytest
pytestmark = pytest.mark.gpu


def test_loss_path_gradients_vs_fp64(cuda):

    import sys, torch, torch.nn.functional as F
    from oracle import torch_ref, cases
    from handwriting_line_generation_amd import rng, ops
    from handwriting_line_generation_amd import model as M
    dev = torch.device('cuda:0')
    rng.set_mode('host')
    G = M.SpacedGenerator(80, 128, 256, n_style_trans=6, append_style=True); H = M.CNNOnlyHWR(80, norm='batch'); D = M.DiscriminatorAP(64, use_low=True); E = M.Encoder2(32)
    sds = {}
    for name, m, seed in (('G', G, 31), ('H', H, 32), ('D', D, 33), ('E', E, 34)):
        sds[name] = torch_ref.seeded_state_dict(m, seed); m.load_state_dict(sds[name]); m.train().to(dev)
    g = torch.Generator().manual_seed(5)
    T, B = 58, 4
    idx = torch.randint(0, 80, (T, B), generator=g); content = F.one_hot(idx, 80).float()
    style = torch.randn(B, 128, generator=g)
    image = torch.rand(B, 1, 64, 4 * T, generator=g) * 2 - 1
    labels = torch.randint(1, 80, (B, 12), generator=g)

    def run(kind, hip, dt=torch.float32, perturb=None):
        torch.manual_seed(77)
        if hip:
            st = style.to(dev).requires_grad_(True)
            for m in (G, H, D, E): m.zero_grad()
            recon = G(content.to(dev), st)
            img = image.to(dev)
        else:
            sd = {k: {kk: (vv.clone().to(dt) if vv.dtype.is_floating_point else vv.clone()) for kk, vv in v.items()} for k, v in sds.items()}
            for k in ('G',):
                for kk, vv in sd[k].items():
                    if vv.dtype.is_floating_point and 'running' not in kk and not kk.endswith(('weight_flip',)) and not ('conv1.2.weight' in kk or 'conv1.1.weight' in kk and vv.shape[1:] == (1, 3, 3)): vv.requires_grad_(True)
            if perturb is not None:    # conditioning probe: every weight moved by 1e-6 relative
                gp = torch.Generator().manual_seed(perturb)
                with torch.no_grad():
                    for net in sd.values():
                        for vv in net.values():
                            if vv.dtype.is_floating_point:
                                vv.mul_(1 + 1e-6 * torch.randn(vv.shape, generator=gp, dtype=vv.dtype))
            st = style.clone().to(dt).requires_grad_(True)
            if dt == torch.float64:
                _rl = torch.randn_like
                torch.randn_like = lambda t: _rl(t.float()).double()   # identical noise values, widened
            recon = torch_ref.generator(sd['G'], content.to(dt), st)
            if dt == torch.float64:
                torch.randn_like = _rl
            img = image.to(dt)
        if kind == 'l1':
            loss = ops.l1_loss(recon, img) if hip else F.l1_loss(recon, img)
        elif kind == 'perc':
            both = torch.cat((img, recon), 0)
            feats = E(both) if hip else torch_ref.encoder2(sd['E'], both)
            loss = 0
            for f in feats:
                a, b = f[:B], f[B:]
                loss = loss + (ops.l1_loss(b, a) if hip else F.l1_loss(b, a))
        elif kind == 'ctc':
            pred = H(recon) if hip else torch_ref.hwr(sd['H'], recon)
            Tn = pred.shape[0]
            loss = ops.ctc_loss(pred, labels, [Tn] * B, [12] * B) if hip else F.ctc_loss(pred, labels, torch.tensor([Tn] * B), torch.tensor([12] * B))
        elif kind == 'adv':
            outs = D(recon) if hip else torch_ref.discriminator(sd['D'], recon)
            loss = 0
            for o in outs:
                loss = loss - (ops.mean_loss(o) if hip else o.mean())
            loss = loss / len(outs)
        loss.backward()
        if hip:
            grads = {k: p.grad.detach().cpu().clone() for k, p in G.named_parameters() if p.grad is not None}
        else:
            grads = {k: v.grad.clone() for k, v in sd['G'].items() if v.requires_grad and v.grad is not None}
        return float(loss), st.grad.detach().cpu().clone(), grads

    rel = lambda a, b: float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
    bad = []
    for kind in ('l1', 'perc', 'ctc', 'adv'):
        lh, dsh, gh = run(kind, True)
        lo, dso, go = run(kind, False)
        ld, dsd, gd = run(kind, False, torch.float64)
        # conditioning of this loss path: how far the fp64 gradients move when every weight changes by 1e-6 relative (the L1 terms have sign()
        # derivatives, the recogniser / encoder stacks ReLU and pooling gates); an independent fp32 implementation cannot be closer than that
        sens, gsens = 0.0, {}
        for trial in (1, 2, 3, 4):
            _, dsp, gp_ = run(kind, False, torch.float64, perturb=trial)
            sens = max(sens, rel(dsp, dsd))
            for k in gd:
                gsens[k] = max(gsens.get(k, 0.0), rel(gp_[k], gd[k]))
        if abs(lh - lo) > 1e-5 * max(abs(lo), 1e-3):
            bad.append('%s loss %.8g vs %.8g' % (kind, lh, lo))
        eh, eo = rel(dsh, dsd), rel(dso, dsd)
        if eh > max(4 * eo + 2e-5, 3 * sens):
            bad.append('%s dstyle err vs fp64: HIP %.2e, fp32 oracle %.2e, sensitivity %.2e' % (kind, eh, eo, sens))
        for k in ('conv.0.conv2.weight', 'conv.2.conv2.weight', 'conv.4.conv2.weight', 'conv.3.conv1.0.weight'):
            eh, eo = rel(gh[k], gd[k]), rel(go[k], gd[k])
            if eh > max(4 * eo + 2e-5, 3 * gsens[k]):
                bad.append('%s d%s err vs fp64: HIP %.2e, fp32 oracle %.2e, sensitivity %.2e' % (kind, k, eh, eo, gsens[k]))
    rng.set_mode('device')
    assert not bad, '; '.join(bad)


def test_count_lesson_recogniser_gradients_and_gate_flips(cuda):
    """The `count` lesson reaches the recogniser only through the style extractor's `recog` input (reference: model/hw_with_style.py:281-300,
    model/char_style.py:193-309). Round 3 recorded recogniser gradients 1000x further from fp64 than the reference's own fp32 run in that
    group and attributed it to flipped discrete decisions WITHOUT measuring them. Measured here on the full-size model under two kernel
    schedules (the planner's and the all-direct one, HWG_WINO=0), with every discrete decision of the path recorded on the way:
      * the arg-max map of the log-probs (which expert sees which window), against the fp64 oracle's map;
      * the sign pattern behind every ReLU (bias_act / fused norm activations) and the winner of every max-pool window with a positive
        maximum, in the recogniser and the style extractor - schedule against schedule.
    Assertions: at least one schedule is within max(1e-4, 3 x the fp32 oracle's own error) of fp64 (the kernels' arithmetic is as good as
    the reference's); a schedule further away than that must differ from the clean one in at least one recorded decision (the excess is
    flipped gates, counted and printed) and stay below 1e-2; the arg-max maps never differ without being counted."""
    import torch
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

    def rel(a, b):
        return float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-300))

    def oracle(dt):
        s = {}
        for k, v in sd.items():
            t = v.detach().clone().to(dt) if v.dtype.is_floating_point else v.clone()
            if k.startswith("hwr.") and v.dtype.is_floating_point and k in pnames:
                t.requires_grad_(True)
            s[k] = t
        img = image.to(dt)
        pred = torch_ref.hwr(s, img, prefix="hwr.")
        T = pred.shape[0]
        ci = img.reshape(B // A, A, 64, W).permute(0, 2, 1, 3).reshape(B // A, 1, 64, A * W)
        cr = pred.permute(1, 2, 0).reshape(B // A, A, pred.shape[2], T).permute(0, 2, 1, 3).reshape(B // A, pred.shape[2], A * T)
        style = torch_ref.style_extractor(s, ci, cr, prefix="style_extractor.")
        (style * wsty.to(dt)).sum().backward()
        grads = {k: s[k].grad for k in s if k.startswith("hwr.") and s[k].grad is not None and float(s[k].grad.norm()) > 1e-12}
        return grads, cr.detach().argmax(dim=1).numpy()          # [B/A, A*T]: the map the style extractor derives

    g64, amax64 = oracle(torch.float64)
    g32, amax32 = oracle(torch.float32)
    e32 = {k: rel(g32[k], g64[k]) for k in g64}
    pooled32 = (sum(v * v for v in e32.values()) / len(e32)) ** 0.5
    enc = model.style_extractor

    # gate recorder: wraps the forward of the three op classes every ReLU / max-pool of these networks goes through
    rec = []
    saved = {c: c.forward for c in (ops._BiasAct, ops._Norm, ops._MaxPool)}

    def wrap_act(cls, act_index):
        f = saved[cls]

        def fwd(ctx, *a):
            y = f(ctx, *a)
            if a[act_index] in (ops.ACT_RELU, ops.ACT_LRELU):
                rec.append(("act", (y > 0)))
            return y
        return staticmethod(fwd)

    def fwd_pool(ctx, *a):
        y = saved[ops._MaxPool](ctx, *a)
        rec.append(("pool", ctx.to_save[0].clone(), (y > 0)))
        return y

    def hip():
        del rec[:]
        for p in model.parameters():
            p.grad = None
        model.pred = None
        ops._BiasAct.forward = wrap_act(ops._BiasAct, 3)
        ops._Norm.forward = wrap_act(ops._Norm, 7)
        ops._MaxPool.forward = staticmethod(fwd_pool)
        try:
            style = model.extract_style(image.to(dev), None, A)
        finally:
            for c, f in saved.items():
                c.forward = f
        (style.view(B // A, A, 128)[:, 0] * wsty.to(dev)).sum().backward()
        torch.cuda.synchronize()
        got = {k: p.grad.detach().clone() for k, p in model.named_parameters() if k in g64 and p.grad is not None}
        errs = {k: rel(got[k], g64[k]) for k in g64}
        return (sum(v * v for v in errs.values()) / len(errs)) ** 0.5, enc.last_argmax.copy(), list(rec)

    bound = max(1e-4, 3 * pooled32)
    runs = []
    try:
        for tag, env in (("planner's schedule", {}), ("all-direct schedule (HWG_WINO=0)", {"HWG_WINO": "0"})):
            with ops.tuning(**env):
                runs.append((tag,) + hip())
    finally:
        rng.set_mode("device")
    clean = min(runs, key=lambda r: r[1])
    lines = ["count-lesson recogniser gradients through the style extractor: pooled relative error vs fp64; fp32 oracle %.2e (its arg-max map differs from "
             "fp64's in %d columns), bound %.2e" % (pooled32, int((amax32 != amax64).sum()), bound)]
    assert clean[1] <= bound, "no schedule is within %.2e of fp64: best %.2e (%s)" % (bound, clean[1], clean[0])
    for tag, err, amax, gates in runs:
        assert amax.shape == amax64.shape
        aflips = int((amax != amax64).sum())
        assert len(gates) == len(clean[3])
        act_flips = pool_flips = n_act = n_pool = 0
        for a_, b_ in zip(gates, clean[3]):
            assert a_[0] == b_[0]
            if a_[0] == "act":
                act_flips += int((a_[1] != b_[1]).sum()); n_act += a_[1].numel()
            else:
                live = a_[2] | b_[2]             # windows whose maximum is positive somewhere: only those pass a gradient on
                pool_flips += int(((a_[1] != b_[1]) & live).sum()); n_pool += int(live.sum())
        lines.append("   %-34s error %.2e; arg-max columns flipped vs fp64: %d of %d; vs the clean schedule: ReLU signs %d of %d, max-pool winners %d of %d"
                     % (tag, err, aflips, amax.size, act_flips, n_act, pool_flips, n_pool))
        if err > bound:
            assert aflips + act_flips + pool_flips > 0, "%s: %.2e from fp64 with every recorded decision equal to the clean schedule's" % (tag, err)
            assert err <= 1e-2, "%s: %.2e" % (tag, err)
    print("\n".join(lines))
    _summary(lines)


def _summary(lines):
    import os
    if os.environ.get("HWG_PARITY_SUMMARY"):
        with open(os.environ["HWG_PARITY_SUMMARY"], "a") as fh:
            fh.write("\n".join(lines) + "\n\n")


def test_adversarial_generator_gradients_are_as_far_from_fp64_as_the_references(cuda):
    """Second flagged group of round 3 (`no-step+gen`, adversarial set: generator gradients 100x further from fp64 than the reference's fp32 run
    in ONE teacher-forced unit, equal or better in the three others). Measured on the isolated path -mean(D(G(content, style))), full-width
    generator and discriminator, three seeds: the reference arithmetic (fp32 oracle) is itself ~1e-3 from fp64 on this path (stacks of
    LeakyReLU gates in D and G), draw by draw; the HIP gradients must be no further than 3 x the worst of the oracle's draws, and the
    forward image within 1e-4."""
    import math
    import torch
    import torch.nn.functional as F
    from oracle import cases, torch_ref
    from handwriting_line_generation_amd import model as M, ops, rng
    dev = torch.device("cuda:0")
    rng.set_mode("host")
    rows = []
    try:
        for seed in (31, 41, 51):
            G = M.SpacedGenerator(80, 128, dim=256, n_style_trans=6, append_style=True)
            D = M.DiscriminatorAP(64, use_low=True, use_med=True)
            gsd, dsd = torch_ref.seeded_state_dict(G, seed), torch_ref.seeded_state_dict(D, seed + 1)
            G.load_state_dict(gsd); D.load_state_dict(dsd)
            G.train().to(dev); D.train().to(dev)
            gen = torch.Generator().manual_seed(seed + 2)
            Bn, T = 4, 122
            content = F.one_hot(torch.randint(0, 80, (T, Bn), generator=gen), 80).float()
            style = torch.randn(Bn, 128, generator=gen)
            gnames = [k for k, p in G.named_parameters() if p.requires_grad]
            torch.manual_seed(cases.FWD_SEED)
            img = G(content.to(dev), style.to(dev))
            loss = 0
            for p in D(img):
                t = ops.mean_loss(p, ops.LOSS_MEAN, -1.0)
                loss = t if isinstance(loss, int) else ops.add(loss, t)
            ops.scale(loss, 0.5).backward()
            torch.cuda.synchronize()
            hipg = {k: p.grad.detach().double().cpu() for k, p in G.named_parameters() if p.grad is not None}

            def oracle(dt):
                gs = {k: (v.detach().to(dt).clone() if v.dtype.is_floating_point else v.clone()) for k, v in gsd.items()}
                ds = {k: (v.detach().to(dt).clone() if v.dtype.is_floating_point else v.clone()) for k, v in dsd.items()}
                for k in gnames:
                    gs[k].requires_grad_(True)
                rl = torch.randn_like
                torch.randn_like = lambda t, **kw: rl(t.to(torch.float32), **kw).to(dt)
                torch.manual_seed(cases.FWD_SEED)
                try:
                    im = torch_ref.generator(gs, content.to(dt), style.to(dt))
                    outs = torch_ref.discriminator(ds, im)
                finally:
                    torch.randn_like = rl
                (-sum(o.mean() for o in outs) / len(outs)).backward()
                return {k: gs[k].grad.double() for k in gnames if gs[k].grad is not None}, im.detach().double()
            o32, _ = oracle(torch.float32)
            o64, im64 = oracle(torch.float64)
            assert float((img.detach().double().cpu() - im64).abs().max() / im64.abs().max()) < 1e-4
            keys = [k for k in gnames if k in o64 and float(o64[k].norm()) > 0]
            eh = math.sqrt(sum((float((hipg[k] - o64[k]).norm()) / float(o64[k].norm())) ** 2 for k in keys) / len(keys))
            eo = math.sqrt(sum((float((o32[k] - o64[k]).norm()) / float(o64[k].norm())) ** 2 for k in keys) / len(keys))
            rows.append((seed, eh, eo))
    finally:
        rng.set_mode("device")
    line = ("adversarial generator gradients -mean(D(G(.))), full width, pooled relative error vs fp64 per seed (HIP / fp32 oracle): " +
            ", ".join("%d: %.2e / %.2e" % r for r in rows))
    print(line)
    _summary([line])
    worst_oracle = max(r[2] for r in rows)
    for seed, eh, eo in rows:
        assert eh <= max(1e-4, 3 * worst_oracle), "seed %d: HIP %.2e, the oracle's worst draw %.2e" % (seed, eh, worst_oracle)
