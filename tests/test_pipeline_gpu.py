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
