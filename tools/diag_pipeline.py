"""GPU diagnostic: gradients of each loss path of the 'auto' lesson w.r.t. generator inputs/params, HIP vs oracle on the same host."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
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

def run(kind, hip, dt=torch.float32):
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

for kind in ('l1', 'perc', 'ctc', 'adv'):
    lh, dsh, gh = run(kind, True)
    lo, dso, go = run(kind, False)
    ld, dsd, gd = run(kind, False, torch.float64)
    rel = lambda a, b: float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
    worst = sorted(((rel(gh[k], go[k]), k) for k in go if k in gh), reverse=True)[:3]
    print('%-5s vs fp64 truth: dstyle HIP %.2e | fp32-oracle %.2e ;  out.0.conv.bias HIP %.2e | oracle %.2e ; conv.2.conv2.bias HIP %.2e | oracle %.2e' % (
        kind, rel(dsh, dsd), rel(dso, dsd), rel(gh['out.0.conv.bias'], gd['out.0.conv.bias']), rel(go['out.0.conv.bias'], gd['out.0.conv.bias']),
        rel(gh['conv.2.conv2.bias'], gd['conv.2.conv2.bias']), rel(go['conv.2.conv2.bias'], gd['conv.2.conv2.bias'])))
    print('%-5s loss %.7g/%.7g  dstyle l2 %.2e  worst G param grads: %s' % (kind, lh, lo, rel(dsh, dso), ', '.join('%s %.1e' % (k, e) for e, k in worst)))
