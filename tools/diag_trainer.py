import sys, json, random, numpy as np, torch
sys.path.insert(0, '.')
from oracle import torch_ref
from handwriting_line_generation_amd import rng
from handwriting_line_generation_amd.harness import build_gan_trainer
from handwriting_line_generation_amd.model import Autoencoder, HWWithStyle
gold = json.load(open('tests/golden/trainer_cycle.json'))
cfg_model = json.load(open('tests/golden/model_config_iam.json'))
msd = torch_ref.seeded_state_dict(HWWithStyle(cfg_model), gold['wseed_model'])
esd = torch_ref.seeded_state_dict(Autoencoder({'type': '2tight', 'hwr': 80}), gold['wseed_enc'])
rng.set_mode('host')
trainer, cfg = build_gan_trainer('iam_gan', 2, 2, width=gold['W'], label_len=gold['label_len'], model_state=msd, encoder_state=esd)
torch.manual_seed(0); np.random.seed(0); random.seed(0)
f = trainer.flat
orig_clip = f.clip_
cur_it = [0]
def spy_clip(v):
    ref = gold.get('pre_clip_grads', {}).get(str(cur_it[0]))
    if ref is not None:
        rows = []
        names = dict((id(p), k) for k, p in trainer.model.named_parameters())
        for k_, pi in enumerate(f.order):
            p = f.params[pi]; name = names[id(p)]
            r = ref[name]
            if not p.requires_grad: continue
            g = p.grad.double()
            mine = None if not f.touched[k_] else (g.sum().item(), g.abs().sum().item())
            if (r is None) != (mine is None):
                rows.append((9.9, name, mine, r)); continue
            if r is None: continue
            rows.append((abs(mine[1] - r[1]) / max(r[1], 1e-30), name, mine, r))
        rows = [r for r in rows if r[3] is None or r[3][1] > 1e-5]
        rows.sort(key=lambda t: -t[0])
        from collections import Counter
        print('   >1e-3 by top module', dict(Counter(r[1].split('.')[0] for r in rows if r[0] > 1e-3)), ' all', dict(Counter(r[1].split('.')[0] for r in rows)))
        for top in ('generator', 'style_extractor', 'discriminator', 'hwr'):
            sub = [r for r in rows if r[1].startswith(top)]
            med = sorted(r[0] for r in sub)[len(sub) // 2] if sub else -1
            print('   %s median rel dev %.2e, n=%d' % (top, med, len(sub)))
            for r in sub[:3]: print('      %.2e %s mine %s ref %s' % r)
        print('  it%d balanced grads: %d tensors deviate >1e-3 of %d; worst:' % (cur_it[0], sum(1 for r in rows if r[0] > 1e-3), len(rows)))
        for r in rows[:6]: print('    %.2e %s mine %s ref %s' % r)
    return orig_clip(v)
f.clip_ = spy_clip
for it, ref in enumerate(gold['logs']):
    cur_it[0] = it
    snap = {k: v.detach().clone() for k, v in trainer.model.named_parameters()}
    log = trainer._train_iteration(it)
    if str(it) in gold.get('per_tensor_update', {}):
        rows = []
        for k, p in trainer.model.named_parameters():
            d = (p.detach() - snap[k]).double()
            rs, ra = gold['per_tensor_update'][str(it)][k]
            rows.append((abs(d.abs().sum().item() - ra) / max(ra, 1e-12) if ra > 0 else d.abs().sum().item(), k, d.sum().item(), d.abs().sum().item(), rs, ra, p.numel()))
        rows.sort(reverse=True)
        nbad = sum(1 for r in rows if r[0] > 1e-3)
        print('  it%d update: %d/%d tensors deviate >1e-3 in |delta|; worst:' % (it, nbad, len(rows)))
        from collections import Counter
        import re
        cnt = Counter(re.sub(r"\.\d+\.", ".N.", r[1].split(".")[0] + "." + ".".join(r[1].split(".")[1:3])) for r in rows if r[0] > 1e-3)
        print('   by module:', dict(cnt))
        ex = sorted(set(int(r[1].split(".")[2]) for r in rows if r[0] > 1e-3 and "char_extractor" in r[1]))
        act = sorted(set(int(r[1].split(".")[2]) for r in rows if r[5] > 0 and "char_extractor" in r[1]))
        print('   deviating experts', ex, ' active experts (ref)', act)
        for r in rows[:4]:
            print('    %.2e %s sum %.4g/%.4g abs %.4g/%.4g n=%d' % (r[0], r[1], r[2], r[4], r[3], r[5], r[6]))
    print(it, ' '.join('%s %.7g/%.7g (%.1e)' % (k, log[k], ref[k], abs(log[k]-ref[k])/max(abs(ref[k]),1e-9)) for k in ref if k not in ('CER','WER')))
