"""ORACLE (test infrastructure only). Module-level parity cases shared by tools/gen_golden.py (reference side),
tests/test_oracle_golden.py (oracle vs reference goldens) and tests/test_modules_gpu.py (HIP vs oracle vs goldens).

A case is defined by constructor arguments, a weight seed and input seeds; weights and inputs are regenerated from the
seeds on every side (torch's CPU generator is deterministic), so only outputs and gradient fingerprints are stored."""
import torch
import torch.nn.functional as F


def _g(seed):
    return torch.Generator().manual_seed(seed)


def peaked_logprobs(T, B, C, seed, blank_frac=0.6):
    """recogniser-like output: mostly blanks, confident characters elsewhere -> log-softmax [T,B,C]"""
    g = _g(seed)
    cls = torch.randint(1, C, (T, B), generator=g)
    cls[torch.rand(T, B, generator=g) < blank_frac] = 0
    logits = torch.randn(T, B, C, generator=g) + 6.0 * F.one_hot(cls, C).float()
    return F.log_softmax(logits, dim=2)


CASES = {
    "generator": dict(ctor=dict(n_class=80, style_size=128, dim=64, n_style_trans=6, append_style=True), wseed=11,
                      T=12, B=2, n_class=80, style=128),
    "discriminator": dict(ctor=dict(dim=16, use_low=True, use_med=True), wseed=12, N=3, W=96),
    "hwr": dict(ctor=dict(nclass=80, norm="batch", small=False, pad=False), wseed=13, B=2, W=64),
    "spacer": dict(ctor=dict(class_size=80, style_size=128, hidden_size=128, n_out=2), wseed=14, L=9, B=3),
    "style_extractor": dict(ctor=dict(input_dim=1, dim=16, style_dim=32, char_dim=32, char_style_dim=0, norm="group", activ="relu",
                                      pad_type="replicate", n_class=80, global_pool=True, average_found_char_style=1.0, window=2),
                            wseed=15, B=2, W=128),
    "encoder2": dict(ctor=dict(out_dim=32), wseed=16, N=2, W=64),
    "decoder": dict(ctor=dict(input_dim=32), wseed=17, N=2, Wc=4),
    "e_hwr": dict(ctor=dict(n_class=80, n_in=32), wseed=18, N=2, Wc=9),
    # RIMES (BASELINE configs[4]): 78 classes (206 / 334-channel generator layers, 78-way recogniser), a padded batch of lines
    # of different widths (the collate's -1 padding on the right of the shorter lines)
    "generator_rimes": dict(kind="generator", ctor=dict(n_class=78, style_size=128, dim=64, n_style_trans=6, append_style=True), wseed=21,
                            T=17, B=2, n_class=78, style=128),
    "discriminator_rimes": dict(kind="discriminator", ctor=dict(dim=16, use_low=True, use_med=True), wseed=22, N=3, W=176, widths=(176, 120, 88)),
    "hwr_rimes": dict(kind="hwr", ctor=dict(nclass=78, norm="batch", small=False, pad=False), wseed=23, B=3, W=160, widths=(160, 96, 132)),
    "style_extractor_rimes": dict(kind="style_extractor", ctor=dict(input_dim=1, dim=16, style_dim=32, char_dim=32, char_style_dim=0, norm="group",
                                                                     activ="relu", pad_type="replicate", n_class=78, global_pool=True,
                                                                     average_found_char_style=1.0, window=2),
                                  wseed=25, B=2, W=192, widths=(192, 140), n_class=78),
    # one 64x512 line through the FULL-WIDTH generator (dim 256) and discriminator (dim 64) of the shipped IAM GAN config
    "generator_full": dict(kind="generator", ctor=dict(n_class=80, style_size=128, dim=256, n_style_trans=6, append_style=True), wseed=31,
                           T=128, B=1, n_class=80, style=128),
    "discriminator_full": dict(kind="discriminator", ctor=dict(dim=64, use_low=True, use_med=True), wseed=32, N=1, W=512),
}


def kind(name):
    """which network a case exercises (the RIMES variants reuse the forward / gradient-input definitions of their base case)"""
    return CASES[name].get("kind", name)


def _padded(x, widths):
    """collate's padding: columns beyond a line's own width hold -1"""
    if widths:
        for b, w in enumerate(widths):
            x[b, :, :, w:] = -1.0
    return x


def inputs(name):
    c = CASES[name]
    seed_shift = 1000 if name != kind(name) else 0
    name = kind(name)
    if name == "generator":
        g = _g(101 + seed_shift)
        idx = torch.randint(0, c["n_class"], (c["T"], c["B"]), generator=g)
        content = F.one_hot(idx, c["n_class"]).float()
        style = torch.randn(c["B"], c["style"], generator=g)
        return dict(content=content, style=style)
    if name == "discriminator":
        return dict(x=_padded(torch.rand(c["N"], 1, 64, c["W"], generator=_g(102 + seed_shift)) * 2 - 1, c.get("widths")))
    if name == "hwr":
        return dict(image=_padded(torch.rand(c["B"], 1, 64, c["W"], generator=_g(103 + seed_shift)) * 2 - 1, c.get("widths")))
    if name == "spacer":
        g = _g(104)
        idx = torch.randint(0, 80, (c["L"], c["B"]), generator=g)
        return dict(onehot=F.one_hot(idx, 80).float(), style=torch.randn(c["B"], 128, generator=g))
    if name == "style_extractor":
        g = _g(105 + seed_shift)
        x = _padded(torch.rand(c["B"], 1, 64, c["W"], generator=g) * 2 - 1, c.get("widths"))
        T = c["W"] // 4 - 6
        recog = peaked_logprobs(T, c["B"], c.get("n_class", 80), 205 + seed_shift).permute(1, 2, 0).contiguous()   # [B,C,T]
        return dict(x=x, recog=recog)
    if name == "encoder2":
        return dict(x=torch.rand(c["N"], 1, 64, c["W"], generator=_g(106)) * 2 - 1)
    if name == "decoder":
        return dict(x=torch.randn(c["N"], 32, 1, c["Wc"], generator=_g(107)))
    if name == "e_hwr":
        return dict(x=torch.randn(c["N"], 32, 1, c["Wc"], generator=_g(108)))
    raise KeyError(name)


FWD_SEED = 4242  # torch.manual_seed before each forward (noise / dropout draws)


def probe_weights(outs, seed=999):
    """fixed random cotangents so that loss = sum_i <out_i, w_i> exercises the whole backward"""
    g = _g(seed)
    return [torch.randn(o.shape, generator=g) for o in outs]


def fingerprint(named_grads):
    """per-tensor (sum, abs-sum) gradient fingerprint, ordered by name"""
    names = sorted(named_grads)
    vals = torch.tensor([[float(named_grads[n].double().sum()), float(named_grads[n].double().abs().sum())] for n in names], dtype=torch.float64)
    return names, vals
