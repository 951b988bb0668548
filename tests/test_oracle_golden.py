"""CPU: the oracle restatement (oracle/) reproduces the golden vectors that tools/gen_golden.py recorded from the
reference itself. This is what pins the oracle; the GPU tests then compare the HIP path with the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, seq_oracle, torch_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")

class _ByKind(dict):
    """case name -> entry of the network it exercises (the RIMES cases reuse their base case's definitions)"""

    def __getitem__(self, name):
        return dict.__getitem__(self, cases.kind(name))


ORACLE_FWD = _ByKind({
    "generator": lambda sd, i: [torch_ref.generator(sd, i["content"], i["style"])],
    "discriminator": lambda sd, i: torch_ref.discriminator(sd, i["x"]),
    "hwr": lambda sd, i: [torch_ref.hwr(sd, i["image"])],
    "spacer": lambda sd, i: [torch_ref.spacer(sd, i["onehot"], i["style"])],
    "style_extractor": lambda sd, i: [torch_ref.style_extractor(sd, i["x"], i["recog"])],
    "encoder2": lambda sd, i: list(torch_ref.encoder2(sd, i["x"])),
    "decoder": lambda sd, i: [torch_ref.decoder_noskip(sd, i["x"])],
    "e_hwr": lambda sd, i: [torch_ref.e_hwr(sd, i["x"])],
})
GRAD_INPUTS = _ByKind({"generator": ["style"], "discriminator": ["x"], "hwr": ["image"], "spacer": ["style"], "style_extractor": ["recog"],
                       "encoder2": ["x"], "decoder": ["x"], "e_hwr": ["x"]})


def product_module(name):
    """the product's module class, used here on CPU only for its parameter names/shapes (no forward is run)"""
    from handwriting_line_generation_amd import model as M
    cls = dict(generator=M.SpacedGenerator, discriminator=M.DiscriminatorAP, hwr=M.CNNOnlyHWR, spacer=M.CountCNN,
               style_extractor=M.CharStyleEncoder, encoder2=M.Encoder2, decoder=M.DecoderNoSkip, e_hwr=M.E_HWR)[cases.kind(name)]
    return cls(**cases.CASES[name]["ctor"])


def oracle_run(name, sd):
    inp = cases.inputs(name)
    for k in GRAD_INPUTS[name]:
        inp[k] = inp[k].clone().requires_grad_(True)
    torch.manual_seed(cases.FWD_SEED)
    outs = ORACLE_FWD[name](sd, inp)
    ws = cases.probe_weights(outs)
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    return outs, {k: inp[k].grad for k in GRAD_INPUTS[name]}


def rel(a, b):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))


PERTURB = 1e-6      # relative size of the conditioning probe: what fp32 kernels of different summation order differ by at intermediate layers


def _oracle_grads64(name, sd, pnames, ws, loss_fn=None, inputs=None, perturb=None):
    """parameter and input gradients of the oracle evaluated in fp64 with the draws of the fp32 run (noise is drawn in fp32 and widened;
    Dropout2d's Bernoulli masks do not depend on the dtype)"""
    sd64 = {k: (v.detach().double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    for k in pnames:
        sd64[k].requires_grad_(True)
    oin = inputs if inputs is not None else cases.inputs(name)
    oin = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in oin.items()}
    if perturb is not None:   # fp32-rounding sized relative noise on every continuous input and weight: probes how well conditioned the gradients are
        g = torch.Generator().manual_seed(perturb)
        oin = {k: (v * (1 + PERTURB * torch.randn(v.shape, generator=g, dtype=torch.float64)) if v.dtype.is_floating_point else v) for k, v in oin.items()}
        for k in pnames:
            with torch.no_grad():
                sd64[k].mul_(1 + PERTURB * torch.randn(sd64[k].shape, generator=g, dtype=torch.float64))
    for k in GRAD_INPUTS[name]:
        oin[k] = oin[k].clone().requires_grad_(True)
    rl, rn = torch.randn_like, torch.randn
    torch.randn_like = lambda t, **kw: rl(t.to(torch.float32), **kw).double()
    torch.manual_seed(cases.FWD_SEED)
    try:
        outs = ORACLE_FWD[name](sd64, oin)
    finally:
        torch.randn_like = rl
    if loss_fn is None:
        sum((o * w.double()).sum() for o, w in zip(outs, ws)).backward()
    else:
        loss_fn(outs).backward()
    return {k: sd64[k].grad for k in pnames}, {k: oin[k].grad for k in GRAD_INPUTS[name]}



@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference_golden(name):
    gold = np.load(os.path.join(GOLD, "module_%s.npz" % name))
    m = product_module(name)
    sd = torch_ref.seeded_state_dict(m, cases.CASES[name]["wseed"])
    pnames = [k for k, p in m.named_parameters() if p.requires_grad]
    for k in pnames:
        sd[k] = sd[k].clone().requires_grad_(True)
    outs, igr = oracle_run(name, sd)
    for i, o in enumerate(outs):
        assert rel(o.detach(), gold["out%d" % i]) < 2e-5, "%s out%d" % (name, i)
    names = json.loads(str(gold["pgrad_names"]))
    assert sorted(pnames) == names, "parameter names differ from the reference's"
    grads = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in pnames}
    _, fp = cases.fingerprint(grads)
    ref = torch.from_numpy(gold["pgrad_fp"])
    scale = ref[:, 1].clamp_min(1e-6)
    tight = all(rel(g, gold["igrad_" + k]) < 2e-5 for k, g in igr.items()) and float(((fp - ref).abs() / scale[:, None]).max()) < 1e-4
    if tight:
        return
    # The goldens were recorded by the reference on the build container's CPU. On another CPU (other vector ISA, other oneDNN kernels) the
    # oracle's fp32 arithmetic rounds differently, and the stacks of ReLU / max-pool gates turn that into percent-level gradient changes
    # (one gate flips). Then the golden is held to the fp64 yardstick instead: it must be as close to this host's fp64 oracle as this host's
    # fp32 oracle is, or within the measured conditioning (1e-6 relative perturbations of weights and inputs, worst of four).
    ws = cases.probe_weights([o.detach() for o in outs])
    base = {k: v.detach() for k, v in sd.items()}
    g64, ig64 = _oracle_grads64(name, base, pnames, ws)
    _, fp64 = cases.fingerprint({k: (g64[k] if g64[k] is not None else torch.zeros_like(base[k], dtype=torch.float64)) for k in pnames})
    cond_fp = torch.zeros_like(fp64)
    cond_ig = {k: 0.0 for k in igr}
    for trial in (1, 2, 3, 4):
        gp, igp = _oracle_grads64(name, base, pnames, ws, perturb=trial)
        _, fpp = cases.fingerprint({k: (gp[k] if gp[k] is not None else torch.zeros_like(base[k], dtype=torch.float64)) for k in pnames})
        cond_fp = torch.maximum(cond_fp, (fpp - fp64).abs())
        for k in igr:
            cond_ig[k] = max(cond_ig[k], rel(igp[k], ig64[k]))
    for k, g in igr.items():
        e_gold, e_cur = rel(torch.from_numpy(gold["igrad_" + k]), ig64[k]), rel(g, ig64[k])
        assert e_gold < max(2e-5, 3 * e_cur, 3 * cond_ig[k]), "%s d%s: golden %.2e from fp64, this host's fp32 oracle %.2e, conditioning %.2e" % (
            name, k, e_gold, e_cur, cond_ig[k])
    e_gold, e_cur = (ref - fp64).abs(), (fp - fp64).abs()
    bound = torch.maximum(torch.maximum(1e-4 * scale[:, None].expand_as(ref), 3 * e_cur), 3 * cond_fp)
    worst = float((e_gold / bound).max())
    assert worst <= 1.0, "%s: golden parameter-gradient fingerprints are %.1fx further from fp64 than fp32 rounding / conditioning explains" % (name, worst)


def test_state_dict_schema_matches_reference():
    """full-size HWWithStyle: the 1411 state-dict entries / 47,358,027 parameters of the reference"""
    from handwriting_line_generation_amd.model import HWWithStyle
    cfg = json.load(open(os.path.join(GOLD, "model_config_iam.json")))
    m = HWWithStyle(cfg)
    schema = json.load(open(os.path.join(GOLD, "state_dict_schema_iam.json")))
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert got == schema
    assert sum(p.numel() for p in m.parameters()) == 47358027 and len(list(m.parameters())) == 1324


def test_seq_edge_known_answers():
    kat = np.load(os.path.join(GOLD, "seq_kat_edges.npz"))
    n = 0
    while "dtw%d_pred" % n in kat:
        out = seq_oracle.correct_pred(torch.from_numpy(kat["dtw%d_pred" % n]), torch.from_numpy(kat["dtw%d_label" % n]))
        assert np.array_equal(out.numpy(), kat["dtw%d_out" % n]), n
        if hasattr(seq_oracle, "correct_pred_c"):
            outc = seq_oracle.correct_pred_c(torch.from_numpy(kat["dtw%d_pred" % n]), torch.from_numpy(kat["dtw%d_label" % n]))
            assert np.array_equal(outc.numpy(), kat["dtw%d_out" % n]), "C oracle, case %d" % n
        n += 1
    assert n >= 8


def test_seq_known_answers():
    kat = np.load(os.path.join(GOLD, "seq_kat.npz"))
    for n in range(5):
        pred = torch.from_numpy(kat["dtw%d_pred" % n]); label = torch.from_numpy(kat["dtw%d_label" % n])
        out = seq_oracle.correct_pred(pred, label)
        assert np.array_equal(out.numpy(), kat["dtw%d_out" % n])
        dec = json.loads(str(kat["dec%d" % n]))
        for b in range(pred.shape[1]):
            assert seq_oracle.naive_decode(pred[:, b].numpy())[0] == dec[b]


def test_gate_matcher_locates_flips_beyond_the_near_zero_lists_by_hash_search():
    """oracle/gates.py Matcher.feed(margin=...): a block whose hash differs from the record's is searched by toggling the other side's
    smallest-|pre-activation| decisions until the 64-bit block hash equals the reference's - single flips, two flips in one block, several
    samples; with an EMPTY near-zero list in the record (the case the search exists for)."""
    from oracle import gates
    rng = np.random.default_rng(0)
    N, M = 2, 20000
    pre64 = rng.standard_normal((N, M))
    dec64 = (pre64 > 0).astype(np.uint8)
    rec = {"name": "net.layer", "kind": "act", "N": N, "M": M, "hashes": gates.block_hashes(dec64), "nz_idx": np.full(gates.NZ, -1),
           "nz_val": np.zeros(gates.NZ), "nz_code": np.zeros(gates.NZ, np.uint8), "ref32_flips": 0}
    flips = [(0, 123), (0, 9000), (1, 17000), (1, 5), (1, 77)]            # the last two share a block
    pre32 = pre64.copy()
    for k, (n, i) in enumerate(flips):
        pre32[n, i] = -np.sign(pre64[n, i]) * (1 + k) * 1e-7
    m = gates.Matcher([rec])
    m.feed("act", (pre32 > 0).astype(np.uint8), margin=pre32)
    assert sorted((s[2], s[3], s[5]) for s in m.sites) == sorted((n, i, int(dec64[n, i])) for n, i in flips)
    assert m.flips == {"net.layer": 5}
    m2 = gates.Matcher([rec])                                              # without margins nothing can be located, the blocks still count
    m2.feed("act", (pre32 > 0).astype(np.uint8))
    assert m2.sites == [] and m2.flips == {"net.layer": 4}
