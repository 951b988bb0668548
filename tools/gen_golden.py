"""Generate tests/golden/* by running the UNMODIFIED reference (imported from /root/reference, CPU) on seeded inputs.

Run in the build container only:  python tools/gen_golden.py [modules] [seq] [trainer]
The reference's source never leaves /root/reference; only numeric inputs/outputs are written.
Each module case is also evaluated with the oracle restatement (oracle/torch_ref.py) and the differences are
printed and asserted, which is what "pins" the oracle.
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
warnings.filterwarnings("ignore")

import ref_bootstrap  # noqa: E402

ref_bootstrap.bootstrap()  # chdir to /root/reference, stub cv2 & co.

import torch  # noqa: E402

from oracle import cases, seq_oracle, torch_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def ref_modules():
    from model.pure_gen import SpacedGenerator
    from model.discriminator_ap import DiscriminatorAP
    from model.cnn_only_hwr import CNNOnlyHWR
    from model.count_cnn import CountCNN
    from model.char_style import CharStyleEncoder
    from model.autoencoder import Encoder2, DecoderNoSkip, E_HWR
    return dict(generator=SpacedGenerator, discriminator=DiscriminatorAP, hwr=CNNOnlyHWR, spacer=CountCNN, style_extractor=CharStyleEncoder,
                encoder2=Encoder2, decoder=DecoderNoSkip, e_hwr=E_HWR)


ORACLE_FWD = {
    "generator": lambda sd, i: [torch_ref.generator(sd, i["content"], i["style"])],
    "discriminator": lambda sd, i: torch_ref.discriminator(sd, i["x"]),
    "hwr": lambda sd, i: [torch_ref.hwr(sd, i["image"])],
    "spacer": lambda sd, i: [torch_ref.spacer(sd, i["onehot"], i["style"])],
    "style_extractor": lambda sd, i: [torch_ref.style_extractor(sd, i["x"], i["recog"])],
    "encoder2": lambda sd, i: list(torch_ref.encoder2(sd, i["x"])),
    "decoder": lambda sd, i: [torch_ref.decoder_noskip(sd, i["x"])],
    "e_hwr": lambda sd, i: [torch_ref.e_hwr(sd, i["x"])],
}


def module_fwd(name, m, i):
    name = cases.kind(name)
    if name == "generator":
        return [m(i["content"], i["style"])]
    if name == "discriminator":
        return m(i["x"])
    if name == "hwr":
        return [m(i["image"], None)]
    if name == "spacer":
        return [m(i["onehot"], i["style"])]
    if name == "style_extractor":
        return [m(i["x"], i["recog"])]
    if name == "encoder2":
        return list(m(i["x"]))
    if name == "decoder":
        return [m(i["x"], None)]
    if name == "e_hwr":
        return [m(i["x"])]
    raise KeyError(name)


GRAD_INPUTS = {"generator": ["style"], "discriminator": ["x"], "hwr": ["image"], "spacer": ["style"], "style_extractor": ["recog"],
               "encoder2": ["x"], "decoder": ["x"], "e_hwr": ["x"]}


def run_case(name, fwd, params, inp):
    """common protocol: seed, forward, probe loss, backward -> (outs, input grads, param-grad fingerprint)"""
    for k in GRAD_INPUTS[cases.kind(name)]:
        inp[k] = inp[k].clone().requires_grad_(True)
    torch.manual_seed(cases.FWD_SEED)
    outs = fwd(inp)
    ws = cases.probe_weights(outs)
    loss = sum((o * w).sum() for o, w in zip(outs, ws))
    loss.backward()
    igr = {k: inp[k].grad.detach().clone() for k in GRAD_INPUTS[cases.kind(name)]}
    pgr = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in params.items() if p.requires_grad}
    return [o.detach().clone() for o in outs], igr, pgr


def gen_modules():
    mods = ref_modules()
    for name, case in cases.CASES.items():
        if os.environ.get("HWG_GOLDEN_ONLY") and name not in os.environ["HWG_GOLDEN_ONLY"].split(","):
            continue
        m = mods[cases.kind(name)](**case["ctor"])
        m.train()
        sd = torch_ref.seeded_state_dict(m, case["wseed"])
        m.load_state_dict(sd)
        outs, igr, pgr = run_case(name, lambda i: module_fwd(name, m, i), dict(m.named_parameters()), cases.inputs(name))
        # oracle on the same weights (fresh copies: spectral u/v and BN running stats are mutated by a forward)
        sd2 = {k: v.clone() for k, v in sd.items()}
        oparams = {k: sd2[k].requires_grad_(True) for k, p in m.named_parameters() if p.requires_grad}
        oouts, oigr, opgr = run_case(name, lambda i: ORACLE_FWD[cases.kind(name)](sd2, i), oparams, cases.inputs(name))
        worst = 0.0
        for a, b in zip(outs, oouts):
            worst = max(worst, float((a - b).abs().max() / max(a.abs().max(), 1e-6)))
        for k in igr:
            worst = max(worst, float((igr[k] - oigr[k]).abs().max() / max(igr[k].abs().max(), 1e-6)))
        for k in pgr:
            worst = max(worst, float((pgr[k] - opgr[k]).abs().max() / max(pgr[k].abs().max(), 1e-6)))
        print("%-16s oracle vs reference: worst relative error %.2e (outputs, input grads, %d param grads)" % (name, worst, len(pgr)))
        assert worst < 2e-5, name
        names, fp = cases.fingerprint(pgr)
        post = {}
        if cases.kind(name) == "discriminator":  # spectral-norm vectors after the forward
            post = {k.replace(".", "__"): v.detach().numpy() for k, v in m.state_dict().items() if k.endswith("weight_u")}
        if cases.kind(name) == "hwr":
            post = {k.replace(".", "__"): v.detach().numpy() for k, v in m.state_dict().items() if "running_mean" in k}
        np.savez_compressed(os.path.join(GOLD, "module_%s.npz" % name),
                            **{"out%d" % i: o.numpy() for i, o in enumerate(outs)},
                            **{"igrad_" + k: v.numpy() for k, v in igr.items()},
                            pgrad_fp=fp.numpy(), pgrad_names=np.array(json.dumps(names)), **{"post_" + k: v for k, v in post.items()})


def gen_seq():
    """known-answer vectors for the integer algorithms, from the reference's own functions"""
    from model.hw_with_style import correct_pred
    import utils.string_utils as su
    recs = {}
    for n, (T, B, Lr, C, seed) in enumerate([(30, 3, 9, 20, 1), (20, 2, 30, 20, 2), (61, 4, 12, 80, 3), (122, 2, 30, 80, 4), (10, 2, 3, 5, 5)]):
        pred = cases.peaked_logprobs(T, B, C, 300 + seed, 0.5)
        g = torch.Generator().manual_seed(400 + seed)
        label = torch.randint(1, C, (Lr, B), generator=g)
        if n == 0:
            label[Lr - 2:, 0] = 0
        if n == 4:  # exact ties: uniform predictions
            pred = torch.full((T, B, C), -1.6094379)
        out = correct_pred(pred, label)
        recs["dtw%d_pred" % n] = pred.numpy(); recs["dtw%d_label" % n] = label.numpy(); recs["dtw%d_out" % n] = out.numpy()
        dec = [su.naive_decode(pred[:, b].numpy()) for b in range(B)]
        recs["dec%d" % n] = np.array(json.dumps([[int(x) for x in d[0]] for d in dec]))
    # insert_spaces of the reference's HWWithStyle (numpy noise + Python round), seeded
    import json as _json
    from model import HWWithStyle
    cfgm = _json.load(open(os.path.join(GOLD, "model_config_iam.json")))
    hm = HWWithStyle(cfgm)
    gi = torch.Generator().manual_seed(77)
    ilabel = torch.randint(1, 80, (11, 3), generator=gi)
    icounts = torch.rand(11, 3, 2, generator=gi) * 3
    ilens = [11, 8, 5]
    np.random.seed(123)
    ispaced, ipadded = hm.insert_spaces(ilabel, ilens, icounts)
    recs.update(ins_seed=np.array(123), ins_label=ilabel.numpy(), ins_counts=icounts.numpy(), ins_lens=np.array(ilens), ins_spaced=ispaced.numpy(),
                ins_padded=np.array(ipadded))
    np.random.seed(123)
    osp, opad = seq_oracle.insert_spaces(ilabel, ilens, icounts, 80, hm.count_std, hm.dup_std)
    assert torch.equal(osp, ispaced) and opad == ipadded
    # gt counts via the trainer's loop, restated through the reference trainer code path is heavy; the scan is pinned in gen_trainer()
    np.savez_compressed(os.path.join(GOLD, "seq_kat.npz"), **recs)
    # oracle must reproduce them exactly
    for n in range(5):
        o = seq_oracle.correct_pred(torch.from_numpy(recs["dtw%d_pred" % n]), torch.from_numpy(recs["dtw%d_label" % n]))
        assert np.array_equal(o.numpy(), recs["dtw%d_out" % n]), n
    print("seq_kat: 5 DTW cases, oracle exact")


EDGE_DTW = [(1, 2, 1, "rand"), (1, 3, 4, "rand"), (2, 2, 5, "rand"), (3, 1, 1, "rand"), (7, 2, 10, "rand"), (9, 3, 4, "flat"), (40, 2, 3, "flat"),
            (16, 2, 8, "onehot")]


def edge_dtw_case(n):
    """inputs of the n-th degenerate alignment geometry (shared with tests/test_ops_gpu.py through the stored arrays)"""
    import torch.nn.functional as F
    T, B, Lr, kind = EDGE_DTW[n]
    C = 12
    g = torch.Generator().manual_seed(500 + n)
    if kind == "rand":
        pred = F.log_softmax(torch.randn(T, B, C, generator=g) * 3, dim=2)
    elif kind == "flat":
        pred = torch.full((T, B, C), -2.5)
    else:
        pred = F.log_softmax(20.0 * F.one_hot(torch.randint(0, C, (T, B), generator=g), C).float(), dim=2)
    label = torch.randint(1, C, (Lr, B), generator=g)
    if Lr > 2:
        label[Lr - 1:, 0] = 0
    return pred, label


def gen_seq_edges():
    """degenerate geometries of correct_pred (one prediction step, fewer steps than the blank-interleaved label, all-equal costs, one-hot
    predictions, zero-padded tails) through the reference's own function; the oracle must agree exactly"""
    from model.hw_with_style import correct_pred
    recs = {}
    for n in range(len(EDGE_DTW)):
        pred, label = edge_dtw_case(n)
        out = correct_pred(pred, label)
        assert torch.equal(seq_oracle.correct_pred(pred, label), out), EDGE_DTW[n]
        recs["dtw%d_pred" % n] = pred.numpy(); recs["dtw%d_label" % n] = label.numpy(); recs["dtw%d_out" % n] = out.numpy()
    np.savez_compressed(os.path.join(GOLD, "seq_kat_edges.npz"), **recs)
    print("seq_kat_edges: %d DTW cases, oracle exact" % len(EDGE_DTW))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["modules", "seq"]
    if "modules" in what:
        gen_modules()
    if "seq" in what:
        gen_seq()
    if "seq_edges" in what or "seq" in what:
        gen_seq_edges()
    if "trainer" in what:
        import gen_golden_trainer
        gen_golden_trainer.main()
