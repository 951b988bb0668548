"""CPU: the gate-decision records (oracle/gates.py) - block hashes, near-zero lists and the content matcher count exactly the flips
that were planted, whatever the order of calls and samples; the committed records of the reference's fp64 runs (tests/golden/tf_*_gates.npz,
tools/gen_golden_tf.py) load and carry every gated module of the reference's hot path."""
import os

import numpy as np

from oracle import gates

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _records(rng, shapes):
    recs, pres = [], []
    for i, (name, kind, N, M) in enumerate(shapes):
        if kind == "act":
            pre = rng.standard_normal((N, M))
            dec = (pre > 0).astype(np.uint8)
            nzi, nzv = gates.near_zero(pre.reshape(-1))
        else:
            pre = None
            dec = gates.pool_codes(rng.integers(0, 4, (N, M)), rng.random((N, M)) > 0.3)
            nzi, nzv = np.full(gates.NZ, -1, np.int64), np.zeros(gates.NZ)
        recs.append({"name": name, "kind": kind, "N": N, "M": M, "hashes": gates.block_hashes(dec), "nz_idx": nzi, "nz_val": nzv, "ref32_flips": 0})
        pres.append(dec)
    return recs, pres


def test_matcher_counts_planted_flips_under_any_call_and_sample_order():
    rng = np.random.default_rng(0)
    shapes = [("hwr.cnn.relu0", "act", 4, 70000), ("hwr.cnn.pooling0", "pool", 4, 17500), ("generator.style_emb.2", "act", 4, 128),
              ("discriminator.convs1.1", "act", 8, 70000), ("style_extractor.char_extractor.5.conv1.0", "act", 9, 1280),
              ("style_extractor.char_extractor.7.conv1.0", "act", 3, 1280)]
    recs, decs = _records(rng, shapes)
    m = gates.Matcher(recs)
    planted = {}
    order = rng.permutation(len(shapes))
    for i in order:
        name, kind, N, M = shapes[i]
        d = decs[i].copy()
        if kind == "act":
            # flip the decisions with the smallest |pre-activation| (what another fp32 implementation would flip), 0 - 3 of them
            k = int(rng.integers(0, 4))
            for j in recs[i]["nz_idx"][:k]:
                d.reshape(-1)[j] ^= 1
            planted[name] = k
        else:
            d[1, 5] = (d[1, 5] + 1) % 5
            planted[name] = 1
        perm = rng.permutation(N)
        # the other side runs half of the samples in one call and the rest in another, in another order
        m.feed(kind, d[perm[: N // 2]])
        m.feed(kind, d[perm[N // 2:]])
    assert m.unmatched_other == 0 and m.unmatched_reference() == []
    for name, k in planted.items():
        assert m.flips.get(name, 0) == k, (name, m.flips.get(name, 0), k)
    assert m.flips_by_network()["hwr"] == planted["hwr.cnn.relu0"] + 1
    # a tensor the reference has no record for stays unmatched and is reported, an optional one silently
    m.feed("act", np.ones((2, 333), np.uint8))
    m.feed("act", np.ones((2, 333), np.uint8), optional=True)
    assert m.unmatched_other == 2


def test_hashes_do_not_depend_on_the_batch_a_sample_sits_in():
    rng = np.random.default_rng(1)
    d = (rng.random((6, 5000)) > 0.5).astype(np.uint8)
    h = gates.block_hashes(d)
    assert h.shape == (6, 10) and h.dtype == np.uint64          # block_size(5000) = 512
    assert np.array_equal(gates.block_hashes(d[2:4]), h[2:4])
    d2 = d.copy(); d2[3, 4999] ^= 1
    h2 = gates.block_hashes(d2)
    assert (h2 != h).sum() == 1 and h2[3, 9] != h[3, 9]


def test_committed_gate_records_of_the_reference_cover_its_gated_modules():
    for case, units in (("tf_full", 5), ("tf_trained", 5)):
        g = gates.load(os.path.join(GOLD, "%s_gates.npz" % case))
        assert sorted(g) == ["0:0", "1:1", "1:2", "2:3", "3:4", "3:5", "4:6"]
        nets = {r["name"].split(".")[0] for recs in g.values() for r in recs}
        assert {"hwr", "generator", "discriminator", "style_extractor", "spacer", "encoder"} <= nets
        for recs in g.values():
            for r in recs:
                assert r["hashes"].shape[0] == r["N"] and r["ref32_flips"] >= 0
                if r["kind"] == "act":
                    k = (r["nz_idx"] >= 0).sum()
                    assert k == min(gates.NZ, r["N"] * r["M"]) and np.all(np.diff(np.abs(r["nz_val"][:k])) >= 0)
        # the auto lesson walks the recogniser's 7 ReLU + 4 ReLU(1-D) + 4 max-pool gates twice (real lines, reconstruction)
        auto = [r["name"] for r in g["1:2"] if r["name"].startswith("hwr.")]
        assert len(auto) == 30
