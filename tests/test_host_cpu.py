"""CPU: host-side logic of the product path (no kernel is executed) and the C-ABI export check."""
import json
import os
import random
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from handwriting_line_generation_amd import _lib
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T hwg_" in l}
    assert set(_lib.DECLS) <= exported, sorted(set(_lib.DECLS) - exported)
    assert exported <= set(_lib.DECLS), "exported but undeclared: %s" % sorted(exported - set(_lib.DECLS))
    assert _lib.abi_version() == 6 and len(_lib.DECLS) >= 100


def test_kernels_refuse_cpu_tensors():
    from handwriting_line_generation_amd import _lib, ops
    with pytest.raises(_lib.HwgError):
        ops.conv2d(torch.zeros(1, 4, 4, 16), torch.zeros(16, 16, 3, 3))


def test_curriculum_cycle_of_the_shipped_gan_config():
    from handwriting_line_generation_amd.harness import load_config
    from handwriting_line_generation_amd.utils.curriculum import Curriculum
    cur = Curriculum(load_config("iam_gan")["trainer"]["curriculum"])
    got = [cur.getLesson(i) for i in range(9)]
    cyc = [["count"], ["no-step", "gen"], ["auto", "auto-gen"], ["disc"], ["no-step", "gen"], ["auto", "auto-gen"], ["disc"]]
    assert got == cyc + cyc[:2]
    assert set(cur.getValid()) == {"count", "auto", "no-step", "valid"} and "valid" in cur.getValid()
    c2 = Curriculum({"0": [[2, "a"], ["b"]], "5": [["c"]]})
    assert [c2.getLesson(i) for i in range(7)] == [["a"], ["a"], ["b"], ["a"], ["a"], ["c"], ["c"]]


def test_insert_spaces_matches_oracle_and_reference_rounding():
    from handwriting_line_generation_amd.model import HWWithStyle
    from oracle import seq_oracle
    cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "model_config_iam.json")))
    m = HWWithStyle(cfg)
    g = torch.Generator().manual_seed(3)
    label = torch.randint(1, 80, (9, 3), generator=g)
    counts = torch.rand(9, 3, 2, generator=g) * 3
    lens = [9, 7, 4]
    np.random.seed(5)
    a, pa = m.insert_spaces(label, lens, counts)
    np.random.seed(5)
    b, pb = seq_oracle.insert_spaces(label, lens, counts, 80, m.count_std, m.dup_std)
    assert torch.equal(a, b) and pa == pb
    # and the vector recorded from the reference's own insert_spaces (tests/golden/seq_kat.npz)
    kat = np.load(os.path.join(ROOT, "tests", "golden", "seq_kat.npz"))
    np.random.seed(int(kat["ins_seed"]))
    c, pc = m.insert_spaces(torch.from_numpy(kat["ins_label"]), kat["ins_lens"].tolist(), torch.from_numpy(kat["ins_counts"]))
    assert np.array_equal(c.numpy(), kat["ins_spaced"]) and np.allclose(pc, kat["ins_padded"])


def test_text_data_and_string_utils(tmp_path):
    from handwriting_line_generation_amd.data.synthetic import write_synthetic_corpus
    from handwriting_line_generation_amd.data.text_data import TextData
    from handwriting_line_generation_amd.harness import CHAR_FILES
    from handwriting_line_generation_amd.utils import string_utils as su
    corpus = str(tmp_path / "c.txt")
    write_synthetic_corpus(corpus, CHAR_FILES["iam"], 5000)
    td = TextData(corpus, CHAR_FILES["iam"], batch_size=4, max_len=20)
    random.seed(1); np.random.seed(1)
    inst = td.getInstance()
    assert inst["image"] is None and inst["label"].dtype == torch.int32 and inst["label"].shape[1] == 4
    assert all(17 <= len(t) <= 20 for t in inst["gt"]) and inst["label_lengths"].tolist() == [len(t) for t in inst["gt"]]
    logits = np.full((6, 4), -5.0); logits[[0, 1, 2, 3, 4, 5], [1, 1, 0, 2, 2, 1]] = 0
    assert su.naive_decode(logits)[0] == [1, 2, 1]
    assert su.cer("abc", "abd") == pytest.approx(1 / 3) and su.wer("a b", "a c") == 0.5 and su.cer("", "xy") == 2


def test_flat_params_bookkeeping_on_cpu():
    from handwriting_line_generation_amd.trainer.flat_params import FlatParams
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    ps = list(net.parameters())
    f = FlatParams(ps, {"main": ps[:2], "disc": ps[2:]})
    net(torch.randn(5, 4)).sum().backward()
    assert f.touched.all() and float(f.flat_grad.abs().sum()) > 0
    st = f.stash()
    assert float(f.flat_grad.abs().sum()) == 0 and st[1].all() and float(st[0].abs().sum()) > 0
    f.zero_grad("main")
    assert f.touched.tolist() == [False, False, True, True]
    assert ps[0].grad.data_ptr() == f.flat_grad.data_ptr()


def test_getcer_known_answers_from_the_reference():
    """getCER = greedy CTC decode + CER / WER (reference: trainer/hw_with_style_trainer.py:894-914, utils/error_rates.py:2-26): the answers of
    the unmodified reference on oracle/cer_kats.py's inputs (tests/golden/valid_gan.json), case sensitive and insensitive"""
    import json
    from oracle import cer_kats
    from handwriting_line_generation_amd.trainer.hw_with_style_trainer import HWWithStyleTrainer
    from handwriting_line_generation_amd.trainer.auto_trainer import AutoTrainer
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "valid_gan.json")))
    pkg = os.path.join(os.path.dirname(os.path.dirname(__file__)), "handwriting_line_generation_amd", "data", "IAM_char_set.json")
    idx_to_char = {int(k): v for k, v in json.load(open(pkg))["idx_to_char"].items()}
    stub = type("T", (), {"idx_to_char": idx_to_char, "casesensitive": True})()
    cases = cer_kats.cases(idx_to_char, len(idx_to_char) + 1)
    assert len(cases) == len(gold["cer_kats"]) == 6
    for (pred, texts, casesens), ref in zip(cases, gold["cer_kats"]):
        stub.casesensitive = casesens
        cer, wer, strs = HWWithStyleTrainer.getCER(stub, texts, pred)
        assert strs == ref["strs"]
        assert abs(cer - ref["cer"]) < 1e-12 and abs(wer - ref["wer"]) < 1e-12, (cer, ref["cer"], wer, ref["wer"])
        if casesens:      # the autoencoder trainer's getCER has no case switch (trainer/auto_trainer.py:321-342)
            cer2, wer2, strs2 = AutoTrainer.getCER(stub, texts, pred)
            assert strs2 == strs and abs(cer2 - ref["cer"]) < 1e-12 and abs(wer2 - ref["wer"]) < 1e-12
    assert any(r["cer"] > 0 for r in gold["cer_kats"])       # the corrupted trials really have errors


def test_call_thunks_refuse_values_that_do_not_fit_the_c_types():
    """the generated fast-call thunks narrow Python ints to the `int` / `size_t` the C-ABI declares: out-of-range values raise instead of wrapping"""
    from handwriting_line_generation_amd import _lib as L
    with pytest.raises(OverflowError):
        L.call("hwg_prof_start", 1 << 40)                       # int capacity
    with pytest.raises(OverflowError):
        L.call("hwg_conv_pack_weight", None, None, 1 << 33, *([1] * 9), None)   # int A (13 arguments: checked before the call is made)


def test_replay_executor_runs_a_call_list_and_classifies_stream_parameters():
    """the C launch-list executor behind replay.py (csrc/hwg_pycall.c, generated): a list of status-returning entry points runs in one call, stops at
    the first non-zero status and reports its index; malformed lists are refused before anything is called; the table it is driven by marks every
    parameter hwg.h names *stream as a stream (the recorder re-points exactly those at the replaying pass's streams, nothing else)"""
    from handwriting_line_generation_amd import _lib as L
    funcs = L._hwgcall.replay_functions()
    assert funcs["hwg_stream_fork"][1] == "ss" and funcs["hwg_stream_join"][1] == "ss"
    header = open(os.path.join(os.path.dirname(__file__), "..", "include", "hwg.h")).read()
    assert sum(k.count("s") for _, k in funcs.values()) == header.count("void* stream") + 4     # + fork / join's two each
    for name, (fid, kinds) in funcs.items():
        assert set(kinds) <= set("psif"), (name, kinds)
    reload_id, abi_id = funcs["hwg_tuning_reload"][0], funcs["hwg_abi_version"][0]
    empty64, table = np.zeros(1, np.int64), np.zeros(1, np.uint64)
    run = lambda recs: L._hwgcall.replay(np.asarray(recs, np.int32).reshape(-1, 3), np.zeros(0, np.uint8), empty64, empty64, table)
    assert run([(reload_id, 0, 0), (reload_id, 0, 0)]) == (0, -1)
    assert run([(reload_id, 0, 0), (abi_id, 0, 0), (reload_id, 0, 0)]) == (L._FN["hwg_abi_version"](), 1)      # a non-zero status stops the list
    assert run([(len(funcs), 0, 0)])[0] == -1002 and run([(funcs["hwg_prof_start"][0], 0, 0)])[0] == -1002   # unknown id; wrong argument count
    recs = np.asarray([(funcs["hwg_prof_start"][0], 0, 1)], np.int32)
    assert L._hwgcall.replay(recs, np.asarray([3], np.uint8), np.asarray([5], np.int64), empty64, table)[0] == -1003   # table slot out of range


def test_winograd_planner_balanced_schedule_bookkeeping():
    """the balanced schedule of the 64 x 64 Winograd kernel (csrc/conv_wino.hip, wino_balance): the planner's bookkeeping - which tiles run
    whole, how many workgroups share the rest, the most pieces a tile is cut into (= workspace images), the most tiles a workgroup touches -
    against an independent recomputation, and the closed form the kernel and the reduce pass use to find a unit's owner against the runs
    themselves (workgroup g owns units [U g / G, U (g + 1) / G)). No launch: hwg_wino_conv_describe only plans."""
    from handwriting_line_generation_amd import ops, _lib as L

    def describe(N, H, W, C, K, R, S, stride, pad, P, Q, tr=0):
        d = ops._desc(N, H, W, C, K, R, S, stride, pad, (1, 1), P, Q, tr)
        out = np.full(8, -7, dtype=np.int32)
        L.call("hwg_wino_conv_describe", d.ptr, out.ctypes.data)
        return [int(v) for v in out], d

    def check(out, d, workspace_fn):
        cfg, ns, G, lead, pieces, tiles, chunks, segs = out
        assert cfg == 6 and G > 0 and lead % 256 == 0 and 0 <= lead < tiles, out
        units = (tiles - lead) * chunks
        owner = lambda u: ((u + 1) * G - 1) // units                      # noqa: E731 - the closed form of wino_bal_owner
        runs = [(units * g // G, units * (g + 1) // G) for g in range(G)]
        assert runs[0][0] == 0 and runs[-1][1] == units and all(a[1] == b[0] for a, b in zip(runs, runs[1:]))
        for g, (u0, u1) in enumerate(runs):
            assert all(owner(u) == g for u in (u0, u1 - 1) if u1 > u0), (g, u0, u1)
        cut = [owner(t * chunks + chunks - 1) - owner(t * chunks) + 1 for t in range(tiles - lead)]
        assert max(cut) == pieces, (max(cut), pieces)
        assert max((u1 - 1) // chunks - u0 // chunks + 1 for u0, u1 in runs if u1 > u0) == segs
        out_bytes = 4 * d.N * d.P * d.Q * d.K
        assert L.query(workspace_fn, d.ptr) == (pieces * out_bytes if pieces > 1 else 0)

    with ops.tuning(HWG_WINO="1", HWG_WINO_BAL=None, HWG_WINO_FORCE=None, HWG_WINO_S2="1"):
        # the recogniser layer that used half the chip (132 tiles of 32 chunks): all of it balanced over 256 workgroups
        out, d = describe(8, 8, 129, 512, 256, 3, 3, (1, 1), (1, 1), 8, 129)
        assert out[:4] == [6, 1, 256, 0] and out[5:7] == [132, 32], out
        check(out, d, "hwg_wino_conv_workspace")
        # 260 tiles: one whole round of 256 + the 4 leftover tiles cut into single-chunk runs
        out, d = describe(4, 13, 256, 256, 256, 3, 3, (1, 1), (2, 2), 15, 258)
        assert out[2] > 0 and out[3] == 256 and out[5] == 260, out
        check(out, d, "hwg_wino_conv_workspace")
        # a layer that fills its rounds keeps the uniform schedule
        out, d = describe(8, 16, 128, 256, 256, 3, 3, (1, 1), (1, 1), 16, 128)
        assert out[0] == 6 and out[2] == 0 and out[5] == 256, out
        # the style extractor's 4x4 stride-2 data gradient as a two-tap problem (described like the fractionally strided product)
        out, d = describe(4, 15, 256, 256, 128, 4, 4, (2, 2), (0, 0), 32, 514, tr=1)
        assert out[0] == 6 and out[6] == 16, out
        if out[2] > 0:
            check(out, d, "hwg_wino_s2_workspace")
    for force in ("3", "7", "40", "64,0", "9,256"):
        with ops.tuning(HWG_WINO="2", HWG_WINO_BAL=force):
            out, d = describe(8, 66, 130, 32, 64, 3, 3, (1, 1), (1, 1), 66, 130)       # 269 tiles of 2 chunks
            assert out[5] == 269 and out[6] == 2 and out[2] == min(int(force.split(",")[0]), (269 - out[3]) * 2), (force, out)
            check(out, d, "hwg_wino_conv_workspace")
    # a 4x4 stride-2 layer the two-tap path does not take (odd image: the data gradient's last row would never be written) reports -1s
    out, _ = describe(4, 15, 256, 256, 128, 4, 4, (2, 2), (0, 0), 33, 514, tr=1)
    assert out == [-1] * 8
