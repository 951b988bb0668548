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
