"""GPU: checkpoint round trip of the GAN trainer in the reference's format (base/base_trainer.py:340-479: arch, iteration, optimizer,
config, state_dict with the reference's 1411 keys): save after a few lessons, resume into a fresh trainer, same weights and Adam state."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_save_resume_round_trip(cuda, tmp_path):
    import json
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    rng.set_mode("device", seed=5)
    torch.manual_seed(0)
    a, _ = build_gan_trainer("iam_gan", 1, 2, width=256, label_len=12, workdir=str(tmp_path / "a"))
    for it in range(4):
        a._train_iteration(it)
    a.iteration = 3
    path = a._save_checkpoint(3, {})
    ck = torch.load(path, map_location="cpu", weights_only=False)
    schema = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_schema_iam.json")))
    assert set(ck["state_dict"]) == set(schema), "checkpoint keys differ from the reference's state-dict schema"
    assert ck["arch"] == "HWWithStyle" and ck["iteration"] == 3 and "optimizer" in ck and "config" in ck
    torch.manual_seed(1)     # different random init: everything must come from the checkpoint
    b, _ = build_gan_trainer("iam_gan", 1, 2, width=256, label_len=12, workdir=str(tmp_path / "b"), resume=path)
    assert b.start_iteration == 4
    sa, sb = a.model.state_dict(), b.model.state_dict()
    for k in sa:
        assert torch.equal(sa[k].cpu(), sb[k].cpu()), k
    oa, ob = a.optimizer.state_dict(), b.optimizer.state_dict()
    assert set(oa["state"]) == set(ob["state"]) and len(oa["state"]) > 0
    for j in oa["state"]:
        for f in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(oa["state"][j][f].cpu(), ob["state"][j][f].cpu()), (j, f)
    # the resumed trainer keeps training
    log = b._train_iteration(4)
    assert all(v == v for v in log.values() if isinstance(v, float))


def test_validation_epoch_runs(cuda, tmp_path):
    """_valid_epoch (trainer :437-486): the validation lesson of the curriculum on a held-out loader under no_grad, weighted losses averaged"""
    import math
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.data.synthetic import SyntheticAuthorDataset, SyntheticLoader
    from handwriting_line_generation_amd.harness import build_gan_trainer
    rng.set_mode("device", seed=5)
    torch.manual_seed(0)
    tr, cfg = build_gan_trainer("iam_gan", 1, 2, width=256, label_len=12, workdir=str(tmp_path))
    dl = cfg["data_loader"]
    ds = SyntheticAuthorDataset(dl["char_file"], 1, 2, width=256, label_len=12, num_batches=3, seed=900)
    tr.valid_data_loader = SyntheticLoader(ds)
    before = {k: v.detach().clone() for k, v in list(tr.model.named_parameters())[:20]}
    val = tr._valid_epoch()
    assert val and all(k.startswith("val_") and math.isfinite(v) for k, v in val.items()), val
    for k, v in list(tr.model.named_parameters())[:20]:
        assert torch.equal(v, before[k]), "validation changed parameter %s" % k
    log = tr._train_iteration(0)     # back to training mode afterwards
    assert tr.model.training and all(math.isfinite(v) for v in log.values() if isinstance(v, float))


def test_eval_writer_and_style_dump(cuda, tmp_path):
    """evaluate.eval_writer (new_eval.py / get_styles.py semantics): recogniser CER on real lines equals what getCER gives on the recogniser's
    eval-mode output, generated lines are scored against the same texts, one style per author, deterministic under fixed host seeds"""
    import math
    import pickle
    import numpy as np
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.data.synthetic import SyntheticAuthorDataset, SyntheticLoader
    from handwriting_line_generation_amd.evaluate import dump_styles, eval_writer
    from handwriting_line_generation_amd.harness import build_gan_trainer
    rng.set_mode("host")
    try:
        torch.manual_seed(0)
        tr, cfg = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=10, workdir=str(tmp_path))
        ds = SyntheticAuthorDataset(cfg["data_loader"]["char_file"], 2, 2, width=256, label_len=10, num_batches=2, seed=700)
        torch.manual_seed(1); np.random.seed(1)
        r1 = eval_writer(tr, SyntheticLoader(ds))
        torch.manual_seed(1); np.random.seed(1)
        r2 = eval_writer(tr, SyntheticLoader(ds))
        assert tr.model.training
        for k in ("cer_real", "wer_real", "cer_gen", "wer_gen"):
            assert math.isfinite(r1[k]) and r1[k] >= 0 and r1[k] == r2[k]
        assert r1["styles"].shape == (4, 128) and len(r1["authors"]) == 4 and np.array_equal(r1["styles"], r2["styles"])
        # the real-line CER is getCER of the eval-mode recogniser output
        inst = ds.batch(0)
        tr.model.eval()
        with torch.no_grad():
            pred = tr.model.hwr(inst["image"].to(cuda), None).cpu().numpy()
        tr.model.train()
        inst1 = ds.batch(1)
        tr.model.eval()
        with torch.no_grad():
            pred1 = tr.model.hwr(inst1["image"].to(cuda), None).cpu().numpy()
        tr.model.train()
        want = (tr.getCER(inst["gt"], pred)[0] + tr.getCER(inst1["gt"], pred1)[0]) / 2
        assert abs(r1["cer_real"] - want) < 1e-9
        path = str(tmp_path / "styles.pkl")
        dump_styles(r1, path)
        back = pickle.load(open(path, "rb"))
        assert set(back) == {"styles", "authors"} and back["styles"].shape == (4, 128)
    finally:
        rng.set_mode("device")
