"""CPU: checkpoints written by the REFERENCE's `_save_checkpoint` (tests/golden/ref_ckpt_*.pth.xz, made by tools/gen_golden_checkpoint.py) unpickle
here - the pickled `logger.logger.Logger` resolves through the shim - and carry the layout `_resume_checkpoint` expects."""
import json
import lzma
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def unpack(name, tmp_path):
    dst = os.path.join(str(tmp_path), "ref_ckpt_%s.pth" % name)
    with open(dst, "wb") as f:
        f.write(lzma.decompress(open(os.path.join(GOLD, "ref_ckpt_%s.pth.xz" % name), "rb").read()))
    return dst


@pytest.mark.parametrize("name,arch,iteration", [("gan", "HWWithStyle", 25000), ("hwr", "HWWithStyle", 100000), ("auto", "Autoencoder", 60000)])
def test_reference_checkpoint_unpickles(tmp_path, name, arch, iteration):
    from handwriting_line_generation_amd.logger import Logger, load_checkpoint
    ck = load_checkpoint(unpack(name, tmp_path))
    assert ck["arch"] == arch and ck["iteration"] == iteration and ck["monitor_best"] == 0.5
    assert isinstance(ck["logger"], Logger) and ck["logger"].entries == {1: {"iteration": iteration, "loss": 1.25}}
    assert set(ck) >= {"arch", "iteration", "logger", "optimizer", "monitor_best", "config", "state_dict"}
    st = ck["optimizer"]["state"]
    assert len(st) > 0 and all(set(v) >= {"step", "exp_avg", "exp_avg_sq"} for v in st.values())
    assert ck["optimizer"]["param_groups"][0]["params"] == list(range(len(ck["optimizer"]["param_groups"][0]["params"])))


def test_reference_gan_checkpoint_matches_this_models_schema(tmp_path):
    """same keys and shapes as a freshly built (width-reduced) HWWithStyle of this package; the Adam state covers the main optimizer's parameters"""
    from handwriting_line_generation_amd.harness import load_config
    from handwriting_line_generation_amd.logger import load_checkpoint
    from handwriting_line_generation_amd.model import HWWithStyle
    ck = load_checkpoint(unpack("gan", tmp_path))
    cfg = dict(load_config("iam_gan")["model"], pretrained_hwr=None, **json.load(open(os.path.join(GOLD, "ref_ckpt_reduced_model.json"))))
    m = HWWithStyle(cfg)
    want = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    got = {k: tuple(v.shape) for k, v in ck["state_dict"].items() if not k.startswith("style_from_normal")}
    assert got == want
    n_main = sum(1 for k, p in m.named_parameters() if "discriminator" not in k and not k.startswith("hwr."))
    assert len(ck["optimizer"]["state"]) == n_main
