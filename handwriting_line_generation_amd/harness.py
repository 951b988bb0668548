"""Builds a ready-to-step trainer on synthetic author batches from one of the shipped configs.
Shared by bench.py, __graft_entry__.smoke() and the GPU tests (no dataset / checkpoint can be downloaded here)."""
import copy
import json
import os
import tempfile

import torch

from .data.synthetic import SyntheticAuthorDataset, SyntheticLoader, write_synthetic_corpus
from .model import Autoencoder, HWWithStyle
from .model import loss as loss_fns
from .trainer import HWWithStyleTrainer

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.dirname(os.path.abspath(__file__))
CONFIGS = {
    "iam_gan": "cf_IAMslant_noMask_charSpecSingleAppend_GANMedMT_autoAEMoPrcp2tightNewCTCUseGen_balB_hCF0.75_sMG.json",
    "rimes_gan": "cf_RIMESLinesslant_noMask_charSpecSingleAppend_GANMedMT_autoAEMoPrcp2tightNewCTCUseGen_balB_hCF0.75_sMG.json",
    "iam_hwr": "cf_IAM_hwr_cnnOnly_batchnorm_aug.json",
    "iam_auto": "cf_IAM_auto_2tight_newCTC.json",
}
CHAR_FILES = {"iam": os.path.join(PKG, "data", "IAM_char_set.json"), "rimes": os.path.join(PKG, "data", "RIMES_characterset_lines.json")}


def load_config(which):
    with open(os.path.join(REPO, "configs", CONFIGS[which])) as f:
        return json.load(f)


def synthetic_gan_config(which="iam_gan", batch_size=None, a_batch_size=None, workdir=None, gpu=0):
    """the shipped GAN config with only the file-system entries redirected to synthetic stand-ins"""
    cfg = copy.deepcopy(load_config(which))
    workdir = workdir or tempfile.mkdtemp(prefix="hwg_")
    cfg["cuda"], cfg["gpu"] = True, gpu
    dl = cfg["data_loader"]
    if batch_size is not None:
        dl["batch_size"] = batch_size
    if a_batch_size is not None:
        dl["a_batch_size"] = a_batch_size
    dl["char_file"] = CHAR_FILES["rimes" if "rimes" in which else "iam"]
    cfg["model"]["pretrained_hwr"] = None
    tr = cfg["trainer"]
    tr["save_dir"] = os.path.join(workdir, "saved")
    tr["print_dir"] = None
    corpus = os.path.join(workdir, "corpus.txt")
    if not os.path.exists(corpus):
        write_synthetic_corpus(corpus, dl["char_file"])
    tr["text_data"] = corpus
    tr["encoder_weights"] = os.path.join(workdir, "encoder.pth")
    return cfg, workdir


def build_gan_trainer(which="iam_gan", batch_size=None, a_batch_size=None, width=512, label_len=30, min_width=None, workdir=None,
                      gpu=0, rank=0, world=1, model_state=None, encoder_state=None, data_seed=100, resume=None, curriculum=None):
    if workdir is not None:
        os.makedirs(workdir, exist_ok=True)
    cfg, workdir = synthetic_gan_config(which, batch_size, a_batch_size, workdir, gpu)
    tr = cfg["trainer"]
    if curriculum is not None:
        tr["curriculum"] = {"0": curriculum}
    if not os.path.exists(tr["encoder_weights"]):
        ae = Autoencoder({"type": tr.get("encoder_type", "2tight"), "hwr": cfg["model"]["num_class"]})
        sd = encoder_state if encoder_state is not None else ae.state_dict()   # keys 'encoder.*' are the ones the trainer reads
        torch.save({"state_dict": sd}, tr["encoder_weights"])
    model = HWWithStyle(cfg["model"])
    if model_state is not None:
        model.load_state_dict(model_state)
    dl = cfg["data_loader"]
    ds = SyntheticAuthorDataset(dl["char_file"], dl["batch_size"], dl.get("a_batch_size", 1), width=width, label_len=label_len,
                                min_width=min_width, seed=data_seed)
    loader = SyntheticLoader(ds, rank, world)
    losses = {name: getattr(loss_fns, fn) for name, fn in cfg["loss"].items()}
    trainer = HWWithStyleTrainer(model, losses, [], resume, cfg, loader, None, None)
    return trainer, cfg


def build_simple_trainer(which, batch_size=None, width=512, label_len=30, workdir=None, gpu=0, rank=0, world=1, model_state=None, data_seed=100):
    """trainers of the two pre-training configs: 'iam_hwr' (CTC recogniser, BASELINE configs[0]) and 'iam_auto' (autoencoder, configs[1])"""
    from .trainer import AutoTrainer
    cfg = copy.deepcopy(load_config(which))
    workdir = workdir or tempfile.mkdtemp(prefix="hwg_")
    cfg["cuda"], cfg["gpu"] = True, gpu
    dl = cfg["data_loader"]
    if batch_size is not None:
        dl["batch_size"] = batch_size
    dl["char_file"] = CHAR_FILES["iam"]
    cfg["trainer"]["save_dir"] = os.path.join(workdir, "saved")
    model = (Autoencoder if cfg["arch"] == "Autoencoder" else HWWithStyle)(cfg["model"])
    if model_state is not None:
        model.load_state_dict(model_state)
    ds = SyntheticAuthorDataset(dl["char_file"], dl["batch_size"], dl.get("a_batch_size", 1), width=width, label_len=label_len, seed=data_seed)
    losses = {name: getattr(loss_fns, fn) for name, fn in cfg["loss"].items()}
    cls = AutoTrainer if cfg["trainer"]["class"] == "AutoTrainer" else HWWithStyleTrainer
    return cls(model, losses, [], None, cfg, SyntheticLoader(ds, rank, world), None, None), cfg
