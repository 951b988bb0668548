"""Evaluation loops (reference: new_eval.py:49-, get_styles.py:19-, trainer/hw_with_style_trainer.py:894-914): how well the recogniser reads
real lines and lines generated from their text in their own extracted style, and the per-author style vectors for later generation."""
import pickle

import numpy as np
import torch

from . import ops


def eval_writer(trainer, loader, max_batches=None):
    """-> {"cer_real", "wer_real", "cer_gen", "wer_gen", "styles" [n, style_dim], "authors": [n]} over the batches of `loader`.
    Per batch: recogniser on the real lines; style of every author (lines side by side, generate.get_style semantics); the same texts rendered
    in that style; recogniser on the rendered lines. Everything under no_grad in eval mode (running BatchNorm statistics, no dropout)."""
    model = trainer.model
    was_training = model.training
    model.eval()
    tot = {"cer_real": 0.0, "wer_real": 0.0, "cer_gen": 0.0, "wer_gen": 0.0}
    styles, authors, n = [], [], 0
    try:
        with torch.no_grad():
            for bi, inst in enumerate(loader):
                if max_batches is not None and bi >= max_batches:
                    break
                image, label = trainer._to_tensor(inst)
                a = inst.get("a_batch_size", 1)
                model.pred = model.spaced_label = model.spaced_label_index = None
                pred = model.hwr(image, None)
                model.pred = pred
                style = model.extract_style(image, label, a)                 # [B, style_dim], one vector per author repeated over its lines
                gen = model(label, inst["label_lengths"], style)
                gen_pred = model.hwr(gen, None)
                cr, wr, _ = trainer.getCER(inst["gt"], pred.cpu().numpy())
                cg, wg, _ = trainer.getCER(inst["gt"], gen_pred.cpu().numpy())
                tot["cer_real"] += cr; tot["wer_real"] += wr; tot["cer_gen"] += cg; tot["wer_gen"] += wg
                styles.append(style[::a].cpu())
                authors += list(inst["author"][::a])
                model.pred = model.spaced_label = model.spaced_label_index = None
                n += 1
    finally:
        if was_training:
            model.train()
    out = {k: v / max(n, 1) for k, v in tot.items()}
    out["styles"] = torch.cat(styles, 0).numpy() if styles else np.zeros((0, model.style_dim), dtype=np.float32)
    out["authors"] = authors
    return out


def dump_styles(result, path):
    """the style pickle get_styles.py writes: {"styles": float array [n, style_dim], "authors": [n]}"""
    with open(path, "wb") as f:
        pickle.dump({"styles": result["styles"], "authors": list(result["authors"])}, f)
