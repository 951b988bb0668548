"""Autoencoder pre-training step on HIP kernels (reference: trainer/auto_trainer.py:79-177, 255-319).
Batch -> pad the width to a multiple of 8 with -1 -> Autoencoder (Encoder2 -> DecoderNoSkip, E_HWR) -> L1 + CTC -> backward
-> clip_grad_value_(2) -> Adam. The perceptual encoder of the GAN trainer is the encoder trained here."""
import json

import numpy as np
import torch

from .. import ops
from ..base.base_trainer import BaseTrainer
from ..utils import string_utils
from .flat_params import allreduce_gradient_sets
from .hw_with_style_trainer import PADDING_CONSTANT, _pad_w


class AutoTrainer(BaseTrainer):
    def __init__(self, model, loss, metrics, resume, config, data_loader, valid_data_loader=None, train_logger=None):
        super().__init__(model, loss, metrics, resume, config, train_logger)
        tr = config["trainer"]
        self.loss_params = dict(config.get("loss_params", {}))
        for name in self.loss:
            self.loss_params.setdefault(name, {})
        self.lossWeights = config.get("loss_weights", {"auto": 1, "recog": 1})
        if data_loader is not None:
            self.batch_size = data_loader.batch_size
            self.data_loader = data_loader
            self.data_loader_iter = iter(data_loader)
        self.valid_data_loader = valid_data_loader
        self.valid = valid_data_loader is not None
        with open(config["data_loader"]["char_file"]) as f:
            self.idx_to_char = {int(k): v for k, v in json.load(f)["idx_to_char"].items()}
        self.num_class = len(self.idx_to_char) + 1
        # the reference hard-codes both (trainer/auto_trainer.py:40-41): width differences are padded on the right, no fg-mask weighting
        self.center_pad = False
        self.no_bg_loss = False
        import torch.distributed as dist
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        ops.SIDE_WGRAD = bool(tr.get("side_stream_wgrad", False))   # process-global switch: every trainer states its own choice
        import os
        self._defer_reduce = bool(int(tr.get("defer_wgrad_reduce", os.environ.get("HWG_DEFER_REDUCE", "1")) or 0))

    def _next_instance(self):
        try:
            return next(self.data_loader_iter)
        except StopIteration:
            self.data_loader_iter = iter(self.data_loader)
            return next(self.data_loader_iter)

    def _train_iteration(self, iteration):
        if not self.model.training:
            self.model.train()
        instance = self._next_instance()
        self.optimizer.zero_grad()
        losses = self.run_gen(instance)
        loss = 0
        scaled = {}
        for name, v in losses.items():
            v = ops.scale(v, self.lossWeights[name[:-4]])
            scaled[name] = v
            loss = v if isinstance(loss, int) else ops.add(loss, v)
        ops.DEFER_REDUCE = self._defer_reduce     # partial-image sums of the pass in one table-driven launch (ops.flush_deferred_reduce)
        try:
            loss.backward()
        finally:
            ops.DEFER_REDUCE = False
            ops.join_side_stream()
        allreduce_gradient_sets(self.flat, (), self.world, self.gpu)
        if getattr(self, "pre_clip_hook", None) is not None:     # parity tests read the gradients where the reference clips them
            self.pre_clip_hook(iteration)
        self.flat.clip_(2)
        self.optimizer.step()
        names = list(scaled)
        host = torch.cat([scaled[n].detach().reshape(1) for n in names]).cpu().tolist()
        log = dict(zip(names, host))
        return {"loss": sum(log.values()), **log}

    def run_gen(self, instance, get=[]):
        image = ops.h2d(instance["image"], self.gpu)
        label = ops.h2d(instance["label"], self.gpu)
        if image.size(3) % 8 > 0:
            p = 8 - image.size(3) % 8
            image = _pad_w(image, p // 2, p // 2 + p % 2, "constant", PADDING_CONSTANT)
        if "recog" in self.loss:
            recon, pred = self.model(image)
        else:
            recon, pred = self.model(image), None
        losses = {}
        if "auto" in self.loss:
            d = recon.size(3) - image.size(3)
            if d > 0:
                image = _pad_w(image, d // 2, d // 2 + d % 2, "constant", PADDING_CONSTANT) if self.center_pad else _pad_w(image, 0, d, "constant", PADDING_CONSTANT)
            elif d < 0:
                recon = _pad_w(recon, (-d) // 2, (-d) // 2 + (-d) % 2, "constant", PADDING_CONSTANT) if self.center_pad else _pad_w(recon, 0, -d, "constant", PADDING_CONSTANT)
            losses["autoLoss"] = self.loss["auto"](recon, image, **self.loss_params["auto"])
        if pred is not None:
            B = pred.size(1)
            losses["recogLoss"] = self.loss["recog"](pred, label.permute(1, 0), [pred.size(0)] * B, instance["label_lengths"])
        if get:
            return losses, {k: v for k, v in (("recon", recon), ("pred", pred)) if k in get}
        return losses

    def getCER(self, gt, pred, individual=False):
        """mean CER and WER over the lines of a batch + the decoded strings (trainer/auto_trainer.py:321-342)"""
        cer = wer = 0
        pred_strs, all_cer = [], []
        for i, g in enumerate(gt):
            s, _ = string_utils.naive_decode(pred[:, i])
            s = string_utils.label2str_single(s, self.idx_to_char, False)
            this = string_utils.cer(g, s)
            cer += this
            all_cer.append(this)
            wer += string_utils.wer(g, s)
            pred_strs.append(s)
        if individual:
            return cer / len(gt), wer / len(gt), pred_strs, all_cer
        return cer / len(gt), wer / len(gt), pred_strs

    def _valid_epoch(self):
        """validation pass (trainer/auto_trainer.py:199-245): weighted losses of every validation batch under no_grad, averaged over the
        batches; CER / WER of the E_HWR head when the config has a 'recog' loss"""
        self.model.eval()
        totals = {}
        total_loss = total_cer = total_wer = 0.0
        n = 0
        with torch.no_grad():
            for instance in self.valid_data_loader:
                losses, got = self.run_gen(instance, ["pred"] if "recog" in self.loss else ["none"])
                for name, v in losses.items():
                    w = float(v) * self.lossWeights[name[:-4]]
                    total_loss += w
                    totals["val_" + name] = totals.get("val_" + name, 0.0) + w
                if "recog" in self.loss:
                    cer, wer, _ = self.getCER(instance["gt"], got["pred"].detach().cpu().numpy())
                    total_cer += cer
                    total_wer += wer
                n += 1
        n = max(n, 1)
        out = {"val_loss": total_loss / n, **{k: v / n for k, v in totals.items()}}
        if "recog" in self.loss:
            out["val_CER"] = total_cer / n
            out["val_WER"] = total_wer / n
        return out
