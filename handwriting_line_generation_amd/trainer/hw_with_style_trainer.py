"""GAN / recogniser training step on HIP kernels (reference: trainer/hw_with_style_trainer.py:21-1023).

`_train_iteration` keeps the reference's control flow - curriculum lesson, loss weighting, the up-to-three backward
passes of the gradient-balancing scheme, gradient stashing on "no-step" lessons, clip to +-2, optimizer choice - but
every per-parameter Python loop is a multi-tensor kernel over the flat gradient buffer (flat_params.py) and the loss
values are fetched with one device->host copy per iteration instead of one per loss.

Data parallelism (absent from the reference): with torch.distributed initialised every rank runs the same lesson on
its own author shard; gradient sets are averaged with one all-reduce (RCCL over xGMI) at the points where the reference
reads them (before balancing / clipping), and the `None`-gradient masks are OR-ed so all ranks update the same tensors.
"""
import collections
import os
import json
import random
from collections import defaultdict

import numpy as np
import torch
import torch.distributed as dist

from .. import ops
from ..base.base_trainer import BaseTrainer
from ..logger import load_checkpoint
from . import flat_params as flat_params_mod
from .flat_params import allreduce_gradient_sets, start_stash_allreduce
from ..data.text_data import TextData
from ..model.autoencoder import Encoder2
from ..model.hw_with_style import correct_pred
from ..utils import string_utils

PADDING_CONSTANT = -1  # datasets/hw_dataset.py


def _pad_w(x, left, right, mode="constant", value=0.0):
    """F.pad on the width of an NCHW single-channel image, through the pad kernel"""
    assert x.shape[1] == 1
    y = ops.pad2d(ops.to_nhwc(x), left, right, 0, 0, mode, value)
    return ops.to_nchw(y)


class HWWithStyleTrainer(BaseTrainer):
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
        self.valid_data_loader = None if self.val_step < 0 else valid_data_loader
        self.valid = self.valid_data_loader is not None

        with open(config["data_loader"]["char_file"]) as f:
            char_set = json.load(f)
        self.idx_to_char = {int(k): v for k, v in char_set["idx_to_char"].items()}
        self.num_class = len(self.idx_to_char) + 1

        self.gan_loss = "discriminator" in config["model"]
        text_bs = tr.get("text_data_batch_size", config["data_loader"]["batch_size"])
        if "a_batch_size" in config["data_loader"]:
            self.a_batch_size = config["data_loader"]["a_batch_size"]
            text_bs *= self.a_batch_size
        else:
            self.a_batch_size = 1
        self.text_data = None
        if data_loader is not None and "text_data" in tr:
            max_len = tr.get("text_data_max_len", self.data_loader.dataset.max_len())
            self.text_data = TextData(tr["text_data"], config["data_loader"]["char_file"], text_bs, max_len=max_len,
                                      words=tr.get("text_words", False), characterBalance=tr.get("character_balance", False))

        self.balance_loss = tr.get("balance_loss", False)
        if self.balance_loss:
            self.balance_var_x = tr.get("balance_var_x")
            if isinstance(self.balance_loss, str) and self.balance_loss.startswith("sign_preserve_x"):
                self.balance_x = float(self.balance_loss[self.balance_loss.find("x") + 1:])
            self.saved_grads = []
        self.style_detach = tr.get("detach_style", tr.get("style_detach", False))

        self.interpolate_gen_styles = tr.get("interpolate_gen_styles", False)
        if isinstance(self.interpolate_gen_styles, str) and self.interpolate_gen_styles.startswith("extra-"):
            e = float(self.interpolate_gen_styles[6:])
            self.interpolate_gen_styles_low, self.interpolate_gen_styles_high = -e, 1 + e
        else:
            self.interpolate_gen_styles_low, self.interpolate_gen_styles_high = 0, 1
        self.prev_styles_size = tr.get("prev_style_size", 100)
        self.prev_styles = []
        self.prev_g_styles = []
        self.sometimes_interpolate = tr.get("sometimes_interpolate", False)
        self.interpolate_freq = tr.get("interpolate_freq", 0.5)
        # NB looked up at the top level of the config (where the shipped configs do not have it) -> False, as in the reference
        self.no_bg_loss = tr["no_bg_loss"] if "no_bg_loss" in config else False

        if "encoder_weights" in tr:
            etype = tr.get("encoder_type", "normal")
            dims = {"2": 256, "2tight": 32, "2tighter": 16}
            if etype not in dims:
                raise NotImplementedError("perceptual encoder type %r: only the Encoder2 family is on the accelerated path" % etype)
            snap = load_checkpoint(tr["encoder_weights"])
            enc_sd = {k[8:]: v for k, v in snap["state_dict"].items() if k.startswith("encoder.")}
            self.encoder = Encoder2(dims[etype])
            self.encoder.load_state_dict(enc_sd)
            self.encoder = self.encoder.to(self.gpu)   # stays in train mode: its Dropout2d is live in the perceptual loss

        self.print_dir = None  # sample image dumps need torchvision; not part of the accelerated path
        self.casesensitive = tr.get("casesensitive", True)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # control-plane group for host-side decisions every rank must take together (skip an iteration): CPU tensors over gloo, so
        # the exchange never waits for the GPU stream the way a device collective + .item() would
        self._dp = self.world > 1 or (flat_params_mod.FORCE_DP and dist.is_available() and dist.is_initialized())
        self._ctl_group = flat_params_mod.control_group() if self._dp else None
        # pipelined logging: 0 / False = the reference's behaviour (each iteration returns its own losses: the host drains the GPU every
        # iteration), n >= 1 = iteration i returns the losses of iteration i - n, so the host may run up to n iterations ahead of the GPU
        self.async_log = tr.get("async_log", False)
        # optional: weight-gradient kernels on a second HIP stream (fills the CUs the data-gradient chain leaves idle: +2.7 % steps/s);
        # off by default because co-running kernels inflate the per-kernel durations the roofline measurement relies on
        # "auto": only in the lessons with the long backward passes (auto / auto-gen: three passes over every network), where the GPU is
        # the bottleneck; the short lessons are bound by the host's enqueue rate and every extra call costs there
        side = tr.get("side_stream_wgrad", os.environ.get("HWG_SIDE_WGRAD", "0"))
        self._side_wgrad = "auto" if side == "auto" else bool(int(side or 0)) if isinstance(side, str) else bool(side)
        ops.SIDE_WGRAD = self._side_wgrad is True
        self._pending_log = collections.deque()
        self.pre_clip_hook = None
        self._defer_reduce = bool(int(tr.get("defer_wgrad_reduce", os.environ.get("HWG_DEFER_REDUCE", "1")) or 0))
        # clip + NaN check + Adam of a stepping lesson as ONE launch (HipAdam.step(clip)); 0 = the three separate passes (A/B timing, tests)
        self._fused_step = bool(int(tr.get("fused_step", os.environ.get("HWG_FUSED_STEP", "1")) or 0)) and hasattr(self.optimizer, "flat")
        # launch-list replay of the frozen recogniser (replay.py): the switches its backward passes run under
        from .. import replay as _replay
        _replay.BACKWARD_FLAGS = {(self._defer_reduce, False), (self._defer_reduce, True)}
        # the two or three gradients a balanced lesson sends through the generator go through it in one pass (see _generator_backward)
        self._batch_gen_backward = bool(int(tr.get("batch_gen_backward", os.environ.get("HWG_BATCH_GEN_BWD", "1")) or 0))
        # recogniser-on-real-lines + style extractor on a tape as well: each loss group's pass through them runs on a stream of its own
        self._tape_style = bool(int(tr.get("tape_style", os.environ.get("HWG_TAPE_STYLE", "1")) or 0))
        self._concurrent_style_passes = bool(int(tr.get("concurrent_style_passes", os.environ.get("HWG_CONCURRENT_STYLE", "1")) or 0))
        self._trim_every = int(tr.get("alloc_trim_every", os.environ.get("HWG_ALLOC_TRIM_EVERY", "1000")) or 0)
        self.allocator_trims = 0
        self.hbm_peak_bytes = 0
        self._style_streams = None
        # Dead-gradient elimination (off by default = the reference's launches). The reference computes two families of parameter gradients
        # that nothing ever reads: the frozen recogniser's (it is in no optimizer; 11 backward traversals per cycle) and the
        # discriminator's in gen / auto lessons (optimizer_discriminator.zero_grad() drops them before the next disc lesson reads any).
        # With the switch on those weight-gradient kernels are not launched: losses, every optimizer update and all weights stay
        # bit-identical (tests/test_trainer_gpu.py), only `.grad` of those never-stepped tensors stays None.
        self.skip_unused_grads = bool(tr.get("skip_unused_grads", int(os.environ.get("HWG_SKIP_UNUSED_GRADS", "0") or 0)))
        self._grad_switch = None

    # ------------------------------------------------------------------------------------------
    def _to_tensor(self, instance):
        image, label = instance["image"], instance["label"]
        if image is not None:
            image = ops.h2d(image, self.gpu)
        if label is not None:
            label = ops.h2d(label, self.gpu)
        return image, label

    def _next_instance(self, lesson):
        if self.curriculum and all(l[:3] == "gen" or l == "no-step" for l in lesson) and self.text_data is not None:
            return self.text_data.getInstance()
        try:
            return next(self.data_loader_iter)
        except StopIteration:
            self.data_loader_iter = iter(self.data_loader)
            return next(self.data_loader_iter)

    def _skip_together(self, local_skip):
        if self._ctl_group is None:
            return bool(local_skip)
        flag = torch.tensor([1 if local_skip else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._ctl_group)
        return bool(flag.item())

    def _allreduce_grads(self, stashes=()):
        allreduce_gradient_sets(self.flat, stashes, self.world, self.gpu)

    def _stash(self, key=None):
        """clone-and-zero the current gradients (trainer :305-338); data parallel: their all-reduce starts right away (`key` = (lesson,
        position of the set in the lesson): the ranges to reduce are agreed between the ranks once per key, see start_stash_allreduce)"""
        return start_stash_allreduce(self.flat.stash(), self.world, self.flat, key=key)

    def _trim_allocator(self):
        """Generated lines have a width drawn per batch (4 T', T' from `insert_spaces`), so the activations of every step ask torch's caching
        allocator for block sizes it has not seen: over thousands of steps the cache splinters (15.8 GB reserved for 4.1 GB allocated after
        5600 steps at the bench batch) and the same logical buffers land on ever different addresses - `tools/mem_watch.py`: 69 steps/s at
        step 5600 against 75-76 with the cache released every 700 steps (the rest of the decline from 80 is the workload itself: the spacer
        learns longer lines, T' 110 -> 150). Every `trainer.alloc_trim_every` iterations (default 1000, 0 = never): if the cache holds more
        than twice the peak in use since the last trim, give it back (a device synchronisation and a few re-allocations, once per trim)."""
        self.hbm_peak_bytes = max(self.hbm_peak_bytes, torch.cuda.max_memory_allocated(self.gpu))     # (the allocator's own peak restarts below)
        if torch.cuda.memory_reserved(self.gpu) > 2 * torch.cuda.max_memory_allocated(self.gpu):
            torch.cuda.synchronize(self.gpu)
            torch.cuda.empty_cache()
            self.allocator_trims += 1
        torch.cuda.reset_peak_memory_stats(self.gpu)

    def _train_iteration(self, iteration):
        if self._trim_every and iteration and iteration % self._trim_every == 0 and self.gpu is not None:
            self._trim_allocator()
        if not self.model.training:   # nn.Module.train() walks all ~3000 sub-modules; only do it when the mode actually changes
            self.model.train()
        lesson = self.curriculum.getLesson(iteration) if self.curriculum else None
        if self._side_wgrad == "auto":
            ops.SIDE_WGRAD = bool(lesson) and "auto" in lesson
        instance = self._next_instance(lesson or [])
        produced = self._forward_backward(instance, lesson)
        if produced is None:
            return {}
        return self._apply_step(lesson, iteration, instance, *produced)

    def _set_unused_grads(self, lesson):
        """skip_unused_grads: the recogniser's parameters never require a gradient (when it is frozen), the discriminator's only in the
        lessons that step it; restored to the reference's behaviour when the switch is off"""
        if not self.curriculum:
            return
        want_d = (not self.skip_unused_grads) or any("disc" in l for l in lesson)
        want_h = not (self.skip_unused_grads and self.hwr_frozen)
        if self._grad_switch == (want_d, want_h):
            return
        self._grad_switch = (want_d, want_h)
        for name, p in self.model.named_parameters():
            if name.startswith("discriminator.") and not name.endswith(("weight_u", "weight_v")):
                p.requires_grad_(want_d)
            elif name.startswith("hwr."):
                p.requires_grad_(want_h)

    def _forward_backward(self, instance, lesson):
        """Gradient production of one iteration (trainer :236-338): zero the gradients of the optimizer(s) in play, run the lesson's forward
        graph, weight the losses and run the up-to-three backward passes of the balancing scheme; every separately balanced gradient set is
        stashed (clone-and-zero) as it is produced, and a "no-step" lesson stashes its main set as well. Everything a rank contributes to a
        data-parallel step is in `self.flat` and `self.saved_grads` afterwards. Returns (scaled losses, pred) or None when the batch is skipped."""
        self.optimizer.zero_grad()
        if self.curriculum and any("disc" in l for l in lesson):
            self.optimizer_discriminator.zero_grad()
        self._set_unused_grads(lesson)

        if self.curriculum:
            # the reference skips a batch without any text; data parallel: if one rank has to skip, all do (a rank that returned
            # alone would leave its peers waiting in the gradient all-reduce)
            if self._skip_together(all(l == 0 for l in instance["label_lengths"])):
                return None
            gen = getattr(self.model, "generator", None)
            taped = self._batch_gen_backward and self.balance_loss and gen is not None and hasattr(gen, "tape_mode")
            if taped:
                gen.tape_mode = True
                self.model.style_tape_mode = self._tape_style
            try:
                losses = self.run_gen(instance, lesson)
            finally:
                if taped:
                    gen.tape_mode = False
                    self.model.style_tape_mode = False
            pred = None
        else:
            pred, losses = self.run_hwr(instance)
        if losses is None:
            return None

        # weight the losses and sum them per balancing group (trainer :280-298): one launch per group (ops.weighted_sum: products and partial
        # sums rounded like the reference's `loss += losses[name] * lossWeights[name]` chain) instead of a scale and an add launch per loss,
        # forward and backward; `scaled` (what is logged) holds views of the groups' scaled-term vectors
        groups = {"loss": [], "recog": [], "autogen": []}
        for name in losses:
            if self.balance_loss and "generator" in name and "auto-gen" in lesson:
                groups["autogen"].append(name)
            elif self.balance_loss and "Recog" in name:
                groups["recog"].append(name)
            else:
                groups["loss"].append(name)
        scaled, sums = {}, {}
        for gname, names in groups.items():
            if not names:
                sums[gname] = 0
                continue
            if len(names) == 1 and float(self.lossWeights[names[0][:-4]]) == 1.0:      # nothing to multiply, nothing to add: no launch
                sums[gname] = scaled[names[0]] = losses[names[0]]
                continue
            total, vec = ops.weighted_sum([losses[n] for n in names], [self.lossWeights[n[:-4]] for n in names])
            sums[gname] = total
            for i, n in enumerate(names):
                scaled[n] = vec[i]
        scaled = {n: scaled[n] for n in losses}          # (the reference's log order)
        loss, recogLoss, autoGenLoss = sums["loss"], sums["recog"], sums["autogen"]

        lkey = tuple(lesson) if lesson else ()
        # the sums of the weight-gradient partial images of a backward pass are queued and made by one table-driven launch at the
        # join_side_stream() behind it (ops.DEFER_REDUCE; bit-identical, ~65 small launches per step less)
        ops.DEFER_REDUCE = self._defer_reduce
        gtapes = self._gen_tapes()
        stapes = self.model.take_style_tapes() if hasattr(self.model, "take_style_tapes") else []
        taped = bool(gtapes or stapes)
        sets = []           # per separately balanced loss group: (stash or None = the current set, key, gradients left on the taped leaves)
        try:
            if self.balance_loss:
                for pos, part in enumerate((autoGenLoss, recogLoss)):
                    if not isinstance(part, int):
                        part.backward(self._one(part), retain_graph=True)
                        ops.join_side_stream()
                        if taped:
                            st = self.flat.stash()                        # its all-reduce starts once the taped networks' share has been added
                            sets.append((st, (lkey, pos), self._take_leaf_grads(gtapes, stapes)))
                        else:
                            self.saved_grads.append(self._stash((lkey, pos)))
            else:
                for part in (recogLoss, autoGenLoss):
                    if not isinstance(part, int):
                        loss = part if isinstance(loss, int) else ops.add(loss, part)
            if not isinstance(loss, int):
                loss.backward(self._one(loss))
                ops.join_side_stream()
            if taped:
                sets.append((None, None, self._take_leaf_grads(gtapes, stapes)))
                self._taped_backward(gtapes, stapes, sets)
        finally:
            ops.DEFER_REDUCE = False
            ops.join_side_stream()      # (an exception inside backward must not leave queued sums behind)
        if self.balance_loss and "no-step" in lesson:
            self.saved_grads.append(self._stash((lkey, 2)))
        return scaled, pred

    # -- taped networks: generator (all loss groups in one pass) and recogniser + style extractor (one pass per group, concurrently) ------
    def _gen_tapes(self):
        """the taped generator forwards of this iteration (model.generator.tape_mode, switched on in run_gen for balanced training lessons)"""
        gen = getattr(self.model, "generator", None)
        return gen.take_tapes() if gen is not None and hasattr(gen, "take_tapes") else []

    @staticmethod
    def _take_leaf_grads(gtapes, stapes):
        out = ([], [])
        for t in gtapes:
            out[0].append(t.image.grad)
            t.image.grad = None
        for t in stapes:
            out[1].append(t.style.grad)
            t.style.grad = None
        return out

    def _taped_backward(self, gtapes, stapes, sets):
        """The reference walks the generator - and, behind it, the style extractor and the recogniser that fed it - once per balanced loss group
        (trainer :300-338: up to three backward() calls on the same graph). Here every group's backward pass stopped at the taped leaves (the
        generated image, pure_gen.GenTape; the extracted style, hw_with_style.StyleTape). Now (1) the gradients the groups left on an image
        go through the generator TOGETHER, stacked along the batch axis - its layers fill a quarter of the chip at 8 lines - each group's
        parameter gradients accumulating into that group's own buffer (ops.grad_set: the stash of the group, or the current gradient set for
        the last one); (2) each group's style gradient goes through the style extractor and the recogniser, one pass per group, the passes
        on streams of their own: they read the same activations, write different buffers, and a single pass leaves much of the chip idle
        (130-workgroup convolutions, latency-bound normalisation passes)."""
        targets = [None if st is None else (st[0], st[1]) for st, _, _ in sets]
        style_g = {id(t.style): [g[1][j] for _, _, g in sets] for j, t in enumerate(stapes)}
        for ti, tape in enumerate(gtapes):
            members = [si for si, (_, _, g) in enumerate(sets) if g[0][ti] is not None]
            if not members:
                continue
            dstyles = tape.backward_sets([sets[si][2][0][ti] for si in members], [targets[si] for si in members])
            if tape.style_src is None or dstyles is None:
                continue
            mine = style_g.get(id(tape.style_src))
            for k, si in enumerate(members):
                if mine is not None:            # the style came from a taped extraction: collected per group, walked below
                    mine[si] = dstyles[k] if mine[si] is None else ops.add(mine[si], dstyles[k])
                else:                           # an autograd style (tape mode off for the extractor): its graph, under the group's redirect
                    v0 = self.flat.flat_grad._version
                    with ops.grad_set(targets[si]):
                        tape.style_src.backward(dstyles[k], retain_graph=k + 1 < len(members))
                        ops.join_side_stream()
                    if targets[si] is not None and self.flat.flat_grad._version != v0:
                        # every op behind the style accumulates its parameter gradients through ops._grad_buffer (which follows the redirect);
                        # a gradient RETURNED to autograd would have been added to param.grad, i.e. to the wrong set
                        raise RuntimeError("autograd accumulated a parameter gradient into the current set during a redirected backward pass")
        ops.join_side_stream()
        for tape in stapes:
            members = [(si, g) for si, g in enumerate(style_g[id(tape.style)]) if g is not None]
            if len(members) > 1 and self._concurrent_style_passes:
                self._style_passes_on_streams(tape, members, targets)
            else:
                for si, g in members:
                    tape.backward(g, targets[si])
                    ops.join_side_stream()
        for st, key, _ in sets:
            if st is not None:
                self.saved_grads.append(start_stash_allreduce(st, self.world, self.flat, key=key))

    def _style_passes_on_streams(self, tape, members, targets):
        main = torch.cuda.current_stream()
        if self._style_streams is None:
            self._style_streams = [torch.cuda.Stream(device=self.gpu) for _ in range(4)]
        side_wgrad, ops.SIDE_WGRAD = ops.SIDE_WGRAD, False      # the passes are each other's filler; one shared side stream would chain them
        ops.DEFER_KEEP_ARENA = True                              # a pass's flush must not hand the arena's start to the next pass
        used = []
        try:
            for k, (si, g) in enumerate(members):
                s = self._style_streams[k % len(self._style_streams)]
                s.wait_stream(main)                              # the style gradients (and every activation) are complete on the main stream
                with torch.cuda.stream(s):
                    tape.backward(g, targets[si])
                    ops.join_side_stream()                       # this pass's deferred sums, on its own stream
                used.append(s)
        finally:
            for s in used:
                main.wait_stream(s)
            ops.DEFER_KEEP_ARENA = False
            ops.reset_defer_arena()
            ops.SIDE_WGRAD = side_wgrad

    def _apply_step(self, lesson, iteration, instance, scaled, pred):
        """Gradient consumption (trainer :340-391): data-parallel averaging of every gradient set, balancing of the stashed sets into the
        current one, clip to +-2, NaN scan, optimizer step; then the log."""
        f = self.flat
        if self.balance_loss and "no-step" in lesson:
            pass
        elif self.balance_loss and len(self.saved_grads) > 0:
            self._allreduce_grads(self.saved_grads)
            multipliers = None
            for it, mult in self.balance_var_x.items():
                if int(it) <= iteration:
                    multipliers = mult if isinstance(mult, list) else [mult]
            f.balance(self.saved_grads, multipliers)
            for s in self.saved_grads:
                f.release(s)
            self.saved_grads = []
        elif self._dp:
            self._allreduce_grads()

        flag = None
        if self.curriculum and "no-step" not in lesson:
            if self.pre_clip_hook is not None:    # parity tests read the balanced gradients here, where the reference clips them (:381)
                self.pre_clip_hook(iteration)
            # clip to +-2, NaN check and the Adam update in ONE launch (HipAdam.step(clip): every touched gradient is clipped, the stepping
            # optimizer's tensors are updated, the sticky flag is raised where a written parameter is not finite); the parameters as
            # loaded are scanned once, at this trainer's first stepping lesson
            if self._fused_step:
                flag = f._flag if getattr(f, "_scanned", False) else f.params_nonfinite_flag()
                opt = self.optimizer_discriminator if ("disc" in lesson or "auto-disc" in lesson) else self.optimizer
                opt.step(clip=2)
            else:
                f.clip_(2)
                flag = f.params_nonfinite_flag()
                if "disc" in lesson or "auto-disc" in lesson:
                    self.optimizer_discriminator.step()
                else:
                    self.optimizer.step()
        elif not self.curriculum:
            if self.pre_clip_hook is not None:
                self.pre_clip_hook(iteration)
            self.optimizer.step()

        # one device->host transfer for everything that is logged
        names = list(scaled)
        vals = [scaled[n].detach().reshape(1) for n in names]
        if flag is not None:
            vals.append(flag.float())
        cer = wer = 0
        if pred is not None:
            cer, wer, _ = self.getCER(instance["gt"], pred.detach().cpu().numpy())
        pending = (names, ops.AsyncFetch(torch.cat(vals)) if vals else None, flag is not None, cer, wer)
        if self.async_log:
            # pipelined logging: return the PREVIOUS iteration's values so that this iteration's kernels need not be drained
            # before the next iteration is enqueued (the reference's `.item()` per loss does exactly that drain)
            self._pending_log.append(pending)
            return self._resolve_log(self._pending_log.popleft()) if len(self._pending_log) > int(self.async_log) else {}
        return self._resolve_log(pending)

    def _resolve_log(self, pending):
        names, fetch, has_flag, cer, wer = pending
        host = fetch.get().tolist() if fetch is not None else []
        if has_flag:
            assert host[-1] == 0.0, "a parameter became NaN/inf"
        log_losses = dict(zip(names, host))
        total = sum(log_losses.values())
        assert not (np.isnan(total) or np.isinf(total)), log_losses
        return {"loss": total, **log_losses, "CER": cer, "WER": wer}

    def flush_log(self):
        log = {}
        while self._pending_log:          # every outstanding iteration is checked (NaN / inf assertions); the newest one's losses are returned
            log = self._resolve_log(self._pending_log.popleft())
        return log

    # ------------------------------------------------------------------------------------------
    def run_hwr(self, instance):
        image, label = self._to_tensor(instance)
        pred = self.model.hwr(image, None)
        T, B = pred.shape[0], pred.shape[1]
        recog = self.loss["recog"](pred, label.permute(1, 0), [T] * B, instance["label_lengths"])
        return pred, {"recogLoss": recog}

    def run_gen(self, instance, lesson, get=[]):
        model = self.model
        image, label = self._to_tensor(instance)
        batch_size = label.size(1)
        label_lengths = instance["label_lengths"]
        a_batch_size = self.a_batch_size if "a_batch_size" in instance else None
        losses = {}
        evaluating = "eval" in lesson or "valid" in lesson

        if any(x in lesson for x in ("count", "auto", "disc")) and instance.get("spaced_label") is not None:
            model.spaced_label_index = instance["spaced_label"].to(label.device)
            model.spaced_label = model.onehot(model.spaced_label_index)

        style = recon = None
        if "auto" in lesson:
            if "eval" not in lesson or "recon" in get:
                recon, style = model.autoencode(image, label, a_batch_size)
            if self.interpolate_gen_styles and not evaluating:
                step = a_batch_size or 1
                det = style.detach()
                for i in range(0, batch_size, step):
                    self.prev_styles.append(det[i])
                self.prev_styles = self.prev_styles[-self.prev_styles_size:]

        style_gen = gen_image = None
        if "gen" in lesson or "disc" in lesson or "gen" in get:
            if not evaluating or not self.interpolate_gen_styles:
                style_gen = self.get_style_gen(batch_size, label.device)
            else:
                n_a = batch_size // a_batch_size
                idx = torch.arange(batch_size, device=style.device).view(n_a, a_batch_size)
                nxt = idx.roll(-1, 0).reshape(-1)
                style_gen = 0.5 * style[nxt] + 0.5 * style
            if "eval" not in lesson and label.size(0) > self.text_data.max_len:
                if "auto" not in lesson:
                    label = label[:self.text_data.max_len].contiguous()
                for b in range(batch_size):
                    label_lengths[b] = min(label_lengths[b], self.text_data.max_len)
            gen_image = model(label, label_lengths, style_gen)

        if "auto" in lesson and "auto" in self.loss and "eval" not in lesson:
            if recon.size(3) > image.size(3):
                image = _pad_w(image, 0, recon.size(3) - image.size(3), "constant", PADDING_CONSTANT)
            elif recon.size(3) < image.size(3):
                recon = _pad_w(recon, 0, image.size(3) - recon.size(3), "constant", PADDING_CONSTANT)
            if self.no_bg_loss:
                raise NotImplementedError("fg-mask weighting is unreachable with the shipped configs (key read from the wrong level)")
            losses["autoLoss"] = self.loss["auto"](recon, image, **self.loss_params["auto"])

        if "count" in lesson and "count" in self.loss and "eval" not in lesson:
            if "auto" not in lesson:
                style = model.extract_style(image, label, a_batch_size)
                if "$UNKOWN$" in instance["gt"]:
                    raise NotImplementedError("pseudo-labelled lines ('$UNKOWN$') are a data-pipeline feature outside the accelerated path")
                spaced_idx = model.take_alignment(label)
            else:
                spaced_idx = model.spaced_label_index
            label_onehot = model.onehot(label)
            style_d = style.detach() if self.style_detach else style
            counts = model.spacer(label_onehot, style_d)
            if not model.count_duplicates:
                raise NotImplementedError("only the 'duplicates' spacer of the shipped GAN configs is on the accelerated path")
            gt_counts, meta = ops.gt_counts(spaced_idx, label)
            # counts[pos:] = 0 for the smallest final pos over the batch (the reference zeroes inside its per-line loop);
            # `meta` = [min pos, label/alignment mismatches] - one small D2H read replaces the reference's per-step .item()
            minpos, mismatch = meta.cpu().tolist()
            assert mismatch == 0, "aligned labels disagree with the text"
            if minpos < counts.size(0):
                counts = ops.zero_rows_from(counts, minpos)
            model.counts = counts
            losses["countLoss"] = self.loss["count"](counts, gt_counts, **self.loss_params["count"])

        if "auto" in lesson and "perceptual" in self.loss and "eval" not in lesson:
            if image.size(3) > recon.size(3):
                d = image.size(3) - recon.size(3)
                recon = _pad_w(recon, d // 2, d // 2 + d % 2)
            elif image.size(3) < recon.size(3):
                d = recon.size(3) - image.size(3)
                image = _pad_w(image, d // 2, d // 2 + d % 2)
            both = torch.cat((image, recon), dim=0)
            if both.size(3) < 40:
                d = 40 - both.size(3)
                both = _pad_w(both, d // 2, d // 2 + d % 2)
            feats = self.encoder(both)
            perceptual = 0
            for fmap in feats:
                orig_f, recon_f = fmap[:image.size(0)], fmap[image.size(0):]
                term = self.loss["perceptual"](recon_f, orig_f, **self.loss_params["perceptual"])
                perceptual = term if isinstance(perceptual, int) else ops.add(perceptual, term)
            losses["perceptualLoss"] = perceptual

        if "auto" in lesson and "reconRecog" in self.loss and "eval" not in lesson:
            recon_pred = model.hwr(recon)
            losses["reconRecogLoss"] = self.loss["reconRecog"](recon_pred, label.permute(1, 0), [recon_pred.size(0)] * batch_size, label_lengths)

        if "gen" in lesson and "genRecog" in self.loss and "eval" not in lesson:
            gen_pred = model.hwr(gen_image)
            # the CTC kernel reports a non-finite loss as 0 with zero gradient, which is what skipping the term amounts to
            losses["genRecogLoss"] = self.loss["genRecog"](gen_pred, label.permute(1, 0), [gen_pred.size(0)] * batch_size, label_lengths)

        fake = None
        if "gen" in lesson or "disc" in lesson:
            if ("auto" in lesson or "auto-disc" in lesson) and "eval" not in lesson:
                if recon.size(3) > gen_image.size(3):
                    gen_image = _pad_w(gen_image, 0, recon.size(3) - gen_image.size(3), "replicate")
                elif recon.size(3) < gen_image.size(3):
                    recon = _pad_w(recon, 0, gen_image.size(3) - recon.size(3), "replicate")
                fake = torch.cat((recon, gen_image), dim=0)
            else:
                fake = gen_image
        elif "auto-gen" in lesson:
            fake = recon

        if "disc" in lesson:
            if fake.size(3) > image.size(3):
                image = _pad_w(image, 0, fake.size(3) - image.size(3), "replicate")
            elif fake.size(3) < image.size(3):
                fake = _pad_w(fake, 0, image.size(3) - fake.size(3), "replicate")
            preds = model.discriminator(torch.cat((image, fake.detach()), dim=0))
            n_real = image.size(0)
            disc_loss = 0
            for p in preds:   # hinge loss per resolution head, averaged over heads
                real, fk = p[:n_real], p[n_real:]
                term = ops.add(ops.mean_loss(real, ops.LOSS_HINGE_REAL), ops.mean_loss(fk, ops.LOSS_HINGE_FAKE))
                disc_loss = term if isinstance(disc_loss, int) else ops.add(disc_loss, term)
            losses["discriminatorLoss"] = ops.scale(disc_loss, 1.0 / len(preds))

        predicted_disc = None
        if ("gen" in lesson or "auto-gen" in lesson) and "eval" not in lesson:
            gen_pred = model.discriminator(fake)
            gen_loss = 0
            predicted_disc = []
            for gp in gen_pred:
                term = ops.mean_loss(gp, ops.LOSS_MEAN, -1.0)
                gen_loss = term if isinstance(gen_loss, int) else ops.add(gen_loss, term)
                if "disc" in get:
                    predicted_disc.append(gp.detach().mean(dim=1).cpu())
            losses["generatorLoss"] = ops.scale(gen_loss, 1.0 / len(gen_pred))

        ret = losses
        if get:
            got = {}
            for name in get:
                if name == "recon":
                    got[name] = recon.detach().cpu()
                elif name in ("gen", "gen_image", "gen_img"):
                    got[name] = gen_image.detach().cpu()
                elif name == "pred":
                    if model.pred is None:
                        model.pred = model.hwr(image, None)
                    got[name] = model.pred.detach().cpu()
                elif name == "style":
                    got[name] = style.detach().cpu()
                elif name == "gt":
                    got[name] = instance["gt"]
                elif name == "author":
                    got[name] = instance["author"]
                elif name == "disc":
                    got[name] = predicted_disc
                else:
                    raise ValueError("Unknown get [{}]".format(name))
            ret = (losses, got)
        for attr in ("spaced_label", "spaced_label_index", "mask", "gen_mask", "top_and_bottom", "counts", "pred", "spacing_pred", "mask_pred",
                     "gen_spaced", "spaced_style", "mu", "sigma"):
            setattr(model, attr, None)
        return ret

    def _one(self, like):
        """the seed of a backward pass: autograd's implicit one is a fill launch per call; a cached tensor of the loss's shape is read-only there"""
        key = (like.shape, like.dtype, like.device)
        hit = self._ones.get(key) if hasattr(self, "_ones") else None
        if hit is None:
            if not hasattr(self, "_ones"):
                self._ones = {}
            hit = self._ones[key] = torch.ones(like.shape, dtype=like.dtype, device=like.device)
        return hit

    def get_style_gen(self, batch_size, device):
        """mix two stored styles per line with a weight in [low, high] (trainer :974-988); the style bank stays on the GPU"""
        if (self.interpolate_gen_styles and len(self.prev_styles) > 0) and (not self.sometimes_interpolate or self.interpolate_freq > random.random()):
            indexes = np.random.randint(0, len(self.prev_styles), (batch_size, 2))
            mix = np.random.uniform(self.interpolate_gen_styles_low, self.interpolate_gen_styles_high, batch_size)
            bank = torch.stack(self.prev_styles, dim=0)
            if bank.is_cuda and bank.dtype == torch.float32:
                # one upload (the two index rows as int32 next to the two weight rows as float32) and one launch (hwg_style_mix) instead of two
                # uploads, two index gathers, two products and a sum. The reference multiplies float32 tensors by numpy float64 scalars: the
                # scalar is rounded to float32, the product is float32 - the kernel rounds both products and the sum the same way
                host = np.empty((4, batch_size), dtype=np.float32)
                host[:2].view(np.int32)[:] = indexes.T
                host[2] = mix.astype(np.float32); host[3] = (1 - mix).astype(np.float32)
                dev_ = ops.h2d(host, device)
                out = torch.empty((batch_size, bank.shape[1]), dtype=torch.float32, device=device)
                ops.L.call("hwg_style_mix", bank, dev_[:2].view(torch.int32), dev_[2:], out, bank.shape[0], batch_size, bank.shape[1], ops._stream())
                return out
            ij = ops.h2d(indexes.T.copy(), device)
            a = bank[ij[0]]
            b = bank[ij[1]]
            mm = ops.h2d(np.stack([mix.astype(np.float32), (1 - mix).astype(np.float32)]), device)
            return (a * mm[0][:, None] + b * mm[1][:, None]).contiguous()
        return ops.h2d(torch.randn(batch_size, self.model.style_dim), device)

    def getCER(self, gt, pred, individual=False):
        cer = wer = 0
        pred_strs, all_cer = [], []
        for i, gt_line in enumerate(gt):
            pred_str, _ = string_utils.naive_decode(pred[:, i])
            pred_str = string_utils.label2str_single(pred_str, self.idx_to_char, False)
            this = string_utils.cer(gt_line, pred_str, self.casesensitive)
            cer += this
            all_cer.append(this)
            pred_strs.append(pred_str)
            wer += string_utils.wer(gt_line, pred_str, self.casesensitive)
        cer /= len(gt)
        wer /= len(gt)
        if individual:
            return cer, wer, pred_strs, all_cer
        return cer, wer, pred_strs

    def _valid_epoch(self):
        """validation pass (trainer :437-486): the curriculum's validation lesson (or the recogniser step) on every batch of the validation
        loader under no_grad; weighted losses, CER and WER averaged over the batches"""
        self.model.eval()
        totals = defaultdict(float)
        total_loss = total_cer = total_wer = 0.0
        n = 0
        with torch.no_grad():
            for instance in self.valid_data_loader:
                if self.curriculum:
                    losses, pred = self.run_gen(instance, self.curriculum.getValid()), None
                else:
                    pred, losses = self.run_hwr(instance)
                for name, v in losses.items():
                    w = float(v) * self.lossWeights[name[:-4]]
                    total_loss += w
                    totals["val_" + name] += w
                if pred is not None:
                    cer, wer, _ = self.getCER(instance["gt"], pred.detach().cpu().numpy())
                    total_cer += cer
                    total_wer += wer
                n += 1
        n = max(n, 1)
        return {"val_loss": total_loss / n, "val_CER": total_cer / n, "val_WER": total_wer / n, **{k: v / n for k, v in totals.items()}}
