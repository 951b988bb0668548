"""Iteration loop, optimizer routing, LR schedule and checkpointing (reference: base/base_trainer.py:16-484).

Differences that are deliberate and MI355X-first:
  * optimizers are `HipAdam` instances over a `FlatParams` gradient buffer (one multi-tensor launch per step);
  * with torch.distributed initialised (one process per GPU, RCCL) the gradient buffer is all-reduced by the trainer.
Config keys, their (sometimes wrong-level) lookups and the checkpoint dictionary layout follow the reference.
"""
import json
import logging
import math
import os
import timeit

import torch
import torch.distributed as dist

from ..logger import install_reference_aliases, load_checkpoint
from ..utils.curriculum import Curriculum


def ensure_dir(path):
    os.makedirs(path, exist_ok=True)


def _atomic_write(path, write):
    """write(file) into a sibling temporary file, then rename over `path`: readers never see a torn file"""
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "wb") as f:
        write(f)
    os.replace(tmp, path)


def lr_schedule(kind, tr, iterations):
    """lr multiplier as a function of the scheduler step (reference: base/base_trainer.py:112-164)"""
    warm = tr.get("warmup_steps", 1000)
    if kind == "LR_test":
        slope = (1 - 0.000001) / iterations
        return lambda s: 0.000001 + slope * s
    if kind == "rampup":
        return lambda s: min(1, (s + 0.001) / warm)
    if kind == "detector":
        return lambda s: min((s + 1) ** -0.3, (s + 1) * warm ** -1.3)
    if kind is True:
        return lambda s: min((max(0.000001, s - (warm - 3)) / 100) ** -0.1, s * (1.485 / warm) + .01)
    if kind == "cyclic":
        mn, cyc = tr.get("min_lr_mul", 0.001), tr.get("cycle_size", 500)
        return lambda s: (1 - (1 - mn) * ((s - 1) % cyc) / (cyc - 1))
    if kind == "cyclic-full":
        mn, cyc = tr.get("min_lr_mul", 0.25), tr.get("cycle_size", 500)

        def true_cycle(s):
            if (s // cyc) % 2 == 0:   # rising
                return ((1 - mn) * (s % cyc) / (cyc - 1)) + mn
            return 1 - (1 - mn) * (s % cyc) / (cyc - 1)
        return true_cycle
    if kind == "1cycle":
        low, mn, cyc = tr.get("low_lr_mul", 0.25), tr.get("min_lr_mul", 0.0001), tr.get("cycle_size", 1000)
        trail = iterations - 2 * cyc

        def one_cycle(s):
            if s < cyc:
                return ((1 - low) * (s % cyc) / (cyc - 1)) + low
            if s < 2 * cyc:
                return 1 - (1 - low) * (s % cyc) / (cyc - 1)
            t = s - 2 * cyc
            return low * (trail - t) / trail + mn * t / trail
        return one_cycle
    raise NotImplementedError("learning schedule %r" % (kind,))


class BaseTrainer:
    def __init__(self, model, loss, metrics, resume, config, train_logger=None):
        self.config = config
        self.model = model
        self.logger = logging.getLogger(self.__class__.__name__)
        self.loss = loss
        self.metrics = metrics
        self.name = config["name"]
        tr = config["trainer"]
        self.logged = config.get("super_computer", False)
        self.iterations = tr["iterations"]
        self.val_step = tr["val_step"]
        self.save_step = tr["save_step"]
        self.save_step_minor = tr.get("save_step_minor")
        self.log_step = tr["log_step"]
        self.verbosity = tr["verbosity"]
        if not config["cuda"] or not torch.cuda.is_available():
            raise RuntimeError("this trainer runs on MI355X only: config['cuda'] must be true and a HIP device visible (no CPU path exists)")
        self.with_cuda = True
        self.gpu = torch.device("cuda:" + str(config["gpu"]))
        self.model = self.model.to(self.gpu)

        self.curriculum = Curriculum(tr["curriculum"]) if "curriculum" in tr else None
        self.hwr_frozen = tr["hwr_frozen"] if "hwr_frozen" in tr else config["model"].get("hwr_frozen", False)
        self.style_frozen = tr["style_frozen"] if "style_frozen" in tr else config["model"].get("style_frozen", False)
        self.train_logger = train_logger

        if config["optimizer_type"] != "none":
            if config["optimizer_type"] != "Adam" or config.get("optimizer_type_discriminator", "Adam") != "Adam":
                raise NotImplementedError("only Adam (what every shipped config uses) has a multi-tensor HIP kernel")
            # NB the reference reads these two from the TOP level of the config, where the shipped configs do not put them
            slow_names = tr["slow_param_names"] if "slow_param_names" in config else []
            freeze_names = tr["freeze_param_names"] if "freeze_param_names" in config else []
            only = tr.get("only_params")
            main, disc = [], []
            for name, p in self.model.named_parameters():
                if only is not None and not any(o in name for o in only):
                    continue
                if any(f in name for f in freeze_names):
                    continue
                if "discriminator" in name:
                    disc.append(p)
                elif any(s in name for s in slow_names) or "gen_deform" in name or "conv_offset_mask" in name:
                    raise NotImplementedError("slow parameter groups are never populated by the shipped configs")
                elif ("hwr" in name and self.hwr_frozen) or ("style_extractor" in name and self.style_frozen):
                    continue
                elif "style_extractor" in name and self.curriculum is not None and self.curriculum.need_style_in_disc:
                    disc.append(p)
                else:
                    main.append(p)
            from ..trainer.flat_params import FlatParams, HipAdam   # (imported here: trainer/ imports this module)
            groups = {"main": main}
            if disc:
                groups["disc"] = disc
            self.flat = FlatParams(list(self.model.parameters()), groups, names=[n for n, _ in self.model.named_parameters()])
            self.optimizer = HipAdam(self.flat, "main", **config["optimizer"])
            self.optimizer_discriminator = HipAdam(self.flat, "disc", **config["optimizer_discriminator"]) if disc else None
        else:
            self.flat = None
            self.optimizer = None
            self.optimizer_discriminator = None

        self.useLearningSchedule = tr.get("use_learning_schedule", False)
        self.lr_lambda = self._make_schedule(tr) if self.useLearningSchedule else None
        self._base_lr = config["optimizer"]["lr"] if self.optimizer is not None else None

        self.monitor = tr["monitor"]
        self.monitor_mode = tr["monitor_mode"]
        self.monitor_best = math.inf if self.monitor_mode == "min" else -math.inf
        if tr.get("retry_count", 1) > 1:
            raise NotImplementedError("retry is disabled")
        self.start_iteration = 1
        self.checkpoint_dir = os.path.join(tr["save_dir"], self.name)
        ensure_dir(self.checkpoint_dir)
        # data parallel: one writer. Every rank holds identical weights and optimizer state, so rank 0 alone writes config.json and
        # the checkpoints (N ranks truncating one path at once can leave a torn file)
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if self.rank == 0:
            _atomic_write(os.path.join(self.checkpoint_dir, "config.json"), lambda f: f.write(json.dumps(config, indent=4, sort_keys=False).encode()))
        self.swa = False
        if tr.get("swa") or tr.get("weight_averaging"):
            raise NotImplementedError("weight averaging is not used by any shipped config")
        self._stop = False
        if resume:
            self._resume_checkpoint(resume)

    def request_stop(self):
        """async-signal-safe: train() saves a checkpoint and returns at the next iteration boundary (on every rank together)"""
        self._stop = True

    def _stop_agreed(self):
        """SIGINT may reach one rank only (or the ranks at different iterations): the flag is OR-ed over the gloo control group so that all
        ranks save - and enter _save_checkpoint's barrier - at the same iteration. Single process: just the flag."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return self._stop
        from ..trainer.flat_params import control_group
        flag = torch.tensor([1 if self._stop else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=control_group())
        return bool(flag.item())

    def _make_schedule(self, tr):
        return lr_schedule(self.useLearningSchedule, tr, self.iterations)

    def scheduled_lr(self, iteration):
        """The reference builds a fresh LambdaLR on every start (also on resume) and calls .step() BEFORE each iteration, so the
        k-th iteration of a run trains with base_lr * lambda(k), k = 1, 2, ... (base/base_trainer.py:112-164, 216-217)."""
        return self._base_lr * self.lr_lambda(iteration - self.start_iteration + 1)

    # ------------------------------------------------------------------------------------------
    def train(self):
        sum_log = {}
        for self.iteration in range(self.start_iteration, self.iterations + 1):
            if self._stop_agreed():
                self.iteration -= 1          # the last completed iteration is what the checkpoint holds
                self.save()
                return
            t0 = timeit.default_timer()
            if self.lr_lambda is not None:
                self.optimizer.param_groups[0]["lr"] = self.scheduled_lr(self.iteration)
            result = self._train_iteration(self.iteration)
            result["sec_per_iter"] = timeit.default_timer() - t0
            for k, v in result.items():
                if isinstance(v, (int, float)):
                    sum_log[k] = sum_log.get(k, 0.0) + v
            if self.iteration % self.log_step == 0:
                log = {"iteration": self.iteration, **{("avg_" + k): v / self.log_step for k, v in sum_log.items()}}
                sum_log = {}
                self._minor_log(log)
                if self.train_logger is not None:
                    self.train_logger.add_entry(log)
            if self.val_step > 0 and self.iteration % self.val_step == 0 and getattr(self, "valid", False):
                val = self._valid_epoch()
                self.logger.info("validation: %s", val)
            if self.iteration % self.save_step == 0:
                self._save_checkpoint(self.iteration, {})
            elif self.save_step_minor and self.iteration % self.save_step_minor == 0:
                self._save_checkpoint(self.iteration, {}, minor=True)
        # pipelined logging (trainer.async_log = n): the losses of the last n iterations are still outstanding - resolve them (their
        # non-finite checks run there) before train() returns
        flush = getattr(self, "flush_log", None)
        if flush is not None:
            flush()

    def _minor_log(self, log):
        self.logger.info("Train " + ",\t".join("%s: %s" % kv for kv in log.items()))

    def save(self):
        self._save_checkpoint(getattr(self, "iteration", 0), {})

    def _save_checkpoint(self, iteration, log, save_best=False, minor=False):
        """checkpoint dictionary of the reference (base/base_trainer.py:340-399): a major save writes checkpoint-iteration<N>.pth AND
        refreshes checkpoint-latest.pth, a minor save only the latter. Written by rank 0 through a temporary file + rename; all
        ranks leave together."""
        path = None
        if self.rank == 0:
            install_reference_aliases()    # the pickled train logger must resolve as logger.logger.Logger, in the reference too
            state = {
                "arch": type(self.model).__name__,
                "iteration": iteration,
                "logger": self.train_logger,
                "optimizer": self.optimizer.state_dict() if self.optimizer is not None else None,
                "monitor_best": self.monitor_best,
                "config": self.config,
                "state_dict": {k: v.cpu() for k, v in self.model.state_dict().items()},
            }
            if self.optimizer_discriminator is not None:
                # the reference's _resume_checkpoint reads this key (:456) although its own save omits it (the discriminator's Adam
                # moments restart from zero there); writing it keeps the file loadable by both
                state["optimizer_discriminator"] = self.optimizer_discriminator.state_dict()
            from .. import rng
            state["rng"] = rng.get_state()
            latest = os.path.join(self.checkpoint_dir, "checkpoint-latest.pth")
            path = latest if minor else os.path.join(self.checkpoint_dir, "checkpoint-iteration{}.pth".format(iteration))
            _atomic_write(path, lambda f: torch.save(state, f))
            if not minor:
                _atomic_write(latest, lambda f: torch.save(state, f))
            if save_best:
                os.replace(path, os.path.join(self.checkpoint_dir, "model_best.pth"))
            self.logger.info("Saved checkpoint: %s", path)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        return path

    def _resume_checkpoint(self, resume_path):
        self.logger.info("Loading checkpoint: %s ...", resume_path)
        ckpt = load_checkpoint(resume_path)   # (checkpoints pickle the reference's logger.logger.Logger)
        self.start_iteration = ckpt["iteration"] + 1
        self.monitor_best = ckpt.get("monitor_best", self.monitor_best)
        sd = {k: v for k, v in ckpt["state_dict"].items() if not k.startswith("style_from_normal")}
        self.model.load_state_dict(sd)
        if self.optimizer is not None and ckpt.get("optimizer") is not None:
            self.optimizer.load_state_dict(ckpt["optimizer"])
        if self.optimizer_discriminator is not None and ckpt.get("optimizer_discriminator") is not None:
            self.optimizer_discriminator.load_state_dict(ckpt["optimizer_discriminator"])
        if ckpt.get("rng") is not None:
            from .. import rng
            rng.set_state(ckpt["rng"], rank=self.rank)
        if ckpt.get("logger") is not None:
            self.train_logger = ckpt["logger"]
