"""CPU: host logic of the trainer base (reference: base/base_trainer.py) - learning-rate schedules against torch's LambdaLR
driven the way the reference drives it (.step() before every iteration), checkpoint file handling, logger pickling."""
import os
import pickle
import sys

import pytest
import torch


@pytest.mark.parametrize("kind,tr", [("rampup", {"warmup_steps": 7}), ("detector", {"warmup_steps": 5}), (True, {"warmup_steps": 6}),
                                     ("cyclic", {"cycle_size": 5}), ("cyclic-full", {"cycle_size": 4}), ("LR_test", {}),
                                     ("1cycle", {"cycle_size": 4})])
def test_lr_schedule_matches_lambdalr_stepped_before_each_iteration(kind, tr):
    from handwriting_line_generation_amd.base.base_trainer import BaseTrainer, lr_schedule
    iterations, base_lr = 30, 2e-4
    lam = lr_schedule(kind, tr, iterations)
    for start in (1, 11):    # fresh run, and a resumed run: the reference builds a new LambdaLR at every start
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=base_lr)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lam)
        stub = type("T", (), {"_base_lr": base_lr, "lr_lambda": staticmethod(lam), "start_iteration": start})()
        for it in range(start, start + 12):
            opt.step()
            sched.step()     # reference: self.lr_schedule.step() right before _train_iteration (base_trainer.py:216-217)
            want = opt.param_groups[0]["lr"]
            got = BaseTrainer.scheduled_lr(stub, it)
            assert got == pytest.approx(want, rel=1e-12), (kind, it, got, want)


def test_atomic_write_replaces_whole_file(tmp_path):
    from handwriting_line_generation_amd.base.base_trainer import _atomic_write
    p = tmp_path / "checkpoint-latest.pth"
    p.write_bytes(b"old" * 1000)
    _atomic_write(str(p), lambda f: f.write(b"new"))
    assert p.read_bytes() == b"new"
    assert os.listdir(tmp_path) == ["checkpoint-latest.pth"]      # no temporary left behind


def test_logger_pickles_under_the_reference_module_name():
    from handwriting_line_generation_amd.logger import Logger, install_reference_aliases
    install_reference_aliases()
    lg = Logger()
    lg.add_entry({"iteration": 1, "loss": 0.5})
    blob = pickle.dumps(lg)
    assert b"logger.logger" in blob and b"handwriting_line_generation_amd" not in blob    # loadable by the reference's train.py
    back = pickle.loads(blob)
    assert back.entries == {1: {"iteration": 1, "loss": 0.5}}
    assert "iteration" in str(back)
