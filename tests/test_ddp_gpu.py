"""GPU: the data-parallel training step with two ranks (gloo rendezvous, both ranks on cuda:0 - RCCL needs one GPU per rank, the
collective semantics are the same): after a full curriculum cycle on different author shards the replicated weights of the two
ranks must still be identical, and must differ from a single-rank run (i.e. the exchange really happened)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out, workdir):
    import random
    import numpy as np
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    rng.set_mode("device", seed=7 + rank)
    torch.manual_seed(0)   # same initial weights on every rank
    wd = os.path.join(workdir, "r%d" % rank)
    os.makedirs(wd, exist_ok=True)
    trainer, _ = build_gan_trainer("iam_gan", 1, 2, width=256, label_len=12, workdir=wd, rank=rank, world=world)
    torch.manual_seed(100 + rank); np.random.seed(100 + rank); random.seed(100 + rank)
    for it in range(7):
        trainer._train_iteration(it)
    torch.cuda.synchronize()
    digest = torch.stack([p.detach().double().sum() for p in trainer.model.parameters()]).cpu()
    finite = bool(torch.isfinite(digest).all())
    out[(world, rank)] = (digest.tolist(), finite)
    if world > 1:
        dist.destroy_process_group()


def _run(world, tmp):
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, tmp)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return dict(out)


def test_two_ranks_keep_identical_weights(cuda, tmp_path):
    two = _run(2, str(tmp_path))
    d0, ok0 = two[(2, 0)]
    d1, ok1 = two[(2, 1)]
    assert ok0 and ok1, "non-finite parameters after the cycle"
    assert d0 == d1, "replicated weights diverged between the ranks: max |diff| %.3e" % max(abs(a - b) for a, b in zip(d0, d1))
    one = _run(1, str(tmp_path / "single"))
    s0, _ = one[(1, 0)]
    assert any(abs(a - b) > 1e-9 for a, b in zip(s0, d0)), "two-rank run equals the single-rank run: no gradient exchange happened"
