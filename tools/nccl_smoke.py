"""Single-rank RCCL smoke: the collective calls the data-parallel trainer makes (backend "nccl" = RCCL, device tensors, async handles,
MAX on int32 masks, reductions of narrow views of a flat buffer, a gloo control group next to it), on however many GPUs are visible.
Run as: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/nccl_smoke.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctl = dist.new_group(backend="gloo")
    from handwriting_line_generation_amd.trainer import flat_params
    flat = torch.arange(1 << 20, dtype=torch.float32, device=dev) * (rank + 1)
    ref = torch.arange(1 << 20, dtype=torch.float32) * sum(r + 1 for r in range(world)) / world
    # whole-buffer asynchronous SUM (a stashed gradient set), then the division every rank applies after the wait
    stash = flat.clone()
    work = dist.all_reduce(stash, op=dist.ReduceOp.SUM, async_op=True)
    mask = torch.tensor([rank % 2, 1, 0, rank], dtype=torch.int32, device=dev)
    dist.all_reduce(mask, op=dist.ReduceOp.MAX)
    work.wait()
    stash.div_(world)
    assert torch.allclose(stash.cpu(), ref), "async SUM all-reduce"
    assert mask.cpu().tolist() == [1 if world > 1 else 0, 1, 0, world - 1], mask
    # reductions of narrow views (touched spans) of the flat buffer
    for lo, hi in ((0, 4096), (100000, 100003), (1 << 19, 1 << 20)):
        view = flat[lo:hi]
        dist.all_reduce(view)
        view.div_(world)
    torch.cuda.synchronize()
    assert torch.allclose(flat[:4096].cpu(), ref[:4096]) and torch.allclose(flat[1 << 19:].cpu(), ref[1 << 19:])
    # control-plane decision over gloo while device collectives are in flight
    flag = torch.tensor([1 if rank == world - 1 else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctl)
    assert int(flag) == 1
    dist.barrier()
    if rank == 0:
        print("nccl smoke ok: world %d, backend %s, counters %s" % (world, dist.get_backend(), flat_params.COMM))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
