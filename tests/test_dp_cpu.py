"""Bucketed gradient all-reduce helper on CPU tensors with gloo, world_size 2 (the GPU path uses the same function on
the gradient arena with the nccl = RCCL backend)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from actmi.dist_utils import allreduce_buckets


def _worker(rank, world, port, out):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    nb = allreduce_buckets(t, 300)
    if rank == 0:
        torch.save({"t": t, "nb": nb}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, 29700 + os.getpid() % 200, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["nb"] == 4                                         # ceil(1000 / 300) buckets, ragged last bucket
    assert torch.equal(r["t"], torch.arange(1000, dtype=torch.float32) * 3)


def test_allreduce_without_process_group_is_noop():
    t = torch.ones(10)
    assert allreduce_buckets(t, 4) == 0 and torch.equal(t, torch.ones(10))
