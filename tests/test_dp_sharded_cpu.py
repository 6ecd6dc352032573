"""The sharded data-parallel step (reduce-scatter -> optimizer on the owned slices -> all-gather) on CPU, world
size 2 over gloo: the ORCHESTRATION of actmi/engine.py (bucket plan, owned slices, tails, flag exchange, all-gather) driven on
a stand-in engine whose arenas are CPU tensors and whose "AdamW" is a plain step -- the product kernels need the GPU and have
no CPU fallback; tests/test_gpu_dp.py runs the same code on the real engine."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from actmi.engine import ACTEngine

N_ARENA, N_PHASE1 = 64 * 37 + 0, 64 * 11          # 37 slots of 64 floats; the transformer range ends inside the arena


class _Lib:
    """the three C entry points the sharded step calls, on CPU tensors"""

    def __init__(self, eng):
        self.e = eng

    def actmi_adamw_step_range(self, h, lr, lr_bb, wd, b1, b2, eps, step, off, cnt, sp):
        e = self.e
        assert off % 64 == 0 and off + cnt <= N_ARENA
        if int(e.flags[0]) == 0:                                   # the device-side gate of the real kernel
            e.params[off:off + cnt] -= lr * e.grads[off:off + cnt]
        e.updated.append((off, cnt))
        return 0

    def actmi_refresh_weights(self, h, sp):
        self.e.refreshed += 1
        return 0


class StandIn:
    _shard_plan = ACTEngine._shard_plan
    backward_reduce_scatter = ACTEngine.backward_reduce_scatter
    adamw_step_sharded = ACTEngine.adamw_step_sharded
    sync_flags = ACTEngine.sync_flags

    def __init__(self, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.local_grads = torch.randn(N_ARENA, generator=g)
        self.grads = torch.zeros(N_ARENA)
        self.params = torch.arange(N_ARENA, dtype=torch.float32) * 1e-3
        self.flags = torch.zeros(1, dtype=torch.int32)
        self.device = torch.device("cpu")
        self.h, self.updated, self.refreshed = None, [], 0
        self.lib = _Lib(self)

    def backward(self, loss_scale):
        self.grads.copy_(self.local_grads * loss_scale)

    def grad_arena(self):
        return self.grads

    def param_arena(self):
        return self.params

    def grad_phase_range(self, phase):
        return (0, N_PHASE1) if phase == 1 else (N_PHASE1, N_ARENA - N_PHASE1)

    def flags_tensor(self):
        return self.flags

    def _sp(self):
        return None

    def adamw_step(self, *a, **k):
        raise AssertionError("the sharded step must not fall back to the full update under a process group")


def _worker(rank, world, port, out, bf16):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e = StandIn(rank)
    if rank == 1:
        e.flags[0] = 0                                             # (second run below raises it on one rank only)
    e.backward_reduce_scatter(1.0 / world, bucket_mb=0, comm_dtype=torch.bfloat16 if bf16 else None)      # bucket_mb 0 -> smallest buckets
    world_, rank_, buckets, tails, _ = e._shard
    assert (world_, rank_) == (world, rank) and len(buckets) >= 3 and all((hi - lo) % (64 * world) == 0 for lo, hi, _ in buckets)
    e.adamw_step_sharded(0.5, 0.5, 0.0, step=1)
    assert e.refreshed == 1 and e._shard is None
    owned = sum(c for _, c in e.updated)
    tail = sum(hi - lo for lo, hi in tails)
    assert owned == (N_ARENA - tail) // world + tail               # 1 / world of the bucketed part + the replicated tail
    torch.save({"params": e.params.clone(), "tails": tails}, os.path.join(out, f"r{rank}.pt"))
    # a flag raised on ONE rank stops the update on EVERY rank (the replicas must not diverge)
    e2 = StandIn(rank)
    if rank == 1:
        e2.flags[0] = 4
    before = e2.params.clone()
    e2.backward_reduce_scatter(1.0 / world, bucket_mb=0)
    assert int(e2.flags[0]) == 4
    e2.adamw_step_sharded(0.5, 0.5, 0.0, step=1)
    assert torch.equal(e2.params, before)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bf16", [False, True])
def test_sharded_step_orchestration_world2_gloo(tmp_path, bf16):
    world = 2
    mp.spawn(_worker, args=(world, 29710 + os.getpid() % 50 + (60 if bf16 else 0), str(tmp_path), bf16), nprocs=world, join=True)
    a = torch.load(tmp_path / "r0.pt", weights_only=True)
    b = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert torch.equal(a["params"], b["params"])                   # every rank holds the same, complete parameter arena
    g = sum(torch.randn(N_ARENA, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    exp = torch.arange(N_ARENA, dtype=torch.float32) * 1e-3 - 0.5 * g
    tol = 2e-2 if bf16 else 1e-6                                   # bf16 buckets: the sum is formed in bf16
    assert float((a["params"] - exp).abs().max()) <= tol
    if bf16:
        assert float((a["params"] - exp).abs().max()) > 0          # (and it did go through bf16)
