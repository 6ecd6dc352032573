"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" for CPU tests)."""
import os

import torch
import torch.distributed as dist


def init_from_env():
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* if a launcher set them. Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def barrier():
    """No-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_gather_rows(local: torch.Tensor, counts):
    """Gather variable-length [n_r, k] float tensors from every rank into one [sum n_r, k] CPU tensor
    (the single collective of the eval path: per-episode (return, highest_reward))."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.detach().cpu()
    world = dist.get_world_size()
    nmax = max(counts)
    k = local.shape[1]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    pad = torch.zeros((nmax, k), dtype=torch.float32, device=dev)
    pad[: local.shape[0]] = local.to(dev, torch.float32)
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:c].cpu() for o, c in zip(outs, counts)], dim=0)


def allreduce_buckets(flat: torch.Tensor, bucket_elems: int, group=None):
    """In-place SUM all-reduce of a flat tensor in fixed-size buckets issued back to back (async), then waited.
    470 MB of fp32 gradients as ~8 buckets of 64 MB keeps every xGMI link busy without one giant collective;
    with world_size 1 (or no process group) this is a no-op."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    works = []
    n = flat.numel()
    for lo in range(0, n, bucket_elems):
        works.append(dist.all_reduce(flat[lo:min(n, lo + bucket_elems)], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    return len(works)


class _Done:
    """Work handle of a collective that already completed."""

    def wait(self):
        return True


def reduce_scatter_flat(out: torch.Tensor, inp: torch.Tensor, group=None):
    """SUM reduce-scatter of a flat tensor (rank r receives the r-th of world equal slices of the sum), asynchronous:
    returns a work handle.  RCCL (backend nccl) runs it on device memory, in place when `out` is the r-th slice of `inp`.
    gloo has no device path for it: CUDA tensors are staged through the host (CPU tests and the two-ranks-on-one-GPU test)."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        h_in = inp.detach().cpu()
        h_out = torch.empty(out.numel(), dtype=h_in.dtype)
        dist.reduce_scatter_tensor(h_out, h_in, op=dist.ReduceOp.SUM, group=group)
        out.copy_(h_out)
        return _Done()
    return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group, async_op=True)


def all_gather_flat(out: torch.Tensor, inp: torch.Tensor, group=None):
    """all-gather of equal flat slices into `out` (in place when `inp` is the r-th slice of `out`); see reduce_scatter_flat."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        h_in = inp.detach().cpu()
        h_out = torch.empty(out.numel(), dtype=h_in.dtype)
        dist.all_gather_into_tensor(h_out, h_in, group=group)
        out.copy_(h_out)
        return _Done()
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=True)
