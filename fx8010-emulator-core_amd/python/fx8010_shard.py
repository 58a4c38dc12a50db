"""Instance sharding for multi-GPU runs.

Instances share nothing mutable (only the read-only program and LUTs), so the batch shards by
contiguous instance ranges, one process per GPU, with NO collective on the data path
(SURVEY.md §8e).  Ranks only meet to agree on timing (barrier, MAX of elapsed time) and to add up
counters.  Everything here works with any torch.distributed backend: "nccl" (= RCCL) on GPUs,
"gloo" in the CPU tests.
"""
import os


def shard_range(n_total, world_size, rank):
    """Contiguous, balanced split of n_total instances: returns (first_instance, count)."""
    base, extra = divmod(int(n_total), int(world_size))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def weak_shard(n_per_rank, rank):
    """Weak scaling: every rank owns n_per_rank instances; returns (first_instance, count)."""
    return int(rank) * int(n_per_rank), int(n_per_rank)


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process when absent)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend, device=None):
    """Initialise torch.distributed when WORLD_SIZE > 1; returns the module or None."""
    rank, local, world = env_world()
    if world <= 1 and os.environ.get("FX_FORCE_DIST") != "1":   # (FX_FORCE_DIST=1: a one-rank group - the RCCL code path on a one-GPU box)
        return None
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local) if device is None else device)
        else:
            dist.init_process_group(backend)
    return dist


def reduce_scalar(dist, value, op, device="cpu"):
    """MAX or SUM of one float64 over all ranks (identity without a process group)."""
    if dist is None:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def gather_scalars(dist, value, device="cpu"):
    """[value of rank 0, value of rank 1, ...] on every rank (a one-element list without a process group)."""
    if dist is None:
        return [float(value)]
    import torch

    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]
