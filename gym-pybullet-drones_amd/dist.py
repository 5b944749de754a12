"""Multi-GPU: shard aviaries across ranks (one process per GPU), optional obs all-gather over RCCL.

Aviaries are independent (the only cross-drone coupling, downwash and the MultiHover reductions,
is inside one aviary), so the physics needs NO collective: rank r owns a contiguous block of envs
and runs the same kernel on its own device.  The one optional exchange step is an all-gather of
the `(E_local*D, 12)` observation shards into the concatenated `(E*D, 12)` tensor a centralised
learner would consume — `torch.distributed.all_gather_into_tensor`, which is RCCL over xGMI with
the "nccl" backend on ROCm (gloo on CPU for tests).
"""
import os

import torch
import torch.distributed as dist


def env_shard(total_envs: int, rank: int, world_size: int):
    """Contiguous block [start, stop) of envs owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total_envs, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_from_env(backend: str = None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class ObsAllGather:
    """Pre-allocated all-gather of equally sized observation shards."""

    def __init__(self, shard_rows: int, cols: int = 12, device=None, dtype=torch.float32, group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.full = torch.empty((self.world * shard_rows, cols), dtype=dtype, device=device)

    def __call__(self, shard: torch.Tensor, async_op: bool = False):
        """Gather `shard` (rows, cols) from every rank into `self.full` (world*rows, cols)."""
        if self.world == 1:
            self.full.copy_(shard.reshape(self.full.shape))
            return self.full if not async_op else (self.full, None)
        work = dist.all_gather_into_tensor(self.full, shard.reshape(-1, self.full.shape[1]).contiguous(),
                                           group=self.group, async_op=async_op)
        return (self.full, work) if async_op else self.full


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a python float over all ranks (timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
