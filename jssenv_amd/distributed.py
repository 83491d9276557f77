"""Sharding the env batch over the GPUs of one node.

Env instances are independent: rank r owns the contiguous slice
[r*B_local, (r+1)*B_local) of the global batch and steps it with no data-path
collective.  The only exchange is one all-reduce of four counters (env steps, finished
episodes, sum of makespans, sum of reward numerators) plus a MAX of the wall time per
measurement window -- over RCCL/xGMI on GPUs (backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple


def shard_bounds(global_batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `rank`; the first (global_batch % world_size) ranks get one more env."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(global_batch, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def select_device(local_rank: int, world_size: int, device_count: int, share_device: bool = False) -> int:
    """The GPU index rank `local_rank` of a one-node job drives: its own (one process per GPU, no oversubscription) -- an error
    when the node exposes fewer devices than ranks, unless ``share_device`` (debug: every rank on device 0)."""
    if not 0 <= local_rank < max(1, world_size):
        raise ValueError(f"local rank {local_rank} outside a job of {world_size} rank(s)")
    if share_device:
        return 0
    if device_count < world_size:
        raise RuntimeError(f"{world_size} ranks but only {device_count} GPU(s) visible: one process per GPU, no oversubscription")
    return local_rank


def init_from_env(backend: Optional[str] = None, force: bool = False):
    """Initialise torch.distributed from the torchrun environment (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_*).
    Returns (rank, world_size, local_rank).  No-op for a single process unless ``force`` (a world of one rank still
    goes through the backend then: how the RCCL lines get executed on a 1-GPU box)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        kwargs = {}
        if backend == "nccl":
            dev = select_device(local_rank, world, torch.cuda.device_count())
            torch.cuda.set_device(dev)
            kwargs["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def reduce_counters(counters, wall_seconds: float, force_collectives: bool = False) -> Dict[str, float]:
    """counters: tensor [4] (or [B,4]) of this rank's window; returns the whole-job totals and the
    max-over-ranks wall time.  One SUM and one MAX all-reduce (skipped in a world of one rank unless
    ``force_collectives``)."""
    import torch
    import torch.distributed as dist
    c = counters.sum(dim=0) if counters.dim() == 2 else counters
    c = c.to(torch.float64)
    t = torch.tensor([wall_seconds], dtype=torch.float64, device=c.device)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives):
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    steps, episodes, makespan_sum, reward_num = [float(x) for x in c.tolist()]
    return {"steps": steps, "episodes": episodes, "makespan_sum": makespan_sum, "reward_num_sum": reward_num,
            "seconds": float(t.item()), "steps_per_second": steps / float(t.item()) if t.item() > 0 else 0.0}
