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


def select_device(local_rank: int, local_world_size: int, device_count: int, share_device: bool = False) -> int:
    """The GPU index rank `local_rank` of a node drives: its own (one process per GPU, no oversubscription) -- an error when the
    node exposes fewer devices than it runs ranks, unless ``share_device`` (debug: every rank on device 0).
    ``local_world_size`` is the number of ranks ON THIS NODE (torchrun's LOCAL_WORLD_SIZE): a 2 x 8-GPU job has a world of 16
    and 8 ranks -- and 8 devices -- per node."""
    if not 0 <= local_rank < max(1, local_world_size):
        raise ValueError(f"local rank {local_rank} outside a node of {local_world_size} rank(s)")
    if share_device:
        return 0
    if device_count < local_world_size:
        raise RuntimeError(f"{local_world_size} ranks but only {device_count} GPU(s) visible: one process per GPU, no oversubscription")
    return local_rank


def local_world_size(world_size: int) -> int:
    """Ranks on this node: torchrun's LOCAL_WORLD_SIZE; the whole world when the launcher does not say (one node)."""
    return int(os.environ.get("LOCAL_WORLD_SIZE", world_size))


def gpu_numa_node(device_index: int, sysfs: str = "/sys") -> Optional[int]:
    """NUMA node of GPU `device_index` from sysfs, or None when it cannot be told.  The KFD topology lists the GPUs in the order
    HIP enumerates them (CPU-only nodes carry no `gpu_id`); a node's `drm_render_minor` names its render device, whose PCI
    function's `numa_node` is the answer.  (`<sysfs>/class/drm/card*/device/numa_node` in card order is the fallback.)"""
    def read_int(path):
        try:
            with open(path) as fh:
                return int(fh.read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    try:
        base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
        gpus = []
        for name in sorted(os.listdir(base), key=lambda x: int(x) if x.isdigit() else 1 << 30):
            props = {}
            try:
                with open(os.path.join(base, name, "properties")) as fh:
                    for line in fh:
                        k, _, v = line.strip().partition(" ")
                        props[k] = v
            except OSError:
                continue
            if int(props.get("simd_count", "0") or 0) > 0:                 # a GPU node
                gpus.append(int(props.get("drm_render_minor", "-1") or -1))
        if 0 <= device_index < len(gpus) and gpus[device_index] >= 0:
            node = read_int(os.path.join(sysfs, "class", "drm", f"renderD{gpus[device_index]}", "device", "numa_node"))
            if node is not None:
                return node if node >= 0 else None
    except (OSError, ValueError):
        pass
    try:
        drm = os.path.join(sysfs, "class", "drm")
        cards = sorted((c for c in os.listdir(drm) if c.startswith("card") and c[4:].isdigit()), key=lambda c: int(c[4:]))
        cards = [c for c in cards if read_int(os.path.join(drm, c, "device", "numa_node")) is not None]
        if 0 <= device_index < len(cards):
            node = read_int(os.path.join(drm, cards[device_index], "device", "numa_node"))
            return node if node is not None and node >= 0 else None
    except OSError:
        pass
    return None


def physical_device_index(device_index: int) -> Optional[int]:
    """The index in the node's own enumeration (KFD topology order) of the GPU this process sees as `device_index`:
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES renumber the devices.  None when the list is not plain
    indices (UUIDs) or does not reach that far."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is None or val.strip() == "":
            continue
        items = [x.strip() for x in val.split(",")]
        if not all(x.isdigit() for x in items) or device_index >= len(items):
            return None
        return int(items[device_index])
    return device_index


def numa_cpus(node: int, sysfs: str = "/sys"):
    """CPUs of NUMA node `node` (`<sysfs>/devices/system/node/node<N>/cpulist`, e.g. "0-47,96-143"); empty set when unknown."""
    cpus = set()
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as fh:
            for part in fh.read().strip().split(","):
                if not part:
                    continue
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
    except (OSError, ValueError):
        return set()
    return cpus


def pin_to_gpu_numa_node(device_index: int, local_rank: int = 0, local_world: int = 1, sysfs: str = "/sys") -> Dict[str, object]:
    """Restrict this process to CPUs of its GPU's NUMA node.  A rank's host thread issues 2-3 launches per 13-15 us step and --
    with HSA_ENABLE_INTERRUPT=0 -- polls its completion signals: eight ranks each burn a core, and a core on the far socket
    adds a hop to every doorbell and every signal read.  The ranks that share a node split its CPUs into equal contiguous slices
    (so that two polling ranks never sit on the same core) when there are at least two CPUs per rank, and share the whole node
    otherwise.  Only CPUs the process is already allowed to use are kept; nothing is changed when the topology cannot be read
    or the intersection is empty.  Returns {"numa_node", "cpus", "pinned"} for the bench line."""
    info: Dict[str, object] = {"numa_node": None, "cpus": None, "pinned": False}
    if not hasattr(os, "sched_setaffinity"):
        return info
    phys = physical_device_index(device_index)
    if phys is None:                                   # the visible-devices list names GPUs by UUID: no index to look up
        return info
    mates_phys = [physical_device_index(r) for r in range(max(1, local_world))]
    node = gpu_numa_node(phys, sysfs)
    if node is None:
        return info
    info["numa_node"] = node
    allowed = sorted(numa_cpus(node, sysfs) & set(os.sched_getaffinity(0)))
    if not allowed:
        return info
    # ranks on the same NUMA node: those whose GPU reports the same node
    mates = [r for r, ph in enumerate(mates_phys) if ph is not None and gpu_numa_node(ph, sysfs) == node] or [local_rank]
    if local_rank in mates and len(allowed) >= 2 * len(mates):
        per = len(allowed) // len(mates)
        k = mates.index(local_rank)
        allowed = allowed[k * per:(k + 1) * per]
    try:
        os.sched_setaffinity(0, allowed)
    except OSError:
        return info
    info["cpus"], info["pinned"] = len(allowed), True
    return info


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
            dev = select_device(local_rank, local_world_size(world), torch.cuda.device_count())
            torch.cuda.set_device(dev)
            kwargs["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def reduce_counters(counters, wall_seconds: float, force_collectives: bool = False) -> Dict[str, float]:
    """counters: tensor [4] (or [B,4]) of this rank's window; returns the whole-job totals and the
    max-over-ranks wall time.  One SUM and one MAX all-reduce (skipped in a world of one rank unless
    ``force_collectives``).  The MAX carries three values -- the wall time, this rank's own rate (env steps over ITS wall
    time) and the negated rate -- so that the slowest and the fastest rank come out of the same collective
    (``rank_rate_min`` / ``rank_rate_max``: a straggler GPU is visible in the line of a multi-GPU run)."""
    import torch
    import torch.distributed as dist
    c = counters.sum(dim=0) if counters.dim() == 2 else counters
    c = c.to(torch.float64)
    mine = c[0] / wall_seconds if wall_seconds > 0 else c[0] * 0.0      # (stays on the counters' device: no extra round trip)
    t = torch.stack([torch.tensor(float(wall_seconds), dtype=torch.float64, device=c.device), mine, -mine])
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives):
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    steps, episodes, makespan_sum, reward_num = [float(x) for x in c.tolist()]
    seconds, rate_max, neg_rate_min = [float(x) for x in t.tolist()]
    return {"steps": steps, "episodes": episodes, "makespan_sum": makespan_sum, "reward_num_sum": reward_num,
            "seconds": seconds, "steps_per_second": steps / seconds if seconds > 0 else 0.0,
            "rank_rate_min": -neg_rate_min, "rank_rate_max": rate_max}
