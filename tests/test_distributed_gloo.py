"""World-size-2 gloo test (CPU) of the sharding + counter reduction used by bench.py --gpus N.
Each rank steps its own shard of envs with the oracle as a stand-in stepping engine (the
partitioning, RNG keying by global env id and the collectives are what is under test)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, global_batch, iters, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from jssenv_amd import builtin_instance
    from jssenv_amd.distributed import init_from_env, reduce_counters, shard_bounds
    from oracle import OracleEnv
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = shard_bounds(global_batch, world, rank)
    inst = builtin_instance("ta01")
    counters = torch.zeros(hi - lo, 4, dtype=torch.int64)
    for i, env_id in enumerate(range(lo, hi)):          # env_id_base = lo keys the RNG by GLOBAL env id
        o = OracleEnv(inst, strict=True)
        o.reset()
        res = o.rollout("random", 7, env_id, iters, episode=1)
        counters[i] = torch.tensor([res["steps"], res["episodes"], res["makespan_sum"],
                                    round(res["reward_sum"] * inst.max_time_op)])
    dist.barrier()
    out = reduce_counters(counters, wall_seconds=1.0 + rank)
    q.put((rank, lo, hi, out))
    dist.destroy_process_group()


def test_shard_bounds():
    from jssenv_amd.distributed import shard_bounds
    for B, W in ((65536, 8), (10, 3), (7, 8), (1, 1)):
        spans = [shard_bounds(B, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_two_rank_counters_match_single_process():
    world, B, iters = 2, 6, 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in results] == [(0, 3), (3, 6)]
    # both ranks hold the same whole-job totals, and the wall time is the MAX over ranks
    assert results[0][3] == results[1][3]
    tot = results[0][3]
    assert tot["seconds"] == 2.0
    # single-process ground truth over the whole batch
    sys.path.insert(0, ROOT)
    from jssenv_amd import builtin_instance
    from oracle import OracleEnv
    inst = builtin_instance("ta01")
    steps = episodes = mk = 0
    for env_id in range(B):
        o = OracleEnv(inst, strict=True)
        o.reset()
        res = o.rollout("random", 7, env_id, iters, episode=1)
        steps, episodes, mk = steps + res["steps"], episodes + res["episodes"], mk + res["makespan_sum"]
    assert (tot["steps"], tot["episodes"], tot["makespan_sum"]) == (steps, episodes, mk)
    assert tot["steps_per_second"] == steps / 2.0
