"""World-size-2 tests of the sharding + counter reduction used by bench.py --gpus N (gloo; the GPU box runs
the same worker through the HIP engine with both ranks on its one GPU).

Each rank owns the contiguous slice shard_bounds() gives it, builds its own BatchedJssEnv with
env_id_base = first env of the slice (the RNG is keyed by the GLOBAL env id) and steps it with no data-path
collective; the only exchange is reduce_counters().  The union of the shards must be bit-identical to one
process stepping the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INSTANCES = ("ta01", "ta31", "ta02")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_shard(device, lo, hi, iters, seed):
    from jssenv_amd import BatchedJssEnv
    env = BatchedJssEnv(list(INSTANCES), batch=hi - lo, device=device, seed=seed, env_id_base=lo,
                        table_of_env=(np.arange(lo, hi) % len(INSTANCES)))
    env.reset()
    env.rollout("random", n_iter=iters)             # fused policy + step, auto-restart
    env.rollout_steps("random", steps=5, n_sub=2)   # and the sub-batch form on top
    env.synchronize()
    n = env.backend.numpy
    return env, {k: n(getattr(env, k)) for k in env._STATE_TENSORS}


def _worker(rank, world, port, device, global_batch, iters, seed, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from jssenv_amd.distributed import init_from_env, reduce_counters, shard_bounds
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = shard_bounds(global_batch, world, rank)
    env, tensors = _run_shard(device, lo, hi, iters, seed)
    dist.barrier()
    out = reduce_counters(torch.from_numpy(tensors["counters"]), wall_seconds=1.0 + rank)
    q.put((rank, lo, hi, env.backend.lib.jss_backend().decode(), tensors, out))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks_equal_one_process(device, global_batch, iters, seed=7):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, device, global_batch, iters, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    half = global_batch // 2
    assert [(r[1], r[2]) for r in results] == [(0, global_batch - half), (global_batch - half, global_batch)]
    # both ranks hold the same whole-job totals, and the wall time is the MAX over ranks
    assert results[0][5] == results[1][5] and results[0][5]["seconds"] == 2.0
    # single-process ground truth over the whole batch, same engine
    _, whole = _run_shard(device, 0, global_batch, iters, seed)
    for name, want in whole.items():
        got = np.concatenate([results[0][4][name], results[1][4][name]], axis=0)
        assert np.array_equal(got, want), f"union of the two shards differs from the unsharded batch in {name}"
    tot = results[0][5]
    c = whole["counters"].sum(axis=0)
    assert (tot["steps"], tot["episodes"], tot["makespan_sum"], tot["reward_num_sum"]) == tuple(float(x) for x in c)
    assert tot["steps"] > 0 and tot["steps_per_second"] == tot["steps"] / 2.0
    # the slowest and the fastest rank of the window (each rank's own env steps over ITS wall time: 1 s and 2 s here)
    own = [float(results[r][4]["counters"][:, 0].sum()) / (1.0 + r) for r in range(2)]
    assert tot["rank_rate_min"] == min(own) and tot["rank_rate_max"] == max(own)
    return results[0][3]


def test_shard_bounds():
    from jssenv_amd.distributed import shard_bounds
    for B, W in ((65536, 8), (10, 3), (7, 8), (1, 1)):
        spans = [shard_bounds(B, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_each_rank_selects_its_own_gpu():
    """Rank r of an N-GPU node drives device r (what only a real multi-GPU run executes with r > 0), fewer devices than
    ranks is an error, --share-device puts everybody on device 0 -- and init_from_env("nccl") hands exactly that device to
    torch.cuda.set_device and to the process group (torch mocked: no GPU, no rendezvous)."""
    from unittest import mock
    from jssenv_amd import distributed as D
    assert [D.select_device(r, 8, 8) for r in range(8)] == list(range(8))
    assert [D.select_device(r, 4, 8) for r in range(4)] == [0, 1, 2, 3]          # 4 ranks on an 8-GPU node
    assert [D.select_device(r, 8, 1, share_device=True) for r in range(8)] == [0] * 8
    with pytest.raises(RuntimeError, match="no oversubscription"):
        D.select_device(3, 8, 4)
    with pytest.raises(ValueError):
        D.select_device(8, 8, 8)
    for r in (0, 5, 7):
        env = dict(RANK=str(r), WORLD_SIZE="8", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
        with mock.patch.dict(os.environ, env), mock.patch("torch.cuda.device_count", return_value=8), \
                mock.patch("torch.cuda.set_device") as set_device, mock.patch("torch.distributed.is_initialized", return_value=False), \
                mock.patch("torch.distributed.init_process_group") as init_pg:
            assert D.init_from_env("nccl") == (r, 8, r)
        set_device.assert_called_once_with(r)
        (backend,), kw = init_pg.call_args
        assert backend == "nccl" and kw["rank"] == r and kw["world_size"] == 8 and kw["device_id"] == torch.device("cuda", r)
    with mock.patch.dict(os.environ, dict(RANK="6", WORLD_SIZE="8", LOCAL_RANK="6")), mock.patch("torch.cuda.device_count", return_value=4), \
            mock.patch("torch.distributed.is_initialized", return_value=False), mock.patch("torch.distributed.init_process_group") as init_pg:
        with pytest.raises(RuntimeError, match="8 ranks but only 4"):
            D.init_from_env("nccl")
        init_pg.assert_not_called()


def test_multi_node_job_selects_by_local_world_size():
    """2 nodes x 8 GPUs: WORLD_SIZE = 16, LOCAL_WORLD_SIZE = 8, 8 devices per node -- rank 13 is local rank 5 and drives GPU 5
    (round 5 compared the device count with the global world size and refused such a job)."""
    from unittest import mock
    from jssenv_amd import distributed as D
    env = dict(RANK="13", WORLD_SIZE="16", LOCAL_RANK="5", LOCAL_WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    with mock.patch.dict(os.environ, env), mock.patch("torch.cuda.device_count", return_value=8), \
            mock.patch("torch.cuda.set_device") as set_device, mock.patch("torch.distributed.is_initialized", return_value=False), \
            mock.patch("torch.distributed.init_process_group") as init_pg:
        assert D.init_from_env("nccl") == (13, 16, 5)
    set_device.assert_called_once_with(5)
    (backend,), kw = init_pg.call_args
    assert backend == "nccl" and kw["rank"] == 13 and kw["world_size"] == 16 and kw["device_id"] == torch.device("cuda", 5)
    with mock.patch.dict(os.environ, dict(env, LOCAL_RANK="8")), mock.patch("torch.cuda.device_count", return_value=8), \
            mock.patch("torch.distributed.is_initialized", return_value=False), mock.patch("torch.distributed.init_process_group"):
        with pytest.raises(ValueError, match="outside a node of 8"):
            D.init_from_env("nccl")


def _fake_sysfs(root, gpu_nodes, cpulists, cpu_only_nodes=2):
    """A sysfs tree with `cpu_only_nodes` CPU entries in the KFD topology followed by one GPU entry per element of `gpu_nodes`
    (its NUMA node), render minors 128.., and `cpulists[n]` as NUMA node n's cpulist."""
    topo = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    for i in range(cpu_only_nodes):
        os.makedirs(os.path.join(topo, str(i)))
        with open(os.path.join(topo, str(i), "properties"), "w") as fh:
            fh.write("cpu_cores_count 48\nsimd_count 0\ndrm_render_minor 0\n")
    for g, node in enumerate(gpu_nodes):
        d = os.path.join(topo, str(cpu_only_nodes + g))
        os.makedirs(d)
        with open(os.path.join(d, "properties"), "w") as fh:
            fh.write(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n")
        dev = os.path.join(root, "class", "drm", f"renderD{128 + g}", "device")
        os.makedirs(dev)
        with open(os.path.join(dev, "numa_node"), "w") as fh:
            fh.write(f"{node}\n")
    for n, cl in enumerate(cpulists):
        d = os.path.join(root, "devices", "system", "node", f"node{n}")
        os.makedirs(d)
        with open(os.path.join(d, "cpulist"), "w") as fh:
            fh.write(cl + "\n")


def test_rank_pins_its_host_thread_to_its_gpus_numa_node(tmp_path, monkeypatch):
    """8 GPUs on 2 sockets (GPUs 0-3 on node 0, 4-7 on node 1), 8 ranks: every rank ends up on CPUs of ITS GPU's socket, the
    four ranks of a socket on disjoint slices of it; a topology that cannot be read changes nothing (sysfs mocked, the
    affinity calls mocked: this container has 8 CPUs)."""
    from unittest import mock
    from jssenv_amd import distributed as D
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    root = str(tmp_path)
    _fake_sysfs(root, [0, 0, 0, 0, 1, 1, 1, 1], ["0-47,96-143", "48-95,144-191"])
    assert [D.gpu_numa_node(g, root) for g in range(8)] == [0, 0, 0, 0, 1, 1, 1, 1] and D.gpu_numa_node(8, root) is None
    assert D.numa_cpus(1, root) == set(range(48, 96)) | set(range(144, 192)) and D.numa_cpus(7, root) == set()
    seen = {}
    for r in range(8):
        with mock.patch("os.sched_getaffinity", return_value=set(range(192))), mock.patch("os.sched_setaffinity") as setaff:
            info = D.pin_to_gpu_numa_node(r, r, 8, sysfs=root)
        (pid, cpus), _ = setaff.call_args
        assert pid == 0 and info == {"numa_node": r // 4, "cpus": 24, "pinned": True}
        assert set(cpus) <= D.numa_cpus(r // 4, root)
        seen[r] = set(cpus)
    assert all(not (seen[a] & seen[b]) for a in range(8) for b in range(a))          # nobody shares a core
    # a cgroup that allows only 4 CPUs of the socket: the ranks share them (no slices of less than two CPUs)
    with mock.patch("os.sched_getaffinity", return_value={0, 1, 2, 3, 50}), mock.patch("os.sched_setaffinity") as setaff:
        info = D.pin_to_gpu_numa_node(2, 2, 8, sysfs=root)
    assert setaff.call_args[0] == (0, [0, 1, 2, 3]) and info["cpus"] == 4
    # a visible-devices list renumbers the GPUs: this process's device 1 is the node's GPU 6 (socket 1); UUIDs: hands off
    with mock.patch.dict(os.environ, {"HIP_VISIBLE_DEVICES": "2,6"}), mock.patch("os.sched_getaffinity", return_value=set(range(192))), \
            mock.patch("os.sched_setaffinity") as setaff:
        assert D.physical_device_index(1) == 6 and D.physical_device_index(2) is None
        assert D.pin_to_gpu_numa_node(1, 1, 2, sysfs=root)["numa_node"] == 1 and set(setaff.call_args[0][1]) <= D.numa_cpus(1, root)
    with mock.patch.dict(os.environ, {"ROCR_VISIBLE_DEVICES": "GPU-abc,GPU-def"}), mock.patch("os.sched_setaffinity") as setaff:
        assert D.pin_to_gpu_numa_node(0, 0, 2, sysfs=root)["pinned"] is False
        setaff.assert_not_called()
    # unknown topology / a GPU without a NUMA node (-1) / no allowed CPU on the node: nothing is touched
    with mock.patch("os.sched_setaffinity") as setaff:
        assert D.pin_to_gpu_numa_node(0, 0, 1, sysfs=os.path.join(root, "nowhere"))["pinned"] is False
        with open(os.path.join(root, "class", "drm", "renderD128", "device", "numa_node"), "w") as fh:
            fh.write("-1\n")
        assert D.pin_to_gpu_numa_node(0, 0, 8, sysfs=root) == {"numa_node": None, "cpus": None, "pinned": False}
        with mock.patch("os.sched_getaffinity", return_value={200, 201}):
            assert D.pin_to_gpu_numa_node(5, 5, 8, sysfs=root)["pinned"] is False
        setaff.assert_not_called()


def test_two_ranks_equal_one_process_cpu_twin():
    """CPU (this container): the two ranks step their shards with libjss_cpu.so."""
    assert _two_ranks_equal_one_process("cpu", global_batch=37, iters=400).startswith("cpu")


@pytest.mark.gpu
def test_two_ranks_equal_one_process_hip_engine():
    """GPU box: both ranks drive the HIP engine on the one GPU (gloo for the counters): the union of the two
    shards' state tensors and counters is bit-identical to a single-process batch of twice the size."""
    assert _two_ranks_equal_one_process("cuda:0", global_batch=2 * 4096 + 70, iters=300) == "hip:gfx950"


def test_reduce_counters_single_process_tensor_forms():
    """reduce_counters on [4] and [B,4] tensors, no process group: totals and the rate."""
    from jssenv_amd.distributed import reduce_counters
    c = torch.tensor([[1, 2, 3, 4], [10, 20, 30, 40]], dtype=torch.int64)
    a, b = reduce_counters(c, 2.0), reduce_counters(c.sum(0), 2.0)
    assert a == b and a["steps"] == 11.0 and a["reward_num_sum"] == 44.0 and a["steps_per_second"] == 5.5


@pytest.mark.gpu
def test_reduce_counters_and_agree_max_on_device_tensors():
    """The forms bench.py uses on the GPU: counters as a CUDA tensor through reduce_counters (the .to(float64), the
    MAX of the wall time on the tensor's device) -- the lines only a multi-GPU run otherwise reaches."""
    from jssenv_amd.distributed import reduce_counters
    c = torch.tensor([[5, 1, 100, -7], [6, 0, 0, 9]], dtype=torch.int64, device="cuda:0")
    out = reduce_counters(c, 0.5)
    assert (out["steps"], out["episodes"], out["makespan_sum"], out["reward_num_sum"]) == (11.0, 1.0, 100.0, 2.0)
    assert out["seconds"] == 0.5 and out["steps_per_second"] == 22.0


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu():
    """`bench.py --gpus 8` end to end on the 1-GPU box: 8 ranks under torch.distributed.run sharing cuda:0, gloo for
    the collectives (RCCL needs one GPU per rank).  Exercises the re-exec under torchrun, the rank/shard arithmetic,
    the per-window SUM/MAX reductions and the config-4-sharded extra that only exists with N > 1."""
    import json
    import subprocess
    import tempfile
    detail = os.path.join(tempfile.mkdtemp(), "detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-device", "--dist-backend", "gloo",
           "--steps", "5", "--warmup", "1", "--batch", "4096", "--no-cpu-baseline", "--detail", detail]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    assert res.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096     # the driver's line: last, and small
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 5 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 8 * 4096 and out["value"] > 0
    assert abs(out["roofline"]["frac"] - out["value"] / 8 * out["roofline"]["alg_bytes_per_env_step"] / 8e12) < 1e-4 * out["roofline"]["frac"]
    assert out["configs"]["c4_syn50x20_b65536_sharded"] > 0
    full = json.load(open(detail))                       # everything, unabridged
    assert full["value"] == pytest.approx(out["value"], rel=1e-5)
    c4 = full["config4_sharded"]
    assert c4.get("value"), c4
    assert c4["global_batch"] == 65536 and c4["batch"] == 8192 and c4["n_gpus"] == 8 and c4["scaling"] == "strong"
    assert out["configs"]["c4_syn50x20_b65536_sharded"] == pytest.approx(c4["roofline_frac"], rel=1e-3)
    # Preflight of the real 8-GPU run, host side: 8 ranks issuing at once under this box's CPU quota (every rank's runtime
    # polls its completion signals, HSA_ENABLE_INTERRUPT=0) must still put a launch into its queue far faster than a GPU of
    # its own would retire it -- the headline's kernels take >= 13 us per step of 2-3 launches (one rank alone: ~2.7 us per launch).
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--batch", "4096",
                          "--no-cpu-baseline", "--no-extras", "--detail", detail + ".1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    alone = json.load(open(detail + ".1"))["host_issue_us_per_launch"]
    crowded = full["host_issue_us_per_launch"]                               # MAX over the 8 ranks
    # (bounds with slack: round 6 saw 8.7 us once on a busy box with the old 10-launch measurement; a pathological host -- tens
    #  of microseconds per launch -- is what this is here to catch)
    assert 0 < alone < 6.0 and 0 < crowded < 10.0, (alone, crowded)
    assert crowded < 4.0 * alone + 2.0, (alone, crowded)
    assert full["host"]["hsa_enable_interrupt"] == "0" and out["host_issue_us_per_launch"] == pytest.approx(crowded, rel=0.01)


_RCCL_WORLD1 = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=%r)
import torch, torch.distributed as dist
from jssenv_amd.distributed import init_from_env, reduce_counters
r, w, lr = init_from_env("nccl", force=True)           # "nccl" IS RCCL on ROCm
assert (r, w, lr) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
c = torch.tensor([[5, 1, 100, -7], [6, 0, 0, 9]], dtype=torch.int64, device="cuda:0")
out = reduce_counters(c, 0.5, force_collectives=True)  # SUM + MAX all-reduce on device tensors through RCCL
assert (out["steps"], out["episodes"], out["makespan_sum"], out["reward_num_sum"]) == (11.0, 1.0, 100.0, 2.0), out
assert out["seconds"] == 0.5
t = torch.arange(8, dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert t.tolist() == list(range(8))
dist.destroy_process_group()
print("RCCL-WORLD1-OK")
'''


@pytest.mark.gpu
def test_rccl_process_group_of_one_rank():
    """The lines only a multi-GPU run otherwise reaches -- init_process_group("nccl", device_id=...), all_reduce of
    device tensors, barrier, teardown -- executed through RCCL with a world of one rank on the 1-GPU box."""
    import subprocess
    res = subprocess.run([sys.executable, "-c", _RCCL_WORLD1 % (ROOT, str(_free_port()))], capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0 and "RCCL-WORLD1-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.gpu
def test_bench_forced_process_group_over_rccl():
    """bench.py --force-process-group: the bench's own barrier / agree_max / reduce_counters calls go through RCCL."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-process-group", "--steps", "5", "--warmup", "1",
           "--batch", "4096", "--no-cpu-baseline", "--no-extras"]
    res = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["process_group"] == {"backend": "nccl", "world_size": 1, "forced_at_world_1": True}
    assert out["n_gpus"] == 1 and out["value"] > 0
