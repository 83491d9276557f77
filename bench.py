#!/usr/bin/env python
"""bench.py -- env steps/sec of the batched JSS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-executes itself under torch.distributed.run,
                                                            one rank per GPU, RCCL; fails loudly with fewer GPUs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the batch: for every env, pick a random masked action on the
device, execute step() and write the full gym outputs (real_obs, action_mask, reward, done) to HBM -- exactly
what a reference ``obs, r, done, _, _ = env.step(policy(obs))`` iteration produces, fused in one kernel
(jss_rollout with n_iter = 1).  Envs found done are reset by that pass instead (the iteration is not counted as
an env step).  A pass is ONE launch over the batch (issued by a Python loop, a hipGraph replay or the library's C loop), or
n_sub launches over n_sub contiguous sub-batches on n_sub HIP streams (jss_rollout_steps): step s of a sub-batch depends only
on its own step s-1, so one sub-batch's drain overlaps another's fill; results are identical either way.  Which form: a short
probe ranks them, every contender within 15 % of the best probe gets the full measurement, the best median is reported
(best_form).

Timing: W untimed warm-up steps, then windows of EXACTLY K steps, each bracketed by barrier +
torch.cuda.synchronize() on both sides; per window the wall time is the MAX over ranks and the env steps the SUM
over ranks (one RCCL all-reduce each, outside the timed region).  As many windows as it takes for the timed total to
reach 0.55 s for the headline (0.1 s for the other configs; at least 5, at most 4 000 windows): the driver's K = 20 makes a
window 0.3 ms, and a few of those are noise.  value = median window; n / min / p10 / max are printed too.  roofline.frac is
value x algorithmic bytes / peak -- the wall-clock number anybody can recompute from the line; the HIP-event figure is kept
as roofline.frac_gpu_time.

Workload: BASELINE.json configs[1] shape (ta01, 15x15, one shared instance, random masked policy) at the
north_star's target batch of 65 536 envs per GPU (weak scaling: every rank owns its own 65 536 envs, no data-path
collective).

Output: the LAST line of stdout is ONE compact JSON object (< 4 KB: metric / value / ms_per_step / config / roofline /
cpu_baseline + one roofline fraction per BASELINE config, `compact_line`); everything that was measured goes, unabridged,
to bench_detail.json next to this file (and to gpurun_out/ when that directory exists).  A default run measures the
headline, the plain one-launch-per-step form of it (the kernel duration rocprofv3 reports) and BASELINE configs 2-5 in
their fused one-launch-per-step form.  `--extras` adds the side measurements of earlier rounds -- jss_step alone with
resident actions (step_only), trajectory mode, jss_steps / step sessions with external actions, the B = 1 facade, 4x the
batch, per-env synthetic 15x15 tables, the C-oracle and twin CPU baselines; `--no-extras` keeps the headline only.

With N > 1 the same line also carries config4_sharded: BASELINE config 4 as it is defined -- synthetic 50x20, 65 536
envs split over the N ranks by shard_bounds (strong scaling) -- measured after the weak-scaling headline.

Extra objects: roofline (HBM; frac = algorithmic bytes over the wall clock, frac_gpu_time over HIP events on the launch
stream, single_launch = the one-launch-per-step form whose kernel duration rocprofv3 reports) and cpu_baseline = the
Python / NumPy restatement of the reference's step() on ONE host core (kind "port": oracle/np_restatement.py, the stand-in
for the reference's own speed, which cannot travel to this box), timed on this box in this run (rank 0, N = 1, ~5 s).
--extras adds cpu_baseline_c_oracle (the C oracle, one env per thread) and cpu_baseline_twin (libjss_cpu.so).
"""
import argparse
import gc
import hashlib
import json
import math
import os
import socket
import sys
import time

# Completion signals are polled instead of waited for through interrupts: a 20-step window is 0.25-0.4 ms of GPU
# work, and the interrupt path adds tens of microseconds to every synchronize().  Process-wide HSA runtime setting,
# must be in the environment before the runtime starts; reported in the JSON line (host.hsa_enable_interrupt).
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
# HIP deals streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; two streams that share a queue
# run strictly one after the other.  The pipelined forms use up to 5 streams (main + side streams of sub-batches or
# shape buckets): 8 queues keep them apart whatever was created before (profiles/README.md: the bucketed env measured
# 0.43-0.53 of the roofline inside the full bench with 4 queues, depending on which queues its streams drew).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_PEAK_GBS = 6290.0  # measured copy bandwidth, same guide (SURVEY.md 8(d) asks for both)
# windows per measurement: as many as it takes for the timed total to reach MIN_TIMED_SECONDS (the headline: half a second --
# at the driver's K = 20 a window is 0.3 ms, so that is ~1 600 windows; the other configs 0.1 s each, the step_only legs 0.05 s)
MIN_WINDOWS, MAX_WINDOWS = 5, 4000
MIN_TIMED_SECONDS_HEADLINE, MIN_TIMED_SECONDS, MIN_TIMED_SECONDS_LIGHT = 0.55, 0.1, 0.05
CPU_BASELINE_SECONDS = 5.0


def csrc_hash():
    """sha256 over the kernel sources + ABI header: ties the static counter numbers in profiles/hbm_traffic.json to
    the binary they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "jssenv_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/jss_hip.h"]:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def b_alg(J, M):
    """Algorithmic bytes of one env step: read + write the per-env state
    (7 int32/job, 2 flag bytes/job, int32 + flag per machine, clock + counters), the action,
    one solution entry, the float32 observation, the mask, reward and done."""
    return 89 * J + 10 * M + 40


# ------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1, after the GPU measurement)
# ------------------------------------------------------------------------------------------------------------
def host_cores():
    """What this box gives us: logical CPUs, the affinity mask, and the cgroup CPU quota (a container may see 256
    logical CPUs and be allowed 8 cores' worth of time -- the thread counts below are what was USED, this is what
    was AVAILABLE)."""
    info = {"logical": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_quota_cores": None}
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            info["cgroup_quota_cores"] = float(quota) / float(period)
    except Exception:
        pass
    return info


def usable_threads():
    """Threads worth starting: the cgroup quota when there is one (more only oversubscribes), else the affinity mask."""
    h = host_cores()
    if h["cgroup_quota_cores"]:
        return max(1, int(h["cgroup_quota_cores"] + 0.5))
    return h["affinity"] or h["logical"] or 1



def cpu_baseline_restatement(inst_name, seed, target_seconds=CPU_BASELINE_SECONDS, instance=None):
    """The reference's own way of doing a step -- Python loops over NumPy arrays, one env, one core
    (oracle/np_restatement.py: attribute-for-attribute restatement of jss_env.py:121-653, pinned bit-exactly to the
    reference's golden traces; in the build container it runs at the live reference's speed) -- driven by the
    README's random masked loop."""
    import numpy as np
    from jssenv_amd import builtin_instance
    from oracle.np_restatement import NumpyJssEnv, random_masked_episode
    env = NumpyJssEnv(instance if instance is not None else builtin_instance(inst_name))
    rng = np.random.default_rng(seed)
    random_masked_episode(env, rng)                      # warm
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < target_seconds:
        steps += random_masked_episode(env, rng)
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env steps/s", "cores": 1, "kind": "port",
            "sample": f"{inst_name}, random masked policy (README.md:53-64) + step() to completion, whole episodes for "
                      f"{dt:.1f} s ({steps} env steps), single env, one core, Python {sys.version_info[0]}.{sys.version_info[1]} + NumPy",
            "implementation": "oracle/np_restatement.py (Python loops over NumPy arrays, like JSSEnv/envs/jss_env.py)"}


def cpu_baseline_port(inst_name, seed, target_seconds=8.0):
    """The C oracle (a scalar restatement of the reference's step(), oracle/jss_oracle.c) running
    the same policy+step loop on this box's host cores, one env per thread."""
    import concurrent.futures as cf
    from jssenv_amd import builtin_instance
    from oracle import OracleEnv
    inst = builtin_instance(inst_name)
    threads = max(1, min(usable_threads(), 64))
    envs = [OracleEnv(inst, strict=True) for _ in range(threads)]

    def run(iters):
        for e in envs:
            e.reset()
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(threads) as ex:
            steps = sum(ex.map(lambda i: envs[i].rollout("random", seed, i, iters, episode=1)["steps"], range(threads)))
        return steps, time.perf_counter() - t0

    t0 = time.perf_counter()
    envs[0].reset()
    envs[0].rollout("random", seed, 0, 50000, episode=1)
    per_step = (time.perf_counter() - t0) / 50000
    # size the sample from a short all-threads burst (threads share cores/caches: no linear scaling)
    cal_iters = 100000
    _, cal_dt = run(cal_iters)
    iters = int(max(cal_iters, cal_iters * target_seconds / max(cal_dt, 1e-6)))
    steps, dt = run(iters)
    return {"value": steps / dt, "unit": "env steps/s", "cores": threads, "kind": "port (C oracle)",
            "sample": f"{inst_name} random-masked policy+step, {threads} envs x {iters} iterations "
                      f"({steps} env steps, {dt:.1f} s, one env per thread; 1 thread alone = {1.0 / per_step:.0f} steps/s)"}


def cpu_baseline_twin(inst_name, seed, target_seconds=4.0):
    """libjss_cpu.so (the from-scratch C++/OpenMP twin with the HIP library's C ABI): same policy+step loop over a
    batch of envs, on one core and on all cores."""
    from jssenv_amd import BatchedJssEnv
    from jssenv_amd.env import CpuBackend
    cores = usable_threads()
    out = {"unit": "env steps/s", "kind": "twin", "library": "libjss_cpu.so (C++17 + OpenMP over envs)"}
    for label, threads, batch in (("one_core", 1, 256), ("all_cores", cores, 64 * cores)):
        env = BatchedJssEnv(inst_name, batch=batch, seed=seed, _backend=CpuBackend(threads=threads))
        env.reset()
        env.rollout("random", n_iter=300)
        env.zero_counters()
        t0 = time.perf_counter()
        env.rollout("random", n_iter=200)
        cal = time.perf_counter() - t0
        iters = int(max(200, 200 * target_seconds / max(cal, 1e-6)))
        env.zero_counters()
        t0 = time.perf_counter()
        env.rollout("random", n_iter=iters)
        dt = time.perf_counter() - t0
        steps = env.stats()["steps"]
        out[label] = {"value": steps / dt, "cores": threads,
                      "sample": f"{inst_name} random-masked policy+step, {batch} envs x {iters} iterations ({steps} env steps, {dt:.1f} s)"}
    out["value"], out["cores"] = out["all_cores"]["value"], cores
    return out


# ------------------------------------------------------------------------------------------------------------
# The driver's line.  Round 4 printed everything on one 28 KB line and the driver's parser gave up on it: the line a
# run ENDS with is now a fixed, small set of keys (tests/test_bench_line.py holds it under 4 KB); the rest is the detail file.
# ------------------------------------------------------------------------------------------------------------
COMPACT_MAX_BYTES = 4096
CONFIG_KEYS = {    # detail-file key of a BASELINE config's side run -> its short name in the compact `configs` map
    "config2_ta01_batch4096_random": "c2_ta01_b4096_random",
    "config3_ta41_spt_batch16384": "c3_ta41_b16384_spt",
    "config4_synthetic50x20_batch8192": "c4_syn50x20_b8192_one_gpu_share",
    "config4_synthetic50x20_batch65536_one_gpu": "c4_syn50x20_b65536_one_gpu",
    "config4_sharded": "c4_syn50x20_b65536_sharded",
    "config5_mixed_padded_batch32768": "c5_mixed_b32768_padded",
    "config5_mixed_padded_interleaved_batch32768": "c5_mixed_b32768_padded_interleaved",
    "config5_mixed_bucketed_batch32768": "c5_mixed_b32768_bucketed",
    "synthetic15x15_per_env_tables": "syn15x15_b65536_per_env_tables",
}


def _sig(x, n=5):
    """n significant digits (floats only): keeps the line short without changing what it says"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x == 0.0 or math.isnan(x) or math.isinf(x):
        return x
    return float(f"{x:.{n}g}")


def _pick(d, keys, n=5):
    return {k: _sig(d[k], n) for k in keys if isinstance(d, dict) and k in d}


def compact_line(out, detail_files=()):
    """The one JSON line the driver parses, from the full result dict `out` (pure function: tests/test_bench_line.py
    feeds it a canned dict).  Required keys are always present; optional ones are dropped, last first, until the line
    fits COMPACT_MAX_BYTES."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"), n=7)
    cfg = out.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:300], **_pick(cfg, ("batch_per_gpu", "global_batch", "policy", "parallelism"))}
    line["config"]["launch"] = str(out.get("launch", ""))[:120]
    rf = out.get("roofline") or {}
    line["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "frac_own_bytes", "frac_gpu_time", "traffic", "kernel",
                                  "gpu_ms_per_step_events", "alg_bytes_per_env_step", "env_steps_per_launch", "wait_fraction",
                                  "wave_cycles_per_env_step"))
    single = out.get("single_launch_per_step")
    if isinstance(single, dict) and single.get("value"):
        # the plain form (ONE launch over the whole batch per step): the kernel duration rocprofv3 --kernel-trace reports
        line["roofline"]["single_launch"] = _pick(single, ("value", "gpu_ms_per_step_events", "roofline_frac", "roofline_frac_gpu_time"))
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = None if not isinstance(cb, dict) else {**_pick(cb, ("value", "unit", "cores", "kind")),
                                                                  "sample": str(cb.get("sample", ""))[:160]}
    line["configs"], line["configs_env_steps_per_s"], line["configs_step_only"] = {}, {}, {}
    so = out.get("step_only")
    if isinstance(so, dict):
        line["configs_step_only"]["headline"] = _sig(so.get("roofline_frac"), 4) if so.get("value") else None
    for key, shortname in CONFIG_KEYS.items():
        ent = out.get(key)
        if isinstance(ent, dict):
            line["configs"][shortname] = _sig(ent.get("roofline_frac"), 4) if ent.get("value") else None
            line["configs_env_steps_per_s"][shortname] = _sig(ent.get("value"), 4)
            so = ent.get("step_only")
            if isinstance(so, dict):
                line["configs_step_only"][shortname] = _sig(so.get("roofline_frac"), 4) if so.get("value") else None
    line["configs_note"] = ("roofline.frac (wall clock, K steps per window) of each config's fused one-launch-per-step form on one GPU; "
                            "configs_step_only: the same for jss_step with the CALLER's actions, one launch per step = JssEnv.step(action)")
    line["windows"] = _pick(out.get("windows") or {}, ("n", "min", "p10", "max", "below_90pct_of_median", "timed_seconds_total"))
    if isinstance(out.get("ranks"), dict):
        line["ranks"] = _pick(out["ranks"], ("value_min", "value_max", "numa_pinned"))
    if out.get("mean_makespan") is not None:
        line["mean_makespan"] = _sig(out["mean_makespan"])
    if out.get("process_group"):
        line["process_group"] = out["process_group"]
    if out.get("host_issue_us_per_launch") is not None:
        line["host_issue_us_per_launch"] = _sig(float(out["host_issue_us_per_launch"]), 3)
    line["csrc_sha16"] = out.get("csrc_sha16")
    line["detail"] = list(detail_files)
    text = json.dumps(line, separators=(",", ":"))
    for optional in ("configs_note", "configs_env_steps_per_s", "mean_makespan", "detail", "configs_step_only"):
        if len(text) <= COMPACT_MAX_BYTES:
            break
        line.pop(optional, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_MAX_BYTES:                    # cannot happen with the bounded strings above; never print a long line
        line["config"]["workload"] = line["config"]["workload"][:80]
        line["cpu_baseline"] = _pick(cb or {}, ("value", "unit", "cores", "kind")) or None
        text = json.dumps(line, separators=(",", ":"))
    return text


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=65536, help="envs per GPU (weak scaling) / whole job (strong scaling)")
    ap.add_argument("--instance", default="ta01")
    ap.add_argument("--policy", default="random")
    ap.add_argument("--workload", default="shared",
                    choices=["shared", "synthetic15x15", "synthetic50x20", "mixed"],
                    help="shared: one instance (--instance) for the whole batch [default, the headline]; "
                         "synthetic15x15 / synthetic50x20: one Taillard-LCG instance per env (50x20 = BASELINE config 4); "
                         "mixed: BASELINE config 5, env i <- ta(1 + i %% 80), padded 100x20")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch envs per GPU; strong: --batch envs in total, split by shard_bounds")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager", "sub1", "sub2", "sub3", "sub4"],
                    help="how a step is issued: eager = one ctypes launch; graph = hipGraph replay of the K launches; "
                         "subN = N sub-batches on N streams (jss_rollout_steps; sub1 = one launch per step from the library's C "
                         "loop); auto = fastest on a probe")
    ap.add_argument("--bucketed", action="store_true",
                    help="mixed workload only: one compact sub-batch per shape class (BucketedJssEnv) instead of "
                         "padding every env to 100x20")
    ap.add_argument("--by-shape", action="store_true",
                    help="mixed workload only: the padded batch with its envs ordered by shape class (BatchedJssEnv(order='by_shape'), "
                         "which is also what the default constructor gives a ragged list): same padded tensors, stepped by ONE grid of "
                         "class-specialised bodies instead of the padded extents' kernel")
    ap.add_argument("--interleaved", action="store_true",
                    help="mixed workload only: env i <- ta(1 + i %% 80) (BatchedJssEnv(order='interleaved'): every env on the padded "
                         "extents' kernel, the default of rounds 1-5)")
    ap.add_argument("--bucketed-launch", default="grid", choices=["grid", "streams"],
                    help="--bucketed: ONE grid over all shape classes per step (jss_multi_rollout) or round 3's form, one launch "
                         "per class and step on a stream per class (A/B)")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="debug: every rank uses cuda:0 (exercises the multi-process path on a 1-GPU box; use with "
                         "--dist-backend gloo)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="create the process group and run every barrier / all-reduce through it even with one rank "
                         "(executes the RCCL lines of the multi-GPU path on a 1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="the headline only: no BASELINE configs 2-5, no side measurements")
    ap.add_argument("--extras", action="store_true",
                    help="also the side measurements (step_only, trajectory, external-action forms, facade, 4x batch, synthetic "
                         "15x15, the C-oracle / twin CPU baselines): minutes of extra run time, detail file only")
    ap.add_argument("--detail", default=None, help="where the unabridged JSON goes (default: bench_detail.json next to bench.py)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: become N ranks."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not args.share_device:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); refusing to run "
                         f"{args.gpus} ranks on fewer devices (no oversubscription, no silent fallback to 1 GPU)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    os.execv(sys.executable, cmd)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)

    import torch
    import torch.distributed as dist
    from jssenv_amd import BatchedJssEnv, builtin_instance
    from jssenv_amd.distributed import local_world_size, pin_to_gpu_numa_node, reduce_counters, select_device, shard_bounds
    from jssenv_amd.instances import synthetic_packed

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the GPU path has no CPU fallback")
    try:
        local_rank = select_device(local_rank, local_world_size(world), torch.cuda.device_count(), share_device=args.share_device)
    except (RuntimeError, ValueError) as exc:
        raise SystemExit(f"bench.py: {exc}")
    torch.cuda.set_device(local_rank)
    # N ranks on one host, each polling its completion signals (HSA_ENABLE_INTERRUPT=0): every rank on CPUs of its own GPU's
    # NUMA node, the ranks of a node on disjoint cores (distributed.pin_to_gpu_numa_node; a 1-rank run keeps the whole host for
    # its CPU baselines).  JSS_BENCH_NO_PIN=1: A/B.
    numa = {}
    if world > 1 and not args.share_device and os.environ.get("JSS_BENCH_NO_PIN", "0") != "1":
        try:
            numa = pin_to_gpu_numa_node(local_rank, int(os.environ.get("LOCAL_RANK", "0")), local_world_size(world))
        except Exception as exc:               # (a box whose sysfs looks different must not cost the measurement)
            numa = {"pinned": False, "error": f"{type(exc).__name__}: {exc}"[:120]}
    dev = torch.device("cuda", local_rank)
    backend = args.dist_backend or "nccl"   # "nccl" is RCCL on ROCm
    use_pg = world > 1 or args.force_process_group
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        assert dist.get_world_size() == world
    on_host = use_pg and backend != "nccl"

    # The side streams of the pipelined forms (sub-batches, shape buckets) are process-wide and shared (HipBackend.side_pool):
    # they are created here, back to back, before anything else makes streams -- HIP deals streams onto its hardware queues
    # in creation order, and a bucket stream created late in the run (after graph captures and session streams) has landed
    # on the queue of another bucket's stream: the bucketed extra then measured 0.38 instead of 0.55 inside the full bench.
    from jssenv_amd.env import HipBackend
    HipBackend(dev).side_pool(4)

    def barrier():
        if use_pg:
            dist.barrier()

    def agree_max(values):
        """element-wise MAX over ranks of a list of floats (every rank must take the same decisions)"""
        t = torch.tensor(values, dtype=torch.float64)
        if use_pg:
            t = t if on_host else t.to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    # ---- workloads ------------------------------------------------------------------------------------------
    def describe(workload, instance="ta01"):
        """(mean algorithmic bytes per env step, label, traffic key)"""
        if workload == "synthetic15x15":
            return b_alg(15, 15), "synthetic 15x15 (Taillard LCG), one instance per env", "syn15x15"
        if workload == "synthetic50x20":
            return b_alg(50, 20), "synthetic 50x20 (Taillard LCG), one instance per env", "syn50x20"
        if workload == "mixed":
            insts = [builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
            return sum(b_alg(i.jobs, i.machines) for i in insts) / 80.0, "mixed ta01-ta80 (env i <- ta(1 + i % 80))", "mixed"
        inst = builtin_instance(instance)
        return b_alg(inst.jobs, inst.machines), f"{instance} ({inst.jobs}x{inst.machines}) one instance shared by the batch", instance

    def make_env(workload, batch, first_env, policy, instance="ta01", bucketed=False, spread=True, order=None):
        if workload == "mixed" and bucketed:
            from jssenv_amd import BucketedJssEnv
            insts = [builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
            e = BucketedJssEnv(insts, batch=batch, device=dev, seed=args.seed, env_id_base=first_env, launch=args.bucketed_launch)
            e.reset()
            e.rollout(policy, n_iter=333, autoreset=True)
            e.zero_counters()
            return e
        if workload == "synthetic15x15":
            src = synthetic_packed(batch, 15, 15, first=first_env)
        elif workload == "synthetic50x20":
            src = synthetic_packed(batch, 50, 20, first=first_env)
        elif workload == "mixed":
            src = [builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
        else:
            src = builtin_instance(instance)
        e = BatchedJssEnv(src, batch=batch, device=dev, seed=args.seed, env_id_base=first_env,
                          order=order if workload == "mixed" else None)      # (None: the constructor's default -- by shape for a ragged list)
        e.reset()
        if spread:
            # Spread the episode phases (a fresh batch is in lock step: every env at step 0) so the timed
            # window sees the steady-state mix of episode stages: env i is advanced (i % 16) * 16 extra
            # steps through the separate policy + step kernels, skipping (-1) the envs that are ahead.
            ids = torch.arange(batch, device=dev) % 16
            skip = torch.full((batch,), -1, dtype=torch.int32, device=dev)
            for r in range(15):
                for _ in range(16):
                    e.step(torch.where(ids > r, e.policy(policy), skip))
        e.rollout(policy, n_iter=64, autoreset=True)
        e.zero_counters()
        return e

    # ---- timing ---------------------------------------------------------------------------------------------
    def window(env, policy, n_launch, n_iter, mode, graph=None, run=None, prep=None, events=False, bound=None):
        """Time n_launch steps.  Returns (wall seconds, GPU ms per step from HIP events on the launch stream -- only in
        the windows that ask for them: the windows behind `value` carry nothing but the steps)."""
        if events:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if prep is not None:
            prep()                             # untimed: e.g. put the state back where a recorded action trace starts
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if events:
            ev0.record()
        if bound is not None and not events:
            bound()                            # the window's ONE C call, resolved outside the timed region (measure)
        elif run is not None:
            run(n_launch)
        elif graph is not None:
            graph.replay()
        elif hasattr(env, "rollout_steps") and mode.startswith("sub"):
            # every window starts on an idle device and ends with a device-wide synchronize(): the sub-batch streams need
            # no fork / join events inside it (caller_orders_streams; A/B: JSS_BENCH_FORK_JOIN=1).  The windows that carry
            # HIP events on the launch stream keep the join, or the closing event would not wait for the side streams.
            free = (not events) and os.environ.get("JSS_BENCH_FORK_JOIN", "0") != "1"
            key = (policy, n_launch, mode, free)
            cache = env.__dict__.setdefault("_bench_bound", {})     # lives and dies with the env object
            if key not in cache:                 # arguments resolved once: the timed region holds one C call
                cache[key] = env.bind_rollout_steps(policy, steps=n_launch, n_sub=int(mode[3:]), autoreset=True,
                                                    caller_orders_streams=free)
            cache[key]()
        elif hasattr(env, "buckets"):   # every bucket's launches on its own stream, one host call for the window
            env.rollout_steps(policy, steps=n_launch, n_iter=n_iter, autoreset=True,
                              caller_orders_streams=(not events) and os.environ.get("JSS_BENCH_FORK_JOIN", "0") != "1")
        else:
            for _ in range(n_launch):
                env.rollout(policy, n_iter=n_iter, autoreset=True)
        if events:
            ev1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0          # this rank's time; the job's time is the MAX over ranks (reduce_counters)
        barrier()
        return dt, (ev0.elapsed_time(ev1) / n_launch if events else 0.0)

    def capture(env, policy, n_launch):
        # the launches go to torch's current stream, so a torch CUDAGraph captures them: one host call
        # replays all n_launch kernels (the Python+ctypes enqueue costs ~7 us per launch otherwise)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for _ in range(n_launch):
                    env.rollout(policy, n_iter=1, autoreset=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        return graph

    def pick_mode(env, policy, candidates):
        if hasattr(env, "buckets"):
            if env.launch == "grid" and args.launch in ("auto", "sub2", "sub3"):
                # parts per shape class (a grid per part and step, parts on their own streams): fastest on a short probe
                choices = (1, 2, 3) if args.launch == "auto" else (int(args.launch[3:]),)
                n_probe, probe = max(20, min(60, args.steps)), []
                for n in choices:
                    env.n_sub = n
                    window(env, policy, n_probe, 1, "eager")
                    probe.append(min(window(env, policy, n_probe, 1, "eager")[0] for _ in range(5)))
                probe = agree_max(probe)
                env.n_sub = choices[min(range(len(choices)), key=lambda i: probe[i])]
                if rank == 0:
                    print("bucketed parts-per-class probe: " + ", ".join(f"{n} {t:.6f}" for n, t in zip(choices, probe)), file=sys.stderr)
            return "eager"
        if args.launch != "auto":
            return args.launch
        probe = []
        n_probe = max(20, min(60, args.steps))      # probe at the window length that will be timed (pipelines ramp up and drain)
        for m in candidates:
            g = capture(env, policy, n_probe) if m == "graph" else None
            window(env, policy, n_probe, 1, m, g)                              # warm (streams, graph upload)
            probe.append(min(window(env, policy, n_probe, 1, m, g)[0] for _ in range(5)))
            del g
        probe = agree_max(probe)
        if rank == 0:
            print(f"launch-mode probe (s per {n_probe} steps): " + ", ".join(f"{m} {t:.6f}" for m, t in zip(candidates, probe)), file=sys.stderr)
        order = sorted(range(len(candidates)), key=lambda i: probe[i])
        pick_mode.ranking = [(candidates[i], probe[i]) for i in order]      # (the headline re-measures a close runner-up)
        return candidates[order[0]]

    def measure(env, policy, steps, mode, n_iter=1, windows=None, run=None, prep=None, min_seconds=MIN_TIMED_SECONDS):
        """Windows of `steps` steps each -- as many as it takes for the timed total to reach `min_seconds` (between
        MIN_WINDOWS and MAX_WINDOWS; every rank takes the same number).  Returns the median window and all windows,
        already reduced over ranks.  `run(n)` overrides what a window executes (extras)."""
        graph = capture(env, policy, steps) if mode == "graph" else None
        bound = None
        if run is None and graph is None and mode.startswith("sub") and hasattr(env, "bind_rollout_steps") and not hasattr(env, "buckets") \
                and os.environ.get("JSS_BENCH_FORK_JOIN", "0") != "1":
            # every argument of the window's launch loop resolved here: the timed region holds the call and nothing else
            bound = env.bind_rollout_steps(policy, steps=steps, n_sub=int(mode[3:]), autoreset=True, caller_orders_streams=True)
        window(env, policy, steps, n_iter, mode, graph, run, prep)           # untimed: side streams / graph exist before the first window
        first = min(window(env, policy, steps, n_iter, mode, graph, run, prep, bound=bound)[0] for _ in range(3))   # untimed: sizes the count
        if windows is None:
            t_win = agree_max([first])[0]
            windows = int(min(MAX_WINDOWS, max(MIN_WINDOWS, math.ceil(min_seconds / max(t_win, 1e-6)))))
        rows = []
        # The collector stays out of the timed loop: a full collection of a process that has imported torch takes 20-40 ms --
        # a hundred 0.3 ms windows -- and it fires whenever the allocation counters say so (round 6's first run: one window of
        # 1 420 at 0.037 G env-steps/s next to a median of 4.26 G).  Nothing in the loop makes cycles; it runs once afterwards.
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        for _ in range(windows):
            env.zero_counters()
            dt, _ = window(env, policy, steps, n_iter, mode, graph, run, prep, bound=bound)
            tot = reduce_counters(env.counter_totals().cpu() if on_host else env.counter_totals(), dt, force_collectives=use_pg)
            rows.append({"steps": tot["steps"], "seconds": tot["seconds"], "rate": tot["steps"] / tot["seconds"],
                         "episodes": tot["episodes"], "makespan_sum": tot["makespan_sum"], "reward_num": tot["reward_num_sum"],
                         "rank_rate_min": tot["rank_rate_min"], "rank_rate_max": tot["rank_rate_max"]})
        if gc_was_on:
            gc.enable()
        # the GPU-time view (HIP events on the launch stream around the same K steps): a few extra windows of their own
        ms = sorted(window(env, policy, steps, n_iter, mode, graph, run, prep, events=True)[1] for _ in range(min(5, windows)))
        event_ms = agree_max([ms[len(ms) // 2]])[0]
        del graph
        rows.sort(key=lambda r: r["rate"])
        for r in rows:
            r["gpu_ms_per_step_events"] = event_ms
        med = rows[len(rows) // 2]
        return med, rows

    SINGLE_LAUNCH_FORMS = ("eager", "graph", "sub1")       # one launch per step: they differ in how the host issues it

    def best_form(env, policy, candidates, min_seconds):
        """The launch form of a workload: a short probe ranks the candidates (pick_mode), then every CONTENDER gets the full
        measurement and the best median wins -- the probe is five short windows per form and calls forms within 15 % of each
        other wrongly often enough (round 6, one box: config 4's share 0.404 through a hipGraph where two sub-batches measure
        0.48).  Contenders: the best-probed one-launch-per-step form, and every pipelined form, within 15 % of the best probe.
        Every rank takes the same decisions (probe times and medians are reduced over ranks)."""
        mode = pick_mode(env, policy, candidates)
        ranking = getattr(pick_mode, "ranking", [])
        if args.launch != "auto" or hasattr(env, "buckets") or len(ranking) < 2:
            return (mode, *measure(env, policy, args.steps, mode, min_seconds=min_seconds))
        single = [r for r in ranking if r[0] in SINGLE_LAUNCH_FORMS][:1]
        piped = [r for r in ranking if r[0] not in SINGLE_LAUNCH_FORMS]
        contenders = sorted(single + piped, key=lambda r: r[1])
        contenders = [m for m, t in contenders if t <= 1.15 * contenders[0][1]][:3]
        results = [(m, *measure(env, policy, args.steps, m, min_seconds=min_seconds)) for m in contenders]
        rates = agree_max([r[1]["rate"] for r in results])
        return results[max(range(len(results)), key=lambda i: rates[i])]

    def window_stats(rows, steps):
        med = rows[len(rows) // 2]["rate"]
        return {"n": len(rows), "steps_each": steps, "statistic": "median", "min": rows[0]["rate"], "max": rows[-1]["rate"],
                "below_90pct_of_median": sum(1 for r in rows if r["rate"] < 0.9 * med),
                "p10": rows[len(rows) // 10]["rate"], "timed_seconds_total": sum(r["seconds"] for r in rows)}

    def launch_label(mode):
        if mode == "sub1":
            return "one launch per step, issued by the library's C loop (jss_rollout_steps, one stream)"
        if mode.startswith("sub"):
            return f"{mode[3:]} sub-batches on {mode[3:]} HIP streams per step (jss_rollout_steps)"
        return {"eager": "one launch per step (ctypes, eager)",
                "graph": "one launch per step, hipGraph replay of the K launches"}.get(mode, mode)

    def kernel_name(env):
        if hasattr(env, "buckets"):
            if env.launch == "grid":
                return "jss_multi_kernel<kRollout1>: one grid over the shape classes (16-lane groups, 32-lane groups, wave, wave x2)"
            return "four launches per step: jss_packed_kernel<16|32,kRollout1,*>, jss_kernel<1|2,kRollout1,*>"
        if getattr(env, "_classes", None) is not None:
            return "jss_multi_kernel<kRollout1>: one grid of class-specialised bodies on the padded rows (order='by_shape')"
        tab = ("kTabLdsC" if env.compact else "kTabLds") if env.n_tables == 1 else ("kTabGlobalM" if getattr(env, "medium", False) else "kTabGlobal")
        jm, mm = env.jmax, env.mmax
        if max(jm, mm) <= 32:
            return f"jss_packed_kernel<{16 if max(jm, mm) <= 16 else 32},kRollout1,{tab}>"
        return f"jss_kernel<{1 if jm <= 64 else 2},kRollout1,{tab}>"

    def static_traffic(key, batch):
        """(HBM-side bytes per launch, where they come from, {wave cycles per env step, wait fraction, ...}) from the
        committed rocprofv3 counter summaries -- static numbers, tied to the kernel sources by their hash."""
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if key is None:
            return None, None, {}
        try:
            with open(prof) as fh:
                ent = json.load(fh).get(f"{key}_b{batch}")
            if not ent:
                return None, None, {}
            stale = ent.get("csrc_sha16") != csrc_hash()
            sq = {k: ent[k] for k in ("wave_cycles_per_env_step", "wait_fraction", "valu_per_wave", "salu_per_wave") if k in ent}
            return ent["bytes_per_launch"], (f"profiles/{ent['source']} (static: rocprofv3 PMC passes of round {ent.get('round')}, not re-measured in "
                                             f"this run; kernel sources {'CHANGED since' if stale else 'unchanged since'} -- csrc_sha16 {ent.get('csrc_sha16')})"), sq
        except Exception:
            return None, None, {}

    def roofline(med, alg_per_step, steps, env, key, batch):
        """frac = whole-job env steps per second x algorithmic bytes per env step / (N x peak): the wall-clock figure,
        reproducible from `value` alone.  frac_gpu_time = the same bytes over the HIP-event time of the timed region
        (what the kernels achieve once launched; the difference is launch ramp-up, drain and the synchronisation)."""
        stepped = med["steps"] / world / steps
        achieved = med["rate"] / world * alg_per_step / 1e9
        gpu_time = stepped * alg_per_step / (med["gpu_ms_per_step_events"] * 1e-3) / 1e9
        traffic, src, sq = static_traffic(key, batch)
        step_s = med["seconds"] / steps
        # of the bytes the kernel really moves (counters; a pass in which every env steps): traffic / wall time per step
        own = (traffic / step_s / 1e9) if traffic else None
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_peak": achieved / HBM_MEASURED_PEAK_GBS,
                "frac_own_bytes": (own / HBM_PEAK_GBS) if own else None, "achieved_own_bytes": own,
                "frac_note": "frac = ALGORITHMIC bytes (SURVEY 8(d): 89 J + 10 M + 40 per env step) over the wall clock; the kernels move "
                             "fewer bytes than that (compact records, unchanged records not rewritten), so frac can exceed what a copy of "
                             "the algorithmic bytes could reach -- frac_own_bytes = the counters' bytes (traffic) over the same wall clock",
                **sq,
                "frac_gpu_time": gpu_time / HBM_PEAK_GBS, "achieved_gpu_time": gpu_time,
                "measured_peak": HBM_MEASURED_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                "kernel": kernel_name(env),
                # HIP-event time per step of windows of their own that keep the fork / join events -- GPU time of a STEP (all its
                # launches, overlapped or not), not a kernel duration: that is rocprofv3's, profiles/<round>_*/kernel_stats.csv
                "gpu_ms_per_step_events": med["gpu_ms_per_step_events"],
                "alg_bytes_per_env_step": alg_per_step, "env_steps_per_launch": stepped}

    def traj_measure(env, policy, alg, KT=32):
        """Trajectory mode on `env`: K steps per launch, every transition recorded (what a scripted / random behaviour
        policy collecting rollouts wants) -- no state reload, no K - 1 launch boundaries."""
        try:
            Jm = env.jmax
            bufs = env.trajectory(policy, steps=KT)
            n_l = max(2, args.steps // KT)

            def traj_run(n):
                for _ in range(n):
                    env.trajectory(policy, steps=KT, buffers=bufs)
            medt, rowst = measure(env, policy, n_l, "eager", run=traj_run)
            j_mean = float(env.jobs_per_env.mean())
            m_mean = float(env.machines_per_env.mean())
            rec_bytes = 28 * j_mean + (Jm + 1) + 4 + 4 + 1                      # obs rows + mask row + action + reward + done
            state_bytes = 2 * (32 * j_mean + 4 * m_mean + 16)                   # state read + written once per launch
            del bufs
            return {"value": medt["rate"], "unit": "env steps/s", "steps_per_launch": KT, "launches_per_window": n_l,
                    "windows": window_stats(rowst, n_l * KT),
                    "roofline_frac": medt["rate"] / world * alg / 1e9 / HBM_PEAK_GBS,
                    "bytes_moved_per_env_step": rec_bytes + state_bytes / KT,
                    "achieved_GBs_of_its_own_bytes": medt["rate"] / world * (rec_bytes + state_bytes / KT) / 1e9,
                    "note": "jss_trajectory: policy + step x K per launch with the observation, mask, action, reward and done of "
                            "EVERY step written out ([K][B] buffers); roofline_frac uses the same algorithmic bytes per env step "
                            "as the step-per-launch figure next to it (SURVEY 8(d)), the last two fields its own byte count"}
        except Exception as exc:
            torch.cuda.synchronize()
            return {"value": None, "error": f"{type(exc).__name__}: {exc}"}

    def step_only_measure(env, policy, alg, B, light=False):
        """The boundary entry point itself: jss_step with the actions already resident in HBM, ONE launch per env step,
        next-step auto-reset folded into the action codes.  The actions are a recorded behaviour trajectory (jss_trajectory
        from a snapshot of the state, restored before every window), so every launch executes real, legal steps.  Eager
        ctypes launches and a hipGraph replay of the same K launches; the better one is reported.  (light: the default
        run's leg -- windows of exactly K steps like every figure of the line, 50 ms of them per form.)"""
        try:
            n2 = args.steps if light else max(20, min(100, args.steps))
            secs = MIN_TIMED_SECONDS_LIGHT if light else MIN_TIMED_SECONDS
            env.zero_counters()                          # (the counters live in the arena: the snapshot holds zeros)
            snap = env._arena.clone(), env.solution.clone()
            acts = env.trajectory(policy, steps=n2, record=("action",))["action"]

            def restore():
                env._arena.copy_(snap[0])
                env.solution.copy_(snap[1])

            def replay_steps(n):
                for k in range(n):
                    env.step(acts[k])
            meds, rowss = measure(env, policy, n2, "eager", run=replay_steps, prep=restore, min_seconds=secs)
            restore()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            gs = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(gs, stream=side):
                    replay_steps(n2)
            torch.cuda.current_stream(dev).wait_stream(side)
            medg, rowsg = measure(env, policy, n2, "eager", run=lambda n: gs.replay(), prep=restore, min_seconds=secs)
            del gs
            best, rws, how = (medg, rowsg, "hipGraph replay") if medg["rate"] > meds["rate"] else (meds, rowss, "eager ctypes launches")
            rfs = roofline(best, alg, n2, env, None, B)
            restore()
            del snap
            return {"value": best["rate"], "unit": "env steps/s", "ms_per_step": best["seconds"] / n2 * 1e3,
                    "launch": f"jss_step(actions resident in HBM), one launch per env step, {how}",
                    "roofline_frac": rfs["frac"], "roofline_frac_gpu_time": rfs["frac_gpu_time"],
                    "windows": window_stats(rws, n2), "eager": meds["rate"], "graph": medg["rate"],
                    "note": "the entry point an RL trainer with its own policy network calls; the policy's cost is not in it"}
        except Exception as exc:
            torch.cuda.synchronize()
            return {"value": None, "error": f"{type(exc).__name__}: {exc}"}

    def unfused_pipelined(env, policy, alg, B):
        """The un-fused loop with the policy a launch of its own (jss_policy stands in for a policy network), the actions
        through HBM and the auto-reset folded into the step kernel -- issued by the library over 2 or 3 sub-batches on as
        many streams, so that one sub-batch's policy overlaps another's step (jss_policy_step_steps): what a learner that
        double-buffers its env batch gets per env step."""
        try:
            n2 = max(20, min(100, args.steps))
            best = None
            for n_sub in (1, 2, 3):
                def run(n, n_sub=n_sub):
                    env.policy_step_steps(policy, steps=n, n_sub=n_sub, autoreset=True)
                med, rows = measure(env, policy, n2, "eager", run=run)
                if best is None or med["rate"] > best[0]["rate"]:
                    best = (med, rows, n_sub)
            med, rows, n_sub = best
            return {"value": med["rate"], "unit": "env steps/s", "ms_per_step": med["seconds"] / n2 * 1e3, "n_sub": n_sub,
                    "roofline_frac": roofline(med, alg, n2, env, None, B)["frac"], "windows": window_stats(rows, n2),
                    "note": "jss_policy -> actions in HBM -> jss_step_autoreset per sub-batch and step (jss_policy_step_steps), "
                            "sub-batches on their own streams"}
        except Exception as exc:
            torch.cuda.synchronize()
            return {"value": None, "error": f"{type(exc).__name__}: {exc}"}

    def external_action_forms(env, policy, alg, K):
        """The step entry points that take the caller's actions, with the actions already resident in HBM (a recorded
        behaviour trajectory of `policy` from the current state, consumed window after window -- every launch executes
        real, legal steps and next-step auto-resets): (a) jss_steps, K steps per launch; (b) a step session, K steps
        posted per wait (the resident kernel runs back to back); (c) the same session in lock step, one fused post + wait
        launch per step -- the learner-in-the-loop form.  Windows of K steps between device-wide synchronizes, like
        every other figure; the session is opened once, outside the windows (a learner keeps it open for a whole run)."""
        out = {}
        try:
            B = env.batch
            K = 32                                        # steps per launch / per wait: the trajectory figure's K
            n_win = int(max(6, min(40, 256e6 // (K * B * 4))))
            env.zero_counters()
            snap = env._arena.clone(), env.solution.clone()

            def restore():
                env._arena.copy_(snap[0])
                env.solution.copy_(snap[1])
                torch.cuda.synchronize()

            acts = env.trajectory(policy, steps=(n_win + 1) * K, record=("action",))["action"]
            counts = (acts >= 0).view(n_win + 1, K * B).sum(1).cpu().tolist()
            restore()

            cur = torch.cuda.current_stream(dev)

            def timed(issue):
                # (the CALLER's stream is synchronized, not the device: a device-wide synchronize would wait for the
                # session's resident kernel, which ends when the session is closed)
                rows = []
                for w in range(n_win + 1):
                    cur.synchronize()
                    t0 = time.perf_counter()
                    issue(w)
                    cur.synchronize()
                    rows.append(counts[w] / (time.perf_counter() - t0))
                rows = sorted(rows[1:])                   # window 0 warms the path up
                med = rows[len(rows) // 2]
                return {"value": med, "unit": "env steps/s", "min": rows[0], "max": rows[-1], "windows": len(rows),
                        "steps_each": K, "us_per_step": 1e6 * (sum(counts[1:]) / len(rows)) / med / K,
                        "roofline_frac": med * alg / 1e9 / HBM_PEAK_GBS}

            rec_all = ("real_obs", "action_mask", "reward", "done")
            bufs = env.steps(acts[:K], record=rec_all)
            restore()
            out["steps_per_launch"] = timed(lambda w: env.steps(acts[w * K:(w + 1) * K], record=rec_all, buffers=bufs))
            out["steps_per_launch"]["launch"] = (f"jss_steps: {K} x jss_step per launch, actions resident, state in registers in between, "
                                                 f"EVERY step's real_obs / action_mask / reward / done written ([K][B] buffers)")
            del bufs
            restore()
            out["steps_per_launch_last_obs_only"] = timed(lambda w: env.steps(acts[w * K:(w + 1) * K]))
            out["steps_per_launch_last_obs_only"]["launch"] = (f"jss_steps: {K} x jss_step per launch, reward / done per step, observation and "
                                                               f"mask of the last step only")
            restore()
            try:
                opened = env.session(depth=K)
            except RuntimeError as exc:            # JSS_E_RESIDENT: the batch does not fit the chip as one round of resident workgroups
                out["session_posted_ahead"] = out["session_lockstep"] = {"value": None, "note": f"no session for this batch: {exc}"[:200]}
                restore()
                del snap, acts
                return out
            with opened as sess:
                out["session_posted_ahead"] = timed(lambda w: (sess.post(acts[w * K:(w + 1) * K]), sess.wait()))
            st = sess.host_status()
            out["session_posted_ahead"].update(launch=f"step session: {K} steps posted per wait (one post + one wait kernel per window), "
                                                      f"state resident on the chip", env_sets_per_wavefront=st["env_sets_per_wavefront"],
                                               timeouts=st["session_timeouts"] + st["wait_timeouts"])
            restore()

            def lockstep(w):
                for k in range(K):
                    sess.step(acts[w * K + k])
            with env.session(depth=1) as sess:
                out["session_lockstep"] = timed(lockstep)
            st = sess.host_status()
            out["session_lockstep"].update(launch="step session: one fused post + wait launch per step (the caller's policy would run "
                                                  "between two of them; its cost is not in the figure)",
                                           env_sets_per_wavefront=st["env_sets_per_wavefront"],
                                           timeouts=st["session_timeouts"] + st["wait_timeouts"])
            restore()
            del snap, acts
        except Exception as exc:
            out["error"] = f"{type(exc).__name__}: {exc}"
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        return out

    def side_run(workload, batch, policy, label_extra="", instance="ta01", bucketed=False, modes=("eager", "graph", "sub1", "sub2", "sub3"),
                 first_env=None, keep=False, with_trajectory=False, with_external=False, order=None, with_step_only=False):
        """One extra workload on this GPU, same timing discipline as the headline."""
        alg, label, key = describe(workload, instance)
        env = make_env(workload, batch, first_env if first_env is not None else rank * batch, policy, instance=instance,
                       bucketed=bucketed, order=order)
        if getattr(env, "_classes", None) is not None:
            key = "mixed_by_shape"
        for _ in range(args.warmup):
            env.rollout(policy, n_iter=1, autoreset=True)
        mode, med, rows = best_form(env, policy, list(modes), MIN_TIMED_SECONDS)
        rf = roofline(med, alg, args.steps, env, ("mixed_bucketed" if env.launch == "grid" else None) if bucketed else key, batch)
        out = {"workload": label + label_extra, "batch": batch, "policy": policy, "value": med["rate"],
               "min": rows[0]["rate"], "max": rows[-1]["rate"], "windows": len(rows), "unit": "env steps/s",
               "ms_per_step": med["seconds"] / args.steps * 1e3, "roofline_frac_gpu_time": rf["frac_gpu_time"],
               "launch": ((f"one grid over all shape classes per step and part, {getattr(env, 'n_sub', 1)} part(s) per class (jss_multi_rollout)" if args.bucketed_launch == "grid" else
                           "one launch per shape bucket per step, every bucket on its own HIP stream") if bucketed else launch_label(mode)),
               "kernel": rf["kernel"], "roofline_frac": rf["frac"],
               "roofline_frac_of_measured_peak": rf["frac_of_measured_peak"], "alg_bytes_per_env_step": alg,
               "traffic": rf["traffic"], "mean_makespan": med["makespan_sum"] / med["episodes"] if med["episodes"] else None}
        if with_trajectory and not bucketed:
            out["trajectory"] = traj_measure(env, policy, alg)
        if with_external and not bucketed:
            out["policy_then_step_pipelined"] = unfused_pipelined(env, policy, alg, batch)
            out["step_only"] = step_only_measure(env, policy, alg, batch)
            out["external_actions"] = external_action_forms(env, policy, alg, args.steps)
        elif with_step_only and not bucketed and world == 1:
            out["step_only"] = step_only_measure(env, policy, alg, batch, light=True)
        if keep:
            return out, env
        if hasattr(env, "close"):
            env.close()
        del env
        return out

    # ---- the headline ------------------------------------------------------------------------------------------
    alg_per_step, wl_label, key = describe(args.workload, args.instance)
    if args.scaling == "strong":
        lo, hi = shard_bounds(args.batch, world, rank)
        B, first_env = hi - lo, lo
    else:
        B, first_env = args.batch, rank * args.batch
    env = make_env(args.workload, B, first_env, args.policy, instance=args.instance, bucketed=args.bucketed,
                   order="by_shape" if args.by_shape else "interleaved" if args.interleaved else None)
    if getattr(env, "_classes", None) is not None:
        key = "mixed_by_shape"
    for _ in range(args.warmup):
        env.rollout(args.policy, n_iter=1, autoreset=True)
    torch.cuda.synchronize()
    mode, med, rows = best_form(env, args.policy, ["eager", "sub1", "sub2", "sub3"] if getattr(env, "_classes", None) is not None
                                else ["eager", "graph", "sub1", "sub2", "sub3"], MIN_TIMED_SECONDS_HEADLINE)
    # Host side of a window: what the C launch loop costs per launch (no synchronisation inside: the hardware queue holds a
    # whole window).  With N ranks on one host this is what must stay below the kernel time per launch -- MAX over ranks.
    host_issue_us = None
    if hasattr(env, "bind_rollout_steps") and getattr(env.backend, "name", "") == "hip":
        n_sub_i = int(mode[3:]) if mode.startswith("sub") else 1
        n_issue = max(args.steps, 40)          # (a handful of launches per call would measure the call, not the launches)
        issue = env.bind_rollout_steps(args.policy, steps=n_issue, n_sub=n_sub_i, autoreset=True, caller_orders_streams=True)
        best = float("inf")
        for _ in range(11):
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            issue()
            best = min(best, time.perf_counter() - t0)
            torch.cuda.synchronize()
        host_issue_us = agree_max([best / (n_issue * n_sub_i) * 1e6])[0]
    bucket_note = ", shape-bucketed (no padding)" if (args.bucketed and args.workload == "mixed") else \
        (", padded 100x20, envs ordered by shape class" if (args.workload == "mixed" and getattr(env, "_classes", None) is not None) else
         ", padded 100x20, env i <- ta(1 + i % 80)" if args.workload == "mixed" else "")
    inst0 = builtin_instance(args.instance)
    out = {
        "metric": "env steps/sec (batched)", "value": med["rate"], "unit": "env steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": med["seconds"] / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int32",
        "data": ("ta01 (the reference's Taillard instance; no dataset involved)" if args.workload == "shared" and args.instance == "ta01"
                 else "synthetic" if args.workload.startswith("synthetic") else "reference instances (ta01-ta80)"),
        "windows": window_stats(rows, args.steps),
        "launch": ((f"one grid over all shape classes per step and part, {getattr(env, 'n_sub', 1)} part(s) per class (jss_multi_rollout, C launch loop)" if args.bucketed_launch == "grid" else
                    "one launch per shape bucket per step, every bucket on its own HIP stream (C launch loop)")
                   if hasattr(env, "buckets") else launch_label(mode)),
        "config": {"workload": f"{wl_label}{bucket_note}, {args.policy} masked policy fused with step(), "
                               f"{'batch %d envs per GPU' % B if args.scaling == 'weak' else 'batch %d envs in total (%d on this rank)' % (args.batch, B)}, "
                               f"full obs/mask/reward/done written every step, auto-restart",
                   "batch_per_gpu": B, "global_batch": args.batch * world if args.scaling == "weak" else args.batch,
                   "parallelism": f"env-shard x{world}", "policy": args.policy},
        "roofline": roofline(med, alg_per_step, args.steps, env,
                             ("mixed_bucketed" if env.launch == "grid" else None) if hasattr(env, "buckets") else key, B),
        "host_issue_us_per_launch": host_issue_us,
        # per-rank view of the median window (a straggler GPU shows as value_min well under value / N)
        "ranks": {"value_min": med["rank_rate_min"], "value_max": med["rank_rate_max"], "numa_pinned": numa.get("pinned"),
                  "numa_node_rank0": numa.get("numa_node"), "cpus_rank0": numa.get("cpus")} if world > 1 else None,
        "episodes_finished": med["episodes"],
        "mean_makespan": med["makespan_sum"] / med["episodes"] if med["episodes"] else None,
        "mean_reward_per_step": (med["reward_num"] / inst0.max_time_op / med["steps"]) if (med["steps"] and args.workload == "shared") else None,
    }

    if not args.no_extras and not hasattr(env, "buckets"):
        if mode not in ("eager", "graph", "sub1"):
            # the plain form: ONE launch per step over the whole batch (the kernel duration rocprofv3 reports)
            m1 = pick_mode(env, args.policy, ["eager", "graph", "sub1"]) if args.launch == "auto" else "eager"
            med1, rows1 = measure(env, args.policy, args.steps, m1)
            out["single_launch_per_step"] = {"value": med1["rate"], "min": rows1[0]["rate"], "max": rows1[-1]["rate"],
                                             "gpu_ms_per_step_events": med1["gpu_ms_per_step_events"], "launch": launch_label(m1),
                                             "roofline_frac": roofline(med1, alg_per_step, args.steps, env, key, B)["frac"],
                                             "roofline_frac_gpu_time": roofline(med1, alg_per_step, args.steps, env, key, B)["frac_gpu_time"]}
    if not args.no_extras and not args.extras and world == 1 and not hasattr(env, "buckets"):
        out["step_only"] = step_only_measure(env, args.policy, alg_per_step, B, light=True)
    if args.extras and not hasattr(env, "buckets"):
        # fused multi-step rollout: 64 iterations per launch, state in registers, outputs once per launch
        n_l = max(4, args.steps // 16)
        medf, _ = measure(env, args.policy, n_l, "eager", n_iter=64, windows=3)
        out["fused_rollout"] = {"value": medf["rate"], "unit": "env steps/s", "iterations_per_launch": 64, "launches": n_l,
                                "note": "policy+step x64 per launch, observation written once per launch"}
    if args.extras and world == 1 and not hasattr(env, "buckets"):
        out["step_only"] = step_only_measure(env, args.policy, alg_per_step, B)
        out["trajectory"] = traj_measure(env, args.policy, alg_per_step)
        out["external_actions"] = external_action_forms(env, args.policy, alg_per_step, args.steps)
        # the un-fused path: jss_policy (stand-in for a policy network) then jss_step_autoreset(actions) -- two launches per
        # env step, hipGraph replay
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            g2 = torch.cuda.CUDAGraph()
            n2 = max(20, min(100, args.steps))
            with torch.cuda.stream(side):
                with torch.cuda.graph(g2, stream=side):
                    for _ in range(n2):
                        env.step(env.policy(args.policy), autoreset=True)
            torch.cuda.current_stream(dev).wait_stream(side)
            med2, rows2 = measure(env, args.policy, n2, "eager", run=lambda n: g2.replay())
            del g2
            out["policy_then_step_two_launches"] = {"value": med2["rate"], "unit": "env steps/s", "iterations": n2,
                                                    "windows": window_stats(rows2, n2),
                                                    "roofline_frac": roofline(med2, alg_per_step, n2, env, None, B)["frac"],
                                                    "note": "jss_policy + jss_step_autoreset per env step (two launches, actions "
                                                            "through HBM; the auto-reset is folded into the step kernel), hipGraph replay"}
        except Exception as exc:
            out["policy_then_step_two_launches"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.synchronize()
        out["policy_then_step_pipelined"] = unfused_pipelined(env, args.policy, alg_per_step, B)
        try:
            # ---- the B = 1 drop-in facade (make('jss-v1')): what a user who only swaps the package gets per step()
            from jssenv_amd import make
            from jssenv_amd.dispatching import get_rule
            f = make("jss-v1", env_config={"instance_path": args.instance}, device=dev)
            f.reset()
            rule = get_rule("FIFO")
            done, n_f = False, 0
            acts_f = []
            while not done:                              # a FIFO episode, recorded ...
                a = rule(f)
                acts_f.append(a)
                _, _, done, _, _ = f.step(a)
            f.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a in acts_f:                             # ... and replayed: step() alone, host -> device -> host every call
                f.step(a)
            dt_f = time.perf_counter() - t0
            t0 = time.perf_counter()
            tot_r, mk_f = rule.run_episode(f, device_rng=True, seed=1)
            dt_fused = time.perf_counter() - t0
            out["facade_b1"] = {"us_per_step": dt_f / len(acts_f) * 1e6, "steps": len(acts_f), "instance": args.instance,
                                "reference_us_per_step_build_container": 79.0,
                                "fused_rule_episode_ms": dt_fused * 1e3, "fused_rule_makespan": mk_f,
                                "note": "JssEnv.step(a) on the GPU: the env's arena lives in page-locked host memory the kernel works "
                                        "on in place -- action written, one launch, one stream synchronise, no copy; fused = "
                                        "DispatchingRule.run_episode(env, device_rng=True), whole episode on the device"}
        except Exception as exc:
            out["facade_b1"] = {"us_per_step": None, "error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.synchronize()
    if not args.no_extras and world == 1 and args.workload == "shared" and args.scaling == "weak":
        if hasattr(env, "close"):
            env.close()
        del env
        env = None
        x = bool(args.extras)                 # trajectory mode and the external-action forms of a config: --extras only
        so = dict(with_step_only=not x)       # jss_step with the caller's actions next to every fused figure (light leg; --extras: the full one)
        extras = [
            ("config2_ta01_batch4096_random", dict(workload="shared", batch=4096, policy="random", modes=("eager", "graph", "sub1"),
                                                   with_trajectory=x, with_external=x, **so)),
            ("config3_ta41_spt_batch16384", dict(workload="shared", batch=16384, policy="SPT", instance="ta41", with_trajectory=x,
                                                 with_external=x, **so)),
            ("config4_synthetic50x20_batch8192", dict(workload="synthetic50x20", batch=8192, policy="random", with_trajectory=x,
                                                      with_external=x, **so)),
            ("config4_synthetic50x20_batch65536_one_gpu", dict(workload="synthetic50x20", batch=65536, policy="random",
                                                               label_extra=" -- all of config 4 on one GPU")),
            # config 5 as a caller gets it: BatchedJssEnv([ta01 .. ta80], batch=32768) -- the constructor deals a ragged list out by
            # shape class (order=None) -- and, next to it, the i % 80 order of rounds 1-5 (every env on the padded extents' kernel)
            ("config5_mixed_padded_batch32768", dict(workload="mixed", batch=32768, policy="random", modes=("eager", "sub1", "sub2", "sub3"),
                                                     label_extra=", padded 100x20, default constructor (envs dealt out by shape class: "
                                                                 "class-specialised bodies on the padded rows)",
                                                     with_trajectory=x, with_external=x, **so)),
            ("config5_mixed_padded_interleaved_batch32768", dict(workload="mixed", batch=32768, policy="random", order="interleaved",
                                                                 label_extra=", padded 100x20, order='interleaved' (env i <- ta(1 + i % 80))")),
            ("config5_mixed_bucketed_batch32768", dict(workload="mixed", batch=32768, policy="random", bucketed=True,
                                                       label_extra=", shape-bucketed (no padding)")),
            # north_star's "synthetic Taillard-shaped instances ... ta01-shape (15x15) at batch 65 536": one instance PER ENV (the
            # headline shares ta01's table, staged in LDS, with 16-byte records; this one reads per-env tables, 24-byte records)
            ("synthetic15x15_per_env_tables", dict(workload="synthetic15x15", batch=B, policy=args.policy, with_external=x, **so)),
        ]
        if x:
            extras += [
                ("batch_x4", dict(workload="shared", batch=4 * B, policy=args.policy, instance=args.instance, modes=("eager", "sub2", "sub3"),
                                  label_extra=" -- 4x the batch (about 200 MB of state and outputs with compact records, plus the 236 MB solution tensor written one word per env step)")),
            ]
        for name, kw in extras:
            try:     # an extra that fails (memory on a busy box, ...) is reported, it does not cost the headline
                out[name] = side_run(kw.pop("workload"), kw.pop("batch"), kw.pop("policy"), **kw)
            except Exception as exc:
                out[name] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
                torch.cuda.synchronize()

    if world > 1 and not args.no_extras and args.workload == "shared" and args.scaling == "weak":
        # BASELINE config 4 as it is defined: synthetic 50x20, 65 536 envs sharded over the N GPUs (strong scaling)
        if env is not None and hasattr(env, "close"):
            env.close()
        del env
        env = None
        try:
            lo4, hi4 = shard_bounds(65536, world, rank)
            c4 = side_run("synthetic50x20", hi4 - lo4, "random", first_env=lo4,
                          label_extra=f" -- BASELINE config 4: 65536 envs sharded over {world} GPUs ({hi4 - lo4} on rank {rank})")
            c4["global_batch"], c4["scaling"], c4["n_gpus"] = 65536, "strong", world
            c4["roofline_frac"] = c4["value"] / world * b_alg(50, 20) / 1e9 / HBM_PEAK_GBS
            out["config4_sharded"] = c4
        except Exception as exc:
            out["config4_sharded"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.synchronize()

    if use_pg:
        out["process_group"] = {"backend": backend, "world_size": dist.get_world_size(), "forced_at_world_1": world == 1}
    out["host"] = {"hsa_enable_interrupt": os.environ.get("HSA_ENABLE_INTERRUPT"),
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), **host_cores()}
    out["csrc_sha16"] = csrc_hash()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "shared":
        legs = [("cpu_baseline", cpu_baseline_restatement)]
        if args.extras:
            legs += [("cpu_baseline_c_oracle", cpu_baseline_port), ("cpu_baseline_twin", cpu_baseline_twin)]
        for name, fn in legs:
            try:     # a host-side hiccup (no compiler for a stale checker build, ...) must not cost the GPU measurement
                out[name] = fn(args.instance, args.seed)
                out[name]["host"] = host_cores()
            except Exception as exc:
                out[name] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
        if args.extras:
            try:     # the other BASELINE shapes at the reference's speed: ta41 (config 3), synthetic 50x20 (config 4), ta80 (config 5's largest)
                from jssenv_amd.instances import taillard_instance
                per = {}
                for label, name, inst in (("config3_ta41", "ta41", None), ("config4_synthetic50x20", "synthetic 50x20 (time seed 1, machine seed 2)",
                                                                            taillard_instance(50, 20, 1, 2)), ("config5_ta80", "ta80", None)):
                    r = cpu_baseline_restatement(name, args.seed, target_seconds=4.0, instance=inst)
                    per[label] = {"value": r["value"], "unit": r["unit"], "cores": 1, "kind": r["kind"], "sample": r["sample"]}
                out["cpu_baseline_per_config"] = per
            except Exception as exc:
                out["cpu_baseline_per_config"] = {"error": f"{type(exc).__name__}: {exc}"}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        # everything, unabridged, to the detail file(s); ONE compact line -- the last line of stdout -- for the driver
        detail = json.dumps(out)
        targets = [args.detail or os.path.join(ROOT, "bench_detail.json")]
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and not args.detail:
            targets.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
        written = []
        for t in targets:
            try:
                with open(t, "w") as fh:
                    fh.write(detail + "\n")
                written.append(os.path.relpath(t, ROOT))
            except OSError as exc:
                print(f"bench.py: could not write {t}: {exc}", file=sys.stderr)
        print(compact_line(out, detail_files=written), flush=True)
    # orderly teardown: graphs were local to measure(); drop the envs (and their side streams) while the
    # runtime is still fully alive
    torch.cuda.synchronize()
    if env is not None and hasattr(env, "close"):
        env.close()
    del env
    gc.collect()
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
