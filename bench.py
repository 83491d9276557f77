#!/usr/bin/env python
"""bench.py -- env steps/sec of the batched JSS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-executes itself under torch.distributed.run,
                                                            one rank per GPU, RCCL; fails loudly with fewer GPUs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the batch: for every env, pick a random masked action on the
device, execute step() and write the full gym outputs (real_obs, action_mask, reward, done) to HBM -- exactly
what a reference ``obs, r, done, _, _ = env.step(policy(obs))`` iteration produces, fused in one kernel
(jss_rollout with n_iter = 1).  Envs found done are reset by that pass instead (the iteration is not counted as
an env step).  A pass is ONE launch over the batch, or -- whichever a short probe finds faster -- n_sub launches
over n_sub contiguous sub-batches on n_sub HIP streams (jss_rollout_steps): step s of a sub-batch depends only on
its own step s-1, so one sub-batch's drain overlaps another's fill; results are identical either way.

Timing: W untimed warm-up steps, then >= 5 windows of EXACTLY K steps, each bracketed by barrier +
torch.cuda.synchronize() on both sides; per window the wall time is the MAX over ranks and the env steps the SUM
over ranks (one RCCL all-reduce each, outside the timed region).  value = median window; min/max are printed too.

Workload: BASELINE.json configs[1] shape (ta01, 15x15, one shared instance, random masked policy) at the
north_star's target batch of 65 536 envs per GPU (weak scaling: every rank owns its own 65 536 envs, no data-path
collective).  Extras on the same line (N = 1): the plain one-launch-per-step figure, 4x the batch (beyond the
256 MB Infinity Cache), per-env synthetic 15x15 tables, BASELINE configs 2-5, the fused 64-step rollout.

Extra objects: roofline (HBM; algorithmic bytes per step / HIP-event time per step), cpu_baseline (the C oracle
of oracle/, kind "port") and cpu_baseline_twin (libjss_cpu.so, 1 core and all cores), timed on this box's host cores,
rank 0, N = 1.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_MEASURED_PEAK_GBS = 6290.0  # measured copy bandwidth, same guide (SURVEY.md 8(d) asks for both)
N_WINDOWS = 5


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
    return {"value": steps / dt, "unit": "env steps/s", "cores": threads, "kind": "port",
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
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager", "sub2", "sub3", "sub4"],
                    help="how a step is issued: eager = one ctypes launch; graph = hipGraph replay of the K launches; "
                         "subN = N sub-batches on N streams (jss_rollout_steps); auto = fastest on a probe")
    ap.add_argument("--bucketed", action="store_true",
                    help="mixed workload only: one compact sub-batch per shape class (BucketedJssEnv) instead of "
                         "padding every env to 100x20")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="debug: every rank uses cuda:0 (exercises the multi-process path on a 1-GPU box; use with "
                         "--dist-backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
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
    from jssenv_amd.distributed import reduce_counters, shard_bounds
    from jssenv_amd.instances import synthetic_packed

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the GPU path has no CPU fallback")
    if args.share_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = args.dist_backend or "nccl"   # "nccl" is RCCL on ROCm
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        assert dist.get_world_size() == world
    on_host = world > 1 and backend != "nccl"

    def barrier():
        if world > 1:
            dist.barrier()

    def agree_max(values):
        """element-wise MAX over ranks of a list of floats (every rank must take the same decisions)"""
        t = torch.tensor(values, dtype=torch.float64)
        if world > 1:
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

    def make_env(workload, batch, first_env, policy, instance="ta01", bucketed=False, spread=True):
        if workload == "mixed" and bucketed:
            from jssenv_amd import BucketedJssEnv
            insts = [builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
            e = BucketedJssEnv(insts, batch=batch, device=dev, seed=args.seed, env_id_base=first_env)
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
        e = BatchedJssEnv(src, batch=batch, device=dev, seed=args.seed, env_id_base=first_env)
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
    def window(env, policy, n_launch, n_iter, mode, graph=None):
        """Time n_launch steps.  Returns (wall seconds, GPU ms per step from HIP events on the launch stream)."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        if graph is not None:
            graph.replay()
        elif hasattr(env, "rollout_steps") and mode.startswith("sub"):
            env.rollout_steps(policy, steps=n_launch, n_sub=int(mode[3:]), autoreset=True)
        elif hasattr(env, "buckets"):   # one fork/join around the window, every bucket's launches on its own stream
            env.rollout_steps(policy, steps=n_launch, n_iter=n_iter, autoreset=True)
        else:
            for _ in range(n_launch):
                env.rollout(policy, n_iter=n_iter, autoreset=True)
        ev1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0          # this rank's time; the job's time is the MAX over ranks (reduce_counters)
        barrier()
        return dt, ev0.elapsed_time(ev1) / n_launch

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
        return candidates[probe.index(min(probe))]

    def measure(env, policy, steps, mode, n_iter=1, windows=N_WINDOWS):
        """`windows` windows of `steps` steps each.  Returns the per-window lists, already reduced over ranks."""
        graph = capture(env, policy, steps) if mode == "graph" else None
        window(env, policy, steps, n_iter, mode, graph)     # untimed: side streams / graph exist before the first window
        rows = []
        for _ in range(windows):
            env.zero_counters()
            dt, ms = window(env, policy, steps, n_iter, mode, graph)
            tot = reduce_counters(env.counter_totals().cpu() if on_host else env.counter_totals(), dt)
            ms = agree_max([ms])[0]
            rows.append({"steps": tot["steps"], "seconds": tot["seconds"], "rate": tot["steps"] / tot["seconds"],
                         "kernel_ms": ms, "episodes": tot["episodes"], "makespan_sum": tot["makespan_sum"],
                         "reward_num": tot["reward_num_sum"]})
        del graph
        rows.sort(key=lambda r: r["rate"])
        med = rows[len(rows) // 2]
        return med, rows

    def launch_label(mode):
        if mode.startswith("sub"):
            return f"{mode[3:]} sub-batches on {mode[3:]} HIP streams per step (jss_rollout_steps)"
        return {"eager": "one launch per step (ctypes, eager)",
                "graph": "one launch per step, hipGraph replay of the K launches"}.get(mode, mode)

    def kernel_name(env):
        if hasattr(env, "buckets"):
            return "four launches per step: jss_packed_kernel<16|32,kRollout1,*>, jss_kernel<1|2,kRollout1,*>"
        tab = "kTabLds" if env.n_tables == 1 else "kTabGlobal"
        jm, mm = env.jmax, env.mmax
        if max(jm, mm) <= 32:
            return f"jss_packed_kernel<{16 if max(jm, mm) <= 16 else 32},kRollout1,{tab}>"
        return f"jss_kernel<{1 if jm <= 64 else 2},kRollout1,{tab}>"

    def static_traffic(key, batch):
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if key is None:
            return None, None
        try:
            with open(prof) as fh:
                ent = json.load(fh).get(f"{key}_b{batch}")
            return (ent["bytes_per_launch"], f"profiles/{ent['source']} (static: rocprofv3 PMC passes, not re-measured in this run)") if ent else (None, None)
        except Exception:
            return None, None

    def roofline(med, alg_per_step, steps, env, key, batch):
        stepped = med["steps"] / world / steps
        achieved = stepped * alg_per_step / (med["kernel_ms"] * 1e-3) / 1e9
        traffic, src = static_traffic(key, batch)
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_peak": achieved / HBM_MEASURED_PEAK_GBS,
                "measured_peak": HBM_MEASURED_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                "kernel": kernel_name(env), "kernel_ms": med["kernel_ms"],
                "alg_bytes_per_env_step": alg_per_step, "env_steps_per_launch": stepped}

    def side_run(workload, batch, policy, label_extra="", instance="ta01", bucketed=False, modes=("eager", "graph", "sub2", "sub3")):
        """One extra workload on this GPU: median of N_WINDOWS windows, same timing discipline as the headline."""
        alg, label, key = describe(workload, instance)
        env = make_env(workload, batch, rank * batch, policy, instance=instance, bucketed=bucketed)
        for _ in range(args.warmup):
            env.rollout(policy, n_iter=1, autoreset=True)
        mode = pick_mode(env, policy, list(modes))
        med, rows = measure(env, policy, args.steps, mode)
        rf = roofline(med, alg, args.steps, env, None if bucketed else key, batch)
        out = {"workload": label + label_extra, "batch": batch, "policy": policy, "value": med["rate"],
               "min": rows[0]["rate"], "max": rows[-1]["rate"], "unit": "env steps/s", "ms_per_step": med["seconds"] / args.steps * 1e3,
               "launch": ("one launch per shape bucket per step, every bucket on its own HIP stream" if bucketed else launch_label(mode)),
               "kernel": rf["kernel"], "roofline_frac": rf["frac"],
               "roofline_frac_of_measured_peak": rf["frac_of_measured_peak"], "alg_bytes_per_env_step": alg,
               "traffic": rf["traffic"], "mean_makespan": med["makespan_sum"] / med["episodes"] if med["episodes"] else None}
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
    env = make_env(args.workload, B, first_env, args.policy, instance=args.instance, bucketed=args.bucketed)
    for _ in range(args.warmup):
        env.rollout(args.policy, n_iter=1, autoreset=True)
    torch.cuda.synchronize()
    mode = pick_mode(env, args.policy, ["eager", "graph", "sub2", "sub3"])
    med, rows = measure(env, args.policy, args.steps, mode)
    bucket_note = ", shape-bucketed (no padding)" if (args.bucketed and args.workload == "mixed") else \
        (", padded 100x20" if args.workload == "mixed" else "")
    inst0 = builtin_instance(args.instance)
    out = {
        "metric": "env steps/sec (batched)", "value": med["rate"], "unit": "env steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": med["seconds"] / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "int32",
        "data": ("ta01 (the reference's Taillard instance; no dataset involved)" if args.workload == "shared" and args.instance == "ta01"
                 else "synthetic" if args.workload.startswith("synthetic") else "reference instances (ta01-ta80)"),
        "windows": {"n": len(rows), "steps_each": args.steps, "statistic": "median", "min": rows[0]["rate"], "max": rows[-1]["rate"],
                    "all": [r["rate"] for r in rows]},
        "launch": ("one launch per shape bucket per step, every bucket on its own HIP stream (C launch loop per bucket)"
                   if hasattr(env, "buckets") else launch_label(mode)),
        "config": {"workload": f"{wl_label}{bucket_note}, {args.policy} masked policy fused with step(), "
                               f"{'batch %d envs per GPU' % B if args.scaling == 'weak' else 'batch %d envs in total (%d on this rank)' % (args.batch, B)}, "
                               f"full obs/mask/reward/done written every step, auto-restart",
                   "batch_per_gpu": B, "global_batch": args.batch * world if args.scaling == "weak" else args.batch,
                   "parallelism": f"env-shard x{world}", "policy": args.policy},
        "roofline": roofline(med, alg_per_step, args.steps, env, None if hasattr(env, "buckets") else key, B),
        "episodes_finished": med["episodes"],
        "mean_makespan": med["makespan_sum"] / med["episodes"] if med["episodes"] else None,
        "mean_reward_per_step": (med["reward_num"] / inst0.max_time_op / med["steps"]) if (med["steps"] and args.workload == "shared") else None,
    }

    if not args.no_extras and not hasattr(env, "buckets"):
        if mode != "eager" and mode != "graph":
            # the plain form: ONE launch per step over the whole batch (the kernel duration rocprofv3 reports)
            m1 = pick_mode(env, args.policy, ["eager", "graph"]) if args.launch == "auto" else "eager"
            med1, rows1 = measure(env, args.policy, args.steps, m1)
            out["single_launch_per_step"] = {"value": med1["rate"], "min": rows1[0]["rate"], "max": rows1[-1]["rate"],
                                             "kernel_ms": med1["kernel_ms"], "launch": launch_label(m1),
                                             "roofline_frac": roofline(med1, alg_per_step, args.steps, env, key, B)["frac"]}
        # fused multi-step rollout: 64 iterations per launch, state in registers, outputs once per launch
        n_l = max(4, args.steps // 16)
        medf, _ = measure(env, args.policy, n_l, "eager", n_iter=64, windows=3)
        out["fused_rollout"] = {"value": medf["rate"], "unit": "env steps/s", "iterations_per_launch": 64, "launches": n_l,
                                "note": "policy+step x64 per launch, observation written once per launch"}
    if not args.no_extras and world == 1 and not hasattr(env, "buckets"):
        # the path an RL trainer with its own policy network uses: jss_policy (stand-in for the network) then
        # jss_step(actions) with next-step auto-reset -- two launches + the action select per env step, hipGraph replay
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        g2 = torch.cuda.CUDAGraph()
        n2 = max(20, min(100, args.steps))
        with torch.cuda.stream(side):
            with torch.cuda.graph(g2, stream=side):
                for _ in range(n2):
                    env.step(env.policy(args.policy), autoreset=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        g2.replay()
        torch.cuda.synchronize()
        rates = []
        for _ in range(3):
            env.zero_counters()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g2.replay()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            rates.append(float(env.counter_totals()[0].item()) / dt2)
        del g2
        out["policy_then_step_two_launches"] = {"value": sorted(rates)[1], "unit": "env steps/s", "iterations": n2,
                                                "note": "jss_policy + jss_step(autoreset) per env step (separate launches, "
                                                        "actions through HBM), hipGraph replay; median of 3"}
    if not args.no_extras and world == 1 and args.workload == "shared" and args.scaling == "weak":
        if hasattr(env, "close"):
            env.close()
        del env
        env = None
        extras = [
            ("batch_x4", dict(workload="shared", batch=4 * B, policy=args.policy, instance=args.instance, modes=("eager", "sub2", "sub3"),
                              label_extra=" -- 4x the batch: 420 MB working set, beyond the Infinity Cache")),
            ("synthetic15x15_per_env_tables", dict(workload="synthetic15x15", batch=B, policy=args.policy)),
            ("config2_ta01_batch4096_random", dict(workload="shared", batch=4096, policy="random", modes=("eager", "graph"))),
            ("config3_ta41_spt_batch16384", dict(workload="shared", batch=16384, policy="SPT", instance="ta41")),
            ("config4_synthetic50x20_batch8192", dict(workload="synthetic50x20", batch=8192, policy="random")),
            ("config4_synthetic50x20_batch65536_one_gpu", dict(workload="synthetic50x20", batch=65536, policy="random",
                                                               label_extra=" -- all of config 4 on one GPU")),
            ("config5_mixed_padded_batch32768", dict(workload="mixed", batch=32768, policy="random", label_extra=", padded 100x20")),
            ("config5_mixed_bucketed_batch32768", dict(workload="mixed", batch=32768, policy="random", bucketed=True,
                                                       label_extra=", shape-bucketed (no padding)")),
        ]
        for name, kw in extras:
            try:     # an extra that fails (memory on a busy box, ...) is reported, it does not cost the headline
                out[name] = side_run(kw.pop("workload"), kw.pop("batch"), kw.pop("policy"), **kw)
            except Exception as exc:
                out[name] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
                torch.cuda.synchronize()

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "shared":
        for name, fn in (("cpu_baseline", cpu_baseline_port), ("cpu_baseline_twin", cpu_baseline_twin)):
            try:     # a host-side hiccup (no compiler for a stale checker build, ...) must not cost the GPU measurement
                out[name] = fn(args.instance, args.seed)
                out[name]["host"] = host_cores()
            except Exception as exc:
                out[name] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    # orderly teardown: graphs were local to measure(); drop the envs (and their side streams) while the
    # runtime is still fully alive
    torch.cuda.synchronize()
    if env is not None and hasattr(env, "close"):
        env.close()
    del env
    import gc
    gc.collect()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
