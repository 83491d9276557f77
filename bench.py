#!/usr/bin/env python
"""bench.py -- env steps/sec of the batched JSS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over the batch: ONE launch of the fused
policy+step kernel (jss_rollout with n_iter = 1) that, for every env, picks a random
masked action on the device, executes step() and writes the full gym outputs
(real_obs, action_mask, reward, done) to HBM -- exactly what a reference
``obs, r, done, _, _ = env.step(policy(obs))`` iteration produces.  Envs found done
are reset by that launch instead (the iteration is not counted as an env step).
value = env steps executed by all ranks / max-over-ranks wall time of the K launches.

Workload: BASELINE.json configs[1] shape (ta01, 15x15, one shared instance, random
masked policy) at the north_star's target batch of 65 536 envs per GPU (weak scaling:
every rank owns its own 65 536 envs, no data-path collective; one RCCL all-reduce of
the counters after the timed region).  The configs[1] batch of 4 096 and the fused
multi-step rollout (state kept in registers for 64 iterations) are measured after the
timed region and reported as extra fields.

Extra objects: roofline (HBM; algorithmic bytes per launch / HIP-event kernel time) and
cpu_baseline (the C oracle of oracle/, timed on this box's host cores, rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def b_alg(J, M):
    """Algorithmic bytes of one env step: read + write the per-env state
    (7 int32/job, 2 flag bytes/job, int32 + flag per machine, clock + counters), the action,
    one solution entry, the float32 observation, the mask, reward and done."""
    return 89 * J + 10 * M + 40


def cpu_baseline(inst_name, seed, target_seconds=10.0):
    """The C oracle (a scalar restatement of the reference's step(), oracle/jss_oracle.c) running
    the same policy+step loop on this box's host cores, one env per thread."""
    import concurrent.futures as cf
    from jssenv_amd import builtin_instance
    from oracle import OracleEnv
    inst = builtin_instance(inst_name)
    threads = max(1, min(os.cpu_count() or 1, 64))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--instance", default="ta01")
    ap.add_argument("--policy", default="random")
    ap.add_argument("--workload", default="shared", choices=["shared", "synthetic50x20", "mixed"],
                    help="shared: one instance (--instance) for the whole batch [default, the headline]; "
                         "synthetic50x20: BASELINE config 4, one Taillard-LCG instance per env; "
                         "mixed: BASELINE config 5, env i <- ta(1 + i %% 80), padded 100x20")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager"],
                    help="graph: replay a hipGraph of the K launches; eager: one Python/ctypes launch per step; "
                         "auto: whichever is faster on a short probe (graphs win when the kernel is shorter than "
                         "the ~7 us host enqueue, eager wins at large batches)")
    ap.add_argument("--bucketed", action="store_true",
                    help="mixed workload only: one compact sub-batch per shape class (BucketedJssEnv) instead of "
                         "padding every env to 100x20")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="debug: every rank uses cuda:0 (exercises the multi-process path on a 1-GPU box; use with "
                         "--dist-backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from jssenv_amd import BatchedJssEnv, builtin_instance
    from jssenv_amd.distributed import reduce_counters

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = args.dist_backend or "nccl"   # "nccl" is RCCL on ROCm
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    def barrier():
        if world > 1:
            dist.barrier()

    inst = builtin_instance(args.instance)
    B = args.batch

    def instances_for(batch):
        """(instances, mean algorithmic bytes per env step, label)."""
        if args.workload == "synthetic50x20":
            from jssenv_amd import synthetic_batch
            return synthetic_batch(batch, 50, 20, first=rank * batch), b_alg(50, 20), "synthetic 50x20 (Taillard LCG), one instance per env"
        if args.workload == "mixed":
            insts = [builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
            mean = sum(b_alg(i.jobs, i.machines) for i in insts) / 80.0
            return insts, mean, "mixed ta01-ta80 (env i <- ta(1 + i % 80)), padded 100x20"
        return inst, b_alg(inst.jobs, inst.machines), f"{args.instance} ({inst.jobs}x{inst.machines}) shared instance"

    def make_env(batch):
        insts, _, _ = instances_for(batch)
        if args.bucketed and args.workload == "mixed":
            from jssenv_amd import BucketedJssEnv
            e = BucketedJssEnv(insts, batch=batch, device=dev, seed=args.seed, env_id_base=rank * batch)
            e.reset()
            e.rollout(args.policy, n_iter=333, autoreset=True)
            e.zero_counters()
            return e
        e = BatchedJssEnv(insts, batch=batch, device=dev, seed=args.seed, env_id_base=rank * batch)
        e.reset()
        # Spread the episode phases (a fresh batch is in lock step: every env at step 0) so the timed
        # window sees the steady-state mix of episode stages: env i is advanced (i % 16) * 16 extra
        # steps through the separate policy + step kernels, skipping (-1) the envs that are ahead.
        ids = torch.arange(batch, device=dev) % 16
        for r in range(15):
            for _ in range(16):
                a = e.policy(args.policy)
                a = torch.where(ids > r, a, torch.full_like(a, -1))
                e.step(a)
        e.rollout(args.policy, n_iter=64, autoreset=True)
        e.zero_counters()
        return e

    def timed(env, n_launch, n_iter, mode):
        """Time n_launch launches of jss_rollout(n_iter).  Returns (max-over-ranks-able wall seconds,
        GPU ms per launch from HIP events on the launch stream)."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        graph = None
        bucketed = hasattr(env, "rollout_steps")
        if mode == "graph" and not bucketed:
            # the launches go to torch's current stream, so a torch CUDAGraph captures them: one host call
            # replays all n_launch kernels (the Python+ctypes enqueue costs ~7 us per launch otherwise)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(n_launch):
                        env.rollout(args.policy, n_iter=n_iter, autoreset=True)
            torch.cuda.current_stream(dev).wait_stream(side)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        if graph is not None:
            graph.replay()
        elif bucketed:   # one fork/join around the window, every bucket's launches on its own stream
            env.rollout_steps(args.policy, steps=n_launch, n_iter=n_iter, autoreset=True)
        else:
            for _ in range(n_launch):
                env.rollout(args.policy, n_iter=n_iter, autoreset=True)
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        return dt, ev0.elapsed_time(ev1) / n_launch

    env = make_env(B)
    for _ in range(args.warmup):
        env.rollout(args.policy, n_iter=1, autoreset=True)
    torch.cuda.synchronize()

    def pick_mode(e):
        if hasattr(e, "rollout_steps"):
            return "eager"
        if args.launch != "auto":
            return args.launch
        probe = {m: timed(e, 40, 1, m)[0] for m in ("graph", "eager")}
        t = torch.tensor([probe["graph"], probe["eager"]], dtype=torch.float64)
        if world > 1:   # every rank must take the same decision
            t = t.to(dev) if backend == "nccl" else t
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return "graph" if float(t[0]) <= float(t[1]) else "eager"

    mode = pick_mode(env)
    env.zero_counters()
    dt, kernel_ms = timed(env, args.steps, 1, mode)
    # the only collectives: SUM of the 4 counters and MAX of the wall time, over RCCL/xGMI
    on_host = world > 1 and backend != "nccl"
    tot = reduce_counters(env.counter_totals().cpu() if on_host else env.counter_totals(), dt)
    steps_total, episodes, makespan_sum, reward_num = tot["steps"], tot["episodes"], tot["makespan_sum"], tot["reward_num_sum"]
    dt_max = tot["seconds"]
    value = steps_total / dt_max

    # roofline of the dominant (only) kernel: algorithmic bytes per launch / HIP-event time per launch
    stepped_per_launch = steps_total / world / args.steps
    _, alg_per_step, wl_label = instances_for(1 if args.workload != "mixed" else 80)
    alg_bytes = stepped_per_launch * alg_per_step
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    jm, mm = env.jmax, env.mmax
    kernel_name = ("jss_packed_kernel<%d,kRollout1>" % (16 if max(jm, mm) <= 16 else 32)
                   if max(jm, mm) <= 32 else "jss_kernel<%d,kRollout1>" % (1 if jm <= 64 else 2))
    if args.bucketed and args.workload == "mixed":
        kernel_name = "four launches per step: jss_packed_kernel<16|32,kRollout1>, jss_kernel<1|2,kRollout1>"
        wl_label += ", shape-bucketed (no padding)"
    traffic = None
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.isfile(prof):
        try:
            with open(prof) as fh:
                traffic = json.load(fh).get(f"{args.instance}_b{B}", {}).get("bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "env steps/sec (batched)", "value": value, "unit": "env steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "launch": ("per-bucket streams, eager launches, one fork/join around the K steps" if hasattr(env, "rollout_steps")
                   else "eager (one ctypes launch per step)" if mode == "eager" else "hipGraph replay of the K launches"),
        "config": {"workload": f"{wl_label}, {args.policy} masked "
                               f"policy fused with step(), batch {B} envs per GPU, one launch per env step, "
                               f"full obs/mask/reward/done written every step, auto-restart",
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"env-shard x{world}",
                   "policy": args.policy},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": kernel_name, "kernel_ms": kernel_ms,
                     "alg_bytes_per_env_step": alg_per_step,
                     "env_steps_per_launch": stepped_per_launch},
        "episodes_finished": episodes,
        "mean_makespan": makespan_sum / episodes if episodes else None,
        "mean_reward_per_step": reward_num / inst.max_time_op / steps_total if steps_total else None,
    }

    if not args.no_extras:
        # fused multi-step rollout: 64 iterations per launch, state in registers, outputs once per launch
        env.zero_counters()
        n_l = max(4, args.steps // 16)
        dtf, _ = timed(env, n_l, 64, "eager")
        totf = reduce_counters(env.counter_totals().cpu() if on_host else env.counter_totals(), dtf)
        out["fused_rollout"] = {"value": totf["steps_per_second"], "unit": "env steps/s",
                                "iterations_per_launch": 64, "launches": n_l,
                                "note": "policy+step x64 per launch, observation written once per launch"}
        if world == 1 and args.workload == "shared":
            # twice the contract batch: same kernel, four rounds of resident waves instead of two
            env2x = make_env(2 * B)
            for _ in range(args.warmup):
                env2x.rollout(args.policy, n_iter=1, autoreset=True)
            mode2 = pick_mode(env2x)
            env2x.zero_counters()
            dt2, ms2 = timed(env2x, args.steps, 1, mode2)
            steps2 = float(env2x.counter_totals()[0].item())
            out["batch_x2"] = {"batch": 2 * B, "value": steps2 / dt2, "unit": "env steps/s", "kernel_ms": ms2,
                               "roofline_frac": steps2 / args.steps * alg_per_step / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "launch": mode2}
            del env2x
            env4k = make_env(4096)
            for _ in range(args.warmup):
                env4k.rollout(args.policy, n_iter=1, autoreset=True)
            mode4 = pick_mode(env4k)
            env4k.counters.zero_()
            dt4, ms4 = timed(env4k, args.steps, 1, mode4)
            out["configs1_batch4096"] = {"value": float(env4k.counters[:, 0].sum().item()) / dt4,
                                         "unit": "env steps/s", "kernel_ms": ms4, "launch": mode4}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "shared":
        out["cpu_baseline"] = cpu_baseline(args.instance, args.seed)
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    # orderly teardown: graphs were local to timed(); drop the envs (and the bucketed env's side streams)
    # while the runtime is still fully alive
    torch.cuda.synchronize()
    if hasattr(env, "close"):
        env.close()
    del env
    import gc
    gc.collect()
    torch.cuda.synchronize()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
