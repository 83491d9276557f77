"""GPU box: env-steps/s of jss_trajectory (K steps per launch, everything recorded) and of jss_steps (the same K steps replayed
from the recorded actions, everything recorded) on the benchmark workloads -- the A/B harness of the kernels that loop over
steps with the state in registers (JSSENV_AMD_LIB=variants/<x>.so python tools/gpu_traj_probe.py)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv, builtin_instance  # noqa: E402
from jssenv_amd.instances import synthetic_packed  # noqa: E402

K = int(os.environ.get("JSS_KT", "32"))
ONLY = os.environ.get("JSS_PROBE_ONLY", "")                     # e.g. "synthetic50x20"; JSS_PROBE_RECORDS = full | medium for the synthetic ones
for what, B in (("ta01", 65536), ("ta01", 4096), ("synthetic50x20", 8192), ("mixed", 32768), ("synthetic15x15", 65536), ("ta41", 16384)):
    if ONLY and what != ONLY:
        continue
    if what == "mixed":
        env = BatchedJssEnv([builtin_instance(f"ta{k:02d}") for k in range(1, 81)], batch=B, device="cuda:0")
    elif what.startswith("synthetic"):
        J, M = (int(x) for x in what[len("synthetic"):].split("x"))
        env = BatchedJssEnv(synthetic_packed(B, J, M), device="cuda:0", records=os.environ.get("JSS_PROBE_RECORDS") or None)
    else:
        env = BatchedJssEnv(what, batch=B, device="cuda:0")
    env.reset()
    env.rollout("random", n_iter=100)
    bufs = env.trajectory("random", steps=K)
    best = 0.0
    for rep in range(4):
        env.zero_counters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            env.trajectory("random", steps=K, buffers=bufs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, env.stats()["steps"] / dt)
    lib = os.path.basename(os.environ.get('JSSENV_AMD_LIB', 'shipped')) + (" records=" + os.environ["JSS_PROBE_RECORDS"] if os.environ.get("JSS_PROBE_RECORDS") else "")
    print(f"traj  K={K} {what} B={B}: {best / 1e9:.3f} G env-steps/s  lib={lib}", flush=True)
    # jss_steps: the recorded actions of one trajectory launch, replayed from the state they were recorded in
    snap = env._arena.clone(), env.solution.clone()
    acts = env.trajectory("random", steps=K, record=("action",))["action"]
    n_steps = int((acts >= 0).sum().item())
    rec = ("real_obs", "action_mask", "reward", "done")
    sb = None
    best = 0.0
    for rep in range(5):
        env._arena.copy_(snap[0])
        env.solution.copy_(snap[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sb = env.steps(acts, record=rec, buffers=sb)
        torch.cuda.synchronize()
        best = max(best, n_steps / (time.perf_counter() - t0))
    print(f"steps K={K} {what} B={B}: {best / 1e9:.3f} G env-steps/s  lib={lib}", flush=True)
    # the fused rollout: K steps per launch, nothing but the final state and the counters written
    best = 0.0
    for rep in range(4):
        env.zero_counters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            env.rollout("random", n_iter=K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, env.stats()["steps"] / dt)
    print(f"fused K={K} {what} B={B}: {best / 1e9:.3f} G env-steps/s  lib={lib}", flush=True)
    del env, bufs, sb
