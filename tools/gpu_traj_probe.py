"""GPU box: env-steps/s of jss_trajectory (K steps per launch, everything recorded) on the benchmark workloads."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv, builtin_instance  # noqa: E402
from jssenv_amd.instances import synthetic_packed  # noqa: E402

K = int(os.environ.get("JSS_KT", "32"))
for what, B in (("ta01", 65536), ("ta01", 4096), ("synthetic50x20", 8192), ("mixed", 32768), ("synthetic15x15", 65536), ("ta41", 16384)):
    if what == "mixed":
        env = BatchedJssEnv([builtin_instance(f"ta{k:02d}") for k in range(1, 81)], batch=B, device="cuda:0")
    elif what.startswith("synthetic"):
        J, M = (int(x) for x in what[len("synthetic"):].split("x"))
        env = BatchedJssEnv(synthetic_packed(B, J, M), device="cuda:0")
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
    print(f"traj K={K} {what} B={B}: {best / 1e9:.3f} G env-steps/s  lib={os.path.basename(os.environ.get('JSSENV_AMD_LIB', 'shipped'))}", flush=True)
    del env, bufs
