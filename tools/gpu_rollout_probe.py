"""jss_rollout(n_iter = 64): policy + step x 64 per launch, outputs once per launch, on the one-wavefront-per-env workloads --
A/B harness for kernel-library builds (JSSENV_AMD_LIB=...).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch
from jssenv_amd import BatchedJssEnv, builtin_instance
from jssenv_amd.instances import synthetic_packed
for label, src, batch in (("synthetic50x20 x 8192", lambda: synthetic_packed(8192, 50, 20), 8192),
                          ("mixed padded x 32768", lambda: [builtin_instance(f"ta{k:02d}") for k in range(1, 81)], 32768)):
    env = BatchedJssEnv(src(), batch=batch, device="cuda:0", seed=0)
    env.reset()
    env.rollout("random", n_iter=100)
    best = 0.0
    for rep in range(5):
        env.zero_counters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            env.rollout("random", n_iter=64)
        torch.cuda.synchronize()
        best = max(best, env.stats()["steps"] / (time.perf_counter() - t0))
    print(f"rollout n_iter=64 {label}: {best / 1e9:.3f} G env-steps/s  lib={os.environ.get('JSSENV_AMD_LIB', 'shipped')[-40:]}", flush=True)
    del env
