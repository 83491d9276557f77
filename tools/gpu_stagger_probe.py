"""GPU box: does starting the second sub-batch of jss_rollout_steps out of phase pay on launches of ONE round of resident wavefronts?
JSSENV_AMD_LIB=variants/stagger.so (tools/build_instrumented.py stagger: -DJSS_EXP_STAGGER) delays sub-batch i by i x
JSS_EXP_STAGGER_NS in front of its first step.  Windows of K steps, synchronised on both sides (what bench.py times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from jssenv_amd import BatchedJssEnv, builtin_instance
from jssenv_amd.instances import synthetic_packed

dev = torch.device("cuda", 0)
for label, make in (("c4 syn50x20 x 8192", lambda: BatchedJssEnv(synthetic_packed(8192, 50, 20), device=dev, seed=0)),
                    ("c3 ta41 x 16384", lambda: BatchedJssEnv("ta41", batch=16384, device=dev, seed=0)),
                    ("c1 ta01 x 65536", lambda: BatchedJssEnv("ta01", batch=65536, device=dev, seed=0))):
    env = make()
    env.reset()
    env.rollout("random", n_iter=100)
    for K in (20, 200):
        for ns in (0, 1500, 3000, 4500, 6000, 0):
            os.environ["JSS_EXP_STAGGER_NS"] = str(ns)
            if ns == 0:
                os.environ.pop("JSS_EXP_STAGGER_NS")
            ts = []
            for rep in range(200 if K == 20 else 30):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                env.rollout_steps("random", steps=K, n_sub=2)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / K * 1e6)
            ts.sort()
            print(f"{label:22s} K={K:3d} stagger {ns:5d} ns: median {ts[len(ts)//2]:6.2f} us/step  p10 {ts[len(ts)//10]:6.2f}  p90 {ts[len(ts)*9//10]:6.2f}", flush=True)
    del env
