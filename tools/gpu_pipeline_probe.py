"""Profiling aid (GPU box): the sub-batch pipeline of jss_rollout_steps.  For n_sub in 1..8: host enqueue time
per step (how long the C launch loop keeps the host busy) and GPU time per step, eager and -- optionally --
captured into one hipGraph (multi-stream capture: fork/join events become graph edges).

    python tools/gpu_pipeline_probe.py [batch] [instance|synthetic50x20|synthetic15x15] [--graph] [--wave]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv  # noqa: E402
from jssenv_amd.instances import synthetic_packed  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 65536
what = args[1] if len(args) > 1 else "ta01"
use_graph = "--graph" in sys.argv
kernel = "wave" if "--wave" in sys.argv else None      # force one wavefront per env (A/B against the packed kernels)
if what == "mixed":                                      # BASELINE config 5: env i <- ta(1 + i % 80), padded 100x20
    from jssenv_amd import builtin_instance
    env = BatchedJssEnv([builtin_instance(f"ta{k:02d}") for k in range(1, 81)], batch=B, device="cuda:0", kernel=kernel)
elif what.startswith("synthetic"):
    J, M = (int(x) for x in what[len("synthetic"):].split("x"))
    env = BatchedJssEnv(synthetic_packed(B, J, M), device="cuda:0", kernel=kernel)
else:
    env = BatchedJssEnv(what, batch=B, device="cuda:0", kernel=kernel)
env.reset()
ids = torch.arange(B, device="cuda:0") % 16
skip = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
for r in range(15):
    for _ in range(16):
        env.step(torch.where(ids > r, env.policy("random"), skip))
env.rollout("random", n_iter=64)
print(f"{what} B={B} lib={os.environ.get('JSSENV_AMD_LIB', 'shipped')}", flush=True)
if os.environ.get("JSS_ABLATE"):      # instrumented build only (JSSENV_AMD_LIB=variants/profiling.so): phase ablation mask
    assert env.lib.jss_profiling_set(1, int(os.environ["JSS_ABLATE"])) == 0
    print("ablation mask", os.environ["JSS_ABLATE"], "(results are wrong by construction; timing only)")
K = int(os.environ.get("JSS_K", "200"))
WARM = int(os.environ.get("JSS_WARM", "0"))
for n_sub in [int(x) for x in os.environ.get("JSS_NSUB", "1,2,3,4,6,8").split(",")]:
    env.rollout_steps("random", steps=K, n_sub=n_sub)          # warm (stream creation)
    torch.cuda.synchronize()
    best = None
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if WARM:                                                  # keep the GPU busy right up to the window's opening sync
            env.rollout_steps("random", steps=WARM, n_sub=n_sub)
        env.zero_counters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        env.rollout_steps("random", steps=K, n_sub=n_sub)
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_wall = time.perf_counter() - t0
        gpu = e0.elapsed_time(e1) / K * 1e3
        steps = env.stats()["steps"]
        row = (gpu, t_host / K * 1e6, steps / (gpu * 1e-6 * K), t_wall / K * 1e6)
        best = row if best is None or row[0] < best[0] else best
    print(f"eager  n_sub={n_sub}: GPU {best[0]:7.2f} us/step   host enqueue {best[1]:6.2f} us/step   {best[2] / 1e9:.3f} G env-steps/s   wall {best[3]:6.2f} us/step (K={K}, warm={WARM})", flush=True)
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                env.rollout_steps("random", steps=K, n_sub=n_sub)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / K * 1e3)
        print(f"graph  n_sub={n_sub}: GPU {best:7.2f} us/step", flush=True)
        del g
torch.cuda.synchronize()
env.close()
print("done")
