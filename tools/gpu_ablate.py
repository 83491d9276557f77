"""Profiling aid (GPU box): time the benchmarked kernel with phases switched off.  Needs the instrumented build:

    python tools/build_instrumented.py profiling
    JSSENV_AMD_LIB=$PWD/variants/profiling.so python tools/gpu_ablate.py [batch] [instance|synthetic50x20]

Results of ablated runs are WRONG by construction; only the timing is meaningful."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
inst = sys.argv[2] if len(sys.argv) > 2 else "ta01"
PROF_ABLATE = 1
if inst == "synthetic50x20":
    from jssenv_amd.instances import synthetic_packed
    env = BatchedJssEnv(synthetic_packed(B, 50, 20), device="cuda:0")
else:
    env = BatchedJssEnv(inst, batch=B, device="cuda:0")
env.reset()
ids = torch.arange(B, device="cuda:0") % 16
for r in range(15):
    for _ in range(16):
        a = env.policy("random")
        env.step(torch.where(ids > r, a, torch.full_like(a, -1)))
snapshot = [t.clone() for t in (env.env_header, env.job_state, env.machine_state)]


def run(mask, n=200):
    for t, s in zip((env.env_header, env.job_state, env.machine_state), snapshot):
        t.copy_(s)
    assert env.lib.jss_profiling_set(PROF_ABLATE, mask) == 0
    for _ in range(10):
        env.rollout("random", n_iter=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                env.rollout("random", n_iter=1)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    env.lib.jss_profiling_set(PROF_ABLATE, 0)
    return e0.elapsed_time(e1) / n * 1e3


names = {0: "full", 1: "-check_no_op", 2: "-prioritize", 4: "-obs", 8: "-select", 16: "-advance", 31: "-all five"}
base = None
for mask, name in names.items():
    us = run(mask)
    base = base or us
    print(f"{name:16s} {us:8.2f} us/launch  ({us - base:+.2f})", flush=True)
# host launch cost: eager loop
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    env.rollout("random", n_iter=1)
t_enq = (time.perf_counter() - t0) / 500 * 1e6
torch.cuda.synchronize()
print(f"eager enqueue+run {t_enq:.2f} us per launch (host path)")

# the two-kernel path an RL trainer with its own policy uses: jss_policy + jss_step
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):
        for _ in range(100):
            env.step(env.policy("random"))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"policy + step (2 launches) {e0.elapsed_time(e1) / 100 * 1e3:.2f} us per env step")
