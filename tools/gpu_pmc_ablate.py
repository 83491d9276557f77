"""Profiling aid (GPU box, instrumented build): run the benchmarked kernel with one ablation mask so that rocprofv3
PMC passes can attribute instruction counts to phases.  Driven by tools/gpu_pmc_ablate.sh.

    JSSENV_AMD_LIB=$PWD/variants/profiling.so python tools/gpu_pmc_ablate.py <mask> [batch] [instance]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv  # noqa: E402

mask = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
inst = sys.argv[3] if len(sys.argv) > 3 else "ta01"
if inst.startswith("synthetic"):
    from jssenv_amd.instances import synthetic_packed
    J, M = (int(x) for x in inst[len("synthetic"):].split("x"))
    env = BatchedJssEnv(synthetic_packed(B, J, M), device="cuda:0")
else:
    env = BatchedJssEnv(inst, batch=B, device="cuda:0")
env.reset()
ids = torch.arange(B, device="cuda:0") % 16
skip = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
for r in range(15):
    for _ in range(16):
        env.step(torch.where(ids > r, env.policy("random"), skip))
torch.cuda.synchronize()
assert env.lib.jss_profiling_set(1, mask) == 0
for _ in range(60):
    env.rollout("random", n_iter=1)
torch.cuda.synchronize()
env.lib.jss_profiling_set(1, 0)
