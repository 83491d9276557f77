"""Profiling aid: 30 launches of the benchmarked kernel per ablation mask, in a fixed order, so that a
rocprofv3 --pmc pass can be chunked into per-mask instruction counts (tools/gpu_pmc_ablate.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv, _abi  # noqa: E402

B = 65536
env = BatchedJssEnv("ta01", batch=B, device="cuda:0")
env.reset()
env.rollout("random", n_iter=120)       # mid-episode state (one launch of another kernel: kRollout)
snap = [t.clone() for t in (env.env_header, env.job_state, env.machine_state)]
for mask in (0, 1, 2, 4, 8, 16, 31):
    for t, s in zip((env.env_header, env.job_state, env.machine_state), snap):
        t.copy_(s)
    env.lib.jss_set_option(_abi.OPT_ABLATE, mask)
    for _ in range(30):
        env.rollout("random", n_iter=1)
    torch.cuda.synchronize()
