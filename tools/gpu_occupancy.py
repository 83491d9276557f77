"""Profiling aid (GPU box): kernel time vs occupancy cap (extra LDS per workgroup) and batch size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv, _abi  # noqa: E402


def time_launches(env, n=200):
    for _ in range(10):
        env.rollout("random", n_iter=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        env.rollout("random", n_iter=1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (65536, 131072, 262144):
    env = BatchedJssEnv("ta01", batch=B, device="cuda:0")
    env.reset()
    env.rollout("random", n_iter=300)
    for pad, label in ((0, "8 blocks/CU (no cap)"), (19000, "6"), (24000, "5"), (32000, "4"), (45000, "3"), (70000, "2")):
        env.lib.jss_set_option(_abi.OPT_LDS_PAD, pad)
        us = time_launches(env)
        print(f"B={B:7d} lds_pad={pad:6d} (~{label} blocks/CU): {us:7.2f} us/launch  {B / us / 1e3:.3f} G steps/s", flush=True)
    env.lib.jss_set_option(_abi.OPT_LDS_PAD, 0)
    del env
