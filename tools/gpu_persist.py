"""Profiling aid: launch time of the benchmarked path vs JSS_OPT_PERSIST (waves per SIMD of the persistent kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv, _abi  # noqa: E402


def time_launches(env, n=300):
    for _ in range(20):
        env.rollout("random", n_iter=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        env.rollout("random", n_iter=1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for inst, B in (("ta01", 65536), ("ta01", 131072), ("ta01", 32768), ("ta41", 16384), ("ta41", 65536)):
    env = BatchedJssEnv(inst, batch=B, device="cuda:0")
    env.reset()
    env.rollout("random", n_iter=300)
    for k in (0, 1, 2, 3, 4):
        env.lib.jss_set_option(_abi.OPT_PERSIST, k)
        us = time_launches(env)
        print(f"{inst} B={B:7d} persist={k}: {us:7.2f} us/launch  {B / us / 1e3:.3f} G steps/s", flush=True)
    env.lib.jss_set_option(_abi.OPT_PERSIST, 0)
    del env
