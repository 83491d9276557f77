"""Profiling aid: per-phase s_memtime stamps of three waves (first, middle, last workgroup) of the benchmarked
kernel, from an instrumented build (variants/timing.so, built ad hoc -- see profiles/README.md)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv  # noqa: E402

names = ["entry", "loads issued + table staged + barrier", "state arrived (unpack)", "policy (select)", "alloc prologue",
         "advance loop", "prioritize", "check_no_op", "state stored", "obs stored"]
for B in (4096, 65536):
    env = BatchedJssEnv("ta01", batch=B, device="cuda:0")
    env.reset()
    env.rollout("random", n_iter=100)
    lib = env.lib
    lib.jss_debug_times.argtypes = [ctypes.c_void_p]
    acc = np.zeros((3, 16))
    n = 40
    for _ in range(n):
        env.rollout("random", n_iter=1)
        torch.cuda.synchronize()
        buf = np.zeros(64, dtype=np.uint64)
        assert lib.jss_debug_times(buf.ctypes.data) == 0
        t = buf.reshape(4, 16)[:3].astype(np.float64)
        acc += t - t[:, :1]
    acc /= n
    print(f"== B={B}: cycles since kernel entry of that wave (s_memtime ticks, 100 MHz? see ratio), mean of {n} launches")
    for w, label in enumerate(("first workgroup", "middle workgroup", "last workgroup")):
        prev = 0.0
        print(f"  {label}")
        for i in range(1, 10):
            print(f"    {names[i]:42s} +{acc[w, i] - prev:9.0f}   (t = {acc[w, i]:9.0f})")
            prev = acc[w, i]
