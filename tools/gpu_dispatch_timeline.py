"""Profiling aid: start/end of every workgroup of one launch of the benchmarked kernel on the chip-wide 100 MHz
real-time counter (instrumented build variants/timeline.so)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import BatchedJssEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = BatchedJssEnv("ta01", batch=B, device="cuda:0")
env.reset()
env.rollout("random", n_iter=100)
lib = env.lib
lib.jss_debug_timeline.argtypes = [ctypes.c_void_p]
nb = (B + 15) // 16
for rep in range(3):
    for _ in range(5):
        env.rollout("random", n_iter=1)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 2, dtype=np.uint64)
    assert lib.jss_debug_timeline(buf.ctypes.data) == 0
    t = buf.reshape(8192, 2)[:nb].astype(np.float64)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0     # microseconds (100 MHz counter)
    life = end - start
    print(f"B={B} rep {rep}: {nb} workgroups; kernel span {end.max():.2f} us; workgroup lifetime mean {life.mean():.2f} "
          f"min {life.min():.2f} max {life.max():.2f} us")
    order = np.argsort(start)
    qs = [0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0]
    print("   start-time quantiles (us): " + "  ".join(f"{q:.2f}:{np.quantile(start, q):6.2f}" for q in qs))
    print("   end-time   quantiles (us): " + "  ".join(f"{q:.2f}:{np.quantile(end, q):6.2f}" for q in qs))
    # concurrency profile
    for x in (1, 3, 5, 8, 10, 12, 15, 18, 20, 22):
        if x < end.max():
            print(f"   t={x:3d} us: started {int((start <= x).sum()):5d}  finished {int((end <= x).sum()):5d}  resident {int(((start <= x) & (end > x)).sum()):5d}")
    print("   start(us) of block ids 0,1,2,3,8,64,512,1024,2048,4095:", [round(float(start[i]), 2) for i in (0, 1, 2, 3, 8, 64, 512, 1024, 2048, min(4095, nb - 1))])
