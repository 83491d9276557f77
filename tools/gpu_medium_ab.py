"""GPU box: 24-byte medium records against full 32-byte records on the per-env-table packed workloads, same process, same
minute (synthetic 15x15 x 65 536, the ta01..ta10 mix x 65 536): microseconds per one-launch step, K = 100 launches."""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from jssenv_amd import BatchedJssEnv, builtin_instance  # noqa: E402
from jssenv_amd.instances import synthetic_packed  # noqa: E402

dev = torch.device("cuda", 0)
for label, src, B in (("synthetic 15x15, one table per env", synthetic_packed(65536, 15, 15), 65536),
                      ("ta01..ta10 mix (env -> instance map)", [builtin_instance(f"ta{k:02d}") for k in range(1, 11)], 65536),
                      ("ta21..ta30 mix 20x20 (G32)", [builtin_instance(f"ta{k:02d}") for k in range(21, 31)], 32768),
                      ("synthetic 50x20 x 8192 (config 4's share, one wavefront per env)", synthetic_packed(8192, 50, 20), 8192),
                      ("synthetic 50x20 x 65536 (config 4 on one GPU)", synthetic_packed(65536, 50, 20), 65536),
                      ("mixed ta01..ta80 padded x 32768 (config 5, two jobs per lane)", [builtin_instance(f"ta{k:02d}") for k in range(1, 81)], 32768),
                      ("ta51..ta70 mix 50x15 / 50x20 x 16384", [builtin_instance(f"ta{k:02d}") for k in range(51, 71)], 16384)):
    res = {}
    for name, kw in (("medium", {"records": "medium"}), ("full", {"records": "full"})):
        env = BatchedJssEnv(src, batch=B, device=dev, seed=0, **kw)
        env.reset()
        env.rollout("random", n_iter=100)
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                env.rollout("random", n_iter=1)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
        res[name] = best
        del env
    print(f"{label}: medium {res['medium']:.2f} us/step, full {res['full']:.2f} us/step  ({res['full'] / res['medium']:.3f}x)", flush=True)
