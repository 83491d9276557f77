"""GPU box, instrumented build: does a one-step launch that is exactly ONE round of resident wavefronts run faster as TWO
overlapping rounds?  The occupancy of the benchmarked kernels is capped through extra dynamic LDS per workgroup
(JSS_PROF_LDS_PAD) and the step time measured -- one launch per step and two / three sub-batches on as many streams.

    JSSENV_AMD_LIB=$PWD/variants/profiling.so python tools/gpu_occupancy_probe.py [K]
"""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from jssenv_amd import BatchedJssEnv, builtin_instance  # noqa: E402
from jssenv_amd.instances import synthetic_packed  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
JSS_PROF_LDS_PAD = 2


def us_per_step(fn, reps=6):
    best = 1e9
    fn()
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    return best


cases = (("c4 share: synthetic 50x20 x 8192 (one wavefront per env)", lambda: synthetic_packed(8192, 50, 20), 8192, "random"),
         ("c3: ta41 SPT x 16384 (two envs per wavefront)", lambda: builtin_instance("ta41"), 16384, "SPT"),
         ("c2: ta01 x 4096", lambda: builtin_instance("ta01"), 4096, "random"),
         ("headline: ta01 x 65536", lambda: builtin_instance("ta01"), 65536, "random"))
for label, src, B, pol in cases:
    env = BatchedJssEnv(src(), batch=B, device=dev, seed=0)
    lib = env.backend.lib
    if not hasattr(lib, "jss_profiling_set"):
        raise SystemExit("needs the instrumented build: JSSENV_AMD_LIB=variants/profiling.so")
    env.reset()
    env.rollout(pol, n_iter=170)
    print(f"== {label} ==  (K = {K}; us per step: one launch / 2 sub-batches / 3 sub-batches)", flush=True)
    for wg_per_cu in (8, 7, 6, 5, 4, 3):
        # 160 KB of LDS per CU: a pad that leaves room for exactly wg_per_cu workgroups (the kernels' own LDS is < 8 KB)
        pad = 0 if wg_per_cu == 8 else (160 * 1024) // wg_per_cu - 8 * 1024
        if pad > 56 * 1024:
            continue
        assert lib.jss_profiling_set(JSS_PROF_LDS_PAD, pad) == 0
        row = [us_per_step(lambda: [env.rollout(pol, n_iter=1) for _ in range(K)])]
        for n_sub in (2, 3):
            run = env.bind_rollout_steps(pol, steps=K, n_sub=n_sub, caller_orders_streams=True)
            row.append(us_per_step(run))
        print(f"  <= {wg_per_cu} workgroups per CU ({4 * wg_per_cu // 4} wavefronts per SIMD; LDS pad {pad:6d} B): "
              + "  ".join(f"{x:7.2f}" for x in row), flush=True)
    lib.jss_profiling_set(JSS_PROF_LDS_PAD, 0)
    del env
