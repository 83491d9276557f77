"""GPU box: two envs per wavefront (jss_kernel_two) against one env per wavefront
(JssDesc.kernel | JSS_KERNEL_ONE_ENV_PER_WAVE), same process, same minute, interleaved repetitions.  Microseconds per step
of the fused one-launch-per-step form (jss_rollout n_iter = 1) -- ONE launch per step, and two sub-batches on two streams
(jss_rollout_steps, windows of K steps between device-wide synchronizes like bench.py) -- and of jss_step with resident actions.

    python tools/gpu_two_per_wave_ab.py [K]
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
B_ALG = lambda J, M: 89 * J + 10 * M + 40   # noqa: E731


def us_per_step(fn, reps=7):
    best = 1e9
    fn()
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    return best


mixed = [builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
cases = (("c4 share: synthetic 50x20 x 8192", lambda: synthetic_packed(8192, 50, 20), 8192, B_ALG(50, 20), {}),
         ("synthetic 50x20 x 16384", lambda: synthetic_packed(16384, 50, 20), 16384, B_ALG(50, 20), {}),
         ("c4 whole: synthetic 50x20 x 65536", lambda: synthetic_packed(65536, 50, 20), 65536, B_ALG(50, 20), {}),
         ("ta51..ta70 mix 50x15 / 50x20 x 16384", lambda: [builtin_instance(f"ta{k:02d}") for k in range(51, 71)], 16384,
          sum(B_ALG(i.jobs, i.machines) for i in mixed[50:70]) / 20.0, {}),
         ("c5: mixed ta01-80 x 32768, default constructor (by shape class)", lambda: mixed, 32768,
          sum(B_ALG(i.jobs, i.machines) for i in mixed) / 80.0, {}))
for label, src, B, alg, kw in cases:
    envs = {}
    for name, kernel in (("two", "auto-2env"), ("one", "auto-1env")):
        e = BatchedJssEnv(src(), batch=B, device=dev, seed=0, kernel=kernel, **kw)
        e.reset()
        e.rollout("random", n_iter=170)
        envs[name] = e
    rows = {}
    for rep in range(2):                       # interleaved: a slow minute of the box hits both forms
        for name, e in envs.items():
            r = rows.setdefault(name, {})
            r["eager"] = min(r.get("eager", 1e9), us_per_step(lambda: [e.rollout("random", n_iter=1) for _ in range(K)]))
            for n_sub in (2, 3):
                run = e.bind_rollout_steps("random", steps=K, n_sub=n_sub, caller_orders_streams=True)
                r[f"sub{n_sub}"] = min(r.get(f"sub{n_sub}", 1e9), us_per_step(run))
            acts = e.trajectory("random", steps=K, record=("action",))["action"]
            snap = e._arena.clone(), e.solution.clone()

            def replay():                      # (jss_step replays the recorded actions from the state they were recorded in)
                e._arena.copy_(snap[0])
                e.solution.copy_(snap[1])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(K):
                    e.step(acts[k])
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / K * 1e6
            replay()
            r["step"] = min([r.get("step", 1e9)] + [replay() for _ in range(4)])
            e._arena.copy_(snap[0])
            e.solution.copy_(snap[1])
    print(f"== {label} ==  (K = {K}; us per step, fraction of the 8 TB/s roofline)")
    for form in ("eager", "sub2", "sub3", "step"):
        a, b = rows["one"][form], rows["two"][form]
        f = lambda us: B * alg / (us * 1e-6) / 8e12   # noqa: E731
        print(f"  {form:6s} one env per wavefront {a:7.2f} ({f(a):.3f})   two envs per wavefront {b:7.2f} ({f(b):.3f})   {a / b:.3f}x", flush=True)
    del envs
