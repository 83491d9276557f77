"""GPU box: full (32-byte) vs medium (24-byte) job records on the one-wavefront-per-env flavour with per-env tables:
BASELINE config 4's share (synthetic 50 x 20 x 8 192) and the whole of it (x 65 536).  One-step rollouts, hipGraph replay of K
launches and jss_rollout_steps with 2 sub-batches.  kernel="auto": two envs per wavefront from 20 480 envs per launch on;
"auto-1env": one env per wavefront whatever the size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch
from jssenv_amd import BatchedJssEnv
from jssenv_amd.instances import synthetic_packed

K = 100
dev = torch.device("cuda", 0)
from jssenv_amd import builtin_instance
ta = lambda a, b: [builtin_instance(f"ta{k:02d}") for k in range(a, b + 1)]
CASES = {"c4": [("syn50x20 x 8192", lambda: synthetic_packed(8192, 50, 20), None), ("syn50x20 x 65536", lambda: synthetic_packed(65536, 50, 20), None)],
         # config 5's one-wavefront-per-env classes on their own, and its 32-lane class
         "c5": [("ta51-70 x 8192 (50 jobs)", lambda: ta(51, 70), 8192), ("ta71-80 x 4096 (100 jobs)", lambda: ta(71, 80), 4096),
                ("ta11-50 x 12288 (20-30 jobs)", lambda: ta(11, 50), 12288)]}
for label, make, batch in CASES[sys.argv[1] if len(sys.argv) > 1 else "c4"]:
    src = make()
    for records, kernel in (("full", "auto"), ("medium", "auto"), ("medium", "auto-1env"), ("full", "auto"), ("medium", "auto"), ("medium", "auto-1env")):
        env = BatchedJssEnv(src, batch=batch, device=dev, seed=0, records=records, kernel=kernel)
        env.reset()
        env.rollout("random", n_iter=100)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for k in range(K):
                    env.rollout("random", n_iter=1)
        torch.cuda.current_stream(dev).wait_stream(side)
        ts, t2 = [], []
        for rep in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e6)
        for rep in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            env.rollout_steps("random", steps=K, n_sub=2)
            torch.cuda.synchronize()
            t2.append((time.perf_counter() - t0) / K * 1e6)
        ts.sort(); t2.sort()
        print(f"{label:30s} records={records:6s} kernel={kernel:9s}: graph replay {ts[len(ts)//2]:6.2f} us/step   2 sub-batches {t2[len(t2)//2]:6.2f} us/step", flush=True)
        del g, env
