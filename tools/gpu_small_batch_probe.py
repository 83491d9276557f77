"""GPU box: below how many envs is the one-wavefront-per-env flavour faster than the packed one on a small instance?  ta01 (15 x 15):
packed = 4 envs per wavefront (B / 4 wavefronts of ~860 dependent instructions), wave = one env per wavefront (B wavefronts of
~730).  A small batch is latency-bound -- one wavefront per SIMD or fewer -- so the shorter chain should win there.
One-step rollouts (policy + step, observation written), hipGraph replay of K launches; also jss_rollout_steps with 2 sub-batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch
from jssenv_amd import BatchedJssEnv

K = 100
dev = torch.device("cuda", 0)
inst = sys.argv[1] if len(sys.argv) > 1 else "ta01"
for B in (256, 1024, 2048, 4096, 8192, 16384):
    for kernel in ("auto", "wave"):
        env = BatchedJssEnv(inst, batch=B, device=dev, seed=0, kernel=kernel)
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
        for rep in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / K * 1e6)
        for rep in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            env.rollout_steps("random", steps=K, n_sub=2)
            torch.cuda.synchronize()
            t2.append((time.perf_counter() - t0) / K * 1e6)
        ts.sort(); t2.sort()
        print(f"{inst} x {B:6d} kernel={kernel:5s}: graph replay {ts[len(ts)//2]:6.2f} us/step   2 sub-batches {t2[len(t2)//2]:6.2f} us/step", flush=True)
        del g, env
