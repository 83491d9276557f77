"""One jss_step launch per env step, actions resident (a recorded behaviour trajectory), hipGraph replay of K launches:
microseconds per step for BASELINE config 4's share, config 5 padded, config 3, per-env 15x15 tables and the headline -- the A/B harness for kernel-library
builds (JSSENV_AMD_LIB=variants/<x>/libjss_hip.so python tools/gpu_step_probe.py).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import torch
from jssenv_amd import BatchedJssEnv, builtin_instance
from jssenv_amd.instances import synthetic_packed

K = 100
dev = torch.device("cuda", 0)
for label, src, batch, pol in (("c4 syn50x20 x 8192", lambda: synthetic_packed(8192, 50, 20), 8192, "random"),
                               ("c5 mixed padded x 32768", lambda: [builtin_instance(f"ta{k:02d}") for k in range(1, 81)], 32768, "random"),
                               ("c3 ta41 x 16384", lambda: builtin_instance("ta41"), 16384, "SPT"),
                               ("syn15x15 x 65536", lambda: synthetic_packed(65536, 15, 15), 65536, "random"),
                               ("c1 ta01 x 65536", lambda: builtin_instance("ta01"), 65536, "random")):
    env = BatchedJssEnv(src(), batch=batch, device=dev, seed=0)
    env.reset()
    env.rollout(pol, n_iter=150)
    snap = env._arena.clone(), env.solution.clone()
    acts = env.trajectory(pol, steps=K, record=("action",))["action"]
    env._arena.copy_(snap[0]); env.solution.copy_(snap[1])
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for k in range(K):
                env.step(acts[k])
    torch.cuda.current_stream(dev).wait_stream(side)
    ts = []
    for rep in range(12):
        env._arena.copy_(snap[0]); env.solution.copy_(snap[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K * 1e6)
    ts.sort()
    print(f"{label:28s} jss_step graph replay: median {ts[len(ts)//2]:.2f} us/step  min {ts[0]:.2f}  ({os.environ.get('JSSENV_AMD_LIB', 'shipped')})", flush=True)
    del g, env
