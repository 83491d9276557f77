import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import parity_cases as P
from jssenv_amd.env import HipBackend, BatchedJssEnv
from oracle import OracleEnv
be = HipBackend("cuda:0")
rng = np.random.default_rng(0)
insts = [P.random_instance(rng, 128, 64, max_dur=999) for _ in range(3)]
env = BatchedJssEnv(insts, seed=1, _backend=be)
env.reset(); env.rollout("random", n_iter=500); env.synchronize()
for i, inst in enumerate(insts):
    o = OracleEnv(inst, strict=True); o.reset(); o.rollout("random", 1, i, 500, episode=1)
    P.assert_matches_oracle(env.host_state(i), o, f"128x64 env {i}")
print("128x64 per-env tables OK, steps", env.stats())
