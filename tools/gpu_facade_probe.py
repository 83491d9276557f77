"""GPU box: microseconds per JssEnv.step() of the B = 1 facade, staged copy vs zero-copy action (JSSENV_AMD_ZEROCOPY)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from jssenv_amd import make  # noqa: E402
from jssenv_amd.dispatching import get_rule  # noqa: E402

for inst in ("ta01", "ta80"):
    f = make("jss-v1", env_config={"instance_path": inst}, device="cuda:0")
    f.reset()
    rule, acts, done = get_rule("FIFO"), [], False
    t0 = time.perf_counter()
    while not done:
        a = rule(f)
        acts.append(a)
        _, _, done, _, _ = f.step(a)
    t_rule = (time.perf_counter() - t0) / len(acts) * 1e6
    best = 1e9
    for rep in range(3):
        f.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in acts:
            f.step(a)
        best = min(best, (time.perf_counter() - t0) / len(acts) * 1e6)
    mk = f.current_time_step
    t0 = time.perf_counter()
    tot, mk2 = rule.run_episode(f, device_rng=True, seed=3)
    t_fused = (time.perf_counter() - t0) * 1e3
    print(f"{inst}: zero_copy={f._zero_copy} step() {best:.1f} us, rule(env)+step() {t_rule:.1f} us per step, {len(acts)} steps, "
          f"makespan {mk}; fused episode {t_fused:.2f} ms (makespan {mk2})", flush=True)
