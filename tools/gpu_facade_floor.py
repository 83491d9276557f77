"""GPU box: where the B = 1 facade's microseconds go -- the bare launch + synchronise of jss_step on a host-arena env, the
same plus the action write and stream lookup, and the whole JssEnv.step()."""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from jssenv_amd import make  # noqa: E402
from jssenv_amd.dispatching import get_rule  # noqa: E402

f = make("jss-v1", env_config={"instance_path": "ta01"}, device="cuda:0")
f.reset()
rule, acts, done = get_rule("FIFO"), [], False
while not done:
    a = rule(f)
    acts.append(a)
    _, _, done, _, _ = f.step(a)
f.reset()
f.step(acts[0])
act, jss_step, sync_check, d, s, o, a_ptr, stream_of, views, lib, b = f._fast
stream = stream_of()
N = 200
for label, fn in (("jss_step + jss_sync_check only", lambda a: (jss_step(d, s, a_ptr, o, stream), sync_check(stream))),
                  ("+ action write + stream lookup", lambda a: (act.__setitem__(0, a), jss_step(d, s, a_ptr, o, stream_of()), sync_check(stream))),
                  ("JssEnv.step()", f.step)):
    best = 1e9
    for rep in range(3):
        f.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in acts[:N]:
            fn(a)
        best = min(best, (time.perf_counter() - t0) / N * 1e6)
    print(f"{label}: {best:.1f} us", flush=True)
t0 = time.perf_counter()
for _ in range(2000):
    sync_check(stream)
print(f"jss_sync_check on an idle stream: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us")
