"""Lock-step stepping with external actions, four ways, on BASELINE configs 2, 3 and config 4's share:

  step      one jss_step launch per env step over the whole batch (the form a learner gets from BatchedJssEnv.step)
  lockstep  ONE step session over the whole batch, one fused post + wait launch per step (jss_session_step)
  halves    TWO step sessions, one per half of the batch, in alternation: wait A, post A', wait B, post B' -- the hand-off
            of one half (post -> resident kernel -> progress word -> waiter) hides behind the step of the other; in a learner
            the policy network of half A would run where `post A'` is issued
  halves+policy  the same with a stand-in policy kernel (a torch elementwise op over the half's observation) between wait and
            post -- what the alternation is for

Actions are recorded behaviour trajectories, resident in HBM; every form executes the same env steps.  Prints microseconds
per whole-batch step.  GPU box:  python tools/gpu_session_halves.py [config ...]
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

dev = torch.device("cuda", 0)
N = 240                                                            # steps per measurement


def make(src, batch, first, policy):
    inst = synthetic_packed(batch, 50, 20, first=first) if src is None else builtin_instance(src)
    env = BatchedJssEnv(inst, batch=batch, device=dev, seed=0, env_id_base=first)
    env.reset()
    env.rollout(policy, n_iter=150)
    snap = env._arena.clone(), env.solution.clone()
    acts = env.trajectory(policy, steps=N, record=("action",))["action"]
    env._arena.copy_(snap[0])
    env.solution.copy_(snap[1])
    torch.cuda.synchronize()
    return env, acts, snap


def timed(fn, reps=5):
    cur = torch.cuda.current_stream(dev)
    best = float("inf")
    for _ in range(reps):
        cur.synchronize()
        t0 = time.perf_counter()
        fn()
        cur.synchronize()
        best = min(best, (time.perf_counter() - t0) / N * 1e6)
    return best


for cfg in [int(a) for a in sys.argv[1:]] or [2, 3, 4]:
    src, batch, policy, slots = {2: ("ta01", 4096, "random", 1), 3: ("ta41", 16384, "SPT", 2), 4: (None, 8192, "random", 2)}[cfg]
    env, acts, snap = make(src, batch, 0, policy)

    def restore(e, s):
        e._arena.copy_(s[0])
        e.solution.copy_(s[1])

    def step_loop():
        restore(env, snap)
        for k in range(N):
            env.step(acts[k])
    t_step = timed(step_loop)
    restore(env, snap)
    torch.cuda.synchronize()
    with env.session(depth=1) as s:
        k0 = [0]

        def lock():
            for k in range(N):
                s.step(acts[k])
        t_lock = timed(lock, reps=3)        # (the actions are replayed on a state that has moved on: legal-or-flagged, same cost)
    st = s.host_status()
    del env
    h = batch // 2
    a_env, a_acts, a_snap = make(src, h, 0, policy)
    b_env, b_acts, b_snap = make(src, batch - h, h, policy)
    res = {}
    for with_policy in (False, True):
        sa = a_env.session(depth=2, slots=slots, timeout_ms=1500)
        sb = b_env.session(depth=2, slots=slots, timeout_ms=1500)
        scratch_a = torch.empty_like(a_env.real_obs)
        scratch_b = torch.empty_like(b_env.real_obs)

        def halves():
            base_a, base_b = sa.posted, sb.posted
            sa.post(a_acts[0])
            sb.post(b_acts[0])
            for k in range(1, N):
                sa.wait()
                if with_policy:
                    torch.mul(a_env.real_obs, 0.5, out=scratch_a)       # the learner's policy on half A (stand-in)
                sa.post(a_acts[k])
                sb.wait()
                if with_policy:
                    torch.mul(b_env.real_obs, 0.5, out=scratch_b)
                sb.post(b_acts[k])
            sa.wait()
            sb.wait()
        try:
            res[with_policy] = timed(halves, reps=3)
        except RuntimeError as exc:             # a starved resident grid gives up after timeout_ms: reported, not fatal
            res[with_policy] = float("nan")
            print(f"config {cfg}: two half sessions failed: {exc}", flush=True)
        finally:
            sa.close(check=False)
            sb.close(check=False)
        sta, stb = sa.host_status(), sb.host_status()
        res[(with_policy, "timeouts")] = sta["session_timeouts"] + sta["wait_timeouts"] + stb["session_timeouts"] + stb["wait_timeouts"]
    print(f"config {cfg}: jss_step per step {t_step:.2f} us | one session, lock step {t_lock:.2f} us (env sets per wavefront {st['env_sets_per_wavefront']}, "
          f"timeouts {st['session_timeouts'] + st['wait_timeouts']}) | two half sessions in alternation {res[False]:.2f} us (timeouts {res[(False, 'timeouts')]}) "
          f"| with a stand-in policy kernel per half {res[True]:.2f} us (timeouts {res[(True, 'timeouts')]})", flush=True)
    del a_env, b_env
