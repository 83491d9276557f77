"""Tooling (CPU): how deep the look-ahead walks of _check_no_op (jss_env.py:324-401) go, measured on the oracle
with the random masked policy.  This is the statistic behind carrying a job's next three ops in its state record
(include/jss_hip.h JSS_F_CUR / JSS_F_NEXT / bits 10-31 of JSS_F_TODO): how often a walk needs an op table entry
beyond them.

    python tools/walk_depth.py [steps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jssenv_amd import builtin_instance, synthetic_batch  # noqa: E402
from oracle import OracleEnv  # noqa: E402


def stats(inst, nsteps, seed=0):
    o = OracleEnv(inst, strict=True)
    o.reset()
    J, M, dur = inst.jobs, inst.machines, inst.duration
    gate_open = steps = lanes = 0
    beyond = np.zeros(6, int)        # walking lanes by number of ops needed beyond (cur, next)
    worst = np.zeros(6, int)         # per open gate: the deepest lane
    ep, st = 1, 0
    for _ in range(nsteps):
        if o.nb_legal_actions == 0:
            o.reset()
            ep, st = ep + 1, 0
            continue
        o.step(o.policy("random", seed=seed, env_id=0, episode=ep, step=st))
        st, steps = st + 1, steps + 1
        legal = o.legal_actions[:-1]
        tm, t = o.time_until_available_machine, o.current_time_step
        if not 1 <= legal.sum() <= 4 or not (tm > 0).any():
            continue
        todo, left = o.todo_time_step_job, o.time_until_finish_current_op_jobs
        blocked, need = o.action_illegal_no_op, o.needed_machine_jobs
        machines = set(need[legal].tolist())
        if len(machines) > 3:
            continue
        if min(t + dur[j, todo[j]] for j in range(J) if legal[j]) < t + tm[tm > 0].min():
            continue
        gate_open += 1
        horizon = {m: t + inst.max_time_op for m in machines}
        mh = t
        for j in range(J):
            if legal[j]:
                horizon[need[j]] = min(horizon[need[j]], t + dur[j, todo[j]])
                mh = max(mh, horizon[need[j]])
        deepest = 0
        for j in range(J):
            if legal[j] or todo[j] >= M:
                continue
            case_a = left[j] > 0 and todo[j] + 1 < M
            case_b = not case_a and not blocked[j]
            if not (case_a or case_b):
                continue
            k = todo[j] + 1 if case_a else todo[j]
            tn = t + left[j] if case_a else t + tm[need[j]]
            while k < M - 1 and mh > tn:
                tn += dur[j, k]
                k += 1
            extra = max(0, (k - 1) - (todo[j] + 1))
            beyond[min(extra, 5)] += 1
            lanes += 1
            deepest = max(deepest, extra)
        worst[min(deepest, 5)] += 1
    print(f"{inst.name:16s} gate open on {gate_open / steps:5.1%} of steps; walking lanes needing 0/1/2/3/4/5+ ops beyond "
          f"(cur, next): {np.round(beyond / max(1, lanes), 3)};  per open gate, the deepest lane: {np.round(worst / max(1, gate_open), 3)}")


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    for inst in (builtin_instance("ta01"), synthetic_batch(1, 50, 20)[0], builtin_instance("ta41"), builtin_instance("ta80")):
        stats(inst, n)
