"""Pins the CPU oracle (oracle/jss_oracle.c) against golden vectors captured from
the live reference (tools/make_golden.py).  CPU only."""
import numpy as np
import pytest

import golden_util as G
from jssenv_amd import instances as I
from oracle import OracleEnv, POLICY_IDS


@pytest.mark.parametrize("inst", G.PUBLISHED)
def test_published_schedules(inst):
    """G1: the 12 known-answer schedules of the reference's tests/test_solutions.py."""
    g = G.load(f"published_{inst}")
    env = OracleEnv(I.builtin_instance(inst))
    env.reset()
    G.replay(env, g)
    assert env.current_time_step == G.PUBLISHED_MAKESPAN[inst] == int(g["makespan"])
    assert (env.solution == g["solution"]).all()
    assert (env.solution >= 0).all()
    assert (env.todo_time_step_job == env.machines).all()


@pytest.mark.parametrize("inst", G.PUBLISHED)
def test_numpy_restatement_published_schedules(inst):
    """The Python / NumPy restatement (bench.py's reference-speed CPU baseline) on the same known answers."""
    from oracle.np_restatement import NumpyJssEnv
    g = G.load(f"published_{inst}")
    env = NumpyJssEnv(I.builtin_instance(inst))
    env.reset()
    G.replay(env, g)
    assert env.current_time_step == G.PUBLISHED_MAKESPAN[inst] == int(g["makespan"])
    assert (env.solution == g["solution"]).all()


@pytest.mark.parametrize("inst", G.RANDOM)
def test_numpy_restatement_random_traces(inst):
    """... and on the random traces with forced NOPEs: every integer, the float64 observation and the reward bit-equal."""
    from oracle.np_restatement import NumpyJssEnv
    g = G.load(f"random_{inst}")
    env = NumpyJssEnv(I.builtin_instance(inst))
    G.replay(env, g)
    assert (env.solution == g["solution"]).all()


@pytest.mark.parametrize("inst", G.RANDOM)
def test_random_traces_with_forced_nope(inst):
    """G2: random masked traces incl. NOPEs forced while action_mask[J] is False."""
    g = G.load(f"random_{inst}")
    env = OracleEnv(I.builtin_instance(inst))
    G.replay(env, g)
    assert (env.solution == g["solution"]).all()


def test_rule_table():
    """G3: deterministic dispatching rules (exploration off)."""
    g = G.load("rules")
    rules, insts = [str(r) for r in g["rules"]], [str(i) for i in g["instances"]]
    for ri, rule in enumerate(rules):
        if rule not in POLICY_IDS:
            continue
        for ii, inst in enumerate(insts):
            env = OracleEnv(I.builtin_instance(inst))
            env.reset()
            done, total, trace = False, 0.0, []
            while not done:
                a = env.policy(rule)
                trace.append(a)
                _, r, done, _, _ = env.step(a)
                total += r
            assert trace == g[f"trace_{rule}_{inst}"].tolist(), (rule, inst)
            assert env.current_time_step == g["makespan"][ri, ii], (rule, inst)
            assert total == g["total_reward"][ri, ii], (rule, inst)
            # identity from SURVEY 8(a10): sum of rewards = (2*sum_op - M*makespan)/max_time_op
            ident = (2 * env.sum_op - env.machines * env.current_time_step) / env.max_time_op
            assert abs(total - ident) < 1e-9


def test_seeded_rule_traces():
    """G4: stochastic rule traces are replayed as actions (NumPy's MT19937 is not emulated)."""
    g = G.load("rules_seeded")
    for key in [k for k in g if k.startswith("trace_")]:
        _, rule, inst, seed = key.split("_")
        env = OracleEnv(I.builtin_instance(inst))
        env.reset()
        done = False
        for a in g[key]:
            assert not done
            _, _, done, _, _ = env.step(int(a))
        assert done and env.current_time_step == int(g[f"makespan_{rule}_{inst}_{seed}"])


def test_state_invariants_random_episodes():
    """The invariants of the reference's tests/test_state.py:8-76, 20 random episodes on ta01."""
    env = OracleEnv(I.builtin_instance("ta01"), strict=True)
    for ep in range(20):
        obs = env.reset()
        done, step = False, 0
        while not done:
            a = env.policy("random", seed=123, env_id=ep, episode=0, step=step)
            assert obs["action_mask"][a]
            obs, r, done, _, _ = env.step(a)
            step += 1
            s = obs["real_obs"]
            assert s.min() >= 0.0 and s.max() <= 1.0 and np.isfinite(s).all()
            legal = env.legal_actions
            assert legal[:-1].sum() == env.nb_legal_actions
            assert len({int(m) for m, l in zip(env.needed_machine_jobs, legal[:-1]) if l}) == env.nb_machine_legal
        assert env.err == 0
        assert len(env.next_time_step) == 0
        assert (env.solution >= 0).all()
        assert (env.todo_time_step_job == env.machines).all()


def test_total_semantics_outside_the_reference_contract():
    env = OracleEnv(I.builtin_instance("ta01"), strict=True)
    env.reset()
    # NOPE at t=0 with nothing busy: the reference raises IndexError (jss_env.py:517)
    _, r, done, _, _ = env.step(env.jobs)
    assert env.err & 2 and done and r == 0.0
    env.reset()
    env.step(3)
    before = (env.todo_time_step_job.copy(), env.legal_actions.copy(), env.current_time_step)
    _, r, _, _, _ = env.step(3)  # job 3 is running: outside the mask -> ignored + flagged
    assert env.err & 1 and r == 0.0
    assert (before[0] == env.todo_time_step_job).all() and (before[1] == env.legal_actions).all()
    _, r, _, _, _ = env.step(99)
    assert env.err & 4


@pytest.mark.refcheck
def test_oracle_vs_live_reference_lockstep():
    """Build container only: drive reference and oracle side by side (float64-exact)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import lockstep as L
    rng = np.random.default_rng(5)
    for name in ("ta03", "ta33", "dmu18"):
        ref, orc = L.make_pair(name)
        done, n = False, 0
        while not done:
            m = np.asarray(ref.legal_actions)
            if rng.random() < 0.05 and len(ref.next_time_step) > 0 and m[:-1].any():
                a = ref.jobs
            else:
                a = int(rng.choice(np.flatnonzero(m)))
            try:
                _, _, done = L.step_both(ref, orc, a, f"{name} step {n}")
            except IndexError:
                break
            n += 1
