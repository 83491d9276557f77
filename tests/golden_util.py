"""Helpers shared by the golden-vector tests (oracle on CPU, HIP path on GPU)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PUBLISHED = ["ta01", "ta41", "ta42", "ta43", "ta44", "ta45", "ta46", "ta47", "ta48", "ta49", "ta50", "ta51"]
PUBLISHED_MAKESPAN = dict(zip(PUBLISHED, [1231, 2006, 1939, 1846, 1979, 2000, 2006, 1889, 1937, 1963, 1923, 2760]))
RANDOM = ["ta01", "ta41", "dmu16", "ta51", "ta80", "ta25"]

# row order of the golden "job_state" block
JOB_FIELDS = ("todo_time_step_job", "needed_machine_jobs", "time_until_finish_current_op_jobs",
              "total_perform_op_time_jobs", "total_idle_time_jobs", "idle_time_jobs_last_op")


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def check_env_against_row(env, g, i, where):
    """env: anything with the reference's attribute names; g: golden dict; i: row."""
    assert int(env.current_time_step) == int(g["clock"][i]), f"{where}: clock"
    for f, name in enumerate(JOB_FIELDS):
        got = np.asarray(getattr(env, name)).astype(np.int64)
        assert (got == g["job_state"][i, f]).all(), f"{where}: {name}\n got={got}\nwant={g['job_state'][i, f]}"
    assert (np.asarray(env.time_until_available_machine) == g["tm"][i]).all(), f"{where}: tm"
    assert (np.asarray(env.legal_actions).astype(bool) == g["legal"][i]).all(), \
        f"{where}: legal\n got={np.asarray(env.legal_actions).astype(int)}\nwant={g['legal'][i].astype(int)}"
    assert (np.asarray(env.action_illegal_no_op).astype(bool) == g["blocked"][i]).all(), f"{where}: blocked"
    assert (np.asarray(env.machine_legal).astype(bool) == g["machine_legal"][i]).all(), f"{where}: machine_legal"
    assert int(env.nb_legal_actions) == int(g["nb_legal"][i]), f"{where}: nb_legal"
    assert int(env.nb_machine_legal) == int(g["nb_machine_legal"][i]), f"{where}: nb_machine_legal"
    assert len(env.next_time_step) == int(g["queue_len"][i]), f"{where}: queue length"


def replay(env, g, obs_tol=None, check_state=True):
    """Replay a golden action trace through `env`; compare everything recorded.

    obs_tol None -> observation must be bit-equal float64 (oracle);
    otherwise |diff| <= obs_tol (float32 device path: 1e-6 per north_star).
    """
    obs_rows = {int(s): k for k, s in enumerate(g["obs_step"])}
    J = env.jobs
    for i, a in enumerate(g["action"]):
        a = int(a)
        where = f"row {i} action {a}"
        if a == -2:
            env.reset()
        elif a == -1:
            env.increase_time_step()
        else:
            _, r, d, _, _ = env.step(a)
            if obs_tol is None:
                assert r == g["reward"][i], f"{where}: reward {r} != {g['reward'][i]}"
            else:
                assert abs(r - g["reward"][i]) <= obs_tol, f"{where}: reward {r} vs {g['reward'][i]}"
            assert bool(d) == bool(g["done"][i]), f"{where}: done"
        if check_state:
            check_env_against_row(env, g, i, where)
        if i in obs_rows:
            want = g["obs"][obs_rows[i]]
            got = np.asarray(env.state, dtype=np.float64)
            if a == -1:
                # direct increase_time_step(): the reference refreshes column 0 only inside
                # step()/reset() (jss_env.py:130); compare the other six columns
                got, want = got[:, 1:], want[:, 1:]
            if obs_tol is None:
                assert (got == want).all(), f"{where}: obs not bit-equal, max diff {np.abs(got - want).max()}"
            else:
                assert np.abs(got - want).max() <= obs_tol, f"{where}: obs max diff {np.abs(got - want).max()}"
    return env
