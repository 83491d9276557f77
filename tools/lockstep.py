"""Tooling (build container only): drive the live reference and the C oracle in
lockstep and compare every public attribute after every call."""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import refload  # noqa: E402
from jssenv_amd import instances as I  # noqa: E402
from oracle import OracleEnv  # noqa: E402

INT_ATTRS = ("todo_time_step_job", "needed_machine_jobs", "time_until_finish_current_op_jobs",
             "total_perform_op_time_jobs", "total_idle_time_jobs", "idle_time_jobs_last_op",
             "time_until_available_machine", "solution")
BOOL_ATTRS = ("legal_actions", "action_illegal_no_op", "machine_legal", "illegal_actions")
SCALARS = ("current_time_step", "nb_legal_actions", "nb_machine_legal")


def compare(ref, orc, where=""):
    for a in SCALARS:
        assert int(getattr(ref, a)) == int(getattr(orc, a)), f"{where}: {a} {getattr(ref, a)} != {getattr(orc, a)}"
    for a in INT_ATTRS:
        r, o = np.asarray(getattr(ref, a)), np.asarray(getattr(orc, a))
        assert r.shape == o.shape and (r == o).all(), f"{where}: {a}\nref={r}\norc={o}"
    for a in BOOL_ATTRS:
        r, o = np.asarray(getattr(ref, a)).astype(bool), np.asarray(getattr(orc, a)).astype(bool)
        assert r.shape == o.shape and (r == o).all(), f"{where}: {a}\nref={r}\norc={o}"
    assert list(ref.next_time_step) == list(orc.next_time_step), f"{where}: queue {ref.next_time_step} {orc.next_time_step}"
    rs, os_ = np.asarray(ref.state, dtype=np.float64), orc.state
    assert (rs == os_).all(), f"{where}: state (float64 exact)\n{rs - os_}"


def make_pair(name):
    JssEnv, _ = refload.load_reference()
    ref = JssEnv({"instance_path": refload.reference_instance_path(name)})
    orc = OracleEnv(I.builtin_instance(name))
    ref.reset()
    orc.reset()
    compare(ref, orc, f"{name} reset")
    return ref, orc


def step_both(ref, orc, action, where):
    o1, r1, d1, _, _ = ref.step(action)
    o2, r2, d2, _, _ = orc.step(action)
    assert r1 == r2, f"{where}: reward {r1} != {r2}"
    assert bool(d1) == bool(d2), f"{where}: done"
    compare(ref, orc, where)
    return o1, r1, d1


def published_sequences():
    """{instance: (per-machine job sequences, 'advance'|'nope')} pulled out of the
    reference's tests/test_solutions.py (data literals only)."""
    path = os.path.join(refload.REFERENCE_ROOT, "tests", "test_solutions.py")
    src = open(path).read()
    tree = ast.parse(src)
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name.startswith("test_optimum_"):
            inst = node.name[len("test_optimum_"):]
            seq, ub = None, None
            for sub in ast.walk(node):
                if isinstance(sub, ast.Assign) and getattr(sub.targets[0], "id", "") == "solution_sequence":
                    seq = ast.literal_eval(sub.value)
            seg = ast.get_source_segment(src, node)
            mode = "advance" if "env.increase_time_step()" in seg else "nope"
            # the asserted makespan is the only 3-4 digit literal compared with current_time_step
            for sub in ast.walk(node):
                if isinstance(sub, ast.Call) and getattr(sub.func, "attr", "") == "assertEqual" and len(sub.args) == 2:
                    a0, a1 = sub.args
                    if isinstance(a0, ast.Attribute) and a0.attr == "current_time_step" and isinstance(a1, ast.Constant) and a1.value:
                        ub = int(a1.value)
            out[inst] = (seq, mode, ub)
    return out


def replay_published(env, seq, mode, on_action):
    """The driver loop of tests/test_solutions.py:36-70, recording the flat action
    trace: job id, J for NOPE, -1 for a direct increase_time_step()."""
    machine_nb, job_nb = len(seq), len(seq[0])
    index_machine = [0] * machine_nb
    done = False
    while not done:
        no_op = True
        for machine in range(machine_nb):
            if done:
                break
            if env.machine_legal[machine] and index_machine[machine] < job_nb:
                a = seq[machine][index_machine[machine]]
                if env.needed_machine_jobs[a] == machine and env.legal_actions[a]:
                    no_op = False
                    done = on_action(a)
                    index_machine[machine] += 1
        if no_op and not done:
            done = on_action(-1 if mode == "advance" else env.jobs)
    assert sum(index_machine) == machine_nb * job_nb
