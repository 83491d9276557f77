"""ctypes view of oracle/libjss_oracle.so with the reference's attribute names.

TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/jss_oracle.h).
``OracleEnv`` mirrors the public surface of the reference ``JssEnv``
(JSSEnv/envs/jss_env.py:121-181, :403-481, :495-637) so a test written against
the reference runs unchanged against the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjss_oracle.so")
_lib = None

POLICY_IDS = {"random": 0, "FIFO": 1, "SPT": 2, "MWR": 3, "LWR": 4, "MOR": 5, "LOR": 6, "CR": 7}


def build_oracle(force: bool = False) -> str:
    src = os.path.join(_HERE, "jss_oracle.c")
    hdr = os.path.join(_HERE, "jss_oracle.h")
    stale = (not os.path.isfile(_LIB_PATH)) or any(
        os.path.isfile(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libjss_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    lib = C.CDLL(_LIB_PATH)
    P = C.c_void_p
    i32p, u8p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
    lib.orc_create.restype = P
    lib.orc_create.argtypes = [C.c_int, C.c_int, i32p, i32p]
    lib.orc_destroy.argtypes = [P]
    lib.orc_reset.argtypes = [P]
    lib.orc_step.restype = C.c_int
    lib.orc_step.argtypes = [P, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.orc_increase_time_step.restype = C.c_int
    lib.orc_increase_time_step.argtypes = [P, C.POINTER(C.c_int)]
    for name in ("jobs", "machines", "current_time_step", "nb_legal_actions", "nb_machine_legal",
                 "next_time_step_len", "err", "max_time_op", "max_time_jobs", "sum_op"):
        f = getattr(lib, "orc_" + name)
        f.restype, f.argtypes = C.c_int, [P]
    lib.orc_last_reward_numerator.restype, lib.orc_last_reward_numerator.argtypes = C.c_long, [P]
    for name in ("todo_time_step_job", "needed_machine_jobs", "time_until_finish_current_op_jobs",
                 "total_perform_op_time_jobs", "total_idle_time_jobs", "idle_time_jobs_last_op",
                 "time_until_available_machine", "solution", "next_time_step"):
        f = getattr(lib, "orc_" + name)
        f.restype, f.argtypes = i32p, [P]
    for name in ("legal_actions", "action_illegal_no_op", "machine_legal", "illegal_actions"):
        f = getattr(lib, "orc_" + name)
        f.restype, f.argtypes = u8p, [P]
    lib.orc_state.restype, lib.orc_state.argtypes = f64p, [P]
    lib.orc_rng_u32.restype = C.c_uint32
    lib.orc_rng_u32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    lib.orc_policy.restype = C.c_int
    lib.orc_policy.argtypes = [P, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]
    lib.orc_policy_explore.restype = C.c_int
    lib.orc_policy_explore.argtypes = [P, C.c_int, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]
    lib.orc_rollout.restype = C.c_long
    lib.orc_rollout.argtypes = [P, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_double)]
    lib.orc_rollout_batch.restype = C.c_int
    lib.orc_rollout_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P, C.c_int, C.c_uint64, C.c_uint64, P,
                                      C.c_uint32, C.c_long, C.c_int, C.c_int] + [P] * 11
    _lib = lib
    return lib


def rng_u32(seed: int, env_id: int, episode: int, step: int) -> int:
    return int(_load().orc_rng_u32(seed, env_id, episode, step))


class OracleEnv:
    """Single env with the reference's attribute names, backed by the C oracle."""

    def __init__(self, instance, strict: bool = False):
        lib = _load()
        self._lib = lib
        self.instance = instance
        self.strict = bool(strict)
        m = np.ascontiguousarray(instance.machine, dtype=np.int32)
        d = np.ascontiguousarray(instance.duration, dtype=np.int32)
        i32p = C.POINTER(C.c_int32)
        self._h = lib.orc_create(instance.jobs, instance.machines, m.ctypes.data_as(i32p), d.ctypes.data_as(i32p))
        if not self._h:
            raise ValueError("orc_create failed")
        self.jobs, self.machines = instance.jobs, instance.machines
        self.instance_matrix = instance.instance_matrix
        self.max_time_op = lib.orc_max_time_op(self._h)
        self.max_time_jobs = lib.orc_max_time_jobs(self._h)
        self.sum_op = lib.orc_sum_op(self._h)
        self.episode = 0
        self.step_in_episode = 0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.orc_destroy(h)
            self._h = None

    # -- arrays (copies, so callers can keep history) ----------------------
    def _i32(self, name, n):
        return np.ctypeslib.as_array(getattr(self._lib, "orc_" + name)(self._h), shape=(n,)).copy()

    def _u8(self, name, n):
        return np.ctypeslib.as_array(getattr(self._lib, "orc_" + name)(self._h), shape=(n,)).astype(bool)

    todo_time_step_job = property(lambda s: s._i32("todo_time_step_job", s.jobs))
    needed_machine_jobs = property(lambda s: s._i32("needed_machine_jobs", s.jobs))
    time_until_finish_current_op_jobs = property(lambda s: s._i32("time_until_finish_current_op_jobs", s.jobs))
    total_perform_op_time_jobs = property(lambda s: s._i32("total_perform_op_time_jobs", s.jobs))
    total_idle_time_jobs = property(lambda s: s._i32("total_idle_time_jobs", s.jobs))
    idle_time_jobs_last_op = property(lambda s: s._i32("idle_time_jobs_last_op", s.jobs))
    time_until_available_machine = property(lambda s: s._i32("time_until_available_machine", s.machines))
    solution = property(lambda s: s._i32("solution", s.jobs * s.machines).reshape(s.jobs, s.machines))
    legal_actions = property(lambda s: s._u8("legal_actions", s.jobs + 1))
    action_illegal_no_op = property(lambda s: s._u8("action_illegal_no_op", s.jobs))
    machine_legal = property(lambda s: s._u8("machine_legal", s.machines))
    illegal_actions = property(lambda s: s._u8("illegal_actions", s.jobs * s.machines).reshape(s.machines, s.jobs))
    current_time_step = property(lambda s: s._lib.orc_current_time_step(s._h))
    nb_legal_actions = property(lambda s: s._lib.orc_nb_legal_actions(s._h))
    nb_machine_legal = property(lambda s: s._lib.orc_nb_machine_legal(s._h))
    err = property(lambda s: s._lib.orc_err(s._h))
    last_reward_numerator = property(lambda s: int(s._lib.orc_last_reward_numerator(s._h)))

    @property
    def next_time_step(self):
        n = self._lib.orc_next_time_step_len(self._h)
        return [] if n == 0 else list(self._i32("next_time_step", n))

    @property
    def state(self):
        return np.ctypeslib.as_array(self._lib.orc_state(self._h), shape=(self.jobs * 7,)).copy().reshape(self.jobs, 7)

    # -- reference API -----------------------------------------------------
    def _obs(self):
        return {"real_obs": self.state, "action_mask": self.legal_actions}

    def get_legal_actions(self):
        return self.legal_actions

    def reset(self):
        self._lib.orc_reset(self._h)
        self.episode += 1
        self.step_in_episode = 0
        return self._obs()

    def step(self, action):
        r, d = C.c_double(0.0), C.c_int(0)
        self.last_rc = self._lib.orc_step(self._h, int(action), int(self.strict), C.byref(r), C.byref(d))
        self.step_in_episode += 1
        return self._obs(), r.value, bool(d.value), False, {}

    def increase_time_step(self):
        hole = C.c_int(0)
        rc = self._lib.orc_increase_time_step(self._h, C.byref(hole))
        if rc < 0:
            raise IndexError("pop from empty list")  # what the reference raises (jss_env.py:517)
        return hole.value

    def policy(self, kind, seed=0, env_id=0, episode=0, step=0, explore=0.0):
        k = POLICY_IDS[kind] if isinstance(kind, str) else int(kind)
        q16 = int(round(explore * 65536))
        return int(self._lib.orc_policy_explore(self._h, k, seed, q16, env_id, episode, step))

    def rollout(self, kind, seed, env_id, iterations, episode=0, step_in_episode=0):
        """Returns dict(steps, episodes, makespan_sum, reward_sum, episode, step_in_episode)."""
        k = POLICY_IDS[kind] if isinstance(kind, str) else int(kind)
        ep, st = C.c_uint32(episode), C.c_uint32(step_in_episode)
        counters = (C.c_long * 3)(0, 0, 0)
        rs = C.c_double(0.0)
        n = self._lib.orc_rollout(self._h, k, seed, env_id, C.byref(ep), C.byref(st), iterations, counters, C.byref(rs))
        return dict(steps=int(n), episodes=int(counters[1]), makespan_sum=int(counters[2]), reward_sum=rs.value,
                    episode=int(ep.value), step_in_episode=int(st.value))


def rollout_batch(packed, batch, kind, seed, iterations, table_of_env=None, env_id_base=0, env_ids=None, explore=0.0,
                  autoreset=True, threads=0, with_obs=True):
    """``iterations`` x (policy + step) for ``batch`` independent envs from a fresh reset, all on the C oracle
    (OpenMP over envs): the checker for EVERY env of a full-size device batch.  ``packed`` is a
    jssenv_amd.instances.PackedBatch (only its arrays are read here).  Returns NumPy arrays padded like the
    device tensors: clock, episode, step_in_episode (B,), job_fields (B, 6, jmax) in golden_util.JOB_FIELDS
    order, tm (B, mmax), solution (B, jmax, mmax), mask (B, jmax + 1), blocked (B, jmax), counters (B, 4) =
    steps / episodes / makespan sum / reward numerator sum, obs (B, jmax, 7) float64, err (B,)."""
    lib = _load()
    ops = np.ascontiguousarray(packed.ops)
    n_tables, jmax, mmax = ops.shape
    mach = np.ascontiguousarray(ops >> 16, dtype=np.int32)
    dur = np.ascontiguousarray(ops & 0xFFFF, dtype=np.int32)
    jobs = np.ascontiguousarray(packed.jobs, dtype=np.int32)
    machines = np.ascontiguousarray(packed.machines, dtype=np.int32)
    B = int(batch)
    toe = None if table_of_env is None else np.ascontiguousarray(table_of_env, dtype=np.int32)
    ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int64)
    out = {"clock": np.zeros(B, np.int32), "episode": np.zeros(B, np.int32), "step_in_episode": np.zeros(B, np.int32),
           "job_fields": np.zeros((B, 6, jmax), np.int32), "tm": np.zeros((B, mmax), np.int32),
           "solution": np.zeros((B, jmax, mmax), np.int32), "mask": np.zeros((B, jmax + 1), np.uint8),
           "blocked": np.zeros((B, jmax), np.uint8), "counters": np.zeros((B, 4), np.int64),
           "obs": np.zeros((B, jmax, 7), np.float64) if with_obs else None, "err": np.zeros(B, np.int32)}
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)   # noqa: E731
    k = POLICY_IDS[kind] if isinstance(kind, str) else int(kind)
    rc = lib.orc_rollout_batch(B, n_tables, jmax, mmax, ptr(jobs), ptr(machines), ptr(mach), ptr(dur), ptr(toe), k,
                               int(seed), int(env_id_base), ptr(ids), int(round(explore * 65536)), int(iterations),
                               int(bool(autoreset)), int(threads),
                               *[ptr(out[n]) for n in ("clock", "episode", "step_in_episode", "job_fields", "tm", "solution",
                                                       "mask", "blocked", "counters", "obs", "err")])
    if rc != 0:
        raise RuntimeError(f"orc_rollout_batch failed: {rc}")
    return out
