"""Host side of the batched MI355X job-shop environment.

``BatchedJssEnv`` owns the per-env state as PyTorch-ROCm tensors laid out over a
batch axis (include/jss_hip.h describes the layout) and advances it with the
HIP kernels of ``libjss_hip.so`` through the C ABI.  ``JssEnv`` is the
single-env view with the reference's exact surface
(JSSEnv/envs/jss_env.py: ``reset() -> obs``, ``step(a) -> (obs, reward, done,
False, {})``, ``get_legal_actions()``, ``increase_time_step()`` and the public
attributes its tests and dispatching rules read).

There is no CPU path: constructing an env without a GPU (or without the built
extension) raises.  torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Union

import numpy as np

from . import _abi
from .instances import Instance, PackedBatch, pack_batch, resolve_instance

_NP = {"int32": np.int32, "int64": np.int64, "uint8": np.uint8, "float32": np.float32}


class HipBackend:
    """Device memory = torch.cuda tensors; kernels = libjss_hip.so on torch's current stream."""

    name = "hip"

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("jssenv_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU path")
        path = _abi.library_path()
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        self.torch = torch
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.lib = _abi.bind(C.CDLL(path))

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=getattr(self.torch, dtype), device=self.device)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def ptr(self, x):
        return 0 if x is None else x.data_ptr()

    def numpy(self, x):
        return x.detach().cpu().numpy()

    def stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def sync(self):
        self.torch.cuda.current_stream(self.device).synchronize()

    def as_device(self, x, dtype):
        t = self.torch.as_tensor(x, device=self.device)
        return t.to(getattr(self.torch, dtype)).contiguous()

    def shift_right(self, x, n):
        return x >> n

    def where(self, cond, a, b):
        return self.torch.where(cond, self.torch.as_tensor(a, dtype=b.dtype, device=self.device), b)

    def copy_into(self, dst, src_numpy):
        dst.copy_(self.torch.from_numpy(np.ascontiguousarray(src_numpy)).to(self.device))


class BatchedJssEnv:
    """B independent job-shop envs on one GPU.

    instances  one instance spec (name, path or Instance) shared by the whole batch, or a
               sequence of them.  With a sequence, env i runs instance ``i % len(instances)``
               unless ``table_of_env`` says otherwise.
    batch      number of envs (defaults to len(instances)).
    """

    def __init__(self, instances, batch: Optional[int] = None, device=None, env_id_base: int = 0,
                 table_of_env: Optional[Sequence[int]] = None, seed: int = 0, _backend=None):
        self.backend = be = _backend if _backend is not None else HipBackend(device)
        if isinstance(instances, (str, os.PathLike, Instance)):
            instances = [instances]
        self.instances = [resolve_instance(i) for i in instances]
        n = len(self.instances)
        if n == 0:
            raise ValueError("need at least one instance")
        self.batch = B = int(batch) if batch is not None else n
        if B < 1:
            raise ValueError("batch must be >= 1")
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        pk: PackedBatch = pack_batch(self.instances)
        self.packed = pk
        self.jmax, self.mmax, self.n_tables = pk.jmax, pk.mmax, n
        if table_of_env is None and n != 1 and n != B:
            table_of_env = np.arange(B) % n
        self.table_of_env_host = (np.zeros(B, dtype=np.int32) if n == 1 else
                                  np.arange(B, dtype=np.int32) if table_of_env is None else
                                  np.asarray(table_of_env, dtype=np.int32))
        if self.table_of_env_host.shape != (B,) or self.table_of_env_host.min() < 0 or self.table_of_env_host.max() >= n:
            raise ValueError("table_of_env must hold B indices into instances")
        self.jobs_per_env = pk.jobs[self.table_of_env_host]
        self.machines_per_env = pk.machines[self.table_of_env_host]

        # instance tables
        self._ops = be.from_numpy(pk.ops)
        self._jobs = be.from_numpy(pk.jobs)
        self._machines = be.from_numpy(pk.machines)
        self._max_time_op = be.from_numpy(pk.max_time_op)
        self._max_time_jobs = be.from_numpy(pk.max_time_jobs)
        self._sum_op = be.from_numpy(pk.sum_op)
        self._table_of_env = None if table_of_env is None else be.from_numpy(self.table_of_env_host)
        # compact 16-bit copy of the op tables (machine << 10 | duration) when every duration fits 10 bits:
        # halves the bytes a kernel stages per table, which matters when every env has its own instance
        self._ops16 = None
        if n > 1 and int(pk.max_time_op.max()) <= 1023:
            ops16 = (((pk.ops >> 16) << 10) | (pk.ops & 0xFFFF)).astype(np.uint16)
            self._ops16 = be.from_numpy(ops16.view(np.int16))
        # state (include/jss_hip.h JssState)
        J, M = self.jmax, self.mmax
        self.env_header = be.zeros((B, 4), "int32")          # clock, episode, step_in_episode, status
        self.job_state = be.zeros((B, J, _abi.NF), "int32")  # one 32-byte record per job
        self.machine_state = be.zeros((B, M), "int32")
        self.solution = be.zeros((B, J, M), "int32")
        self.counters = be.zeros((B, 4), "int64")
        # outputs (JssOut)
        self.real_obs = be.zeros((B, J, 7), "float32")
        self.action_mask = be.zeros((B, J + 1), "uint8")
        self.reward = be.zeros((B,), "float32")
        self.done = be.zeros((B,), "uint8")
        self.makespan = be.zeros((B,), "int32")

        p = be.ptr
        self._desc = _abi.JssDesc(B, J, M, n, p(self._ops), p(self._jobs), p(self._machines), p(self._max_time_op),
                                  p(self._max_time_jobs), p(self._sum_op), p(self._table_of_env), self.env_id_base, None,
                                  p(self._ops16))
        self._state = _abi.JssState(p(self.env_header), p(self.job_state), p(self.machine_state), p(self.solution),
                                    p(self.counters))
        self._out = _abi.JssOut(p(self.real_obs), p(self.action_mask), p(self.reward), p(self.done), p(self.makespan))
        self._is_reset = False

    def assign_instances(self, env_indices, table_indices):
        """Give envs ``env_indices`` the instances ``table_indices`` (indices into the ``instances`` this batch
        was built with) and reset exactly those envs -- per-env instance resampling between episodes.  The batch
        must have been built with an explicit or modular env -> instance map (more than one instance)."""
        if self._table_of_env is None:
            raise ValueError("this batch has a fixed env -> instance map (one shared instance, or one instance per env)")
        env_indices = np.asarray(env_indices, dtype=np.int64).reshape(-1)
        table_indices = np.asarray(table_indices, dtype=np.int32).reshape(-1)
        if env_indices.shape != table_indices.shape:
            raise ValueError("env_indices and table_indices must have the same length")
        if env_indices.size and (env_indices.min() < 0 or env_indices.max() >= self.batch or
                                 table_indices.min() < 0 or table_indices.max() >= self.n_tables):
            raise ValueError("index out of range")
        self.table_of_env_host[env_indices] = table_indices
        self.jobs_per_env = self.packed.jobs[self.table_of_env_host]
        self.machines_per_env = self.packed.machines[self.table_of_env_host]
        self.backend.copy_into(self._table_of_env, self.table_of_env_host)
        which = np.zeros(self.batch, dtype=np.uint8)
        which[env_indices] = 1
        return self.reset(which=which)

    def set_env_ids(self, ids):
        """Explicit global env ids (int64, one per env) keying the per-env RNG streams; used by
        BucketedJssEnv, whose buckets hold non-contiguous slices of the global batch."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64))
        if ids.shape != (self.batch,):
            raise ValueError("env ids must have shape (B,)")
        self._env_ids = self.backend.from_numpy(ids)
        self._desc.env_ids = self.backend.ptr(self._env_ids)

    # -- raw ABI handles (bench.py launches through these) -------------------------------
    @property
    def lib(self):
        return self.backend.lib

    def _obs(self):
        # like the reference (jss_env.py:130-134) the returned arrays are the env's own buffers,
        # overwritten by the next call; clone to keep history
        return {"real_obs": self.real_obs, "action_mask": self.action_mask}

    def _mask_arg(self, which):
        if which is None:
            return None
        w = self.backend.as_device(which, "uint8")
        if tuple(w.shape) != (self.batch,):
            raise ValueError("mask must have shape (B,)")
        return w

    # -- API -----------------------------------------------------------------------------
    def reset(self, which=None):
        """reset() of jss_env.py:145-181 for every env (or those with which[i] != 0). Returns the obs dict."""
        be = self.backend
        w = self._mask_arg(which)
        _abi.check(be.lib, be.lib.jss_reset(C.byref(self._desc), C.byref(self._state), C.byref(self._out), be.ptr(w),
                                            be.stream()), "jss_reset")
        self._is_reset = True
        return self._obs()

    def step(self, actions, autoreset: bool = False):
        """step() of jss_env.py:403-481, one action per env (J = NOPE, -1 = leave the env untouched).

        Returns (obs, reward (B,) float32, done (B,) uint8, truncated=False, info={}).
        autoreset=True gives gymnasium.vector "next-step" semantics: an env that reported done on the
        previous call is reset by this call instead of being stepped (its action is ignored, reward 0,
        done 0) -- one extra masked jss_reset launch, no host synchronisation."""
        if not self._is_reset:
            raise RuntimeError("call reset() before step()")
        be = self.backend
        a = be.as_device(actions, "int32")
        if tuple(a.shape) != (self.batch,):
            raise ValueError("actions must have shape (B,)")
        if autoreset:
            was_done = self.done.clone() if hasattr(self.done, "clone") else self.done.copy()
            a = be.where(was_done != 0, -1, a)
        _abi.check(be.lib, be.lib.jss_step(C.byref(self._desc), C.byref(self._state), be.ptr(a), C.byref(self._out),
                                           be.stream()), "jss_step")
        if autoreset:
            _abi.check(be.lib, be.lib.jss_reset(C.byref(self._desc), C.byref(self._state), C.byref(self._out),
                                                be.ptr(was_done), be.stream()), "jss_reset")
        return self._obs(), self.reward, self.done, False, {}

    def increase_time_step(self, which=None):
        """increase_time_step() of jss_env.py:495-637 per env; returns hole_planning (B,) int32."""
        be = self.backend
        w = self._mask_arg(which)
        hole = be.zeros((self.batch,), "int32")
        _abi.check(be.lib, be.lib.jss_advance(C.byref(self._desc), C.byref(self._state), be.ptr(w), be.ptr(hole),
                                              C.byref(self._out), be.stream()), "jss_advance")
        return hole

    def policy(self, kind: Union[str, int] = "random", seed: Optional[int] = None, explore: float = 0.0):
        """Per-env action from the on-device selectors (random masked, FIFO, SPT, MWR, LWR, MOR, LOR)."""
        be = self.backend
        k = _abi.POLICY[kind] if isinstance(kind, str) else int(kind)
        out = be.zeros((self.batch,), "int32")
        _abi.check(be.lib, be.lib.jss_policy(C.byref(self._desc), C.byref(self._state), k,
                                             self.seed if seed is None else int(seed), int(round(explore * 65536)),
                                             be.ptr(out), be.stream()), "jss_policy")
        return out

    def rollout(self, kind: Union[str, int] = "random", n_iter: int = 1, seed: Optional[int] = None,
                autoreset: bool = True, explore: float = 0.0):
        """n_iter x (policy + step) per env in ONE launch (state stays in registers)."""
        if not self._is_reset:
            raise RuntimeError("call reset() before rollout()")
        be = self.backend
        k = _abi.POLICY[kind] if isinstance(kind, str) else int(kind)
        flags = _abi.ROLLOUT_AUTORESET if autoreset else 0
        _abi.check(be.lib, be.lib.jss_rollout(C.byref(self._desc), C.byref(self._state), C.byref(self._out), k,
                                              self.seed if seed is None else int(seed), int(round(explore * 65536)),
                                              int(n_iter), flags, be.stream()), "jss_rollout")
        return self._obs(), self.reward, self.done, False, {}

    def synchronize(self):
        self.backend.sync()

    # -- state views with the reference's names (device arrays, batch first) ---------------
    @property
    def current_time_step(self):
        return self.env_header[:, _abi.H_CLOCK]

    clock = current_time_step

    @property
    def episode(self):
        return self.env_header[:, _abi.H_EPISODE]

    @property
    def step_in_episode(self):
        return self.env_header[:, _abi.H_STEP]

    @property
    def err(self):
        return self.env_header[:, _abi.H_STATUS] & 0xFF

    @property
    def todo_time_step_job(self):
        return self.job_state[:, :, _abi.F_TODO]

    @property
    def needed_machine_jobs(self):
        return self.backend.shift_right(self.job_state[:, :, _abi.F_CUR], 16)

    @property
    def time_until_finish_current_op_jobs(self):
        return self.job_state[:, :, _abi.F_LEFT]

    @property
    def total_perform_op_time_jobs(self):
        return self.job_state[:, :, _abi.F_PERF]

    @property
    def total_idle_time_jobs(self):
        return self.job_state[:, :, _abi.F_IDLE]

    @property
    def idle_time_jobs_last_op(self):
        return self.job_state[:, :, _abi.F_IDLE_LAST]

    @property
    def time_until_available_machine(self):
        return self.machine_state

    @property
    def legal_actions(self):
        return self.action_mask

    @property
    def action_illegal_no_op(self):
        return self.backend.shift_right(self.job_state[:, :, _abi.F_FLAGS], 1) & 1

    def counter_totals(self):
        """Device tensor [4]: env steps, finished episodes, sum of makespans, sum of reward numerators."""
        return self.counters.sum(0)

    def zero_counters(self):
        self.counters[...] = 0

    def stats(self):
        """Host dict of the per-env counters summed over the batch."""
        c = self.backend.numpy(self.counters).sum(axis=0)
        return {"steps": int(c[0]), "episodes": int(c[1]), "makespan_sum": int(c[2]), "reward_num_sum": int(c[3])}

    # -- checkpoint / resume: the state is a handful of tensors --------------------
    _STATE_TENSORS = ("env_header", "job_state", "machine_state", "solution", "counters", "real_obs", "action_mask",
                      "reward", "done", "makespan")

    def state_dict(self):
        """Host copy of everything needed to resume: state + last outputs + the batch description."""
        n = self.backend.numpy
        d = {k: n(getattr(self, k)) for k in self._STATE_TENSORS}
        d["meta"] = {"batch": self.batch, "jmax": self.jmax, "mmax": self.mmax, "seed": self.seed,
                     "env_id_base": self.env_id_base, "instances": [i.name for i in self.instances],
                     "table_of_env": self.table_of_env_host.copy(), "ops": self.packed.ops.copy()}
        return d

    def load_state_dict(self, d):
        m = d["meta"]
        if (m["batch"], m["jmax"], m["mmax"]) != (self.batch, self.jmax, self.mmax) or \
                not np.array_equal(m["ops"], self.packed.ops) or not np.array_equal(m["table_of_env"], self.table_of_env_host):
            raise ValueError("checkpoint belongs to a different batch (shape or instances differ)")
        for k in self._STATE_TENSORS:
            self.backend.copy_into(getattr(self, k), d[k])
        self.seed, self._is_reset = int(m["seed"]), True

    def host_state(self, i: int = 0):
        """Everything about env i as NumPy, sliced to its true (J, M)."""
        n = self.backend.numpy
        J, M = int(self.jobs_per_env[i]), int(self.machines_per_env[i])
        js = n(self.job_state[i])[:J].astype(np.int64).T          # (NF, J): rows = JSS_F_* words
        hdr = n(self.env_header[i])
        return {
            "jobs": J, "machines": M,
            "clock": int(hdr[_abi.H_CLOCK]),
            "job_state": js,
            "tm": n(self.machine_state[i])[:M].astype(np.int64),
            "mask": n(self.action_mask[i])[:J + 1].astype(bool),
            "blocked": (js[_abi.F_FLAGS] & _abi.FLAG_BLOCKED) != 0,
            "solution": n(self.solution[i])[:J, :M].astype(np.int64),
            "obs": n(self.real_obs[i])[:J].astype(np.float32),
            "obs_padding": n(self.real_obs[i])[J:],
            "reward": float(n(self.reward[i:i + 1])[0]),
            "done": bool(n(self.done[i:i + 1])[0]),
            "err": int(hdr[_abi.H_STATUS]) & 0xFF,
            "noop_flag": bool(int(hdr[_abi.H_STATUS]) & _abi.STATUS_NOOP),
            "makespan": int(n(self.makespan[i:i + 1])[0]),
            "episode": int(hdr[_abi.H_EPISODE]),
            "step_in_episode": int(hdr[_abi.H_STEP]),
        }


class JssEnv:
    """Drop-in for ``JSSEnv.envs.jss_env.JssEnv``: one env (B = 1) on the GPU.

    Same constructor argument (``env_config={'instance_path': ...}``, default ta80 as at
    jss_env.py:35-38), same methods and return shapes, same public attributes (NumPy, pulled
    from the device on access).  Differences, all outside what the reference defines:
    a job action outside the mask raises ``ValueError`` (the reference corrupts its counters
    silently); the observation is float32.
    """

    metadata = {"render_modes": ["human"]}

    def __init__(self, env_config=None, device=None, _backend=None):
        if env_config is None:
            env_config = {"instance_path": "ta80"}                         # jss_env.py:35-38
        inst = resolve_instance(env_config["instance_path"])
        self.instance = inst
        self.jobs, self.machines = inst.jobs, inst.machines                # :77
        self.instance_matrix = inst.instance_matrix                        # :78,:85
        self.jobs_length = inst.jobs_length                                # :87
        self.max_time_op = inst.max_time_op                                # :86
        self.max_time_jobs = inst.max_time_jobs                            # :89
        self.sum_op = inst.sum_op                                          # :88
        self.last_time_step = float("inf")                                 # :53
        self.last_solution = None                                          # :52
        self._b = BatchedJssEnv([inst], batch=1, device=device, _backend=_backend)
        self._cache = None
        self._err_seen = 0
        try:  # spaces only when gymnasium is importable (jss_env.py:97, :112-119)
            import gymnasium as gym
            self.action_space = gym.spaces.Discrete(self.jobs + 1)
            self.observation_space = gym.spaces.Dict({
                "action_mask": gym.spaces.Box(0, 1, shape=(self.jobs + 1,)),
                "real_obs": gym.spaces.Box(low=0.0, high=1.0, shape=(self.jobs, 7), dtype=float),
            })
        except Exception:  # pragma: no cover - gymnasium is optional
            self.action_space = self.observation_space = None

    # -- host mirror of the device state ---------------------------------------------------
    def _h(self):
        if self._cache is None:
            self._cache = self._b.host_state(0)
        return self._cache

    def _obs(self):
        h = self._h()
        return {"real_obs": h["obs"], "action_mask": h["mask"]}

    current_time_step = property(lambda s: s._h()["clock"])
    todo_time_step_job = property(lambda s: s._h()["job_state"][_abi.F_TODO])
    needed_machine_jobs = property(lambda s: s._h()["job_state"][_abi.F_CUR] >> 16)
    time_until_finish_current_op_jobs = property(lambda s: s._h()["job_state"][_abi.F_LEFT])
    total_perform_op_time_jobs = property(lambda s: s._h()["job_state"][_abi.F_PERF])
    total_idle_time_jobs = property(lambda s: s._h()["job_state"][_abi.F_IDLE])
    idle_time_jobs_last_op = property(lambda s: s._h()["job_state"][_abi.F_IDLE_LAST])
    time_until_available_machine = property(lambda s: s._h()["tm"])
    solution = property(lambda s: s._h()["solution"])
    legal_actions = property(lambda s: s._h()["mask"])
    action_illegal_no_op = property(lambda s: s._h()["blocked"])
    state = property(lambda s: s._h()["obs"])
    err = property(lambda s: s._h()["err"])

    @property
    def nb_legal_actions(self):            # stored counter in the reference; a popcount here
        return int(self.legal_actions[:-1].sum())

    @property
    def machine_legal(self):               # reference :173-179, :463, :632-634
        out = np.zeros(self.machines, dtype=bool)
        need = self.needed_machine_jobs
        out[need[self.legal_actions[:-1]]] = True
        return out

    @property
    def nb_machine_legal(self):
        return int(self.machine_legal.sum())

    @property
    def next_time_step(self):              # the reference's sorted event list (:449-453, :517)
        tm = self.time_until_available_machine
        return sorted({int(self.current_time_step + v) for v in tm if v > 0})

    @property
    def illegal_actions(self):             # (M, J) matrix of the reference (:171, :427, :464-467)
        out = np.zeros((self.machines, self.jobs), dtype=bool)
        need, bl = self.needed_machine_jobs, self.action_illegal_no_op
        for j in range(self.jobs):
            if bl[j] and need[j] >= 0:
                out[need[j], j] = True
        return out

    # -- reference API -----------------------------------------------------------------------
    def get_legal_actions(self):           # jss_env.py:136-143
        return self.legal_actions

    def reset(self, *, seed=None, options=None):
        """jss_env.py:145-181 -- returns the observation dict only (no info tuple)."""
        self._b.reset()
        self._cache = None
        self._err_seen = 0
        return self._obs()

    def step(self, action):
        """jss_env.py:403-481."""
        action = int(action)
        self._b.step(np.asarray([action], dtype=np.int32))
        self._cache = None
        h = self._h()
        new_err = h["err"] & ~self._err_seen
        self._err_seen = h["err"]
        if new_err & _abi.ERR_BAD_ACTION:
            raise IndexError(f"action {action} out of range for {self.jobs} jobs")
        if new_err & _abi.ERR_NOPE_IDLE:
            raise IndexError("pop from empty list")  # what the reference raises at jss_env.py:517
        if new_err & _abi.ERR_ILLEGAL_ACTION:
            raise ValueError(f"job {action} is not a legal action")
        if h["done"]:                                                       # :649-652
            self.last_time_step = h["clock"]
            self.last_solution = h["solution"]
        return self._obs(), h["reward"], h["done"], False, {}

    def increase_time_step(self):
        """jss_env.py:495-637 -- public in the reference and called directly by its tests."""
        hole = self._b.increase_time_step()
        self._cache = None
        new_err = self._h()["err"] & ~self._err_seen
        self._err_seen = self._h()["err"]
        if new_err & _abi.ERR_NOPE_IDLE:
            raise IndexError("pop from empty list")
        return int(self._b.backend.numpy(hole)[0])

    def render(self, mode: str = "human"):
        """Gantt chart of ``solution`` (jss_env.py:655-693); needs pandas + plotly on the host."""
        from .render import gantt
        return gantt(self)

    # on-device action selectors for the dispatching module
    def _policy(self, kind):
        return int(self._b.backend.numpy(self._b.policy(kind))[0])


def make(env_id: str = "jss-v1", env_config=None, **kwargs):
    """``gym.make('jss-v1', env_config=...)`` without gymnasium (JSSEnv/__init__.py:6-9)."""
    if env_id != "jss-v1":
        raise ValueError(f"unknown env id {env_id!r}; this package registers 'jss-v1'")
    return JssEnv(env_config=env_config, **kwargs)
