"""The single-env view with the reference's exact surface: ``JssEnv`` (JSSEnv/envs/jss_env.py: ``reset() -> obs``,
``step(a) -> (obs, reward, done, False, {})``, ``get_legal_actions()``, ``increase_time_step()``, ``render()`` and the public
attributes its tests and dispatching rules read) over a ``BatchedJssEnv`` of one env, and ``make('jss-v1', ...)``.
"""
from __future__ import annotations

import os

import numpy as np

from . import _abi
from .env import BatchedJssEnv
from .instances import resolve_instance


class _Snap:
    """Env 0 of a host snapshot, decoded lazily: ``step()`` needs the observation, mask, reward, done and the error
    bits; everything else (the per-job arrays the reference exposes as attributes) is unpacked when it is read."""

    def __init__(self, t, J, M, decode):
        self.t, self.J, self.M, self.c, self.decode = t, J, M, {}, decode

    def __contains__(self, k):
        return k in self.c

    def __setitem__(self, k, v):
        self.c[k] = v

    def __getitem__(self, k):
        c = self.c
        if k in c:
            return c[k]
        t, J, M = self.t, self.J, self.M
        if k in ("clock", "err", "noop_flag", "episode", "step_in_episode"):
            hdr = t["env_header"][0]
            st = int(hdr[_abi.H_STATUS])
            c.update(clock=int(hdr[_abi.H_CLOCK]), err=st & 0xFF, noop_flag=bool(st & _abi.STATUS_NOOP),
                     episode=int(hdr[_abi.H_EPISODE]), step_in_episode=int(hdr[_abi.H_STEP]))
        elif k in ("job_state", "next_op", "next2_op", "blocked"):
            js, nxt, nxt2 = self.decode(t["job_state"][0], 0)
            c.update(job_state=js, next_op=nxt, next2_op=nxt2, blocked=(js[7] & 2) != 0)
        elif k == "tm":
            c[k] = (t["machine_state"][0, :M].astype(np.int64) if "machine_state" in t else
                    BatchedJssEnv.clocks_from_jobs(self["job_state"], M))
        elif k == "mask":
            c[k] = t["action_mask"][0, :J + 1] != 0
        elif k == "obs":
            c[k] = t["real_obs"][0, :J].copy()
        elif k == "reward":
            c[k] = float(t["reward"][0])
        elif k == "done":
            c[k] = bool(t["done"][0])
        elif k == "makespan":
            c[k] = int(t["makespan"][0])
        elif k == "counters":
            c[k] = t["counters"][0].copy()
        elif k == "mask_padding":
            c[k] = t["action_mask"][0, J + 1:].copy()
        elif k == "obs_padding":
            c[k] = t["real_obs"][0, J:].copy()
        else:
            raise KeyError(k)
        return c[k]


def gymnasium_base(which: str = "Env"):
    """``gymnasium.Env`` (or ``gymnasium.vector.VectorEnv``) when gymnasium is importable and really has that class,
    ``object`` otherwise: the reference's env IS a ``gym.Env`` (jss_env.py:14) and gymnasium's wrappers assert
    ``isinstance(env, gymnasium.Env)``; without gymnasium the package works all the same."""
    try:
        import gymnasium
        base = getattr(gymnasium, "Env", None) if which == "Env" else getattr(getattr(gymnasium, "vector", None), "VectorEnv", None)
        return base if isinstance(base, type) else object
    except Exception:
        return object


class JssEnv(gymnasium_base("Env")):
    """Drop-in for ``JSSEnv.envs.jss_env.JssEnv``: one env (B = 1) on the GPU (a ``gymnasium.Env`` when gymnasium exists).

    Same constructor argument (``env_config={'instance_path': ...}``, default ta80 as at
    jss_env.py:35-38), same methods and return shapes, same public attributes (NumPy, pulled
    from the device on access).  Differences, all outside what the reference defines:
    a job action outside the mask raises ``ValueError`` (the reference corrupts its counters
    silently); the observation is float32.  ``device='cpu'`` runs the same env on the
    host-core twin (no GPU needed).
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
        import datetime
        import random
        self.start_timestamp = datetime.datetime.now().timestamp()         # :70 (render's time origin, :672)
        self.colors = [tuple(random.random() for _ in range(3)) for _ in range(self.machines)]   # :99-101, used by render :686
        self._alloc_log = []        # the job actions of this episode in call order (next_jobs: who queued an event first)
        self._alloc_log_ok = True   # False once the episode was advanced by a device-side rollout (no per-call log)
        # on the GPU the env's arena (state + outputs, ~1 KB) lives in page-locked host memory the kernel works on in
        # place: step() = one launch + one stream synchronisation, nothing is copied (JSSENV_AMD_HOST_ARENA=0: device
        # memory and one device -> host copy per step, the round-3 form)
        self._b = BatchedJssEnv([inst], batch=1, device=device, _backend=_backend,
                                host_arena=os.environ.get("JSSENV_AMD_HOST_ARENA", "1") != "0")
        self._cache = None
        self._act = np.zeros(1, dtype=np.int32)
        # remaining work of job j from op k on (MWR / LWR / CR on the host): suffix sums of the durations
        self._remaining = np.cumsum(inst.duration[:, ::-1], axis=1)[:, ::-1].astype(np.int64)
        self._act_pinned, self._zero_copy, self._fast = None, False, None
        be = self._b.backend
        if getattr(be, "name", "") == "hip":
            self._act_pinned = be.torch.zeros(1, dtype=be.torch.int32).pin_memory()
            # JSSENV_AMD_ZEROCOPY=1: jss_step reads the action straight from the pinned host word (one H2D copy less per step)
            self._zero_copy = os.environ.get("JSSENV_AMD_ZEROCOPY", "0") == "1"
        try:  # spaces only when gymnasium is importable (jss_env.py:97, :112-119)
            import gymnasium as gym
            self.action_space = gym.spaces.Discrete(self.jobs + 1)
            self.observation_space = gym.spaces.Dict({
                "action_mask": gym.spaces.Box(0, 1, shape=(self.jobs + 1,)),
                "real_obs": gym.spaces.Box(low=0.0, high=1.0, shape=(self.jobs, 7), dtype=float),
            })
        except ImportError:  # gymnasium is optional
            self.action_space = self.observation_space = None

    # -- host mirror of the device state ---------------------------------------------------
    def _h(self):
        if self._cache is None:                    # one device -> host copy per step (the env's arena), decoded lazily
            self._cache = _Snap(self._b.host_tensors(), self.jobs, self.machines, self._b.decode_jobs)
        return self._cache

    def _solution(self):
        h = self._h()
        if "solution" not in h:                    # the start-time table comes over only when somebody reads it
            b = self._b
            h["solution"] = b.backend.numpy(b.solution[0])[:self.jobs, :self.machines].astype(np.int64)
        return h["solution"]

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
    solution = property(lambda s: s._solution())
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
    def next_jobs(self):
        """The reference's list parallel to ``next_time_step`` (jss_env.py:56, :156, :453, :518): entry i is the job whose
        allocation QUEUED event time ``next_time_step[i]`` -- the first job allocated to finish at that time; a later job
        that finishes at the same time adds nothing (:450-453).  Derived: the running ops (start = solution[j][todo[j]],
        end = start + duration > now) sorted by end time; among ops that end together the one allocated first -- the
        earlier start, then the earlier ``step`` call of this episode (the facade logs its job actions; after a device-side
        rollout there is no log and the lower job index stands in)."""
        now = self.current_time_step
        left, todo = self.time_until_finish_current_op_jobs, self.todo_time_step_job
        running = [j for j in range(self.jobs) if left[j] > 0]
        if not running:
            return []
        sol = self._solution()
        rank = {}
        if self._alloc_log_ok:
            for pos, j in enumerate(self._alloc_log):
                rank[j] = pos                                # the LAST allocation of job j is its running op
        first = {}
        for j in running:
            end = int(now + left[j])
            key = (int(sol[j][todo[j]]), rank.get(j, len(self._alloc_log) + j), j)
            if end not in first or key < first[end][0]:
                first[end] = (key, j)
        return [first[t][1] for t in sorted(first)]

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
        self._alloc_log, self._alloc_log_ok = [], True
        return self._obs()

    def _raise_for(self, err, action=None):
        """Turn the kernel's per-env error bits into the reference's exceptions.  The bits are sticky on the
        device, so they are cleared here: every offending call raises, not only the first of an episode."""
        if not err:
            return
        self._b.clear_errors()
        self._cache = None
        if err & _abi.ERR_ILLEGAL_ACTION and self._alloc_log and self._alloc_log[-1] == action:
            self._alloc_log.pop()                  # the job was not allocated
        if err & _abi.ERR_BAD_ACTION:
            raise IndexError(f"action {action} out of range for {self.jobs} jobs")
        if err & _abi.ERR_NOPE_IDLE:
            raise IndexError("pop from empty list")  # what the reference raises at jss_env.py:517
        if err & _abi.ERR_ILLEGAL_ACTION:
            raise ValueError(f"job {action} is not a legal action")

    def step(self, action):
        """jss_env.py:403-481."""
        action = int(action)
        if 0 <= action < self.jobs:
            self._alloc_log.append(action)         # (an action the mask refuses raises below and queues nothing: popped there)
        if getattr(self._b, "host_arena", False):  # GPU, arena in host memory: the action word is part of it
            return self._step_host_arena(action)
        elif self._act_pinned is not None:         # GPU: the action goes out through a pinned word, nothing is allocated
            self._act_pinned[0] = action
            b = self._b
            if self._zero_copy:                    # the kernel reads the pinned word itself
                b.step_raw(self._act_pinned.data_ptr())
            else:
                b._act_in.copy_(self._act_pinned, non_blocking=True)
                b.step_raw(b._act_in.data_ptr())
        else:
            self._act[0] = action
            self._b.step(self._act)
        self._cache = None
        h = self._h()
        self._raise_for(h["err"], action)
        if h["done"]:                                                       # :649-652
            self.last_time_step = h["clock"]
            self.last_solution = self._solution()
        return self._obs(), h["reward"], h["done"], False, {}

    def _step_host_arena(self, action):
        """step() when the env's arena is page-locked host memory: the action is written into it, ONE launch works on it
        in place, one stream synchronisation (through the library: it also surfaces a kernel fault), and the results are
        read where they lie.  Everything that does not change between calls is bound once."""
        fp = self._fast
        if fp is None:
            b, be = self._b, self._b.backend
            if not b._is_reset:
                raise RuntimeError("call reset() before step()")
            lib, (d, s, o) = be.lib, b._refs()
            views = b.host_tensors()
            # torch's current stream of the env's device as a raw handle: the private accessor costs 0.3 us, building a
            # torch.cuda.Stream object to ask for its .cuda_stream 3 us (tools/gpu_facade_floor.py)
            raw = getattr(be.torch._C, "_cuda_getCurrentRawStream", None)
            index = be.device.index
            stream_of = (lambda: raw(index)) if raw is not None else (lambda: be.torch.cuda.current_stream(be.device).cuda_stream)
            fp = self._fast = (b._act_in.numpy(), lib.jss_step, lib.jss_sync_check, d, s, o, b._act_in.data_ptr(),
                               stream_of, views, lib, b)
        act, jss_step, sync_check, d, s, o, a_ptr, stream_of, views, lib, b = fp
        act[0] = action
        stream = stream_of()
        rc = jss_step(d, s, a_ptr, o, stream)
        if rc == 0:
            rc = sync_check(stream)
        if rc:
            _abi.check(lib, rc, "jss_step")
        h = self._cache = _Snap(views, self.jobs, self.machines, b.decode_jobs)
        err = h["err"]
        if err:
            self._raise_for(err, action)
        done = h["done"]
        if done:                                                            # :649-652
            self.last_time_step = h["clock"]
            self.last_solution = self._solution()
        return {"real_obs": h["obs"], "action_mask": h["mask"]}, h["reward"], done, False, {}

    def increase_time_step(self):
        """jss_env.py:495-637 -- public in the reference and called directly by its tests."""
        hole = int(self._b.backend.numpy(self._b.increase_time_step())[0])
        self._cache = None
        self._raise_for(self._h()["err"])
        return hole

    def render(self, mode: str = "human"):
        """Gantt chart of ``solution`` (jss_env.py:655-693); needs pandas + plotly on the host."""
        from .render import gantt
        return gantt(self)

    def close(self):
        """gymnasium.Env.close(): waits for the env's outstanding device work."""
        self._b.synchronize()

    def _run_rule(self, kind, explore: float = 0.0, seed=None):
        """One whole episode of a dispatching rule, rule + step fused on the device (dispatching.py:55-75 with the
        exploration drawn from the counter RNG).  Returns (total reward, makespan) like ``run_episode``."""
        b = self._b
        if seed is not None:
            b.seed = int(seed)
        b.reset()
        b.zero_counters()
        self._alloc_log, self._alloc_log_ok = [], False      # the device picks the actions: no per-call log (next_jobs)
        chunk = self.jobs * self.machines + 16
        for _ in range(64):
            b.rollout(kind, n_iter=chunk, autoreset=False, explore=explore)
            self._cache = None
            h = self._h()
            if h["done"]:
                break
        else:
            raise RuntimeError("episode did not finish")
        self._raise_for(h["err"])
        self.last_time_step = h["clock"]
        self.last_solution = self._solution()
        return float(h["counters"][3]) / self.max_time_op, h["clock"]

    def _rule_best(self, kind, legal_actions, due_date_factor: float = 1.5):
        """arg-best of a dispatching rule over the legal jobs from the host snapshot of this step (dispatching.py's
        strict comparisons: the lowest job index wins ties); -1 when no job is legal.  Same selectors as the device's
        jss_policy (tests hold the two to each other)."""
        legal = np.asarray(legal_actions[:self.jobs], dtype=bool)
        if not legal.any():
            return -1
        h = self._h()
        if "job_state" not in h and kind in ("FIFO", "MOR", "LOR"):
            # these rank on ONE word of the job records: read it where it lies instead of decoding every record
            raw = h.t["job_state"][0][:self.jobs]
            if kind == "FIFO":
                key = raw[:, _abi.FC_IDLE_LAST if self._b.compact else _abi.F_IDLE_LAST].astype(np.float64)
            else:
                todo = raw[:, 0] & (_abi.FC_TODO_MASK if self._b.compact else _abi.TODO_MASK)
                key = ((self.machines - todo) * (1 if kind == "MOR" else -1)).astype(np.float64)
            key[~legal] = -np.inf
            return int(np.argmax(key))
        js = h["job_state"]
        todo = js[_abi.F_TODO]
        if kind == "FIFO":
            key = js[_abi.F_IDLE_LAST]
        elif kind == "SPT":
            key = -(js[_abi.F_CUR] & 0xFFFF)
        elif kind in ("MOR", "LOR"):
            key = (self.machines - todo) * (1 if kind == "MOR" else -1)
        else:
            rem = self._remaining[np.arange(self.jobs), np.minimum(todo, self.machines - 1)]
            if kind == "MWR":
                key = rem
            elif kind == "LWR":
                key = -rem
            elif kind == "CR":     # smallest (1.5 * job length - now) / remaining work, as the reference's floats (:391-398)
                with np.errstate(divide="ignore"):
                    key = -np.where(rem > 0, (self._remaining[:, 0] * due_date_factor - self._h()["clock"]) / np.maximum(rem, 1), np.inf)
            else:
                raise KeyError(kind)
        key = np.where(legal, key, -np.inf)
        return int(np.argmax(key))           # first maximum = lowest index among ties

    # on-device action selectors for the dispatching module
    def _policy(self, kind, cr_factor=None):
        return int(self._b.backend.numpy(self._b.policy(kind, cr_factor=cr_factor))[0])


def make(env_id: str = "jss-v1", env_config=None, **kwargs):
    """``gym.make('jss-v1', env_config=...)`` without gymnasium (JSSEnv/__init__.py:6-9)."""
    if env_id != "jss-v1":
        raise ValueError(f"unknown env id {env_id!r}; this package registers 'jss-v1'")
    return JssEnv(env_config=env_config, **kwargs)
