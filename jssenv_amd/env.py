"""Host side of the batched MI355X job-shop environment.

``BatchedJssEnv`` owns the per-env state as PyTorch-ROCm tensors laid out over a batch axis (include/jss_hip.h describes the
layout) and advances it with the HIP kernels of ``libjss_hip.so`` through the C ABI.  Memory and library handles come from a
backend (``jssenv_amd.backends``: ``HipBackend`` by default -- no silent CPU path -- or, on request, ``CpuBackend``, the
host-core twin); the single-env view with the reference's surface is ``jssenv_amd.facade.JssEnv``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Union

import numpy as np

from . import _abi
from .backends import _NULL_CTX, CpuBackend, HipBackend, _carve_numpy, make_backend  # noqa: F401  (re-exported: the historical home)
from .instances import Instance, PackedBatch, pack_batch, resolve_instance


class BatchedJssEnv:
    """B independent job-shop envs on one GPU.

    instances  one instance spec (name, path or Instance) shared by the whole batch, or a
               sequence of them.  With a sequence of n instances and a batch of another size the envs are dealt onto the
               instances so that every instance gets as many envs as ``i % n`` would give it; WHICH env gets which instance
               is ``order``'s business and is always readable from ``table_of_env_host`` (``instance_of_env(i)``).
               ``batch == n`` (or no batch): env i runs instances[i]; ``table_of_env`` overrides everything.
    batch      number of envs (defaults to len(instances)).
    kernel     "auto" (packed kernel when every env fits a 16/32-lane group) or "wave"
               (one wavefront per env); a per-env-object choice carried in JssDesc.
    order      how a batch deals its envs onto a LIST of instances (``batch != len(instances)``, no ``table_of_env``):
               "by_shape"     a ragged population in ONE set of padded tensors, stepped by class-specialised bodies.  The envs
                              are dealt onto the instances class by class (J, M <= 16, <= 32, J < 64, the rest: env i <-
                              sorted_by_class[...] -- every class a contiguous range of the batch) and reset / policy / step /
                              rollout(n_iter=1) / rollout_steps run as ONE grid over the classes (jss_multi_*, JssDesc.jclass:
                              4 or 2 envs per wavefront for the small classes, one wavefront per env for the others) on the
                              padded rows, instead of the one kernel the padded extents would pick.  Same tensors, same layout,
                              same results per env; an env keeps its class for life (assign_instances: within the class).
               "interleaved"  env i <- instances[i % n]: every env on the kernel of the PADDED shape (BASELINE config 5 as
                              rounds 1-5 ran it by default: 0.49 of the roofline against 0.57 by shape).
               None           (default) "by_shape" when the list holds more than one shape class, else "interleaved" -- the
                              fast form is what a caller gets without asking; ``env.order`` says which one it got.  Chosen
                              this way, the by-class deal costs nothing in generality: an ``assign_instances`` that moves an
                              env to another class (an error under an explicit "by_shape") quietly hands the batch to the
                              padded extents' kernel from then on (``steps_by_shape_class`` turns False).
    """

    def __init__(self, instances, batch: Optional[int] = None, device=None, env_id_base: int = 0,
                 table_of_env: Optional[Sequence[int]] = None, seed: int = 0, kernel: Optional[str] = None,
                 compact: Optional[bool] = None, host_arena: bool = False, records: Optional[str] = None,
                 order: Optional[str] = None, _backend=None):
        self._owns_backend = _backend is None
        self.backend = be = _backend if _backend is not None else make_backend(device)
        if isinstance(instances, PackedBatch):
            pk = instances
            self.instances = None
        else:
            if isinstance(instances, (str, os.PathLike, Instance)):
                instances = [instances]
            self.instances = [resolve_instance(i) for i in instances]
            if len(self.instances) == 0:
                raise ValueError("need at least one instance")
            pk = pack_batch(self.instances)
        n = int(pk.ops.shape[0])
        self.batch = B = int(batch) if batch is not None else n
        if B < 1:
            raise ValueError("batch must be >= 1")
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        self.packed = pk
        self.jmax, self.mmax, self.n_tables = pk.jmax, pk.mmax, n
        if order not in (None, "by_shape", "interleaved"):
            raise ValueError("order must be None, 'by_shape' or 'interleaved'")
        self._class_of_table = None
        # shape class of every instance (BucketedJssEnv's classes; a 64-job instance inside rows wider than 64 goes with
        # the two-jobs-per-lane class: its NOPE flag lives at byte 64 of the mask row)
        wide = 63 if pk.jmax > 64 else 64
        cls = np.array([0 if (j <= 16 and m <= 16) else 1 if (j <= 32 and m <= 32) else 2 if j <= wide else 3
                        for j, m in zip(pk.jobs.tolist(), pk.machines.tolist())])
        self._order_given = order is not None
        if order is None:
            # the default: a list of instances of more than one shape class, dealt out by this constructor, is dealt out by
            # class (a caller that passes no order gets the fast form); everything else keeps i % n
            dealt_here = n > 1 and n != B and table_of_env is None
            order = "by_shape" if (dealt_here and len(set(cls.tolist())) > 1) else "interleaved"
        self.order = order
        if order == "by_shape":
            if n == 1 or table_of_env is not None:
                raise ValueError("order='by_shape' deals the envs onto a LIST of instances itself (no table_of_env)")
            by_class = np.argsort(cls, kind="stable")                    # instances class by class, given order inside a class
            counts = np.bincount(np.arange(B) % n, minlength=n)[by_class]   # envs per instance as i % n would deal them
            table_of_env = np.repeat(by_class, counts)
            self._class_of_table = cls
        if table_of_env is None and n != 1 and n != B:
            table_of_env = np.arange(B) % n
        self.table_of_env_host = (np.zeros(B, dtype=np.int32) if n == 1 else
                                  np.arange(B, dtype=np.int32) if table_of_env is None else
                                  np.asarray(table_of_env, dtype=np.int32))
        if self.table_of_env_host.shape != (B,) or self.table_of_env_host.min() < 0 or self.table_of_env_host.max() >= n:
            raise ValueError("table_of_env must hold B indices into instances")
        self.jobs_per_env = pk.jobs[self.table_of_env_host]
        self.machines_per_env = pk.machines[self.table_of_env_host]
        # job records: 16-byte compact records (no cached ops: they are read from the ONE op table, which every workgroup
        # has in LDS) whenever the batch shares one instance; 32-byte records otherwise
        if records is not None:                       # explicit layout: "compact" (16 B), "medium" (24 B) or "full" (32 B)
            if records not in ("compact", "medium", "full"):
                raise ValueError("records must be 'compact', 'medium' or 'full'")
            compact = records == "compact"
        self.compact = (n == 1) if compact is None else bool(compact)
        if self.compact and n != 1:
            raise ValueError("compact job records need a batch that shares one instance")
        self.kernel = kernel if kernel is not None else getattr(be, "default_kernel", "auto")
        if self.kernel not in _abi.KERNEL:
            raise ValueError(f"kernel must be one of {list(_abi.KERNEL)}")
        # 24-byte medium records (the three cached ops in 21 bits each, no machine clocks) for batches of different instances.
        # The library takes them for every shape with machines <= 32 (both kernel flavours); by default they are used
        # where they measure faster than full records: the 16-lane groups (jobs, machines <= 16: +8-10 % on 15 x 15; with 32-lane
        # groups the three 8-byte accesses and the unpacking cost more than the bytes save, -2 % on 20 x 20 --
        # profiles/README.md) and the shapes of the one-wavefront-per-env flavour when every instance of the batch is of the same
        # class (33..64 jobs: 50 x 20 x 65 536 +7 %, x 8 192 +-0; more than 64: ta71-80 x 4 096 +6-9 % --
        # profiles/r06_misc/medium_records_wave.txt; a batch that mixes shape classes keeps full records, the fused grid has no
        # medium body for these shapes).  `compact=False` / records="full" asks for full records everywhere.
        fits = n != 1 and pk.mmax <= 32                 # 21-bit ops: machines <= 32, any number of jobs, either kernel flavour
        if records == "medium" and not fits:
            raise ValueError("medium job records need a batch of different instances with machines <= 32")
        if records is None and compact is None and fits and getattr(be, "default_records", None) == "medium":
            records = "medium"                         # (test backends: the layout travels with the backend like default_kernel)
        one_wave_class = set(cls.tolist()) in ({2}, {3})
        self.medium = records == "medium" or (records is None and compact is None and fits and self.kernel.startswith("auto")
                                              and ((pk.jmax <= 16 and pk.mmax <= 16) or one_wave_class))
        self.record_ints = _abi.NFC if self.compact else _abi.NFM if self.medium else _abi.NF
        self.no_clocks = self.compact or self.medium           # time_until_available_machine is derived, not stored

        with be.on_device():
            # instance tables
            self._ops = be.from_numpy(pk.ops)
            self._rem = be.from_numpy(pk.rem)
            self._inst = be.from_numpy(pk.inst)
            self._table_of_env = None if table_of_env is None else be.from_numpy(self.table_of_env_host)
            self._env_ids = None
            # state (include/jss_hip.h JssState), outputs (JssOut) and the per-call scratch outputs: ONE allocation,
            # carved into 256-byte aligned views -- nothing is allocated inside the stepping calls, and a small batch
            # (the B = 1 facade) comes back to the host in a single copy (host_state)
            J, M = self.jmax, self.mmax
            specs = [("env_header", (B, _abi.NH), "int32"),     # clock, episode, step_in_episode, status
                     ("env_const", (B, _abi.NC), "int32"),      # the env's instance constants, written by reset (JSS_C_*)
                     ("job_state", (B, J, self.record_ints), "int32"),   # one 32- (or compact: 16-) byte record per job
                     ("machine_state", (B, M), "int32"),      # (not with compact records: derived, see machine_state)
                     ("counters", (B, 4), "int64"),
                     ("real_obs", (B, J, 7), "float32"),
                     ("action_mask", (B, J + 1), "uint8"),
                     ("reward", (B,), "float32"), ("done", (B,), "uint8"), ("makespan", (B,), "int32"),
                     ("_actions_out", (B,), "int32"), ("_hole", (B,), "int32"), ("_act_buf", (B,), "int32"),
                     ("_act_in", (B,), "int32"), ("_which_in", (B,), "uint8")]
            if self.no_clocks:
                specs = [sp for sp in specs if sp[0] != "machine_state"]
            self._layout, off = {}, 0
            for name, shape, dtype in specs:
                self._layout[name] = (off, shape, dtype)
                off += (int(np.prod(shape)) * np.dtype(dtype).itemsize + 255) & ~255
            # host_arena (tiny batches: the B = 1 facade): state and outputs live in page-locked HOST memory that the kernels
            # read and write in place over PCIe -- a step is one launch and one stream synchronisation, no copy either way
            self.host_arena = bool(host_arena) and hasattr(be, "zeros_pinned")
            self._arena = be.zeros_pinned((off,), "uint8") if self.host_arena else be.zeros((off,), "uint8")
            carve = getattr(be, "carve", None) or (lambda a, o, sh, dt: _carve_numpy(a, o, sh, dt))
            for name, (o, shape, dtype) in self._layout.items():
                setattr(self, name, carve(self._arena, o, shape, dtype))
            self._host_arena = None
            self._host_views = None
            self._stream_events = {}                             # fork / join events of rollout_steps, owned by this env
            if self.host_arena:                                  # (atomics across PCIe are not something to lean on: the
                self.counters = be.zeros((B, 4), "int64")        #  counters stay in device memory)
            self.solution = be.zeros((B, J, M), "int32")         # the one large, rarely read tensor stays on its own
            if hasattr(be, "scalar"):
                be.scalar(_abi.ACTION_RESET, "int32")

        p = be.ptr
        self._desc = _abi.JssDesc(B, J, M, n, p(self._ops), p(self._rem), p(self._inst), p(self._table_of_env), None,
                                  self.env_id_base, _abi.KERNEL[self.kernel], int(getattr(be, "threads", 0)),
                                  int(pk.jobs.min()), self.record_ints, 0.0, 0, 0)
        self._state = _abi.JssState(p(self.env_header), p(self.env_const), p(self.job_state),
                                    None if self.no_clocks else p(self.machine_state), p(self.solution),
                                    p(self.counters))
        self._out = _abi.JssOut(p(self.real_obs), p(self.action_mask), p(self.reward), p(self.done), p(self.makespan))
        self._is_reset = False
        self._session = None                                     # an open StepSession: every other call raises meanwhile
        self._classes = None
        if self._class_of_table is not None:
            self._build_class_views()

    @property
    def steps_by_shape_class(self) -> bool:
        """True while reset / policy / step / rollout(n_iter=1) / rollout_steps run the class-specialised bodies (order 'by_shape',
        no assign_instances across classes since); False: the kernel of the padded extents."""
        return self._classes is not None

    def instance_of_env(self, i: int) -> int:
        """Index (into the ``instances`` this batch was built with) of the instance env ``i`` runs -- ``table_of_env_host[i]``:
        the one place that says how the constructor (``order``), ``table_of_env`` or ``assign_instances`` dealt the envs."""
        return int(self.table_of_env_host[i])

    def _build_class_views(self):
        """order='by_shape': one JssDesc / JssState / JssOut per shape class, each describing a contiguous range of THIS
        batch's padded tensors (every per-env pointer moved to the class's first env; the instance tables shared).  Rebuilt
        whenever something they copy changes: ``set_env_ids`` (the RNG keys), ``assign_instances`` (a class's jmin / extents)."""
        be, B, J, M = self.backend, self.batch, self.jmax, self.mmax
        cls_env = self._class_of_table[self.table_of_env_host]
        assert (np.diff(cls_env) >= 0).all()
        p = be.ptr
        descs, states, outs, spans = [], [], [], []
        for k in range(4):
            idx = np.flatnonzero(cls_env == k)
            if idx.size == 0:
                continue
            a, b = int(idx[0]), int(idx[-1]) + 1
            assert b - a == idx.size
            jc, mc = int(self.jobs_per_env[a:b].max()), int(self.machines_per_env[a:b].max())
            off = lambda t, per_env: p(t) + a * per_env * t.dtype.itemsize if t is not None else None     # noqa: E731
            d = _abi.JssDesc(b - a, J, M, self.n_tables, p(self._ops), p(self._rem), p(self._inst), off(self._table_of_env, 1),
                             off(self._env_ids, 1),      # explicit global env ids (set_env_ids) key the RNG of a class like the batch's
                             self.env_id_base + a, _abi.KERNEL[self.kernel], int(getattr(be, "threads", 0)),
                             int(self.jobs_per_env[a:b].min()), self.record_ints, float(self._desc.cr_factor), jc, mc)
            st = _abi.JssState(off(self.env_header, _abi.NH), off(self.env_const, _abi.NC), off(self.job_state, J * self.record_ints),
                               None if self.no_clocks else off(self.machine_state, M), off(self.solution, J * M), off(self.counters, 4))
            o = _abi.JssOut(off(self.real_obs, J * 7), off(self.action_mask, J + 1), off(self.reward, 1), off(self.done, 1),
                            off(self.makespan, 1))
            descs.append(d), states.append(st), outs.append(o), spans.append((a, b, k))
        n = len(descs)
        D, S, O = C.POINTER(_abi.JssDesc), C.POINTER(_abi.JssState), C.POINTER(_abi.JssOut)
        # The single-set calls that loop over steps (rollout(n_iter > 1), trajectory, steps) run ONE launch over all the classes
        # that fit one job per lane (they differ in nothing but the lane group the fused grid gives them) and one over the
        # two-jobs-per-lane class: `ranges` = their (desc, state, out, first env) -- the first one's jclass says "below 64 jobs".
        narrow = [i for i, (_, _, k) in enumerate(spans) if k < 3]
        ranges = []
        if len(narrow) > 1:
            a, b = spans[narrow[0]][0], spans[narrow[-1]][1]
            d0 = descs[narrow[0]]
            dm = _abi.JssDesc(b - a, J, M, self.n_tables, d0.ops, d0.rem, d0.inst, d0.table_of_env, d0.env_ids, d0.env_id_base, d0.kernel,
                              d0.threads, int(self.jobs_per_env[a:b].min()), self.record_ints, float(self._desc.cr_factor),
                              int(self.jobs_per_env[a:b].max()), int(self.machines_per_env[a:b].max()))
            ranges.append((dm, states[narrow[0]], outs[narrow[0]], a))
            ranges += [(descs[i], states[i], outs[i], spans[i][0]) for i in range(n) if i not in narrow]
        else:
            ranges = [(descs[i], states[i], outs[i], spans[i][0]) for i in range(n)]
        self._class_ranges = ranges
        self._classes = {"n": n, "spans": spans, "keep": (descs, states, outs),
                         "sets": ((D * n)(*[C.pointer(x) for x in descs]), (S * n)(*[C.pointer(x) for x in states]),
                                  (O * n)(*[C.pointer(x) for x in outs]))}

    def _class_ptrs(self, t):
        """(void* * n): where each class's rows of the (B, ...) tensor `t` start"""
        be = self.backend
        per_env = int(np.prod(t.shape[1:])) if len(t.shape) > 1 else 1
        return (C.c_void_p * self._classes["n"])(*[be.ptr(t) + a * per_env * t.dtype.itemsize for a, _, _ in self._classes["spans"]])

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
        if self._class_of_table is not None and not np.array_equal(self._class_of_table[self.table_of_env_host[env_indices]],
                                                                   self._class_of_table[table_indices]):
            if self._order_given:
                raise ValueError("order='by_shape': an env keeps its shape class for life -- give it an instance of the same class")
            # The constructor chose the by-class deal on its own (order=None): the caller did not sign up for its one restriction.
            # From here on every env is stepped by the kernel of the padded extents -- same tensors, same layout, same results
            # (the reset below rewrites every row of the moved envs' padded blocks); the class ranges are given up.
            self._classes, self._class_of_table = None, None
        self.table_of_env_host[env_indices] = table_indices
        self.jobs_per_env = self.packed.jobs[self.table_of_env_host]
        self.machines_per_env = self.packed.machines[self.table_of_env_host]
        self.backend.copy_into(self._table_of_env, self.table_of_env_host)
        if self._classes is not None:
            self._build_class_views()                        # a class's smallest J / extents may have changed
        which = np.zeros(self.batch, dtype=np.uint8)
        which[env_indices] = 1
        return self.reset(which=which)

    def set_env_ids(self, ids):
        """Explicit global env ids (int64, one per env) keying the per-env RNG streams; used by
        BucketedJssEnv, whose buckets hold non-contiguous slices of the global batch."""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64))
        if ids.shape != (self.batch,):
            raise ValueError("env ids must have shape (B,)")
        with self.backend.on_device():
            self._env_ids = self.backend.from_numpy(ids)
        self._desc.env_ids = self.backend.ptr(self._env_ids)
        if self._classes is not None:
            self._build_class_views()                        # the class views carry their own (offset) copy of the pointer

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
        return self._stage(self._which_in, which, "uint8")

    def _stage(self, buf, x, dtype):
        be = self.backend
        if hasattr(be, "stage"):
            return be.stage(buf, x)
        w = be.as_device(x, dtype)
        if tuple(w.shape) != (self.batch,):
            raise ValueError(f"expected shape ({self.batch},)")
        return w

    def _refs(self):
        if self._session is not None and not self._session.closed:
            raise RuntimeError("a step session is open on this env: the state is resident in its kernel -- close() it first")
        return C.byref(self._desc), C.byref(self._state), C.byref(self._out)

    # -- API -----------------------------------------------------------------------------
    def reset(self, which=None):
        """reset() of jss_env.py:145-181 for every env (or those with which[i] != 0). Returns the obs dict."""
        be = self.backend
        d, s, o = self._refs()
        with be.on_device():
            w = self._mask_arg(which)
            if self._classes is not None:        # order='by_shape': one grid over the shape classes
                rc = be.lib.jss_multi_reset(self._classes["n"], *self._classes["sets"], None if w is None else self._class_ptrs(w), be.stream())
                _abi.check(be.lib, rc, "jss_multi_reset")
            else:
                _abi.check(be.lib, be.lib.jss_reset(d, s, o, be.ptr(w), be.stream()), "jss_reset")
        self._is_reset = True
        return self._obs()

    def step(self, actions, autoreset: bool = False):
        """step() of jss_env.py:403-481, one action per env (J = NOPE, -1 = leave the env untouched).

        Returns (obs, reward (B,) float32, done (B,) uint8, truncated=False, info={}).
        autoreset=True gives gymnasium.vector "next-step" semantics: an env that reported done on the
        previous call is reset by this call instead of being stepped (its action is ignored, reward 0,
        done 0) -- same launch (jss_step_autoreset), no host synchronisation, no extra kernel."""
        if not self._is_reset:
            raise RuntimeError("call reset() before step()")
        be = self.backend
        d, s, o = self._refs()
        with be.on_device():
            a = self._stage(self._act_in, actions, "int32")   # the caller's int32 tensor itself, or a copy into our buffer
            # autoreset: envs that reported done last time are reset instead of stepped, in the same launch (the kernel
            # looks at the done flags itself: jss_step_autoreset)
            if self._classes is not None:
                cs = self._classes["sets"]
                rc = be.lib.jss_multi_step(self._classes["n"], cs[0], cs[1], self._class_ptrs(a), cs[2],
                                           _abi.ROLLOUT_AUTORESET if autoreset else 0, be.stream())
                _abi.check(be.lib, rc, "jss_multi_step")
            else:
                fn = be.lib.jss_step_autoreset if autoreset else be.lib.jss_step
                _abi.check(be.lib, fn(d, s, be.ptr(a), o, be.stream()), "jss_step")
        return self._obs(), self.reward, self.done, False, {}

    def step_raw(self, actions_ptr: int):
        """jss_step with a caller-owned int32[B] action buffer given by address (device memory, or pinned host memory the
        device can read): no staging, nothing allocated.  The B = 1 facade's path."""
        if not self._is_reset:
            raise RuntimeError("call reset() before step()")
        be = self.backend
        d, s, o = self._refs()
        with be.on_device():
            _abi.check(be.lib, be.lib.jss_step(d, s, actions_ptr, o, be.stream()), "jss_step")

    def increase_time_step(self, which=None):
        """increase_time_step() of jss_env.py:495-637 per env; returns hole_planning (B,) int32 (the env's own
        buffer, overwritten by the next call)."""
        if not self._is_reset:
            raise RuntimeError("call reset() before increase_time_step()")
        be = self.backend
        d, s, o = self._refs()
        with be.on_device():
            w = self._mask_arg(which)
            _abi.check(be.lib, be.lib.jss_advance(d, s, be.ptr(w), be.ptr(self._hole), o, be.stream()), "jss_advance")
        return self._hole

    def policy(self, kind: Union[str, int] = "random", seed: Optional[int] = None, explore: float = 0.0,
               cr_factor: Optional[float] = None):
        """Per-env action from the on-device selectors (random masked, FIFO, SPT, MWR, LWR, MOR, LOR, CR).
        Returns the env's own (B,) int32 action buffer (overwritten by the next policy() call).
        ``cr_factor``: CriticalRatio(due_date_factor=...) with ANY positive float (dispatching.py:337-360) -- the selector then
        evaluates the reference's float64 expression itself (JSS_POLICY_CR_F64); without it "CR" is the default 1.5, and
        ``_abi.cr_kind(f)`` codes the factors p / 2^k that also run inside the fused rollouts."""
        if not self._is_reset:
            raise RuntimeError("call reset() before policy()")
        be = self.backend
        k = _abi.policy_code(kind)
        if cr_factor is not None:
            if (k & 0xFF) != _abi.POLICY["CR"] or not 0.0 < float(cr_factor) < 1e300:
                raise ValueError("cr_factor is CriticalRatio's due-date factor: a positive float, with kind 'CR'")
            k = _abi.POLICY_CR_F64
            self._desc.cr_factor = float(cr_factor)
        d, s, _ = self._refs()
        with be.on_device():
            if self._classes is not None:
                cs = self._classes["sets"]
                for x in self._classes["keep"][0]:
                    x.cr_factor = self._desc.cr_factor
                rc = be.lib.jss_multi_policy(self._classes["n"], cs[0], cs[1], k, self.seed if seed is None else int(seed),
                                             int(round(explore * 65536)), self._class_ptrs(self._actions_out), be.stream())
                _abi.check(be.lib, rc, "jss_multi_policy")
            else:
                _abi.check(be.lib, be.lib.jss_policy(d, s, k, self.seed if seed is None else int(seed),
                                                     int(round(explore * 65536)), be.ptr(self._actions_out), be.stream()),
                           "jss_policy")
        return self._actions_out

    def rollout(self, kind: Union[str, int] = "random", n_iter: int = 1, seed: Optional[int] = None,
                autoreset: bool = True, explore: float = 0.0):
        """n_iter x (policy + step) per env in ONE launch (state stays in registers)."""
        if not self._is_reset:
            raise RuntimeError("call reset() before rollout()")
        be = self.backend
        k = _abi.policy_code(kind)
        flags = _abi.ROLLOUT_AUTORESET if autoreset else 0
        d, s, o = self._refs()
        with be.on_device():
            if self._classes is not None and int(n_iter) == 1:
                streams = (C.c_void_p * 1)(be.stream())
                rc = be.lib.jss_multi_rollout(self._classes["n"], *self._classes["sets"], k, self.seed if seed is None else int(seed),
                                              int(round(explore * 65536)), 1, flags, 1, streams)
                _abi.check(be.lib, rc, "jss_multi_rollout")
                return self._obs(), self.reward, self.done, False, {}
        sd, q16 = self.seed if seed is None else int(seed), int(round(explore * 65536))
        # (by shape class: one launch per range -- below 64 jobs / the rest -- with the kernel of its shape)
        self._over_ranges(lambda dk, sk, ok, _a, stream: be.lib.jss_rollout(dk, sk, ok, k, sd, q16, int(n_iter), flags, stream), "jss_rollout")
        return self._obs(), self.reward, self.done, False, {}

    def rollout_steps(self, kind: Union[str, int] = "random", steps: int = 1, n_sub: int = 2, seed: Optional[int] = None,
                      autoreset: bool = True, explore: float = 0.0):
        """``steps`` consecutive one-step rollouts of the whole batch, issued as ``n_sub`` independent contiguous
        sub-batches on ``n_sub`` streams (the current one + side streams forked from / joined back into it): step s of a
        sub-batch depends only on its own step s-1, so the drain of one sub-batch's launch overlaps the fill of
        another's.  Results are identical to ``steps`` calls of ``rollout(n_iter=1)``; outputs hold the last step."""
        if not self._is_reset:
            raise RuntimeError("call reset() before rollout_steps()")
        if not 1 <= int(n_sub) <= _abi.MAX_SUB_BATCHES:
            raise ValueError(f"n_sub must be in [1, {_abi.MAX_SUB_BATCHES}]")
        be = self.backend
        k = _abi.policy_code(kind)
        flags = _abi.ROLLOUT_AUTORESET if autoreset else 0
        d, s, o = self._refs()
        sd = self.seed if seed is None else int(seed)
        with be.on_device():
            if self._classes is not None:       # order='by_shape': a grid over the shape classes per step and part
                n = min(int(n_sub), 4)
                streams = be.stream_array(n) if hasattr(be, "stream_array") else (C.c_void_p * n)()
                rc = be.lib.jss_multi_rollout(self._classes["n"], *self._classes["sets"], k, sd, int(round(explore * 65536)), int(steps),
                                              flags | (_abi.ROLLOUT_FORK_JOIN if n > 1 and hasattr(be, "stream_array") else 0), n, streams)
                _abi.check(be.lib, rc, "jss_multi_rollout")
                return self._obs(), self.reward, self.done, False, {}
            if hasattr(be, "stream_array"):     # the library forks / joins the side streams itself (two C calls per stream)
                rc = be.lib.jss_rollout_steps(d, s, o, k, sd, int(round(explore * 65536)), int(steps),
                                              flags | _abi.ROLLOUT_FORK_JOIN, int(n_sub), be.stream_array(int(n_sub)))
            else:
                rc = be.with_streams(int(n_sub), lambda streams: be.lib.jss_rollout_steps(
                    d, s, o, k, sd, int(round(explore * 65536)), int(steps), flags, int(n_sub), streams), self._stream_events)
        _abi.check(be.lib, rc, "jss_rollout_steps")
        return self._obs(), self.reward, self.done, False, {}

    def policy_step_steps(self, kind: Union[str, int] = "random", steps: int = 1, n_sub: int = 2, seed: Optional[int] = None,
                          autoreset: bool = True, explore: float = 0.0, caller_orders_streams: bool = False):
        """``steps`` x (``policy`` -> actions in memory -> ``step``): the UN-fused loop of a learner whose policy is a
        launch of its own (``jss_policy`` stands in for it), issued by the library over ``n_sub`` sub-batches on ``n_sub``
        streams so that one sub-batch's policy overlaps another's step (``jss_policy_step_steps``).  Same results as the
        Python loop ``for _ in range(steps): env.step(env.policy(kind), autoreset=autoreset)``."""
        if not self._is_reset:
            raise RuntimeError("call reset() before policy_step_steps()")
        if not 1 <= int(n_sub) <= _abi.MAX_SUB_BATCHES:
            raise ValueError(f"n_sub must be in [1, {_abi.MAX_SUB_BATCHES}]")
        be = self.backend
        k = _abi.policy_code(kind)
        flags = (_abi.ROLLOUT_AUTORESET if autoreset else 0)
        d, s, o = self._refs()
        sd, q16 = self.seed if seed is None else int(seed), int(round(explore * 65536))
        with be.on_device():
            if hasattr(be, "stream_array"):
                flags |= 0 if caller_orders_streams else _abi.ROLLOUT_FORK_JOIN
                rc = be.lib.jss_policy_step_steps(d, s, o, k, sd, q16, be.ptr(self._actions_out), int(steps), flags, int(n_sub),
                                                  be.stream_array(int(n_sub)))
            else:
                rc = be.with_streams(int(n_sub), lambda streams: be.lib.jss_policy_step_steps(
                    d, s, o, k, sd, q16, be.ptr(self._actions_out), int(steps), flags, int(n_sub), streams), self._stream_events)
        _abi.check(be.lib, rc, "jss_policy_step_steps")
        return self._obs(), self.reward, self.done, False, {}

    def bind_rollout_steps(self, kind: Union[str, int] = "random", steps: int = 1, n_sub: int = 2, seed: Optional[int] = None,
                           autoreset: bool = True, explore: float = 0.0, caller_orders_streams: bool = False):
        """``rollout_steps`` with every argument resolved now: returns a zero-argument callable that issues the same
        launches on the stream that is current NOW and on the side streams (one C call, no Python-side work between
        the call and the first launch).  For loops that issue the same window over and over (bench.py).
        ``caller_orders_streams=True``: the library does not fork / join the side streams -- the caller guarantees that
        the device is idle when the call is made and synchronises the whole device (not just its stream) before it
        touches the results (include/jss_hip.h: "the caller orders streams[] against its own stream")."""
        if not self._is_reset:
            raise RuntimeError("call reset() before rollout_steps()")
        be = self.backend
        if not hasattr(be, "stream_array"):
            return lambda: self.rollout_steps(kind, steps, n_sub, seed, autoreset, explore)
        k = _abi.policy_code(kind)
        flags = (_abi.ROLLOUT_AUTORESET if autoreset else 0) | (0 if caller_orders_streams else _abi.ROLLOUT_FORK_JOIN)
        d, s, o = self._refs()
        with be.on_device():
            streams = be.stream_array(int(n_sub))
        fn, sd, q16, n_steps, n = be.lib.jss_rollout_steps, self.seed if seed is None else int(seed), int(round(explore * 65536)), int(steps), int(n_sub)
        lib = be.lib
        if self._classes is not None:
            n = min(n, 4)
            with be.on_device():
                streams = be.stream_array(n)
            if n == 1:
                flags &= ~_abi.ROLLOUT_FORK_JOIN
            n_sets, sets, fm = self._classes["n"], self._classes["sets"], be.lib.jss_multi_rollout

            def issue_classes():
                rc = fm(n_sets, *sets, k, sd, q16, n_steps, flags, n, streams)
                if rc:
                    _abi.check(lib, rc, "jss_multi_rollout")
            return issue_classes

        def issue():
            rc = fn(d, s, o, k, sd, q16, n_steps, flags, n, streams)
            if rc:
                _abi.check(lib, rc, "jss_rollout_steps")
        return issue

    def trajectory(self, kind: Union[str, int] = "random", steps: int = 1, seed: Optional[int] = None,
                   autoreset: bool = True, explore: float = 0.0, buffers: Optional[dict] = None,
                   record=("real_obs", "action_mask", "action", "reward", "done")):
        """``steps`` x (policy + step) per env in ONE launch, like ``rollout(n_iter=steps)``, with every iteration's
        transition written out step-major: ``real_obs`` (K, B, J, 7) and ``action_mask`` (K, B, J + 1) = what the
        policy saw in slot k, ``action`` (K, B) what it did (``-2``: the env was found done and was reset instead --
        gymnasium.vector next-step auto-reset; ``-1``: found done with autoreset off), ``reward`` / ``done`` (K, B)
        what it got.  State is read and written once per call: the (s, a, r, d) stream of a scripted / random
        behaviour policy without K launch boundaries and K - 1 state round trips.

        ``buffers`` (the dict a previous call returned) is reused when its shapes fit; otherwise zero-filled
        tensors are allocated here (outside the hot loop: allocate once, pass them back in)."""
        if not self._is_reset:
            raise RuntimeError("call reset() before trajectory()")
        be = self.backend
        K, B, J = int(steps), self.batch, self.jmax
        shapes = {"real_obs": ((K, B, J, 7), "float32"), "action_mask": ((K, B, J + 1), "uint8"),
                  "action": ((K, B), "int32"), "reward": ((K, B), "float32"), "done": ((K, B), "uint8")}
        out = {}
        with be.on_device():
            for name in record:
                shape, dtype = shapes[name]
                t = None if buffers is None else buffers.get(name)
                out[name] = t if t is not None and tuple(t.shape) == shape else be.zeros(shape, dtype)
        k = _abi.policy_code(kind)
        flags = _abi.ROLLOUT_AUTORESET if autoreset else 0
        sd, q16 = self.seed if seed is None else int(seed), int(round(explore * 65536))
        # (by shape class: one launch per range of the batch -- below 64 jobs / the rest -- each with the kernel of ITS shape,
        #  recording into its columns of the whole batch's [K][B] buffers)
        keep = []

        def call(d, s, o, a, stream):
            keep.append(self._traj_at(out, a))
            return be.lib.jss_trajectory(d, s, o, C.byref(keep[-1]), k, sd, q16, K, flags, stream)
        self._over_ranges(call, "jss_trajectory")
        return out

    def _ranges(self):
        """[(desc, state, out, first env)] a single-set call that loops over steps covers the batch with: the batch itself,
        or -- dealt out by shape class -- the range of the classes below 64 jobs and the range of the rest, each with a JssDesc
        of its own (jclass: the library picks the kernel of that shape: one job per lane / two)."""
        if self._classes is None:
            d, s, o = self._refs()
            return [(d, s, o, 0)]
        self._refs()                                  # (refuses while a session is open)
        return [(C.byref(d), C.byref(s), C.byref(o), a) for d, s, o, a in self._class_ranges]

    def _over_ranges(self, call, what):
        """call(desc, state, out, first_env, stream) for every range of `_ranges()`: one range on the current stream; several
        on the current stream + side streams forked from / joined back into it (a launch that loops over K steps lasts K step
        latencies however few envs it holds -- two of them one after the other would take twice that)."""
        be = self.backend
        rs = self._ranges()
        with be.on_device():
            if len(rs) == 1 or not hasattr(be, "stream_array"):
                for d, s, o, a in rs:
                    _abi.check(be.lib, call(d, s, o, a, be.stream()), what)
                return

            def issue(streams):
                for i, (d, s, o, a) in enumerate(rs):
                    rc = call(d, s, o, a, streams[i])
                    if rc:
                        return rc
                return 0
            _abi.check(be.lib, be.with_streams(len(rs), issue, self._stream_events), what)

    def _traj_at(self, bufs, a):
        """JssTraj over the step-major [K][B] buffers `bufs` for the range of the batch that starts at env `a`."""
        be, J = self.backend, self.jmax
        per_env = {"real_obs": J * 7 * 4, "action_mask": J + 1, "action": 4, "reward": 4, "done": 1}
        ptr = lambda n: (be.ptr(bufs[n]) + a * per_env[n]) if bufs.get(n) is not None else None      # noqa: E731
        return _abi.JssTraj(ptr("real_obs"), ptr("action_mask"), ptr("action"), ptr("reward"), ptr("done"), self.batch)

    def steps(self, actions, record=(), buffers: Optional[dict] = None):
        """K consecutive ``step()`` calls per env in ONE launch (``jss_steps``): ``actions`` is a (K, B) int32 array of
        action codes as ``step`` takes them (job, J = NOPE, -1 = skip, -2 = reset) -- a recorded trace, a planned
        open-loop sequence.  The state is read and written once.  ``record`` names the per-step streams to keep,
        step-major: ``real_obs`` (K, B, J, 7) and ``action_mask`` (K, B, J + 1) AFTER each step, ``reward`` / ``done``
        (K, B) -- for an env that a step skips (-1) or restarts (-2) the recorded reward is 0 and the recorded done says
        whether the state it is in has a legal action (``step``'s own ``reward`` / ``done`` tensors keep the carried-over
        values instead).  Returns the dict of recorded buffers (reusable through ``buffers``); the env's own outputs hold
        the last step."""
        if not self._is_reset:
            raise RuntimeError("call reset() before steps()")
        be = self.backend
        shape = tuple(actions.shape)
        if len(shape) != 2 or shape[1] != self.batch:
            raise ValueError(f"expected actions of shape (K, {self.batch}), got {shape}")
        K, B, J = shape[0], self.batch, self.jmax
        shapes = {"real_obs": ((K, B, J, 7), "float32"), "action_mask": ((K, B, J + 1), "uint8"),
                  "reward": ((K, B), "float32"), "done": ((K, B), "uint8")}
        out = {}
        with be.on_device():
            a = be.as_device(actions, "int32")
            for name in record:
                shp, dtype = shapes[name]
                t = None if buffers is None else buffers.get(name)
                out[name] = t if t is not None and tuple(t.shape) == shp else be.zeros(shp, dtype)
        keep = []

        def call(d, s, o, first, stream):         # (by shape class: one launch per range, see trajectory)
            keep.append(self._traj_at({n: out.get(n) for n in ("real_obs", "action_mask", "reward", "done")}, first))
            return be.lib.jss_steps(d, s, o, C.byref(keep[-1]), be.ptr(a) + first * 4, K, stream)
        self._over_ranges(call, "jss_steps")
        self._steps_keep = a                  # alive until the launch has read it
        return out

    def session(self, depth: int = 16, timeout_ms: int = 10000, slots: int = 0):
        """Open a step session (``jssenv_amd.session.StepSession``): the env state stays on the chip between steps, the
        caller posts actions and waits for outputs.  Use as a context manager.  ``timeout_ms`` bounds how long the resident
        kernel waits for the next actions before it gives the session up (close it around anything longer, e.g. a
        training phase)."""
        from .session import StepSession
        return StepSession(self, depth=depth, timeout_ms=timeout_ms, slots=slots)

    def sync_check(self):
        """Wait for the env's stream and raise if a kernel faulted (the launching calls only report launch errors)."""
        be = self.backend
        with be.on_device():
            _abi.check(be.lib, be.lib.jss_sync_check(be.stream()), "jss_sync_check")

    def synchronize(self):
        self.backend.sync()

    def close(self):
        """Wait for outstanding work (the side streams of rollout_steps are process-wide and stay)."""
        self.synchronize()
        if self._owns_backend:
            self.backend.close()

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

    def clear_errors(self):
        """Clear the sticky per-env error bits (the NOPE flag in the same word is kept)."""
        if getattr(self, "host_arena", False):
            self.backend.sync()                    # host memory the kernels work on in place: nothing may be in flight
        self.env_header[:, _abi.H_STATUS] &= ~0xFF

    @property
    def todo_time_step_job(self):
        return self.job_state[:, :, 0] & (_abi.FC_TODO_MASK if self.compact else _abi.FM_TODO_MASK if self.medium else _abi.TODO_MASK)

    def _word(self, f):
        """Word JSS_F_* `f` (LEFT, PERF, IDLE, IDLE_LAST) of every job record as a (B, J) tensor, whichever record layout
        the batch uses (a view of the state for full records, decoded from the packed words for compact ones)."""
        js = self.job_state
        if self.medium:
            if f == _abi.F_LEFT:
                return js[:, :, _abi.FM_LEFT_F4] & 0xFFFF
            if f == _abi.F_PERF:
                return js[:, :, _abi.FM_PERF_NEXT] & _abi.FM_OP_MASK
            return js[:, :, {_abi.F_IDLE: _abi.FM_IDLE, _abi.F_IDLE_LAST: _abi.FM_IDLE_LAST}[f]]
        if not self.compact:
            return js[:, :, f]
        if f == _abi.F_LEFT:
            return js[:, :, _abi.FC_LEFT_F4] & 0xFFFF
        if f == _abi.F_PERF:
            return (js[:, :, _abi.FC_W0] >> _abi.FC_PERF_SHIFT) & 0x3FFFFF
        return js[:, :, {_abi.F_IDLE: _abi.FC_IDLE, _abi.F_IDLE_LAST: _abi.FC_IDLE_LAST}[f]]

    @property
    def needed_machine_jobs(self):
        """(B, J) machine of every job's current op, -1 once the job is finished (same kind of array as the state
        tensors).  With compact records the op is not stored: it is looked up in the batch's one op table."""
        if self.medium:
            return self._current_ops() >> 16
        if not self.compact:
            return self.job_state[:, :, _abi.F_CUR] >> 16
        return self._current_ops() >> 16                      # a finished job's "op" is -1, and -1 >> 16 == -1

    def _current_ops(self):
        """(B, J) op table entry [j][todo_time_step_job[j]] of a compact batch (machine << 16 | duration), -1 where the
        job is finished -- what a full record carries as its JSS_F_CUR word."""
        js = self.job_state
        if self.medium:                                       # the record carries the op itself (21 bits, 0 = job finished)
            cur = (js[:, :, _abi.FM_W0] >> _abi.FM_CUR_SHIFT) & _abi.FM_OP_MASK
            if isinstance(js, np.ndarray):
                return np.where(cur != 0, cur, -1).astype(np.int32)
            import torch
            return torch.where(cur != 0, cur, torch.full_like(cur, -1))
        M, J = int(self.packed.machines[0]), self.jmax
        todo = js[:, :, _abi.FC_W0] & _abi.FC_TODO_MASK
        if isinstance(js, np.ndarray):
            cur = self.packed.ops[0][np.arange(J)[None, :], np.minimum(todo, M - 1)]
            return np.where(todo < M, cur, -1).astype(np.int32)
        import torch
        cur = self._ops[0][torch.arange(J, device=js.device)[None, :], todo.clamp(max=M - 1).long()]
        return torch.where(todo < M, cur, torch.full_like(cur, -1))

    @property
    def time_until_finish_current_op_jobs(self):
        return self._word(_abi.F_LEFT)

    @property
    def total_perform_op_time_jobs(self):
        return self._word(_abi.F_PERF)

    @property
    def total_idle_time_jobs(self):
        return self._word(_abi.F_IDLE)

    @property
    def idle_time_jobs_last_op(self):
        return self._word(_abi.F_IDLE_LAST)

    @property
    def time_until_available_machine(self):
        return self.machine_state

    @property
    def legal_actions(self):
        return self.action_mask

    @property
    def action_illegal_no_op(self):
        return (self.job_state[:, :, 0] >> (8 if self.compact else 7 if self.medium else 9)) & 1

    def __getattr__(self, name):
        # compact batches keep no machine clocks in memory: time_until_available_machine[m] is the time left of the job
        # running on m (both are set to the op's duration at jss_env.py:446-449 and count down together, :521-530)
        if name == "machine_state" and self.__dict__.get("no_clocks"):
            return self._clocks_from_records()
        raise AttributeError(name)

    def _clocks_from_records(self):
        """(B, M) int32 time_until_available_machine of a compact batch, computed from its job records where they live."""
        js, cur = self.job_state, self._current_ops()
        left = js[:, :, _abi.FC_LEFT_F4] & 0xFFFF             # (the same word and bits in the medium record)
        if isinstance(js, np.ndarray):
            tm = np.zeros((self.batch, self.mmax), dtype=np.int32)
            b, j = np.nonzero((cur >= 0) & (left > 0))
            tm[b, cur[b, j] >> 16] = left[b, j]
            return tm
        import torch
        run = (cur >= 0) & (left > 0)                          # at most one running job per machine: the sum is a scatter
        val = torch.where(run, left, torch.zeros_like(left))
        idx = torch.where(run, cur >> 16, torch.zeros_like(cur)).long()
        return torch.zeros((self.batch, self.mmax), dtype=torch.int32, device=js.device).scatter_add_(1, idx, val)

    @staticmethod
    def clocks_from_jobs(js, M):
        """time_until_available_machine[M] from one env's decoded job matrix (``decode_jobs``)."""
        tm = np.zeros(M, dtype=np.int64)
        run = (js[_abi.F_LEFT] > 0) & (js[_abi.F_CUR] >= 0)
        tm[js[_abi.F_CUR][run] >> 16] = js[_abi.F_LEFT][run]
        return tm

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
    _STATE_TENSORS = ("env_header", "env_const", "job_state", "machine_state", "solution", "counters", "real_obs", "action_mask",
                      "reward", "done", "makespan")

    def _saved_tensors(self):                      # a compact / medium batch has no machine clocks to save: they are derived
        return tuple(k for k in self._STATE_TENSORS if k != "machine_state" or not self.no_clocks)

    def _no_open_session(self, what):
        if self._session is not None and not self._session.closed:
            raise RuntimeError(f"{what}: a step session is open on this env -- the state lives in its resident kernel and the "
                               "tensors in memory are stale; close() the session first")

    def state_dict(self):
        """Host copy of everything needed to resume: state + last outputs + the batch description."""
        self._no_open_session("state_dict")
        n = self.backend.numpy
        d = {k: n(getattr(self, k)) for k in self._saved_tensors()}
        d["meta"] = {"abi": _abi.STATE_LAYOUT, "record_ints": self.record_ints, "batch": self.batch, "jmax": self.jmax, "mmax": self.mmax, "seed": self.seed,
                     "env_id_base": self.env_id_base, "table_of_env": self.table_of_env_host.copy(),
                     "ops": self.packed.ops.copy(),
                     # the global env ids key the per-env RNG streams: a resumed run continues them only on the same ids
                     "env_ids": (np.zeros(0, dtype=np.int64) if self._env_ids is None else n(self._env_ids).astype(np.int64))}
        return d

    def load_state_dict(self, d):
        self._no_open_session("load_state_dict")
        m = d["meta"]
        if int(m.get("abi", 0)) != _abi.STATE_LAYOUT:
            raise ValueError(f"checkpoint was written with state layout v{m.get('abi')}, this build is v{_abi.STATE_LAYOUT}")
        if int(m.get("record_ints", _abi.NF)) != self.record_ints:
            raise ValueError("checkpoint uses another job-record layout (compact / medium / full)")
        if (int(m["batch"]), int(m["jmax"]), int(m["mmax"])) != (self.batch, self.jmax, self.mmax) or \
                not np.array_equal(m["ops"], self.packed.ops) or not np.array_equal(m["table_of_env"], self.table_of_env_host):
            raise ValueError("checkpoint belongs to a different batch (shape or instances differ)")
        mine = np.zeros(0, dtype=np.int64) if self._env_ids is None else self.backend.numpy(self._env_ids).astype(np.int64)
        if int(m["env_id_base"]) != self.env_id_base or not np.array_equal(np.asarray(m.get("env_ids", mine)).reshape(-1), mine):
            raise ValueError(f"checkpoint was written by envs with other global ids (env_id_base {int(m['env_id_base'])} "
                             f"vs {self.env_id_base}, or different set_env_ids): the RNG streams would not continue")
        with self.backend.on_device():
            self.backend.sync()
            for k in self._saved_tensors():
                self.backend.copy_into(getattr(self, k), np.asarray(d[k]))
        self.seed, self._is_reset = int(m["seed"]), True

    def save_checkpoint(self, path):
        """state_dict() to one .npz file (NumPy arrays only, no pickling)."""
        d = self.state_dict()
        meta = d.pop("meta")
        flat = {f"state_{k}": v for k, v in d.items()}
        for k, v in meta.items():
            flat[f"meta_{k}"] = np.asarray(v)
        with open(path, "wb") as fh:
            np.savez(fh, **flat)

    def load_checkpoint(self, path):
        with np.load(path, allow_pickle=False) as z:
            d = {k[len("state_"):]: z[k] for k in z.files if k.startswith("state_")}
            d["meta"] = {k[len("meta_"):]: (z[k] if z[k].ndim else z[k].item()) for k in z.files if k.startswith("meta_")}
        self.load_state_dict(d)

    def decode_jobs(self, raw, i: int):
        """One env's job records (rows of ``job_state[i]``, either layout) as an (8, J) int64 matrix in JSS_F_* word order
        with word 0 decoded -- row 0 = todo_time_step_job, row 7 = flags (1 legal, 2 blocked) -- plus the cached next ops
        (which a compact record does not store: they are what the op table says)."""
        J, M = int(self.jobs_per_env[i]), int(self.machines_per_env[i])
        raw = np.asarray(raw)[:J].astype(np.int64)                 # the record's words, signed
        u0 = raw[:, 0] & 0xFFFFFFFF                                 # word 0 as the bit field it is
        js = np.zeros((8, J), dtype=np.int64)
        if self.medium:
            u1, u2, u3 = (raw[:, k] & 0xFFFFFFFF for k in (_abi.FM_LEFT_F4, _abi.FM_PERF_NEXT, _abi.FM_NEXT_NEXT2))
            op = lambda x: np.where(x != 0, x, -1)                  # noqa: E731  21-bit op, 0 = none
            js[_abi.F_TODO], js[7] = u0 & _abi.FM_TODO_MASK, (u0 >> 6) & 3
            js[_abi.F_CUR] = op((u0 >> _abi.FM_CUR_SHIFT) & _abi.FM_OP_MASK)
            nxt, nxt2 = op((u2 >> 21) | ((u3 & 0x3FF) << 11)), op((u3 >> 10) & _abi.FM_OP_MASK)
            js[_abi.F_LEFT], js[_abi.F_PERF] = u1 & 0xFFFF, u2 & _abi.FM_OP_MASK
            js[_abi.F_F4] = np.where(u0 & _abi.FM_FLAG_F4_ONE, _abi.F4_ONE, u1 >> 16)
            js[_abi.F_IDLE], js[_abi.F_IDLE_LAST] = raw[:, _abi.FM_IDLE], raw[:, _abi.FM_IDLE_LAST]
        elif self.compact:
            u1 = raw[:, _abi.FC_LEFT_F4] & 0xFFFFFFFF
            todo = u0 & _abi.FC_TODO_MASK
            js[_abi.F_TODO], js[7] = todo, (u0 >> 7) & 3
            ops = self.packed.ops[int(self.table_of_env_host[i])][:J].astype(np.int64)
            at = lambda k: np.where(todo + k < M, ops[np.arange(J), np.minimum(todo + k, M - 1)], -1)   # noqa: E731
            js[_abi.F_CUR], nxt, nxt2 = at(0), at(1), at(2)
            js[_abi.F_LEFT], js[_abi.F_PERF] = u1 & 0xFFFF, u0 >> _abi.FC_PERF_SHIFT
            js[_abi.F_F4] = np.where(u0 & _abi.FC_FLAG_F4_ONE, _abi.F4_ONE, u1 >> 16)
            js[_abi.F_IDLE], js[_abi.F_IDLE_LAST] = raw[:, _abi.FC_IDLE], raw[:, _abi.FC_IDLE_LAST]
        else:
            js[_abi.F_TODO], js[7] = u0 & _abi.TODO_MASK, (u0 >> 8) & 3
            for f in range(1, 7):
                js[f] = raw[:, f]
            nxt = raw[:, _abi.F_NEXT]
            n2 = u0 >> _abi.NEXT2_SHIFT
            nxt2 = np.where(n2, n2, -1)
        return js, nxt, nxt2

    def host_tensors(self):
        """NumPy copies of the state and output tensors (not ``solution``).  A small batch comes over in ONE
        device -> host copy of the arena they were carved from."""
        self._no_open_session("host_tensors")
        be = self.backend
        if getattr(self, "host_arena", False):        # the arena IS host memory: wait for the kernels, look at it
            be.sync()
            if self._host_views is None:
                flat = self._arena.numpy()
                self._host_views = {k: _carve_numpy(flat, o, sh, dt) for k, (o, sh, dt) in self._layout.items() if not k.startswith("_")}
                self._host_views["counters"] = _LazyRows(lambda: be.numpy(self.counters))   # device memory: fetched when read
            return self._host_views
        if self.batch <= 64 and hasattr(be, "snapshot"):
            with be.on_device():
                host, flat = be.snapshot(self._arena, self._host_arena)
            if self._host_views is None or host is not self._host_arena:      # the staging buffer is reused: so are its views
                self._host_views = {k: _carve_numpy(flat, o, sh, dt) for k, (o, sh, dt) in self._layout.items()
                                    if not k.startswith("_")}
            self._host_arena = host
            return self._host_views
        return {k: be.numpy(getattr(self, k)) for k in self._layout if not k.startswith("_")}

    def host_state(self, i: int = 0, with_solution: bool = True):
        """Everything about env i as NumPy, sliced to its true (J, M).  ``job_state`` rows follow the JSS_F_* word
        order with the packed word 0 decoded: row 0 = todo_time_step_job, row 7 = flags (1 legal, 2 blocked);
        ``next_op`` / ``next2_op`` are the record's cached next ops."""
        n = self.backend.numpy
        J, M = int(self.jobs_per_env[i]), int(self.machines_per_env[i])
        if self.batch <= 64:
            t = self.host_tensors()
            t = {k: v[i] for k, v in t.items()}
        else:
            t = {k: n(getattr(self, k)[i:i + 1])[0] for k in self._layout if not k.startswith("_")}
        js, nxt, nxt2 = self.decode_jobs(t["job_state"], i)
        hdr = t["env_header"]
        out = {
            "jobs": J, "machines": M,
            "clock": int(hdr[_abi.H_CLOCK]),
            "job_state": js,
            "next_op": nxt,
            "next2_op": nxt2,
            "tm": self.clocks_from_jobs(js, M) if self.no_clocks else t["machine_state"][:M].astype(np.int64),
            "mask": t["action_mask"][:J + 1].astype(bool),
            "mask_padding": t["action_mask"][J + 1:].copy(),
            "blocked": (js[7] & 2) != 0,
            "obs": t["real_obs"][:J].astype(np.float32),
            "obs_padding": t["real_obs"][J:].copy(),
            "reward": float(t["reward"]),
            "done": bool(t["done"]),
            "err": int(hdr[_abi.H_STATUS]) & 0xFF,
            "noop_flag": bool(int(hdr[_abi.H_STATUS]) & _abi.STATUS_NOOP),
            "makespan": int(t["makespan"]),
            "episode": int(hdr[_abi.H_EPISODE]),
            "step_in_episode": int(hdr[_abi.H_STEP]),
            "counters": t["counters"].copy(),
        }
        if with_solution:
            out["solution"] = n(self.solution[i])[:J, :M].astype(np.int64)
        return out


class _LazyRows:
    """rows[i] of an array that is only fetched (from the device) when somebody indexes it."""

    def __init__(self, fetch):
        self._fetch = fetch

    def __getitem__(self, i):
        return self._fetch()[i]


def __getattr__(name):
    """``jssenv_amd.env.JssEnv`` / ``make`` / ``gymnasium_base`` (their home until round 6, and the gymnasium entry point
    "jssenv_amd.env:JssEnv"): resolved lazily, facade.py imports this module."""
    if name in ("JssEnv", "make", "gymnasium_base"):
        from . import facade
        return getattr(facade, name)
    raise AttributeError(name)
