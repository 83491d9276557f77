"""Shape-bucketed batches: ragged instance mixes without the padding cost.

``BatchedJssEnv`` pads every env of a batch to the largest (Jmax, Mmax), and the kernel
flavour is chosen from that padded shape: one 15x15 env in a batch that also holds a 100x20
instance runs on the two-jobs-per-lane wave kernel with 85 % of its lanes idle.
``BucketedJssEnv`` splits the envs into shape classes

    class 0: J, M <= 16   -> packed body, 4 envs per wavefront
    class 1: J, M <= 32   -> packed body, 2 envs per wavefront
    class 2: J <= 64      -> one wavefront per env
    class 3: J <= 128     -> one wavefront per env, two jobs per lane

each backed by its own compactly padded ``BatchedJssEnv`` (own tensors).  Every call --
``reset``, ``policy``, ``step``, ``rollout_steps`` -- is ONE launch over all classes
(``jss_multi_*``, include/jss_hip.h: a workgroup of the grid finds its class by its index and
runs that class's body), on the caller's stream: no side streams, no per-class launches whose
overlap depends on how the runtime deals streams onto hardware queues.  Env ``i`` of the
wrapper lives at ``slot[i]`` of bucket ``bucket_of[i]``; the per-env RNG key stays the global
env id, so a bucketed run is the same random process as the padded one.  Outputs stay per
bucket (an RL policy on ragged instances batches by shape anyway); ``host_state(i)`` reads one
env wherever it lives.

``launch="streams"`` keeps round 3's form of ``rollout_steps`` (one launch per class and step,
every class on its own HIP stream, ``jss_rollout_steps_multi``) for A/B measurements.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _abi
from .env import BatchedJssEnv
from .instances import resolve_instance


def shape_class(jobs: int, machines: int) -> int:
    if jobs <= 16 and machines <= 16:
        return 0
    if jobs <= 32 and machines <= 32:
        return 1
    return 2 if jobs <= 64 else 3


class BucketedJssEnv:
    def __init__(self, instances: Sequence, batch: Optional[int] = None, device=None, seed: int = 0,
                 env_id_base: int = 0, launch: str = "grid", _backend=None):
        from .env import make_backend
        if launch not in ("grid", "streams"):
            raise ValueError("launch must be 'grid' or 'streams'")
        if _backend is None:
            _backend = make_backend(device)     # one backend (library handle, device) shared by the buckets
        insts = [resolve_instance(i) for i in instances]
        n = len(insts)
        self.batch = B = int(batch) if batch is not None else n
        self.seed = int(seed)
        self.table_of_env = np.arange(B) % n
        cls = np.array([shape_class(i.jobs, i.machines) for i in insts])
        self.bucket_of = cls[self.table_of_env]
        self.slot = np.zeros(B, dtype=np.int64)
        self.buckets: List[Optional[BatchedJssEnv]] = [None] * 4
        self.members: List[np.ndarray] = [np.zeros(0, dtype=np.int64)] * 4
        self.jmax = max(i.jobs for i in insts)
        self.mmax = max(i.machines for i in insts)
        for k in range(4):
            env_ids = np.flatnonzero(self.bucket_of == k)
            self.members[k] = env_ids
            if env_ids.size == 0:
                continue
            self.slot[env_ids] = np.arange(env_ids.size)
            tables = sorted(set(self.table_of_env[env_ids].tolist()))
            remap = {t: i for i, t in enumerate(tables)}
            bucket_insts = [insts[t] for t in tables]
            if len(bucket_insts) == 1:
                # a class with ONE instance would take the shared-table layout (op table staged in LDS, compact records),
                # which has no body in the fused grid: listing the instance twice keeps the class on per-env tables
                bucket_insts = bucket_insts * 2
            # (the one-wavefront-per-env classes on full records: the fused grid has no medium body for them)
            sub = BatchedJssEnv(bucket_insts, batch=env_ids.size, device=device, seed=seed, records="full" if k >= 2 else None,
                                table_of_env=[remap[t] for t in self.table_of_env[env_ids]], _backend=_backend)
            # RNG streams are keyed by the GLOBAL env id: give the bucket an explicit id list when its
            # members are not a contiguous range
            sub.set_env_ids(env_id_base + env_ids)
            self.buckets[k] = sub
        self._backend = _backend
        self.launch = launch
        self.n_sub = 1                               # parts per class in rollout_steps (see there)
        each = self._each()
        n_sets = len(each)
        D, S, O = C.POINTER(_abi.JssDesc), C.POINTER(_abi.JssState), C.POINTER(_abi.JssOut)
        # the argument arrays of the jss_multi_* calls: built once, the structs they point to live in the buckets
        self._sets = ((D * n_sets)(*[C.pointer(b._desc) for _, b in each]), (S * n_sets)(*[C.pointer(b._state) for _, b in each]),
                      (O * n_sets)(*[C.pointer(b._out) for _, b in each]))
        self._n_sets = n_sets
        p = _backend.ptr
        self._policy_out = (C.c_void_p * n_sets)(*[p(b._actions_out) for _, b in each])
        if launch == "streams" and hasattr(_backend, "side_pool"):
            _backend.side_pool(n_sets - 1)       # created now, back to back (HipBackend.side_pool)

    def close(self):
        """Wait for outstanding work."""
        self.synchronize()

    def _each(self):
        return [(k, b) for k, b in enumerate(self.buckets) if b is not None]

    def _check_reset(self, what):
        if not all(b._is_reset for _, b in self._each()):
            raise RuntimeError(f"call reset() before {what}()")

    def _seed(self, seed):
        return self.seed if seed is None else int(seed)

    def reset(self):
        """reset() of every env, one launch.  Returns {class: obs dict}."""
        be = self._backend
        with be.on_device():
            _abi.check(be.lib, be.lib.jss_multi_reset(self._n_sets, *self._sets, None, be.stream()), "jss_multi_reset")
        for _, b in self._each():
            b._is_reset = True
        return {k: b._obs() for k, b in self._each()}

    def rollout(self, kind="random", n_iter=1, seed=None, autoreset=True, explore=0.0):
        """n_iter x (policy + step) per env.  n_iter == 1: one launch over all classes; more: one launch per class (each
        keeps its envs' state in registers for the n_iter iterations), back to back on the caller's stream."""
        if n_iter == 1:
            return self.rollout_steps(kind, steps=1, seed=seed, autoreset=autoreset, explore=explore)
        for _, b in self._each():
            b.rollout(kind, n_iter=n_iter, seed=self._seed(seed), autoreset=autoreset, explore=explore)

    def rollout_steps(self, kind="random", steps=1, n_iter=1, seed=None, autoreset=True, explore=0.0,
                      caller_orders_streams=False, n_sub=None):
        """`steps` consecutive one-step rollouts of every env: `steps` launches, each ONE grid over all shape classes
        (jss_multi_rollout), issued from a C loop.  Same results as `steps` x rollout(n_iter=1) of each bucket.
        n_sub (default: the env's ``n_sub``, 1): every class is cut into n_sub contiguous parts and part i of ALL classes is
        one grid per step on stream i (the current stream + the process-wide side streams), so that the drain of one
        part's step overlaps the fill of another's -- what ``BatchedJssEnv.rollout_steps`` does for a single batch."""
        if n_iter != 1:
            for _ in range(steps):
                self.rollout(kind, n_iter=n_iter, seed=seed, autoreset=autoreset, explore=explore)
            return
        self._check_reset("rollout_steps")
        be = self._backend
        k = _abi.policy_code(kind)
        flags = _abi.ROLLOUT_AUTORESET if autoreset else 0
        q16 = int(round(explore * 65536))
        with be.on_device():
            if self.launch == "streams" and hasattr(be, "stream_array") and self._n_sets > 1:
                # round 3's form: every class on its own stream, the library forks / joins them
                flags |= 0 if caller_orders_streams else _abi.ROLLOUT_FORK_JOIN
                rc = be.lib.jss_rollout_steps_multi(self._n_sets, *self._sets, k, self._seed(seed), q16, int(steps), flags,
                                                    be.stream_array(self._n_sets))
                _abi.check(be.lib, rc, "jss_rollout_steps_multi")
                return
            n = max(1, min(4, int(self.n_sub if n_sub is None else n_sub)))
            if hasattr(be, "stream_array"):
                streams = be.stream_array(n)
                if n > 1 and not caller_orders_streams:
                    flags |= _abi.ROLLOUT_FORK_JOIN
            else:
                streams = (C.c_void_p * n)()
            rc = be.lib.jss_multi_rollout(self._n_sets, *self._sets, k, self._seed(seed), q16, int(steps), flags, n, streams)
        _abi.check(be.lib, rc, "jss_multi_rollout")

    def policy(self, kind="random", seed=None, explore=0.0):
        """Per-class action buffers {class: (B_class,) int32} (each bucket's own preallocated tensor), one launch."""
        self._check_reset("policy")
        be = self._backend
        with be.on_device():
            rc = be.lib.jss_multi_policy(self._n_sets, self._sets[0], self._sets[1], _abi.policy_code(kind), self._seed(seed),
                                         int(round(explore * 65536)), self._policy_out, be.stream())
        _abi.check(be.lib, rc, "jss_multi_policy")
        return {k: b._actions_out for k, b in self._each()}

    def step(self, actions_per_bucket, autoreset: bool = False):
        """step() of every env, one launch; actions_per_bucket = {class: (B_class,) actions}.  Returns {class: (obs, reward,
        done, False, {})} like BatchedJssEnv.step."""
        self._check_reset("step")
        be = self._backend
        each = self._each()
        with be.on_device():
            staged = [b._stage(b._act_in, actions_per_bucket[k], "int32") for k, b in each]
            acts = (C.c_void_p * self._n_sets)(*[be.ptr(a) for a in staged])
            rc = be.lib.jss_multi_step(self._n_sets, self._sets[0], self._sets[1], acts, self._sets[2],
                                       _abi.ROLLOUT_AUTORESET if autoreset else 0, be.stream())
        _abi.check(be.lib, rc, "jss_multi_step")
        return {k: (b._obs(), b.reward, b.done, False, {}) for k, b in each}

    def synchronize(self):
        for _, b in self._each():
            b.synchronize()

    def counter_totals(self):
        parts = [b.counter_totals() for _, b in self._each()]
        tot = parts[0]
        for x in parts[1:]:
            tot = tot + x
        return tot

    def zero_counters(self):
        for _, b in self._each():
            b.zero_counters()

    def stats(self):
        tot = {"steps": 0, "episodes": 0, "makespan_sum": 0, "reward_num_sum": 0}
        for _, b in self._each():
            for key, v in b.stats().items():
                tot[key] += v
        return tot

    def host_state(self, i: int):
        return self.buckets[self.bucket_of[i]].host_state(int(self.slot[i]))
