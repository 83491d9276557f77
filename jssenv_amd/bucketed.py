"""Shape-bucketed batches: ragged instance mixes without the padding cost.

``BatchedJssEnv`` pads every env of a batch to the largest (Jmax, Mmax), and the kernel
flavour is chosen from that padded shape: one 15x15 env in a batch that also holds a 100x20
instance runs on the two-jobs-per-lane wave kernel with 85 % of its lanes idle.
``BucketedJssEnv`` splits the envs into shape classes

    class 0: J, M <= 16   -> packed kernel, 4 envs per wavefront
    class 1: J, M <= 32   -> packed kernel, 2 envs per wavefront
    class 2: J <= 64      -> one wavefront per env
    class 3: J <= 128     -> one wavefront per env, two jobs per lane

each backed by its own compactly padded ``BatchedJssEnv`` (own tensors, own launches, all
on the caller's stream).  Env ``i`` of the wrapper lives at ``slot[i]`` of bucket
``bucket_of[i]``; the per-env RNG key stays the global env id, so a bucketed run is the same
random process as the padded one.  Outputs stay per bucket (an RL policy on ragged
instances batches by shape anyway); ``gather()`` assembles padded (B, Jmax, ...) views on
demand.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

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
                 env_id_base: int = 0, concurrent: bool = True, _backend=None):
        from .env import make_backend
        if _backend is None:
            _backend = make_backend(device)     # one backend (library handle, device) shared by the buckets
        insts = [resolve_instance(i) for i in instances]
        n = len(insts)
        self.batch = B = int(batch) if batch is not None else n
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
            sub = BatchedJssEnv([insts[t] for t in tables], batch=env_ids.size, device=device, seed=seed,
                                table_of_env=[remap[t] for t in self.table_of_env[env_ids]], _backend=_backend)
            # RNG streams are keyed by the GLOBAL env id: give the bucket an explicit id list when its
            # members are not a contiguous range
            sub.set_env_ids(env_id_base + env_ids)
            self.buckets[k] = sub

        # one side stream per bucket: the buckets are independent env sets, so their launches may overlap.
        # Every call forks from / joins back to the caller's current stream, so callers see ordinary
        # stream-ordered semantics (and a graph capture of the caller's stream records the fork/join too).
        self._backend = _backend
        self._multi = None
        self._torch = getattr(_backend, "torch", None) if concurrent else None
        self._streams = None
        self._device = getattr(_backend, "device", None)
        if self._torch is not None and len(self._each()) > 1:
            # the process-wide side streams of the device (HipBackend.side_pool): creating fresh streams per object
            # lets late-created ones alias the caller's hardware queue
            pool = _backend.side_pool(len(self._each()))
            self._streams = {k: pool["streams"][i] for i, (k, _) in enumerate(self._each())}
            # the fork / join events are this object's own (two env objects driven from two host threads must not
            # re-record each other's fork event); the streams are shared, which only serialises such envs
            self._fork_event = self._torch.cuda.Event()
            self._join_events = {k: self._torch.cuda.Event() for k, _ in self._each()}

    def close(self):
        """Wait for outstanding work and stop using the (process-wide) side streams."""
        self.synchronize()
        self._streams = None
        self._fork_event = None
        self._join_events = None

    def _each(self):
        return [(k, b) for k, b in enumerate(self.buckets) if b is not None]

    def _fan_out(self, fn):
        """fn(k, bucket) per bucket, each on its own stream when available."""
        if self._streams is None:
            return {k: fn(k, b) for k, b in self._each()}
        t = self._torch
        main = t.cuda.current_stream(self._device)      # the env's device, which need not be the current one
        self._fork_event.record(main)
        out = {}
        for k, b in self._each():
            st = self._streams[k]
            st.wait_event(self._fork_event)
            with t.cuda.stream(st):
                out[k] = fn(k, b)
            self._join_events[k].record(st)
        for k, _ in self._each():
            main.wait_event(self._join_events[k])
        return out

    def reset(self):
        return self._fan_out(lambda k, b: b.reset())

    def rollout(self, kind="random", n_iter=1, seed=None, autoreset=True, explore=0.0):
        self._fan_out(lambda k, b: b.rollout(kind, n_iter=n_iter, seed=seed, autoreset=autoreset, explore=explore))

    def rollout_steps(self, kind="random", steps=1, n_iter=1, seed=None, autoreset=True, explore=0.0, chunk=8,
                      caller_orders_streams=False):
        """`steps` consecutive rollout(n_iter) launches per bucket with ONE fork/join around the whole
        window: bucket k's launch i+1 depends only on bucket k's launch i, so the buckets run ahead of
        each other on their own streams (no per-step synchronisation, no graph capture needed).

        The launches are issued round-robin over the buckets in chunks of `chunk` steps (with n_iter == 1 each
        chunk is one call of the C loop of jss_rollout_steps, 2.7 us of host time per launch).  Bucket-major order
        -- all of bucket 0's launches, then all of bucket 1's -- is 3x slower whenever two of the streams share a
        hardware queue (HIP deals streams onto 4 queues): the second bucket then waits for the whole window of the
        first.  Interleaved, an aliased pair loses its overlap and nothing more."""
        if self._streams is None:
            for _, b in self._each():
                self._run_bucket(b, kind, steps, n_iter, seed, autoreset, explore)
            return
        each = self._each()
        # the single-call path passes ONE seed for every bucket and at most 16 env sets: with per-bucket seeds that
        # differ (seed=None) or more buckets, the per-bucket path below does the same work in more host calls
        one_seed = seed is not None or len({b.seed for _, b in each}) == 1
        if n_iter == 1 and hasattr(self._backend, "stream_array") and one_seed and len(each) <= 16:
            # ONE host call for the whole window: the library issues the launches step-major over the buckets and
            # forks / joins the side streams itself (jss_rollout_steps_multi)
            import ctypes as C
            from . import _abi
            be = self._backend
            if not all(b._is_reset for _, b in each):
                raise RuntimeError("call reset() before rollout_steps()")
            n = len(each)
            ident = tuple(id(b) for _, b in each)
            if self._multi is None or self._multi[3] != ident:      # (rebuilt if the bucket set ever changes)
                D, S, O = C.POINTER(_abi.JssDesc), C.POINTER(_abi.JssState), C.POINTER(_abi.JssOut)
                self._multi = ((D * n)(*[C.pointer(b._desc) for _, b in each]), (S * n)(*[C.pointer(b._state) for _, b in each]),
                               (O * n)(*[C.pointer(b._out) for _, b in each]), ident)
            k = _abi.POLICY[kind] if isinstance(kind, str) else int(kind)
            # caller_orders_streams: the device is idle now and the caller synchronises the whole device afterwards
            flags = (_abi.ROLLOUT_AUTORESET if autoreset else 0) | (0 if caller_orders_streams else _abi.ROLLOUT_FORK_JOIN)
            sd = each[0][1].seed if seed is None else int(seed)
            with be.on_device():
                rc = be.lib.jss_rollout_steps_multi(n, *self._multi[:3], k, sd, int(round(explore * 65536)), int(steps), flags,
                                                    be.stream_array(n))
            _abi.check(be.lib, rc, "jss_rollout_steps_multi")
            return
        t = self._torch
        main = t.cuda.current_stream(self._device)
        self._fork_event.record(main)
        for k, _ in self._each():
            self._streams[k].wait_event(self._fork_event)
        done = 0
        while done < steps:
            n = min(chunk, steps - done)
            for k, b in self._each():
                with t.cuda.stream(self._streams[k]):
                    self._run_bucket(b, kind, n, n_iter, seed, autoreset, explore)
            done += n
        for k, _ in self._each():
            self._join_events[k].record(self._streams[k])
            main.wait_event(self._join_events[k])

    @staticmethod
    def _run_bucket(b, kind, steps, n_iter, seed, autoreset, explore):
        if n_iter == 1:
            b.rollout_steps(kind, steps=steps, n_sub=1, seed=seed, autoreset=autoreset, explore=explore)
        else:
            for _ in range(steps):
                b.rollout(kind, n_iter=n_iter, seed=seed, autoreset=autoreset, explore=explore)

    def policy(self, kind="random", seed=None, explore=0.0):
        """Per-bucket action buffers (each bucket's own preallocated tensor: nothing is allocated on the side streams)."""
        return self._fan_out(lambda k, b: b.policy(kind, seed=seed, explore=explore))

    def step(self, actions_per_bucket):
        return self._fan_out(lambda k, b: b.step(actions_per_bucket[k]))

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
