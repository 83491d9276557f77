"""Step session: ``obs, r, done, _, _ = env.step(policy(obs))`` with the env state resident on the chip.

``BatchedJssEnv.session()`` opens one (include/jss_hip.h, ``jss_session_*``): a kernel that lives across steps holds
the state of every env of the batch in registers / LDS, takes each step's actions from a device mailbox and writes that
step's observation, mask, reward and done flag.  The caller's stream only runs two small kernels per exchange: ``post``
(the actions of one or more steps into the mailbox) and ``wait`` (returns when those steps' outputs are visible).

    with env.session(depth=8) as s:
        for _ in range(n):
            actions = policy(env.real_obs, env.action_mask)     # the caller's kernels, on the caller's stream
            obs, reward, done = s.step(actions)                 # post + wait
    # the batch is an ordinary batch again: state tensors and counters are current

``post`` may run ahead of ``wait`` by up to ``depth`` steps (an open-loop action sequence, or a learner that tolerates
a lag): the resident kernel then runs back to back, never waiting for the host.

While a session is open, synchronize YOUR stream (``torch.cuda.current_stream().synchronize()``), never the device:
``torch.cuda.synchronize()`` waits for every stream, the resident kernel's included, and that kernel only ends when the
session is closed (it would return when the kernel's own timeout fires, with the session dead).

Between open and close the state tensors (job records, header, machine clocks) and the counters in memory are stale;
the outputs (``real_obs``, ``action_mask``, ``reward``, ``done``, ``makespan``, ``solution``) are current after every
``wait``.  Other calls on the env raise while a session is open.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi


class StepSession:
    def __init__(self, env, depth: int = 16, timeout_ms: int = 10000, slots: int = 0):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if not env._is_reset:
            raise RuntimeError("call reset() before opening a session")
        if getattr(env, "_session", None) is not None:
            raise RuntimeError("this env already has an open session")
        self.env, self.depth = env, int(depth)
        be = self.be = env.backend
        B = env.batch
        with be.on_device():
            self.mail = be.zeros((self.depth, B), "int64")
            self.progress = be.zeros((B,), "int32")
            self.status = be.zeros((4,), "int32")
        self._sess = _abi.JssSession(be.ptr(self.mail), be.ptr(self.progress), be.ptr(self.status), self.depth,
                                     int(timeout_ms), int(slots), 0)
        self.posted = self.waited = 0
        self.closed = False
        self._stream = None
        torch = getattr(be, "torch", None)
        d, s, o = env._refs()
        with be.on_device():
            if torch is not None:
                # The resident kernel gets a stream of its own, behind what the env has queued -- and a PRIORITY of its own:
                # HIP maps the streams of one priority onto a small pool of hardware queues, and a post / wait kernel that
                # lands in the queue the resident kernel occupies would wait for it to END (seen on the first bench run:
                # every wavefront ran into its timeout).  High-priority streams have their own pool of queues.
                self._stream = torch.cuda.Stream(device=be.device, priority=-1)
                self._stream.wait_stream(torch.cuda.current_stream(be.device))
                stream = self._stream.cuda_stream
            else:
                stream = be.stream()
            rc = be.lib.jss_session_open(d, s, o, C.byref(self._sess), stream)
        _abi.check(be.lib, rc, "jss_session_open")
        env._session = self

    # -- the exchange -----------------------------------------------------------------------------------------
    def post(self, actions):
        """Actions of the next step ((B,) int32) or the next n steps ((n, B) int32), device resident: one small kernel
        on the current stream writes them into the mailbox.  At most ``depth`` steps may be posted and not yet waited for."""
        if self.closed:
            raise RuntimeError("the session is closed")
        be, env = self.be, self.env
        shape = tuple(actions.shape)
        if shape == (env.batch,):
            n = 1
        elif len(shape) == 2 and shape[1] == env.batch and shape[0] >= 1:
            n = shape[0]
        else:
            raise ValueError(f"expected actions of shape ({env.batch},) or (n, {env.batch}), got {shape}")
        if self.posted + n - self.waited > self.depth:
            raise RuntimeError(f"mailbox ring overrun: {self.posted + n - self.waited} steps in flight, depth {self.depth}: wait() first")
        a = self._as_actions(actions)
        with be.on_device():
            rc = be.lib.jss_session_post(C.byref(env._desc), C.byref(self._sess), be.ptr(a), self.posted, n, self.waited, be.stream())
        _abi.check(be.lib, rc, "jss_session_post")
        self.posted += n
        return self.posted

    def _as_actions(self, actions):
        be = self.be
        if isinstance(actions, np.ndarray) or not hasattr(be, "torch"):
            a = np.ascontiguousarray(np.asarray(actions), dtype=np.int32)
            if hasattr(be, "torch"):
                a = be.from_numpy(a)
            self._keep = a
            return a
        t = be.torch
        a = actions
        if a.device != be.device or a.dtype != t.int32 or not a.is_contiguous():
            a = a.to(device=be.device, dtype=t.int32).contiguous()
        self._keep = a                     # alive until the post kernel has read it (stream-ordered: the next call at the latest)
        return a

    def wait(self, steps=None):
        """Stream-ordered wait on the current stream until ``steps`` steps (default: everything posted) are finished and
        their outputs visible to the kernels enqueued behind it.  Does not block the host."""
        be, env = self.be, self.env
        n = self.posted if steps is None else int(steps)
        if n > self.posted or n < self.waited:
            raise ValueError(f"cannot wait for {n} steps: {self.waited} already waited for, {self.posted} posted")
        with be.on_device():
            rc = be.lib.jss_session_wait(C.byref(env._desc), C.byref(self._sess), n, be.stream())
        _abi.check(be.lib, rc, "jss_session_wait")
        self.waited = n

    def step(self, actions):
        """post + wait of one step in ONE launch on the current stream; returns (obs dict, reward, done) -- the env's own
        output tensors, current for the kernels enqueued behind this call."""
        if self.closed:
            raise RuntimeError("the session is closed")
        be, env = self.be, self.env
        if tuple(actions.shape) != (env.batch,):
            raise ValueError(f"expected actions of shape ({env.batch},), got {tuple(actions.shape)}")
        if self.posted != self.waited:
            self.wait()
        a = self._as_actions(actions)
        with be.on_device():
            rc = be.lib.jss_session_step(C.byref(env._desc), C.byref(self._sess), be.ptr(a), self.posted, be.stream())
        _abi.check(be.lib, rc, "jss_session_step")
        self.posted += 1
        self.waited = self.posted
        return env._obs(), env.reward, env.done

    # -- life cycle --------------------------------------------------------------------------------------------
    def close(self, check: bool = True):
        """Finish every posted step, have the resident kernel write the state back and leave; the current stream
        continues behind it.  With ``check`` the host waits and raises if a device-side wait ran out of time."""
        if self.closed:
            return
        be, env = self.be, self.env
        self.wait()
        with be.on_device():
            rc = be.lib.jss_session_close(C.byref(env._desc), C.byref(self._sess), self.posted, be.stream())
            _abi.check(be.lib, rc, "jss_session_close")
            if self._stream is not None:
                be.torch.cuda.current_stream(be.device).wait_stream(self._stream)
        self.closed = True
        env._session = None
        if check:
            st = self.host_status()
            if st["session_timeouts"] or st["wait_timeouts"]:
                raise RuntimeError(f"step session timed out on the device: {st}")

    def host_status(self):
        """Blocks until the session's work so far is done; returns the status words."""
        be = self.be
        if self._stream is not None and self.closed:
            self._stream.synchronize()
        be.sync()
        st = be.numpy(self.status)
        return {"session_timeouts": int(st[0]), "wait_timeouts": int(st[1]), "wavefronts_exited": int(st[2]),
                "env_sets_per_wavefront": int(st[3])}

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close(check=exc_type is None)
        return False
