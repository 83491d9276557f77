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

Between open and close the state tensors (job records, header, machine clocks) and the counters in memory are stale, and
every other call on the env (``step``, ``rollout``, ``state_dict``, ``host_tensors`` ...) raises.

The outputs (``real_obs``, ``action_mask``, ``reward``, ``done``, ``makespan``, ``solution``) are single buffers that the
resident kernel overwrites step after step: they hold step n's values -- whole, untorn -- for the kernels enqueued behind a
``wait`` that leaves NOTHING in flight (``wait()`` of everything posted, or ``step()``).  After ``wait(n)`` with more than n
steps posted the kernel is already writing step n + 1 over them: their content is undefined until the caller has waited
for everything it posted (``outputs_current`` says which).  A caller that posts ahead and wants every step's outputs uses
``BatchedJssEnv.steps(actions, record=...)`` instead, which keeps them per step.

A session whose resident kernel saw no mail for ``timeout_ms`` (a long learner update, a debugger, a device-wide
synchronisation) has written its state back and left: ``step`` / ``wait`` notice within ``check_every`` calls (an
asynchronous copy of the status words, no host synchronisation) and raise instead of handing out stale observations.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi


class StepSession:
    def __init__(self, env, depth: int = 16, timeout_ms: int = 10000, slots: int = 0, check_every: int = 64):
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
        # liveness probe: every `check_every` exchanges the status words are copied to pinned host memory behind the caller's
        # work (asynchronous); the copy of the previous probe is looked at when its event has fired
        self.check_every, self._calls, self._probe = max(1, int(check_every)), 0, None
        if torch is not None:
            self._status_host = torch.zeros(4, dtype=torch.int32, pin_memory=True)
            self._probe_event = torch.cuda.Event()
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

    @property
    def outputs_current(self) -> bool:
        """True when the env's output tensors hold exactly the last waited-for step (nothing posted beyond it)."""
        return self.posted == self.waited

    def _alive(self):
        """Raise if the resident kernel has given the session up (see the module docstring); costs a host-side event query,
        and one 16-byte asynchronous copy every `check_every` calls."""
        be = self.be
        torch = getattr(be, "torch", None)
        if torch is None:                        # host twin / emulator: synchronous, the status words are host memory
            st = be.numpy(self.status)
            dead, waits = int(st[0]), int(st[1])
        else:
            dead = waits = 0
            if self._probe is not None and self._probe_event.query():
                dead, waits = int(self._status_host[0]), int(self._status_host[1])
                self._probe = None
            self._calls += 1
            if self._probe is None and self._calls % self.check_every == 0:
                with be.on_device():
                    self._status_host.copy_(self.status, non_blocking=True)
                    self._probe_event.record(torch.cuda.current_stream(be.device))
                self._probe = True
        if dead or waits:
            raise RuntimeError(f"the step session is dead: {dead} wavefront(s) of its resident kernel saw no actions for timeout_ms "
                               f"and left, {waits} wait(s) gave up -- the outputs are stale from there on; close(check=False) it "
                               "and open a new one")

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
        """Stream-ordered wait on the current stream until ``steps`` steps (default: everything posted) are finished.  Does
        not block the host.  The env's output tensors are defined for the kernels enqueued behind it only when this leaves
        nothing in flight (``steps`` == everything posted): see the module docstring."""
        be, env = self.be, self.env
        n = self.posted if steps is None else int(steps)
        if n > self.posted or n < self.waited:
            raise ValueError(f"cannot wait for {n} steps: {self.waited} already waited for, {self.posted} posted")
        self._alive()
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
        else:
            self._alive()
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
        wait_rc = 0
        if self.posted != self.waited:          # (not through wait(): a dead session must still be closable)
            with be.on_device():
                wait_rc = be.lib.jss_session_wait(C.byref(env._desc), C.byref(self._sess), self.posted, be.stream())
            if wait_rc == 0:
                self.waited = self.posted       # (a wait that was never launched has waited for nothing)
        with be.on_device():
            rc = be.lib.jss_session_close(C.byref(env._desc), C.byref(self._sess), self.posted, be.stream())
            if check or rc == 0:
                _abi.check(be.lib, rc, "jss_session_close")
            if self._stream is not None:
                be.torch.cuda.current_stream(be.device).wait_stream(self._stream)
        self.closed = True
        env._session = None
        if check:
            # the wait in front of the close: a launch error or JSS_E_SESSION is the caller's to know (the session is closed
            # either way -- the close granule is posted behind whatever the wait managed to enqueue)
            _abi.check(be.lib, wait_rc, "jss_session_wait")
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
