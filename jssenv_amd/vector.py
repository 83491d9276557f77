"""gymnasium.vector-style facade over BatchedJssEnv.

The reference ships a single env only (its README loop steps one ``JssEnv``); RL code that
wants many of them wraps it in ``gymnasium.vector.SyncVectorEnv``.  ``JssVectorEnv`` offers
that calling convention directly on the batched engine:

    envs = JssVectorEnv("ta01", num_envs=4096, device="cuda:0")
    obs, info = envs.reset(seed=0)
    obs, rewards, terminations, truncations, infos = envs.step(actions)

* observations are the dict ``{"real_obs": (N, J, 7) float32, "action_mask": (N, J+1) bool}``
  as device tensors (``to_numpy=True`` returns host copies);
* autoreset follows gymnasium >= 1.0's default ("next step"): an env that terminated is reset by
  the following ``step`` call, whose action for that env is ignored;
* ``truncations`` is all False (the reference never truncates, jss_env.py:438), ``infos`` is
  ``{}`` (jss_env.py:439); the final makespans are in ``envs.makespan``.
"""
from __future__ import annotations

from typing import Optional

from .env import BatchedJssEnv
from .facade import gymnasium_base


class JssVectorEnv(gymnasium_base("VectorEnv")):
    """(a ``gymnasium.vector.VectorEnv`` when gymnasium is importable; its attributes are set here directly, the base
    class's constructor -- whose signature differs between gymnasium 0.29 and 1.x -- is not called)"""

    def __init__(self, instances, num_envs: Optional[int] = None, device=None, to_numpy: bool = False, order: Optional[str] = None,
                 _backend=None):
        # order="by_shape": a list of instances of different shapes, the envs dealt onto them class by class and stepped by
        # class-specialised kernel bodies on the padded tensors (BatchedJssEnv); `step` stays ONE launch (jss_multi_step)
        self.env = BatchedJssEnv(instances, batch=num_envs, device=device, order=order, _backend=_backend)
        self.num_envs = self.env.batch
        self.to_numpy = to_numpy
        self.jobs_per_env = self.env.jobs_per_env
        try:
            import gymnasium as gym
            J = self.env.jmax
            self.single_action_space = gym.spaces.Discrete(J + 1)
            self.single_observation_space = gym.spaces.Dict({
                "action_mask": gym.spaces.Box(0, 1, shape=(J + 1,)),
                "real_obs": gym.spaces.Box(low=0.0, high=1.0, shape=(J, 7), dtype=float),
            })
            try:      # the batched spaces gymnasium.vector callers look at (real gymnasium only)
                from gymnasium.vector.utils import batch_space
                self.action_space = batch_space(self.single_action_space, self.num_envs)
                self.observation_space = batch_space(self.single_observation_space, self.num_envs)
            except Exception:
                self.action_space = self.observation_space = None
        except Exception:  # gymnasium is optional
            self.single_action_space = self.single_observation_space = None
            self.action_space = self.observation_space = None
        self.closed = False

    def _out(self, x):
        return self.env.backend.numpy(x) if self.to_numpy else x

    def _obs(self):
        return {"real_obs": self._out(self.env.real_obs), "action_mask": self._out(self.env.action_mask != 0)}

    def reset(self, *, seed: Optional[int] = None, options=None):
        if seed is not None:
            self.env.seed = int(seed)
        self.env.reset()
        return self._obs(), {}

    def step(self, actions):
        _, reward, done, _, _ = self.env.step(actions, autoreset=True)
        terminations = done != 0
        truncations = terminations & False
        return self._obs(), self._out(reward), self._out(terminations), self._out(truncations), {}

    def sample_actions(self, kind="random", explore: float = 0.0):
        """Per-env actions from the on-device selectors (random masked, FIFO, SPT, ...)."""
        return self.env.policy(kind, explore=explore)

    @property
    def makespan(self):
        return self._out(self.env.makespan)

    def close(self, **kwargs):
        self.env.synchronize()
        self.closed = True
