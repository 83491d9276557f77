"""Dispatching rules with the reference's surface (JSSEnv/dispatching.py).

Same names (``ShortestProcessingTime`` ... ``CriticalRatio``, ``DISPATCHING_RULES``,
``get_rule``, ``compare_rules``), same call convention (``rule(env) -> action``,
``rule.run_episode(env) -> (total_reward, makespan)``) and the same semantics:

  * NOPE when it is the only legal action (dispatching.py:96-97 and twins);
  * arg-min / arg-max over the legal jobs with strict comparisons, so the lowest job
    index wins ties (:108, :148, ...);
  * then, if NOPE is legal, NOPE with probability 0.1 drawn from NumPy's *global* RNG
    (:113, :153, ...) -- kept on the host exactly so, which makes ``np.random.seed(s)``
    runs reproduce the reference's traces.

The arg-best itself runs on the GPU (``jss_policy``, csrc ``select_action``/``p_select``)
when ``env`` is a ``jssenv_amd.JssEnv``; any other env object exposing the reference's
attributes falls back to the attribute-reading loop of the reference rule (CriticalRatio's float
ratios with its per-episode due-date cache, :327-408, included).

For whole batches use ``BatchedJssEnv.rollout(kind)`` -- rule + step fused on the device.  Two entry points
of this module do that for you on a jssenv_amd env: ``rule.run_episode(env, device_rng=True)`` plays the whole
episode in fused launches (the 10 % NOPE exploration then comes from the device's counter RNG instead of
NumPy's global one -- same distribution, not the same stream), and ``compare_rules`` plays each rule's
``num_episodes`` episodes as ONE batch of ``num_episodes`` envs.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

EXPLORATION_PROBABILITY = 0.1  # dispatching.py:113


class DispatchingRule:
    """Base class (dispatching.py:21-75)."""

    kind: Optional[str] = None   # name of the on-device selector
    larger_wins = False

    def __init__(self, name: str, description: str):
        self.name = name
        self.description = description

    def get_name(self) -> str:
        return self.name

    def get_description(self) -> str:
        return self.description

    # value the rule ranks job `job` by (host path)
    def _value(self, env, job: int):
        raise NotImplementedError("Subclasses must implement __call__")

    def _best_job(self, env, legal_actions) -> int:
        if self.kind is not None and hasattr(env, "_rule_best"):
            # jssenv_amd env: the state of this step is already on the host (one copy per step()); the arg-best over
            # the legal jobs is one vectorised pass over it -- no second launch / copy / sync per decision
            return env._rule_best(self.kind, legal_actions)
        if self.kind is not None and hasattr(env, "_policy"):
            a = env._policy(self.kind)            # device arg-best, lowest index wins ties
            return a if a < env.jobs else -1
        best, best_v = -1, None
        for job in range(env.jobs):
            if legal_actions[job]:
                v = self._value(env, job)
                if best_v is None or (v > best_v if self.larger_wins else v < best_v):
                    best, best_v = job, v
        return best

    def __call__(self, env) -> int:
        legal_actions = env.get_legal_actions()
        if np.sum(legal_actions) == 1 and legal_actions[-1]:
            return env.jobs
        job = self._best_job(env, legal_actions)
        if legal_actions[env.jobs] and np.random.random() < EXPLORATION_PROBABILITY:
            return env.jobs
        return job

    def run_episode(self, env, device_rng: bool = False, seed: Optional[int] = None) -> Tuple[float, int]:
        """dispatching.py:55-75.  ``device_rng=True`` (jssenv_amd envs only): rule + step fused on the device for the
        whole episode, exploration drawn from the counter RNG keyed by ``seed``."""
        if device_rng:
            if self.kind is None or not hasattr(env, "_b") or not _device_rule_ok(self):
                raise ValueError("device_rng=True needs a jssenv_amd.JssEnv and a rule with an on-device selector")
            return env._run_rule(device_kind(self), explore=EXPLORATION_PROBABILITY, seed=seed)
        env.reset()
        done = False
        total_reward = 0.0
        while not done:
            action = self(env)
            _, reward, done, _, _ = env.step(action)
            total_reward += reward
        return total_reward, env.current_time_step


def _device_rule_ok(rule) -> bool:
    """CriticalRatio's device selector takes due-date factors p / q with q a power of two (exact in the reference's
    doubles); any other factor stays on the host."""
    return device_kind(rule) is not None


def device_kind(rule):
    """What the device entry points take as `kind` for this rule (None: no on-device selector)."""
    if rule.kind is None:
        return None
    if hasattr(rule, "due_date_factor"):
        from ._abi import cr_kind
        return cr_kind(rule.due_date_factor)
    return rule.kind


def _remaining_work(env, job: int) -> int:                                # dispatching.py:187-189
    return int(sum(env.instance_matrix[job][op][1] for op in range(env.todo_time_step_job[job], env.machines)))


class ShortestProcessingTime(DispatchingRule):                            # dispatching.py:78-116
    kind, larger_wins = "SPT", False

    def __init__(self):
        super().__init__("SPT", "SPT - among the legal jobs, start the one whose current operation is shortest")

    def _value(self, env, job):
        return env.instance_matrix[job][env.todo_time_step_job[job]][1]


class FirstInFirstOut(DispatchingRule):                                   # dispatching.py:119-156
    kind, larger_wins = "FIFO", True

    def __init__(self):
        super().__init__("FIFO", "FIFO - among the legal jobs, start the one that has been idle longest since its last operation")

    def _value(self, env, job):
        return env.idle_time_jobs_last_op[job]


class MostWorkRemaining(DispatchingRule):                                 # dispatching.py:159-199
    kind, larger_wins = "MWR", True

    def __init__(self):
        super().__init__("MWR", "MWR - among the legal jobs, start the one with the largest sum of remaining durations")

    def _value(self, env, job):
        return _remaining_work(env, job)


class LeastWorkRemaining(DispatchingRule):                                # dispatching.py:202-242
    kind, larger_wins = "LWR", False

    def __init__(self):
        super().__init__("LWR", "LWR - among the legal jobs, start the one with the smallest sum of remaining durations")

    def _value(self, env, job):
        return _remaining_work(env, job)


class MostOperationsRemaining(DispatchingRule):                           # dispatching.py:245-283
    kind, larger_wins = "MOR", True

    def __init__(self):
        super().__init__("MOR", "MOR - among the legal jobs, start the one with the most operations still to do")

    def _value(self, env, job):
        return env.machines - env.todo_time_step_job[job]


class LeastOperationsRemaining(DispatchingRule):                          # dispatching.py:286-324
    kind, larger_wins = "LOR", False

    def __init__(self):
        super().__init__("LOR", "LOR - among the legal jobs, start the one with the fewest operations still to do")

    def _value(self, env, job):
        return env.machines - env.todo_time_step_job[job]


class CriticalRatio(DispatchingRule):                                     # dispatching.py:327-408
    """(due date - now) / remaining work, smallest first; due date = factor x job length, cached
    per job and dropped when ``current_time_step == 0`` (:373-374).  On a jssenv_amd env the arg-min runs on
    the device -- as an exact fraction comparison (csrc ``cr_better``) for factors p / 2^k, as the reference's float64
    expression (``cr_ratio_f64``) for any other: ``BatchedJssEnv.policy("CR", cr_factor=f)``; the float path below
    serves other envs."""

    kind, larger_wins = "CR", False

    def __init__(self, due_date_factor: float = 1.5):
        super().__init__("CR", "CR - among the legal jobs, start the one with the smallest (due date - now) / remaining work")
        self.due_date_factor = due_date_factor
        self._due_dates: Dict[int, float] = {}

    def _calculate_due_date(self, env, job: int) -> float:                # :351-363
        if job not in self._due_dates:
            total = sum(env.instance_matrix[job][op][1] for op in range(env.machines))
            self._due_dates[job] = total * self.due_date_factor
        return self._due_dates[job]

    def _best_job(self, env, legal_actions) -> int:
        # on a jssenv_amd env the arg-min comes from the host snapshot of the step (the reference's float expression,
        # vectorised) or from the device selectors: (p * job_length - q * now) / remaining compared exactly for factors
        # p / q with q a power of two (these also run inside the fused rollouts), the reference's float64 expression
        # itself for any other factor (JSS_POLICY_CR_F64, policy launches only); other envs take the loop below
        if hasattr(env, "_rule_best"):
            return env._rule_best("CR", legal_actions, due_date_factor=self.due_date_factor)
        if hasattr(env, "_policy"):
            code = device_kind(self)          # p / 2^k factors: the integer-exact selector; any other: the float64 one
            a = env._policy(code) if code is not None else env._policy("CR", cr_factor=self.due_date_factor)
            return a if a < env.jobs else -1
        saved, self.kind = self.kind, None
        try:
            return super()._best_job(env, legal_actions)
        finally:
            self.kind = saved

    def _value(self, env, job):
        remaining = _remaining_work(env, job)
        time_remaining = self._calculate_due_date(env, job) - env.current_time_step
        return time_remaining / remaining if remaining > 0 else float("inf")   # :395-398

    def __call__(self, env) -> int:
        legal_actions = env.get_legal_actions()
        if np.sum(legal_actions) == 1 and legal_actions[-1]:
            return env.jobs
        if env.current_time_step == 0:                                    # :373-374
            self._due_dates = {}
        job = self._best_job(env, legal_actions)
        if legal_actions[env.jobs] and np.random.random() < EXPLORATION_PROBABILITY:
            return env.jobs
        return job


DISPATCHING_RULES = {                                                     # dispatching.py:412-420
    "SPT": ShortestProcessingTime(),
    "FIFO": FirstInFirstOut(),
    "MWR": MostWorkRemaining(),
    "LWR": LeastWorkRemaining(),
    "MOR": MostOperationsRemaining(),
    "LOR": LeastOperationsRemaining(),
    "CR": CriticalRatio(),
}


def get_rule(rule_name: str) -> DispatchingRule:                          # dispatching.py:423-439
    try:
        return DISPATCHING_RULES[rule_name]
    except KeyError:
        raise ValueError(f"Rule '{rule_name}' not found. Available rules: {list(DISPATCHING_RULES)}") from None


def compare_rules(env, rules: Optional[List[str]] = None, num_episodes: int = 10,
                  seed: Optional[int] = None) -> Dict[str, Dict[str, float]]:
    """Mean total reward and mean makespan of each rule over ``num_episodes`` episodes on ``env``
    (same result keys as dispatching.py:442-475).

    On a jssenv_amd env the ``num_episodes`` episodes of a rule are one batch of ``num_episodes`` envs of the same
    instance: rule + step fused in the rollout kernel, the rules' 10 % NOPE exploration drawn per env from the
    counter RNG (``seed`` keys it; default: a draw from NumPy's global RNG, so ``np.random.seed`` still makes the
    comparison reproducible).  Any other env object takes the reference's sequential loop."""
    names = list(DISPATCHING_RULES) if rules is None else list(rules)
    on_device = hasattr(env, "_b") and all(get_rule(n).kind is not None and _device_rule_ok(get_rule(n)) for n in names)
    results = {}
    if on_device and num_episodes > 0:
        from .env import BatchedJssEnv
        base = int(np.random.randint(0, 2**31 - 1)) if seed is None else int(seed)
        inst = env.instance
        batch = BatchedJssEnv([inst], batch=int(num_episodes), seed=base, _backend=env._b.backend)
        # an episode is J * M allocations plus its NOPEs; chunks of that size until every env reports done
        chunk = inst.jobs * inst.machines + 16
        for k, name in enumerate(names):
            batch.seed = base + 7919 * k
            batch.reset()
            batch.zero_counters()
            for _ in range(64):
                batch.rollout(device_kind(get_rule(name)), n_iter=chunk, autoreset=False, explore=EXPLORATION_PROBABILITY)
                if bool(batch.backend.numpy(batch.done).all()):
                    break
            else:
                raise RuntimeError(f"rule {name}: episodes did not finish")
            cnt = batch.backend.numpy(batch.counters)
            results[name] = {"avg_reward": float(cnt[:, 3].sum()) / inst.max_time_op / num_episodes,
                             "avg_makespan": float(batch.backend.numpy(batch.makespan).sum()) / num_episodes}
        return results
    for name in names:
        episodes = [get_rule(name).run_episode(env) for _ in range(num_episodes)]
        results[name] = {"avg_reward": sum(r for r, _ in episodes) / num_episodes,
                         "avg_makespan": sum(m for _, m in episodes) / num_episodes}
    return results
