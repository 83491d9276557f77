"""Single-env Python / NumPy restatement of the reference simulator -- TEST INFRASTRUCTURE and the
reference-speed CPU baseline of bench.py (cpu_baseline kind "restatement").  NOT product code: nothing
under jssenv_amd/ imports it.

Why it exists: the reference (prosysscience/JSSEnv v1.1.0, JSSEnv/envs/jss_env.py) is pure Python over
NumPy arrays and cannot travel to the GPU box, so its speed cannot be measured there.  This file does the
same work in the same way -- one Python-level pass over the jobs per event, NumPy arrays as storage, a
sorted Python list as the event queue, the M x J ``illegal_actions`` matrix, stored counters, the float64
observation written at the reference's write points -- so that timing it on the GPU box's host cores
stands in for timing the reference's ``step()`` there.  It was
written from oracle/jss_oracle.c (this repo's C restatement), attribute for attribute, and is pinned the
same way: tests/test_oracle_golden.py replays every golden trace captured from the live reference through
it and demands equality on every integer, on the float64 observation and on the reward.

Reference lines are cited as ``:NNN`` (JSSEnv/envs/jss_env.py).
"""
from __future__ import annotations

import bisect

import numpy as np

INF = float("inf")


class NumpyJssEnv:
    """Same public attributes and methods as the reference's ``JssEnv`` (:27-119 for the constants)."""

    def __init__(self, instance):
        self.instance = instance
        self.jobs, self.machines = int(instance.jobs), int(instance.machines)
        self.instance_matrix = np.asarray(instance.instance_matrix)          # (J, M, 2): machine, duration  :85
        dur = self.instance_matrix[:, :, 1]
        self.max_time_op = int(dur.max())                                     # :86
        self.jobs_length = dur.sum(axis=1)                                    # :87
        self.sum_op = int(dur.sum())                                          # :88
        self.max_time_jobs = int(self.jobs_length.max())                      # :89
        self.reset()

    # -- observation ----------------------------------------------------------------------------------
    def _observe(self):                                                       # :121-134
        self.state[:, 0] = self.legal_actions[:-1]
        return {"real_obs": self.state, "action_mask": self.legal_actions}

    def get_legal_actions(self):                                              # :136-143
        return self.legal_actions

    # -- reset ----------------------------------------------------------------------------------------
    def reset(self):                                                          # :145-181
        J, M = self.jobs, self.machines
        self.current_time_step = 0
        self.next_time_step, self.next_jobs = [], []
        self.nb_legal_actions, self.nb_machine_legal = J, 0
        self.legal_actions = np.ones(J + 1, dtype=bool)
        self.legal_actions[J] = False
        self.solution = np.full((J, M), -1, dtype=int)
        self.time_until_available_machine = np.zeros(M, dtype=int)
        self.time_until_finish_current_op_jobs = np.zeros(J, dtype=int)
        self.todo_time_step_job = np.zeros(J, dtype=int)
        self.total_perform_op_time_jobs = np.zeros(J, dtype=int)
        self.needed_machine_jobs = np.zeros(J, dtype=int)
        self.total_idle_time_jobs = np.zeros(J, dtype=int)
        self.idle_time_jobs_last_op = np.zeros(J, dtype=int)
        self.illegal_actions = np.zeros((M, J), dtype=bool)
        self.action_illegal_no_op = np.zeros(J, dtype=bool)
        self.machine_legal = np.zeros(M, dtype=bool)
        for j in range(J):                                                    # :174-179
            m = self.instance_matrix[j][0][0]
            self.needed_machine_jobs[j] = m
            if not self.machine_legal[m]:
                self.machine_legal[m] = True
                self.nb_machine_legal += 1
        self.state = np.zeros((J, 7), dtype=float)
        return self._observe()

    # -- final-op suppression heuristic -----------------------------------------------------------------
    def _prioritization_non_final(self):                                      # :183-254
        if self.nb_machine_legal < 1:
            return
        last = self.machines - 1
        for m in range(self.machines):
            if not self.machine_legal[m]:
                continue
            finals, n_other, shortest = [], 0, INF
            for j in range(self.jobs):
                if self.needed_machine_jobs[j] != m or not self.legal_actions[j]:
                    continue
                k = self.todo_time_step_job[j]
                if k == last:                                                 # :217
                    finals.append(j)
                    continue
                d = self.instance_matrix[j][k][1]                             # :222
                after = self.instance_matrix[j][k + 1][0]                     # :227
                if self.time_until_available_machine[after] == 0:             # :234
                    shortest = min(shortest, d)
                    n_other += 1
            if n_other:                                                       # :243
                for j in finals:
                    if self.instance_matrix[j][self.todo_time_step_job[j]][1] > shortest:   # :252
                        self.legal_actions[j] = False
                        self.nb_legal_actions -= 1

    # -- is waiting (NOPE) worth offering? ----------------------------------------------------------------
    def _walk(self, j, k, when, horizon, horizon_of, seen):
        """Look-ahead over the future ops of job j (:340-363 and :380-401); True = NOPE became legal."""
        last = self.machines - 1
        while k < last and horizon > when:
            m = self.instance_matrix[j][k][0]
            if horizon_of[m] > when and self.machine_legal[m]:
                seen.add(m)
                if len(seen) == self.nb_machine_legal:                        # :357 / :395
                    self.legal_actions[self.jobs] = True
                    return True
            when += self.instance_matrix[j][k][1]                             # :362 / :400
            k += 1
        return False

    def _check_no_op(self):                                                   # :256-401
        J, M = self.jobs, self.machines
        self.legal_actions[J] = False                                         # :278
        if not (len(self.next_time_step) > 0 and self.nb_machine_legal <= 3 and self.nb_legal_actions <= 4):
            return                                                            # :284-288
        seen = set()
        first_event = self.next_time_step[0]                                  # :293
        horizon = self.current_time_step                                      # :296
        horizon_of = [self.current_time_step + self.max_time_op] * M          # :300-302
        for j in range(J):                                                    # pass 1, ascending job order :305-321
            if self.legal_actions[j]:
                k = self.todo_time_step_job[j]
                m = self.instance_matrix[j][k][0]
                end = self.current_time_step + self.instance_matrix[j][k][1]
                if end < first_event:                                         # :314-315
                    return
                horizon_of[m] = min(horizon_of[m], end)                       # :318
                horizon = max(horizon, horizon_of[m])                         # :321
        for j in range(J):                                                    # pass 2 :324-401
            if self.legal_actions[j]:
                continue
            k = self.todo_time_step_job[j]
            if self.time_until_finish_current_op_jobs[j] > 0 and k + 1 < M:   # :327-330
                when = self.current_time_step + self.time_until_finish_current_op_jobs[j]
                if self._walk(j, k + 1, when, horizon, horizon_of, seen):
                    return
            elif not self.action_illegal_no_op[j] and k < M:                  # :366-369
                m = self.instance_matrix[j][k][0]
                when = self.current_time_step + self.time_until_available_machine[m]
                if self._walk(j, k, when, horizon, horizon_of, seen):
                    return

    # -- step -------------------------------------------------------------------------------------------
    def step(self, action):                                                   # :403-481
        J = self.jobs
        reward = 0.0
        if action == J:                                                       # :419 NOPE
            self.nb_machine_legal = 0
            self.nb_legal_actions = 0
            for j in range(J):                                                # :422-428
                if self.legal_actions[j]:
                    self.legal_actions[j] = False
                    m = self.needed_machine_jobs[j]
                    self.machine_legal[m] = False
                    self.illegal_actions[m][j] = True
                    self.action_illegal_no_op[j] = True
            while self.nb_machine_legal == 0:                                 # :429-430
                reward -= self.increase_time_step()
        else:                                                                 # :441 allocate job `action`
            k = self.todo_time_step_job[action]
            m = self.needed_machine_jobs[action]
            d = self.instance_matrix[action][k][1]
            reward += d                                                       # :445
            self.time_until_available_machine[m] = d
            self.time_until_finish_current_op_jobs[action] = d
            self.state[action][1] = d / self.max_time_op                      # :448
            finish = self.current_time_step + d
            if finish not in self.next_time_step:                             # :449-453
                at = bisect.bisect_left(self.next_time_step, finish)
                self.next_time_step.insert(at, finish)
                self.next_jobs.insert(at, action)
            self.solution[action][k] = self.current_time_step                 # :454
            for j in range(J):                                                # :455-461
                if self.needed_machine_jobs[j] == m and self.legal_actions[j]:
                    self.legal_actions[j] = False
                    self.nb_legal_actions -= 1
            self.nb_machine_legal -= 1                                        # :462
            self.machine_legal[m] = False
            for j in range(J):                                                # :464-467
                if self.illegal_actions[m][j]:
                    self.action_illegal_no_op[j] = False
                    self.illegal_actions[m][j] = False
            while self.nb_machine_legal == 0 and len(self.next_time_step) > 0:   # :469-470
                reward -= self.increase_time_step()
        self._prioritization_non_final()                                      # :432 / :471
        self._check_no_op()                                                   # :433 / :472
        scaled = reward / self.max_time_op                                    # :483-493
        obs = self._observe()
        return obs, scaled, self._is_done(), False, {}

    def _is_done(self):                                                       # :639-653
        if self.nb_legal_actions == 0:
            self.last_time_step = self.current_time_step
            self.last_solution = self.solution
            return True
        return False

    # -- time advance -------------------------------------------------------------------------------------
    def increase_time_step(self):                                             # :495-637
        J, M = self.jobs, self.machines
        hole = 0
        event = self.next_time_step.pop(0)                                    # :517 (IndexError on an empty queue, as there)
        self.next_jobs.pop(0)
        gap = event - self.current_time_step                                  # :521
        self.current_time_step = event
        for j in range(J):                                                    # :525-601
            left = self.time_until_finish_current_op_jobs[j]
            if left > 0:
                done_now = min(gap, left)
                self.time_until_finish_current_op_jobs[j] = max(0, left - gap)
                self.state[j][1] = self.time_until_finish_current_op_jobs[j] / self.max_time_op     # :539
                self.total_perform_op_time_jobs[j] += done_now
                self.state[j][3] = self.total_perform_op_time_jobs[j] / self.max_time_jobs          # :545
                if self.time_until_finish_current_op_jobs[j] == 0:            # :550 the op is finished
                    self.total_idle_time_jobs[j] += gap - left
                    self.state[j][6] = self.total_idle_time_jobs[j] / self.sum_op
                    self.idle_time_jobs_last_op[j] = gap - left
                    self.state[j][5] = self.idle_time_jobs_last_op[j] / self.sum_op
                    self.todo_time_step_job[j] += 1
                    self.state[j][2] = self.todo_time_step_job[j] / M
                    if self.todo_time_step_job[j] < M:                        # :562
                        nm = self.instance_matrix[j][self.todo_time_step_job[j]][0]
                        self.needed_machine_jobs[j] = nm
                        self.state[j][4] = max(0, self.time_until_available_machine[nm] - gap) / self.max_time_op   # :569-578
                    else:
                        self.needed_machine_jobs[j] = -1                      # :581
                        self.state[j][4] = 1.0
                        if self.legal_actions[j]:                             # :589-591
                            self.legal_actions[j] = False
                            self.nb_legal_actions -= 1
            elif self.todo_time_step_job[j] < M:                              # :594 waiting
                self.total_idle_time_jobs[j] += gap
                self.idle_time_jobs_last_op[j] += gap
                self.state[j][5] = self.idle_time_jobs_last_op[j] / self.sum_op
                self.state[j][6] = self.total_idle_time_jobs[j] / self.sum_op
        for m in range(M):                                                    # :604-634
            if self.time_until_available_machine[m] < gap:
                hole += gap - self.time_until_available_machine[m]            # :606-608
            self.time_until_available_machine[m] = max(0, self.time_until_available_machine[m] - gap)
            if self.time_until_available_machine[m] == 0:                     # :616
                for j in range(J):
                    if self.needed_machine_jobs[j] == m and not self.legal_actions[j] and not self.illegal_actions[m][j]:
                        self.legal_actions[j] = True
                        self.nb_legal_actions += 1
                        if not self.machine_legal[m]:
                            self.machine_legal[m] = True
                            self.nb_machine_legal += 1
        return hole


def random_masked_episode(env, rng):
    """The README's loop (README.md:53-64): uniform over the set bits of the mask until done.  Returns env steps."""
    obs = env.reset()
    done, n = False, 0
    while not done:
        mask = obs["action_mask"]
        a = int(rng.choice(len(mask), p=mask / mask.sum()))
        obs, _, done, _, _ = env.step(a)
        n += 1
    return n
