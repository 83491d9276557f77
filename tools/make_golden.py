"""Tooling (build container only): capture golden vectors from the LIVE reference.

Writes small ``.npz`` fixtures under tests/golden/ (data only: inputs and the
reference's outputs).  The reference is imported from /root/reference behind
the gymnasium stand-in of tools/refload.py; nothing of it is copied.

  G1  published_<inst>.npz   the 12 published best schedules of the reference's
                             tests/test_solutions.py replayed with that file's
                             own driver loop (tests/test_solutions.py:36-76):
                             flat action trace (job / J = NOPE / -1 = direct
                             increase_time_step()), full integer state after
                             every call, reward, done, final solution, makespan.
  G2  random_<inst>.npz      seeded random masked traces with ~8 % NOPEs forced
                             against the mask; full integer state per step and
                             the float64 observation (every step for small
                             instances, every 8th for the big ones).
  G3  rules.npz              the 7 dispatching rules with exploration disabled
                             (np.random.random -> 1.0): makespan, total reward,
                             action trace per (rule, instance).
  G4  rules_seeded.npz       FIFO/SPT with the reference's 10 % NOPE exploration
                             under np.random.seed(s): action trace + makespan.

Run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import lockstep as L  # noqa: E402
import refload  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

JOB_FIELDS = ("todo_time_step_job", "needed_machine_jobs", "time_until_finish_current_op_jobs",
              "total_perform_op_time_jobs", "total_idle_time_jobs", "idle_time_jobs_last_op")


class Recorder:
    """Collects the reference's state after every call."""

    def __init__(self, env, obs_every=1):
        self.env, self.obs_every = env, obs_every
        self.rows = {k: [] for k in ("action", "clock", "reward", "reward_num", "done", "nb_legal", "nb_machine_legal",
                                     "queue_len", "job_state", "tm", "legal", "blocked", "machine_legal")}
        self.obs, self.obs_step = [], []
        self.n = 0

    def snap(self, action, reward, done):
        e, r = self.env, self.rows
        r["action"].append(action)
        r["clock"].append(e.current_time_step)
        r["reward"].append(reward)
        r["reward_num"].append(int(round(reward * e.max_time_op)))
        r["done"].append(bool(done))
        r["nb_legal"].append(e.nb_legal_actions)
        r["nb_machine_legal"].append(e.nb_machine_legal)
        r["queue_len"].append(len(e.next_time_step))
        r["job_state"].append(np.stack([np.asarray(getattr(e, f)) for f in JOB_FIELDS]))
        r["tm"].append(np.asarray(e.time_until_available_machine).copy())
        r["legal"].append(np.asarray(e.legal_actions).copy())
        r["blocked"].append(np.asarray(e.action_illegal_no_op).copy())
        r["machine_legal"].append(np.asarray(e.machine_legal).copy())
        if self.n % self.obs_every == 0:
            self.obs.append(np.asarray(e.state, dtype=np.float64).copy())
            self.obs_step.append(self.n)
        self.n += 1

    def arrays(self):
        r = self.rows
        small = lambda x: np.asarray(x).astype(np.int16 if np.abs(np.asarray(x)).max(initial=0) < 32000 else np.int32)  # noqa: E731
        return dict(
            action=np.asarray(r["action"], dtype=np.int16), clock=np.asarray(r["clock"], dtype=np.int32),
            reward=np.asarray(r["reward"], dtype=np.float64), reward_num=np.asarray(r["reward_num"], dtype=np.int32),
            done=np.asarray(r["done"], dtype=bool), nb_legal=small(r["nb_legal"]),
            nb_machine_legal=small(r["nb_machine_legal"]), queue_len=small(r["queue_len"]),
            job_state=small(r["job_state"]), tm=small(r["tm"]), legal=np.asarray(r["legal"], dtype=bool),
            blocked=np.asarray(r["blocked"], dtype=bool), machine_legal=np.asarray(r["machine_legal"], dtype=bool),
            obs=np.asarray(self.obs, dtype=np.float64), obs_step=np.asarray(self.obs_step, dtype=np.int32),
            job_fields=np.asarray(JOB_FIELDS),
        )


def new_ref(name):
    JssEnv, _ = refload.load_reference()
    env = JssEnv({"instance_path": refload.reference_instance_path(name)})
    env.reset()
    return env


def g1_published():
    for inst, (seq, mode, ub) in sorted(L.published_sequences().items()):
        env = new_ref(inst)
        rec = Recorder(env, obs_every=1 if inst == "ta01" else 16)

        def on_action(a):
            if a == -1:
                env.increase_time_step()
                rec.snap(-1, 0.0, False)
                return False
            _, rew, done, _, _ = env.step(a)
            rec.snap(a, rew, done)
            return done

        L.replay_published(env, seq, mode, on_action)
        assert env.current_time_step == ub, (inst, env.current_time_step, ub)
        arrs = rec.arrays()
        arrs.update(sequence=np.asarray(seq, dtype=np.int16), mode=np.asarray(mode), makespan=np.int32(ub),
                    solution=np.asarray(env.solution, dtype=np.int32))
        np.savez_compressed(os.path.join(OUT, f"published_{inst}.npz"), **arrs)
        forced = int(sum(1 for a, l in zip(arrs["action"][1:], arrs["legal"][:-1]) if a == env.jobs and not l[-1]))
        print(f"G1 {inst}: {rec.n} calls, makespan {ub}, NOPEs forced against the mask: {forced}")


def g2_random():
    plan = [("ta01", 3, 1), ("ta41", 2, 1), ("dmu16", 1, 8), ("ta51", 1, 8), ("ta80", 1, 8), ("ta25", 1, 8)]
    for inst, episodes, obs_every in plan:
        rng = np.random.default_rng(sum(map(ord, inst)))
        env = new_ref(inst)
        rec = Recorder(env, obs_every=obs_every)
        ep_start = []
        for ep in range(episodes):
            env.reset()
            ep_start.append(rec.n)
            rec.snap(-2, 0.0, False)  # -2 = reset()
            done = False
            while not done:
                m = np.asarray(env.legal_actions)
                if rng.random() < 0.08 and len(env.next_time_step) > 0 and m[:-1].any():
                    a = env.jobs  # forced against the mask unless m[-1]
                else:
                    a = int(rng.choice(np.flatnonzero(m)))
                # a forced NOPE can dead-lock the reference later (IndexError at jss_env.py:517);
                # such a trace ends the episode here, which is still a valid prefix
                try:
                    _, rew, done, _, _ = env.step(a)
                except IndexError:
                    break
                rec.snap(a, rew, done)
        arrs = rec.arrays()
        arrs.update(episode_start=np.asarray(ep_start, dtype=np.int32), solution=np.asarray(env.solution, dtype=np.int32),
                    makespan=np.int32(env.current_time_step))
        np.savez_compressed(os.path.join(OUT, f"random_{inst}.npz"), **arrs)
        print(f"G2 {inst}: {rec.n} calls, last makespan {env.current_time_step}")


def g3_rules():
    _, disp = refload.load_reference()
    insts = ["ta01", "ta11", "ta21", "ta31", "ta41", "dmu16", "ta51", "ta61", "ta71", "ta80"]
    rules = list(disp.DISPATCHING_RULES.keys())
    real_random = np.random.random
    out = {"rules": np.asarray(rules), "instances": np.asarray(insts)}
    mk = np.zeros((len(rules), len(insts)), dtype=np.int32)
    rw = np.zeros((len(rules), len(insts)), dtype=np.float64)
    try:
        np.random.random = lambda *a, **k: 1.0  # disables the 10 % NOPE exploration (dispatching.py:113 etc.)
        for ri, rule in enumerate(rules):
            for ii, inst in enumerate(insts):
                env = new_ref(inst)
                policy = disp.get_rule(rule)
                env.reset()
                done, total, trace = False, 0.0, []
                while not done:
                    a = policy(env)
                    trace.append(a)
                    _, r, done, _, _ = env.step(a)
                    total += r
                mk[ri, ii], rw[ri, ii] = env.current_time_step, total
                out[f"trace_{rule}_{inst}"] = np.asarray(trace, dtype=np.int16)
            print("G3", rule, dict(zip(insts, mk[ri].tolist())))
    finally:
        np.random.random = real_random
    out["makespan"], out["total_reward"] = mk, rw
    np.savez_compressed(os.path.join(OUT, "rules.npz"), **out)


def g4_rules_seeded():
    _, disp = refload.load_reference()
    out = {}
    for rule in ("FIFO", "SPT"):
        for inst in ("ta01", "ta41"):
            for seed in (0, 1):
                np.random.seed(seed)
                env = new_ref(inst)
                policy = disp.get_rule(rule)
                env.reset()
                done, trace = False, []
                while not done:
                    a = policy(env)
                    trace.append(a)
                    _, _, done, _, _ = env.step(a)
                out[f"trace_{rule}_{inst}_{seed}"] = np.asarray(trace, dtype=np.int16)
                out[f"makespan_{rule}_{inst}_{seed}"] = np.int32(env.current_time_step)
                print("G4", rule, inst, seed, len(trace), env.current_time_step)
    np.savez_compressed(os.path.join(OUT, "rules_seeded.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    g1_published()
    g2_random()
    g3_rules()
    g4_rules_seeded()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"golden fixtures: {total / 1024:.0f} KiB in {OUT}")
