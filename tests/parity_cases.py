"""Parity cases shared by tests/test_kernel_emu.py (kernel source under the SIMT
emulator, CPU) and tests/test_hip_parity.py (the real libjss_hip.so on an MI355X).

Every case drives jssenv_amd through its public host API / the C ABI and compares with
the CPU oracle (oracle/, pinned to the live reference by tests/test_oracle_golden.py) and
with the golden vectors directly.  Bars: integers bit-exact; float32 observation and
reward within 1e-6 of the reference's float64 (north_star)."""
import numpy as np

import golden_util as G
from jssenv_amd import _abi
from jssenv_amd import instances as I
from jssenv_amd.env import BatchedJssEnv, JssEnv
from oracle import OracleEnv, rollout_batch

OBS_TOL = 1e-6


def reward_close(got, want):
    """reward = integer numerator / max_time_op as float32.  It is not confined to [0, 1] (|r| reaches
    tens on wide instances), so float32 rounding alone exceeds 1e-6 absolute there: the bar is 1e-6
    relative (1e-6 absolute below 1).  The exact integer numerator is checked through `counters`."""
    return abs(got - want) <= OBS_TOL * max(1.0, abs(want))


def assert_matches_oracle(h, orc, where, fresh_col0=True, check_outputs=True):
    """h = BatchedJssEnv.host_state(i); orc = OracleEnv in the same state."""
    J = orc.jobs
    assert h["clock"] == orc.current_time_step, f"{where}: clock {h['clock']} != {orc.current_time_step}"
    js = h["job_state"]
    pairs = [(_abi.F_TODO, "todo_time_step_job"), (_abi.F_LEFT, "time_until_finish_current_op_jobs"),
             (_abi.F_PERF, "total_perform_op_time_jobs"), (_abi.F_IDLE, "total_idle_time_jobs"),
             (_abi.F_IDLE_LAST, "idle_time_jobs_last_op")]
    for f, name in pairs:
        want = getattr(orc, name)
        assert (js[f] == want).all(), f"{where}: {name}\n got={js[f]}\nwant={want}"
    need = js[_abi.F_CUR] >> 16
    assert (need == orc.needed_machine_jobs).all(), f"{where}: needed_machine_jobs\n got={need}\nwant={orc.needed_machine_jobs}"
    todo = orc.todo_time_step_job
    for j in range(J):  # the packed current op carries the duration of op todo[j]
        if todo[j] < orc.machines:
            assert (js[_abi.F_CUR][j] & 0xFFFF) == orc.instance_matrix[j, todo[j], 1], f"{where}: cur dur job {j}"
    assert (h["tm"] == orc.time_until_available_machine).all(), f"{where}: time_until_available_machine"
    assert (h["mask"] == orc.legal_actions).all(), \
        f"{where}: legal_actions\n got={h['mask'].astype(int)}\nwant={orc.legal_actions.astype(int)}"
    assert (h["blocked"] == orc.action_illegal_no_op).all(), f"{where}: action_illegal_no_op"
    assert (h["solution"] == orc.solution).all(), f"{where}: solution"
    want_obs = orc.state
    got_obs = h["obs"].astype(np.float64)
    if not fresh_col0:
        want_obs, got_obs = want_obs[:, 1:], got_obs[:, 1:]
    err = np.abs(got_obs - want_obs).max()
    assert err <= OBS_TOL, f"{where}: real_obs max |diff| {err}"
    assert not h["obs_padding"].any(), f"{where}: padding rows of real_obs must be zero"
    assert not h["mask_padding"].any(), f"{where}: action_mask bytes behind the NOPE flag must be zero"
    # the record's cached next op is op table entry [j][todo + 1] (-1 when there is none)
    for j in range(J):
        want_next = -1
        if todo[j] + 1 < orc.machines:
            want_next = (int(orc.instance_matrix[j, todo[j] + 1, 0]) << 16) | int(orc.instance_matrix[j, todo[j] + 1, 1])
        assert h["next_op"][j] == want_next, f"{where}: next op of job {j}"
        want2 = -1
        if todo[j] + 2 < orc.machines:
            want2 = (int(orc.instance_matrix[j, todo[j] + 2, 0]) << 16) | int(orc.instance_matrix[j, todo[j] + 2, 1])
        assert h["next2_op"][j] == want2, f"{where}: next-but-one op of job {j}"
    assert h["noop_flag"] == bool(orc.legal_actions[-1]), f"{where}: NOPE flag in the header"
    if check_outputs:
        assert h["done"] == (orc.nb_legal_actions == 0), f"{where}: done"


# -----------------------------------------------------------------------------------------
# single-env facade (the reference's own API) against golden traces
# -----------------------------------------------------------------------------------------
def replay_golden_through_facade(backend, golden_name, inst, max_rows=None, check_every=1):
    g = G.load(golden_name)
    env = JssEnv({"instance_path": inst}, _backend=backend)
    orc = OracleEnv(I.builtin_instance(inst))
    env.reset()
    orc.reset()
    n = len(g["action"]) if max_rows is None else min(max_rows, len(g["action"]))
    obs_rows = {int(s): k for k, s in enumerate(g["obs_step"])}
    num_before = 0
    for i in range(n):
        a = int(g["action"][i])
        where = f"{golden_name} row {i} action {a}"
        if a == -2:
            env.reset()
            orc.reset()
        elif a == -1:
            h1, h2 = env.increase_time_step(), orc.increase_time_step()
            assert h1 == h2, f"{where}: hole {h1} != {h2}"
        else:
            _, r1, d1, t1, info = env.step(a)
            _, r2, d2, _, _ = orc.step(a)
            assert reward_close(r1, g["reward"][i]) and reward_close(r1, r2), f"{where}: reward {r1} vs {g['reward'][i]}"
            if "reward_num" in g:   # the exact integer numerator: accumulated on the device next to the float reward
                num = int(env._b.backend.numpy(env._b.counters)[0, 3])
                assert num - num_before == int(g["reward_num"][i]), f"{where}: reward numerator {num - num_before} vs {g['reward_num'][i]}"
                num_before = num
            assert d1 == bool(g["done"][i]) == d2, f"{where}: done"
            assert t1 is False and info == {}
        # golden rows (the reference's own values) ...
        G.check_env_against_row(env, g, i, where)
        if i in obs_rows:
            want = g["obs"][obs_rows[i]]
            got = np.asarray(env.state, dtype=np.float64)
            if a == -1:
                want, got = want[:, 1:], got[:, 1:]
            assert np.abs(got - want).max() <= OBS_TOL, f"{where}: obs vs golden {np.abs(got - want).max()}"
        # ... and the full oracle state
        if i % check_every == 0 or i == n - 1:
            assert_matches_oracle(env._b.host_state(0), orc, where, fresh_col0=(a != -1), check_outputs=(a != -1))
    return env, g, n == len(g["action"])


def case_published(backend, inst, max_rows=None):
    env, g, complete = replay_golden_through_facade(backend, f"published_{inst}", inst, max_rows)
    if complete:
        assert env.current_time_step == G.PUBLISHED_MAKESPAN[inst]
        assert (env.solution == g["solution"]).all()
        assert env.last_time_step == G.PUBLISHED_MAKESPAN[inst]
        assert (env.todo_time_step_job == env.machines).all()
        env.reset()
        assert env.current_time_step == 0


def case_random_golden(backend, inst, max_rows=None):
    env, g, complete = replay_golden_through_facade(backend, f"random_{inst}", inst, max_rows)
    if complete:
        assert (env.solution == g["solution"]).all()


# -----------------------------------------------------------------------------------------
# batched API: ragged batch, on-device policies, every env compared with its own oracle
# -----------------------------------------------------------------------------------------
def case_batch_lockstep(backend, inst_names, batch, n_steps, kind="random", seed=11, nope_every=0, check_every=1,
                        explore=0.0, records=None, order="interleaved"):
    # (order: these generic cases hold the kernel of the PADDED extents to the oracle -- env i <- instance i % n; the by-class
    #  deal a ragged list gets by default has its own cases: case_by_shape_padded, case_config5_mixed, FULL_SIZE_CONFIGS[7])
    insts = [I.builtin_instance(n) if isinstance(n, str) else n for n in inst_names]
    env = BatchedJssEnv(insts, batch=batch, seed=seed, env_id_base=1000, records=records, order=order, _backend=backend)
    orcs = [OracleEnv(insts[t], strict=True) for t in env.table_of_env_host]
    env.reset()
    for o in orcs:
        o.reset()
    be = env.backend
    done_seen = np.zeros(batch, dtype=bool)
    for it in range(n_steps):
        acts = be.numpy(env.policy(kind, explore=explore)).astype(np.int64)
        for i, o in enumerate(orcs):  # the device selector must agree with the oracle's (same counter RNG)
            want = o.policy(kind, seed=seed, env_id=1000 + i, episode=o.episode, step=o.step_in_episode, explore=explore)
            if want is not None:
                assert acts[i] == want, f"iter {it} env {i}: policy {kind} chose {acts[i]}, oracle {want}"
        if nope_every and it % nope_every == nope_every - 1:
            # force NOPE against the mask on some envs that have a busy machine (the fixtures do this)
            for i, o in enumerate(orcs):
                if i % 2 == 0 and len(o.next_time_step) > 0 and o.nb_legal_actions > 0:
                    acts[i] = o.jobs
        _, reward, done, trunc, info = env.step(acts)
        reward, done = be.numpy(reward), be.numpy(done)
        for i, o in enumerate(orcs):
            if acts[i] == -1:
                continue
            _, r, d, _, _ = o.step(int(acts[i]))
            assert reward_close(float(reward[i]), r), f"iter {it} env {i}: reward {reward[i]} vs {r}"
            assert bool(done[i]) == d, f"iter {it} env {i}: done"
            done_seen[i] |= d
        if it % check_every == 0 or it == n_steps - 1:
            for i, o in enumerate(orcs):
                h = env.host_state(i)
                assert_matches_oracle(h, o, f"iter {it} env {i} ({o.instance.name}) action {acts[i]}")
                assert h["err"] == o.err, f"iter {it} env {i}: err {h['err']} vs {o.err}"
                assert h["step_in_episode"] == o.step_in_episode
        if done_seen.all():
            break
    return env, orcs


def case_rollout(backend, inst_names, batch, n_iter, kind="random", seed=5, chunks=(1,), autoreset=True, order="interleaved"):
    """jss_rollout (fused policy+step, auto-restart) against orc_rollout: same counter RNG, so
    final state, counters and makespans must agree exactly."""
    insts = [I.builtin_instance(n) for n in inst_names]
    env = BatchedJssEnv(insts, batch=batch, seed=seed, env_id_base=77, order=order, _backend=backend)
    orcs = [OracleEnv(insts[t], strict=True) for t in env.table_of_env_host]
    env.reset()
    tot = [dict(steps=0, episodes=0, makespan_sum=0, reward_sum=0.0) for _ in orcs]
    ep = [1] * batch
    st = [0] * batch
    for o in orcs:
        o.reset()
    for chunk in chunks:
        env.rollout(kind, n_iter=chunk, seed=seed, autoreset=autoreset)
        for i, o in enumerate(orcs):
            if autoreset:
                r = o.rollout(kind, seed, 77 + i, chunk, episode=ep[i], step_in_episode=st[i])
                ep[i], st[i] = r["episode"], r["step_in_episode"]
                for k in ("steps", "episodes", "makespan_sum", "reward_sum"):
                    tot[i][k] += r[k]
            else:
                for _ in range(chunk):
                    if o.nb_legal_actions == 0:
                        break
                    a = o.policy(kind, seed=seed, env_id=77 + i, episode=ep[i], step=st[i])
                    _, rew, d, _, _ = o.step(a)
                    st[i] += 1
                    tot[i]["steps"] += 1
                    tot[i]["reward_sum"] += rew
                    if d:
                        tot[i]["episodes"] += 1
                        tot[i]["makespan_sum"] += o.current_time_step
    cnt = env.backend.numpy(env.counters)
    for i, o in enumerate(orcs):
        h = env.host_state(i)
        assert_matches_oracle(h, o, f"rollout env {i} ({o.instance.name})")
        assert h["episode"] == ep[i] and h["step_in_episode"] == st[i], f"env {i}: rng position"
        assert cnt[i, 0] == tot[i]["steps"], f"env {i}: steps {cnt[i, 0]} vs {tot[i]['steps']}"
        assert cnt[i, 1] == tot[i]["episodes"], f"env {i}: episodes"
        assert cnt[i, 2] == tot[i]["makespan_sum"], f"env {i}: makespan sum"
        assert abs(cnt[i, 3] / o.max_time_op - tot[i]["reward_sum"]) < 1e-6, f"env {i}: reward sum"
        assert h["err"] == 0
    return env, orcs


def case_rule_makespans(backend, rules=("FIFO", "SPT", "MWR", "LWR", "MOR", "LOR", "CR"), insts=("ta01", "ta41")):
    """G3: deterministic rule makespans of the live reference, reproduced by device rollouts."""
    g = G.load("rules")
    rnames, inames = [str(r) for r in g["rules"]], [str(i) for i in g["instances"]]
    for inst in insts:
        inst_obj = I.builtin_instance(inst)
        for rule in rules:
            one = BatchedJssEnv(inst_obj, batch=2, _backend=backend)
            one.reset()
            one.rollout(rule, n_iter=2 * inst_obj.jobs * inst_obj.machines, autoreset=False)
            h = one.host_state(1)
            want = int(g["makespan"][rnames.index(rule), inames.index(inst)])
            assert h["done"] and h["makespan"] == want == h["clock"], f"{rule} {inst}: {h['makespan']} vs {want}"
            total = one.backend.numpy(one.counters)[1, 3] / inst_obj.max_time_op
            assert abs(total - g["total_reward"][rnames.index(rule), inames.index(inst)]) < 1e-6
            assert (h["solution"] >= 0).all()


def case_error_semantics(backend):
    """Behaviour outside the reference's contract is total and flagged (include/jss_hip.h JSS_ERR_*)."""
    inst = I.builtin_instance("ta01")
    env = BatchedJssEnv(inst, batch=5, _backend=backend)
    orcs = [OracleEnv(inst, strict=True) for _ in range(5)]
    env.reset()
    for o in orcs:
        o.reset()
    J = inst.jobs
    # env0: NOPE at t=0 with nothing busy; env1: legal job; env2: skip; env3: out of range; env4: legal job
    acts = np.array([J, 3, -1, J + 5, 7], dtype=np.int32)
    env.step(acts)
    for i in (0, 1, 3, 4):
        orcs[i].step(int(acts[i]))
    errs = [env.host_state(i)["err"] for i in range(5)]
    assert errs == [_abi.ERR_NOPE_IDLE, 0, 0, _abi.ERR_BAD_ACTION, 0], errs
    for i in range(5):
        assert_matches_oracle(env.host_state(i), orcs[i], f"err case env {i}")
    assert env.host_state(0)["done"] is True and env.host_state(2)["step_in_episode"] == 0
    # JSS_ACTION_SKIP leaves reward / done / makespan of that env as they were
    r4 = env.host_state(4)["reward"]
    assert r4 > 0
    env.step(np.array([-1, -1, -1, -1, -1], dtype=np.int32))
    assert env.host_state(4)["reward"] == r4 and env.host_state(0)["done"] is True and env.host_state(4)["step_in_episode"] == 1
    for i in range(5):
        assert_matches_oracle(env.host_state(i), orcs[i], f"after an all-skip step, env {i}")
    # repeat a running job: outside the mask -> ignored + flagged, state untouched
    before = env.host_state(1)
    env.step(np.array([-1, 3, -1, -1, -1], dtype=np.int32))
    orcs[1].step(3)
    after = env.host_state(1)
    assert after["err"] == _abi.ERR_ILLEGAL_ACTION == orcs[1].err
    assert (after["job_state"] == before["job_state"]).all() and after["reward"] == 0.0
    assert_matches_oracle(after, orcs[1], "illegal action")
    # partial reset clears the flags of the chosen envs only
    env.reset(which=np.array([1, 0, 0, 0, 0], dtype=np.uint8))
    orcs[0].reset()
    assert env.host_state(0)["err"] == 0 and env.host_state(3)["err"] == _abi.ERR_BAD_ACTION
    assert_matches_oracle(env.host_state(0), orcs[0], "after partial reset")
    assert env.host_state(0)["episode"] == 2 and env.host_state(1)["episode"] == 1


def case_facade_errors(backend):
    env = JssEnv({"instance_path": "ta01"}, _backend=backend)
    env.reset()
    try:
        env.step(env.jobs)
        raise AssertionError("NOPE with nothing busy must raise like the reference")
    except IndexError:
        pass
    env.reset()
    env.step(0)
    for attempt in range(2):        # the error bits are sticky on the device; the facade must raise every time
        try:
            env.step(0)
            raise AssertionError("illegal job must raise")
        except ValueError:
            pass
    try:
        env.step(env.jobs + 3)
        raise AssertionError("out-of-range action must raise")
    except IndexError:
        pass


def case_state_invariants(backend, episodes=3, inst="ta01"):
    """The reference's tests/test_state.py:8-76 on the facade with the device random policy."""
    env = JssEnv({"instance_path": inst}, _backend=backend)
    for ep in range(episodes):
        state = env.reset()
        assert env.current_time_step == 0
        done = False
        while not done:
            legal = env.get_legal_actions()
            a = env._policy("random")
            assert legal[a]
            assert legal[:-1].sum() == env.nb_legal_actions
            state, r, done, _, _ = env.step(a)
            obs = state["real_obs"]
            assert obs.max() <= 1.0 and obs.min() >= 0.0 and np.isfinite(obs).all()
            avail = {int(env.needed_machine_jobs[j]) for j in range(env.jobs) if env.legal_actions[j]}
            assert len(avail) == env.nb_machine_legal
        assert len(env.next_time_step) == 0
        assert env.solution.min() != -1
        assert (env.todo_time_step_job == env.machines).all()
        assert env.last_time_step == env.current_time_step


# -----------------------------------------------------------------------------------------
# jssenv_amd.dispatching (the reference's rule classes) on the single-env facade
# -----------------------------------------------------------------------------------------
def case_dispatching_seeded(backend, keys=None):
    """G4: np.random.seed(s) + rule(env) loop reproduces the live reference's action trace, because the
    arg-best is deterministic and the 10 % exploration draws from NumPy's global RNG in the same order."""
    from jssenv_amd import dispatching as D
    g = G.load("rules_seeded")
    for key in (keys or [k for k in g if k.startswith("trace_")]):
        _, rule, inst, seed = key.split("_")
        np.random.seed(int(seed))
        env = JssEnv({"instance_path": inst}, _backend=backend)
        env.reset()
        policy = D.get_rule(rule)
        done, trace = False, []
        while not done:
            a = policy(env)
            trace.append(a)
            _, _, done, _, _ = env.step(a)
        assert trace == g[key].tolist(), f"{key}: trace differs"
        assert env.current_time_step == int(g[f"makespan_{rule}_{inst}_{seed}"])


def case_dispatching_deterministic(backend, rules=("SPT", "FIFO", "MWR", "LWR", "MOR", "LOR", "CR"), insts=("ta01",)):
    """G3 through the module API (exploration disabled the way the golden run did: random() -> 1.0)."""
    from jssenv_amd import dispatching as D
    g = G.load("rules")
    rnames, inames = [str(r) for r in g["rules"]], [str(i) for i in g["instances"]]
    real = np.random.random
    np.random.random = lambda *a, **k: 1.0
    try:
        for inst in insts:
            env = JssEnv({"instance_path": inst}, _backend=backend)
            for rule in rules:
                total, makespan = D.get_rule(rule).run_episode(env)
                ri, ii = rnames.index(rule), inames.index(inst)
                # the facade ranks on the host snapshot (JssEnv._rule_best); the device selector must agree with it
                env.reset()
                for _ in range(40):
                    dev = env._policy(rule)                    # job index, or J (NOPE) when no job is legal
                    assert env._rule_best(rule, env.get_legal_actions()) == (dev if dev < env.jobs else -1), rule
                    env.step(D.get_rule(rule)(env))
                assert makespan == g["makespan"][ri, ii], (rule, inst, makespan)
                assert abs(total - g["total_reward"][ri, ii]) < 1e-4, (rule, inst)
    finally:
        np.random.random = real
    with np.testing.assert_raises(ValueError):
        D.get_rule("nope")
    assert set(D.DISPATCHING_RULES) == {"SPT", "FIFO", "MWR", "LWR", "MOR", "LOR", "CR"}
    res = D.compare_rules(JssEnv({"instance_path": "ta01"}, _backend=backend), rules=["SPT"], num_episodes=1)
    assert set(res["SPT"]) == {"avg_reward", "avg_makespan"} and res["SPT"]["avg_makespan"] > 0


# -----------------------------------------------------------------------------------------
# edge shapes: the limits include/jss_hip.h promises (J <= 128, M <= 64, durations <= 65535)
# -----------------------------------------------------------------------------------------
def random_instance(rng, jobs, machines, max_dur=99, permutation=True, name=None):
    if permutation:
        machine = np.stack([rng.permutation(machines) for _ in range(jobs)]).astype(np.int32)
    else:  # machines may repeat inside a job (the reference's parser does not forbid it)
        machine = rng.integers(0, machines, size=(jobs, machines)).astype(np.int32)
    duration = rng.integers(1, max_dur + 1, size=(jobs, machines)).astype(np.int32)
    return I.Instance(name or f"rnd_{jobs}x{machines}", machine, duration)


EDGE_SHAPES = [(1, 2), (2, 2), (3, 7), (5, 16), (16, 16), (16, 5), (17, 3), (32, 32), (32, 2), (33, 4), (8, 40),
               (64, 6), (65, 3), (12, 64), (128, 2)]


def case_edge_shapes(backend, shapes=EDGE_SHAPES, steps=40, seed=2024, batch_per_shape=3):
    rng = np.random.default_rng(seed)
    for (J, M) in shapes:
        insts = [random_instance(rng, J, M, max_dur=(65535 if (J + M) % 5 == 0 else 99),
                                 permutation=((J * M) % 3 != 0)) for _ in range(batch_per_shape)]
        env, orcs = case_batch_lockstep(backend, insts, batch=batch_per_shape, n_steps=steps, kind="random",
                                        seed=seed + J, nope_every=5, check_every=4)
        # and a fused rollout on the same shapes (auto-restart crosses episode boundaries on the tiny ones)
        env2 = BatchedJssEnv(insts, seed=seed, env_id_base=77, _backend=backend)
        orcs2 = [OracleEnv(i, strict=True) for i in insts]
        env2.reset()
        env2.rollout("random", n_iter=steps, seed=seed)
        for i, o in enumerate(orcs2):
            o.reset()
            o.rollout("random", seed, 77 + i, steps, episode=1)
            assert_matches_oracle(env2.host_state(i), o, f"edge {J}x{M} rollout env {i}")


def case_ragged_j64_nope_flag(backend, steps=60, seed=3):
    """A ragged batch padded beyond 64 jobs whose envs include one with EXACTLY 64 jobs: the NOPE flag of that env's
    mask row sits at index 64 -- the first lane of the second job slot.  Every step, mask[J] must equal the header's
    NOPE bit and the oracle's legal_actions[J] (NOPEs are taken whenever legal, to visit such states often)."""
    rng = np.random.default_rng(seed)
    insts = [random_instance(rng, 64, 6, max_dur=9), random_instance(rng, 70, 6, max_dur=9),
             random_instance(rng, 63, 5, max_dur=9), random_instance(rng, 65, 4, max_dur=9)]
    env = BatchedJssEnv(insts, seed=seed, _backend=backend)
    orcs = [OracleEnv(i, strict=True) for i in insts]
    env.reset()
    for o in orcs:
        o.reset()
    n_noop = 0
    for st in range(steps):
        acts = []
        for i, o in enumerate(orcs):
            if o.nb_legal_actions == 0:
                acts.append(_abi.ACTION_SKIP)
                continue
            a = o.jobs if o.legal_actions[o.jobs] else o.policy("random", seed=seed, env_id=i, episode=1, step=st)
            n_noop += int(a == o.jobs)
            o.step(a)
            acts.append(a)
        env.step(np.asarray(acts, dtype=np.int32))
        hdr = env.backend.numpy(env.env_header)
        mask = env.backend.numpy(env.action_mask)
        for i, o in enumerate(orcs):
            flag = bool(hdr[i, _abi.H_STATUS] & _abi.STATUS_NOOP)
            assert bool(mask[i, o.jobs]) == flag == bool(o.legal_actions[o.jobs]), f"step {st} env {i} (J = {o.jobs}): NOPE flag"
            assert not mask[i, o.jobs + 1:].any(), f"step {st} env {i}: bytes behind the NOPE flag"
            assert np.array_equal(mask[i, :o.jobs] != 0, o.legal_actions[:o.jobs]), f"step {st} env {i}: mask"
    for i, o in enumerate(orcs):
        assert_matches_oracle(env.host_state(i), o, f"ragged J64 env {i}", check_outputs=False)
    assert n_noop > 0, "the case never took a NOPE"


def case_vector_env_features(backend):
    """step(autoreset=True) (gymnasium.vector next-step semantics) and state_dict round trip."""
    inst = I.builtin_instance("ta01")
    B, seed = 6, 3
    env = BatchedJssEnv(inst, batch=B, seed=seed, _backend=backend)
    orcs = [OracleEnv(inst, strict=True) for _ in range(B)]
    env.reset()
    for o in orcs:
        o.reset()
    prev_done = np.zeros(B, dtype=bool)
    resets = 0
    ckpt = None
    for it in range(330):
        acts = env.backend.numpy(env.policy("random")).astype(np.int64)
        _, reward, done, _, _ = env.step(acts, autoreset=True)
        done = env.backend.numpy(done).astype(bool)
        for i, o in enumerate(orcs):
            if prev_done[i]:
                o.reset()
                resets += 1
                assert not done[i]
            elif acts[i] >= 0:
                _, r, d, _, _ = o.step(int(acts[i]))
                assert d == done[i]
        prev_done = done
        if it % 30 == 0:
            for i, o in enumerate(orcs):
                assert_matches_oracle(env.host_state(i), o, f"autoreset iter {it} env {i}")
        if it == 100:
            ckpt = env.state_dict()
            expect = [env.host_state(i) for i in range(B)]
    assert resets >= B
    # resume the checkpoint in a fresh env object: same state, and the same future
    env2 = BatchedJssEnv(inst, batch=B, seed=seed, _backend=backend)
    env2.load_state_dict(ckpt)
    for i in range(B):
        h = env2.host_state(i)
        assert h["clock"] == expect[i]["clock"] and (h["job_state"] == expect[i]["job_state"]).all()
        assert (h["solution"] == expect[i]["solution"]).all() and h["step_in_episode"] == expect[i]["step_in_episode"]
    env.load_state_dict(ckpt)
    env.rollout("random", n_iter=50)
    env2.rollout("random", n_iter=50)
    for i in range(B):
        a, b = env.host_state(i), env2.host_state(i)
        assert a["clock"] == b["clock"] and (a["job_state"] == b["job_state"]).all() and (a["obs"] == b["obs"]).all()
    other = BatchedJssEnv(I.builtin_instance("ta02"), batch=B, _backend=backend)
    try:
        other.load_state_dict(ckpt)
        raise AssertionError("checkpoint of another instance must be rejected")
    except ValueError:
        pass


def case_bucketed_equals_padded(backend, n_envs=24, n_iter=260, seed=13):
    """BucketedJssEnv (per-shape-class tensors and kernels) is the same random process as the padded batch."""
    from jssenv_amd.bucketed import BucketedJssEnv, shape_class
    names = ["ta01", "ta11", "ta21", "ta31", "ta41", "ta51", "ta61", "ta71"]
    insts = [I.builtin_instance(n) for n in names]
    assert [shape_class(i.jobs, i.machines) for i in insts] == [0, 1, 1, 1, 1, 2, 2, 3]
    # (the bucketed wrapper deals env i onto instance i % n: the padded batch it is compared with is asked for the same deal --
    #  its default would be by shape class)
    padded = BatchedJssEnv(insts, batch=n_envs, seed=seed, env_id_base=500, order="interleaved", _backend=backend)
    assert padded.order == "interleaved" and padded._classes is None and (padded.table_of_env_host == np.arange(n_envs) % len(insts)).all()
    bucketed = BucketedJssEnv(insts, batch=n_envs, seed=seed, env_id_base=500, _backend=backend)
    padded.reset()
    bucketed.reset()
    padded.rollout("random", n_iter=n_iter)
    bucketed.rollout("random", n_iter=n_iter)
    padded.rollout_steps("random", steps=7, n_sub=2)        # the step-per-launch forms on top: sub-batches / ONE grid over the classes
    bucketed.rollout_steps("random", steps=4)               # 4 launches of jss_multi_rollout's fused grid
    bucketed.rollout_steps("random", steps=3, n_sub=3)      # and 3 steps with every class cut in three parts, a grid per part
    n_iter += 7
    # the un-fused calls, each one launch over all classes: policy -> actions per class -> step (with next-step auto-reset)
    for _ in range(5):
        padded.step(padded.policy("random"), autoreset=True)
        acts = bucketed.policy("random")
        res = bucketed.step(acts, autoreset=True)
        assert set(res) == set(acts) == {0, 1, 2, 3}
    # round 3's form (one launch per class and step, every class on its own stream) stays available and identical
    legacy = BucketedJssEnv(insts, batch=n_envs, seed=seed, env_id_base=500, launch="streams", _backend=backend)
    legacy.reset()
    legacy.rollout("random", n_iter=n_iter - 7)
    legacy.rollout_steps("random", steps=7)
    for _ in range(5):
        legacy.step(legacy.policy("random"), autoreset=True)
    for i in range(n_envs):
        a, b = legacy.host_state(i), bucketed.host_state(i)
        assert a["clock"] == b["clock"] and (a["job_state"] == b["job_state"]).all() and (a["obs"] == b["obs"]).all(), f"env {i}"
    assert legacy.stats() == bucketed.stats()
    for i in range(n_envs):
        a, b = padded.host_state(i), bucketed.host_state(i)
        assert a["clock"] == b["clock"] and (a["job_state"] == b["job_state"]).all(), f"env {i}"
        assert (a["solution"] == b["solution"]).all() and (a["mask"] == b["mask"]).all() and (a["tm"] == b["tm"]).all()
        assert np.abs(a["obs"] - b["obs"]).max() == 0 and a["episode"] == b["episode"]
    sa, sb = padded.stats(), bucketed.stats()
    assert sa == sb and sa["steps"] > 0
    # and both agree with the oracle (the five policy + step pairs with next-step auto-reset are five more rollout iterations)
    for i in (0, 5, 7, 23):
        o = OracleEnv(insts[i % len(insts)], strict=True)
        o.reset()
        o.rollout("random", seed, 500 + i, n_iter + 5, episode=1)
        assert_matches_oracle(bucketed.host_state(i), o, f"bucketed env {i}")


def case_vector_facade(backend):
    from jssenv_amd.vector import JssVectorEnv
    envs = JssVectorEnv("ta01", num_envs=5, to_numpy=True, _backend=backend)
    obs, info = envs.reset(seed=4)
    assert info == {} and obs["real_obs"].shape == (5, 15, 7) and obs["action_mask"].shape == (5, 16)
    assert obs["action_mask"][:, :15].all() and not obs["action_mask"][:, 15].any()
    finished = np.zeros(5, dtype=int)
    prev_term = np.zeros(5, dtype=bool)
    for it in range(600):
        a = envs.env.backend.numpy(envs.sample_actions("random"))
        assert all(obs["action_mask"][i, a[i]] for i in range(5) if not prev_term[i])
        obs, rew, term, trunc, info = envs.step(a)
        assert not trunc.any() and rew.shape == (5,)
        assert not (term & prev_term).any()          # a terminated env is reset by the next step
        finished += term
        prev_term = term
    assert (finished >= 1).all() and (envs.makespan > 1000).all()


def case_vector_facade_by_shape(backend, steps=120):
    """The vector facade over a ragged population ordered by shape class: the gymnasium.vector calling convention, every step
    one launch (jss_multi_step with next-step auto-reset), equal to the oracle env by env."""
    from jssenv_amd.vector import JssVectorEnv
    insts = [I.builtin_instance(n) for n in ("ta01", "ta21", "ta51", "ta71", "ta02")]
    envs = JssVectorEnv(insts, num_envs=10, to_numpy=True, order="by_shape", _backend=backend)
    toe = envs.env.table_of_env_host
    assert sorted(toe.tolist()) == sorted((np.arange(10) % 5).tolist()) and list(envs.jobs_per_env) == sorted(envs.jobs_per_env)
    obs, _ = envs.reset(seed=11)
    assert obs["real_obs"].shape == (10, 100, 7) and obs["action_mask"].shape == (10, 101)
    orcs = [OracleEnv(insts[t], strict=True) for t in toe]
    for o in orcs:
        o.reset()
    rng = np.random.default_rng(3)
    prev = np.zeros(10, dtype=bool)
    for st in range(steps):
        acts = []
        for i, o in enumerate(orcs):
            if prev[i]:
                o.reset()                                    # next-step auto-reset: the action of a terminated env is ignored
                acts.append(0)
                continue
            a = int(rng.choice(np.flatnonzero(o.legal_actions)))
            assert obs["action_mask"][i, a]
            o.step(a)
            acts.append(a)
        obs, rew, term, trunc, _ = envs.step(np.asarray(acts, dtype=np.int32))
        for i, o in enumerate(orcs):
            J = o.jobs
            assert np.array_equal(obs["action_mask"][i, :J + 1], o.legal_actions != 0), (st, i)
            assert np.abs(obs["real_obs"][i, :J].astype(np.float64) - o.state).max() <= OBS_TOL and not obs["real_obs"][i, J:].any()
            assert bool(term[i]) == (o.nb_legal_actions == 0 and not prev[i])
        prev = term.copy()


def case_instance_resampling(backend):
    """assign_instances(): envs switch instance (and shape) between episodes; untouched envs keep running."""
    insts = [I.builtin_instance(n) for n in ("ta01", "ta11", "ta02")]
    env = BatchedJssEnv(insts, batch=6, seed=8, env_id_base=40, order="interleaved", _backend=backend)     # env i <- instance i % 3
    orcs = [OracleEnv(insts[i % 3], strict=True) for i in range(6)]
    env.reset()
    for o in orcs:
        o.reset()

    def step_all(n):
        for _ in range(n):
            acts = env.backend.numpy(env.policy("random")).astype(np.int64)
            env.step(acts)
            for i, o in enumerate(orcs):
                if acts[i] >= 0:
                    o.step(int(acts[i]))

    step_all(25)
    env.assign_instances([1, 4], [2, 1])          # env 1: ta11 -> ta02 (20x15 -> 15x15), env 4: ta11 stays ta11 (fresh)
    orcs[1] = OracleEnv(insts[2], strict=True)
    orcs[4] = OracleEnv(insts[1], strict=True)
    for i in (1, 4):
        orcs[i].reset()
        orcs[i].episode = 2                       # second reset of that env slot
    step_all(25)
    for i, o in enumerate(orcs):
        h = env.host_state(i)
        assert h["jobs"] == o.jobs and h["episode"] == o.episode
        assert_matches_oracle(h, o, f"resampled env {i} ({o.instance.name})")
    # larger -> smaller instance (20 jobs -> 15): the mask bytes behind the new NOPE index must be cleared
    step_all(30)                                   # env 4 (ta11) has NOPE history / legal bits beyond index 15 by now
    env.assign_instances([4], [0])
    orcs[4] = OracleEnv(insts[0], strict=True)
    orcs[4].reset()
    orcs[4].episode = 3
    row = env.backend.numpy(env.action_mask)[4]
    assert row[:15].all() and not row[15:].any(), row
    # ... and so must the rows behind the new J of every state tensor: nothing of the larger instance survives a reset
    js = env.backend.numpy(env.job_state)[4]
    if env.medium:            # 24-byte records: "no job" is an all-zero record (op 0 = none)
        assert not js[15:].any(), js[15:]
    else:
        assert (js[15:, _abi.F_CUR] == -1).all() and (js[15:, _abi.F_NEXT] == -1).all(), js[15:]
        assert not js[15:, [_abi.F_TODO, _abi.F_LEFT, _abi.F_PERF, _abi.F_IDLE, _abi.F_IDLE_LAST, _abi.F_F4]].any(), js[15:]
    assert (env.backend.numpy(env.solution)[4] == -1).all()
    assert not env.backend.numpy(env.machine_state)[4].any()
    cst = env.backend.numpy(env.env_const)[4]
    assert cst[_abi.C_JOBS] == 15 and cst[_abi.C_MACHINES] == 15 and cst[_abi.C_TABLE] == 0, cst
    step_all(10)
    for i, o in enumerate(orcs):
        assert_matches_oracle(env.host_state(i), o, f"after larger->smaller reassignment, env {i}")
    single = BatchedJssEnv(insts[0], batch=2, _backend=backend)
    try:
        single.assign_instances([0], [0])
        raise AssertionError("a shared-instance batch has no env -> instance map to change")
    except ValueError:
        pass
    # The DEFAULT constructor deals this ragged list out by shape class (ta01, ta02: 16-lane groups; ta11: 32-lane groups).  Its
    # one restriction -- an env keeps its class -- is not the caller's problem: a move across classes hands the batch to the
    # padded extents' kernel, results unchanged.  (An explicit order="by_shape" refuses the move.)
    env = BatchedJssEnv(insts, batch=7, seed=8, env_id_base=40, _backend=backend)
    assert env.order == "by_shape" and env.steps_by_shape_class
    toe = env.table_of_env_host.copy()
    assert sorted(np.bincount(toe, minlength=3)) == [2, 2, 3] and (np.diff(env._class_of_table[toe]) >= 0).all()
    orcs = [OracleEnv(insts[t], strict=True) for t in toe]
    env.reset()
    for o in orcs:
        o.reset()
    step_all(20)
    a, b = int(np.flatnonzero(toe == 1)[0]), int(np.flatnonzero(toe == 0)[0])     # a ta11 env and a ta01 env
    env.assign_instances([b], [2])                 # ta01 -> ta02: same class, the class bodies stay
    assert env.steps_by_shape_class
    orcs[b] = OracleEnv(insts[2], strict=True)
    orcs[b].reset()
    orcs[b].episode = 2
    step_all(12)
    env.assign_instances([a, b], [0, 1])           # 20x15 -> 15x15 and 15x15 -> 20x15: across classes
    assert not env.steps_by_shape_class and env.instance_of_env(a) == 0 and env.instance_of_env(b) == 1
    orcs[a], orcs[b] = OracleEnv(insts[0], strict=True), OracleEnv(insts[1], strict=True)
    for i, ep in ((a, 2), (b, 3)):
        orcs[i].reset()
        orcs[i].episode = ep
    step_all(25)
    env.rollout_steps("random", steps=3, n_sub=2)   # (the sub-batch form of the plain kernel on top)
    for i, o in enumerate(orcs):
        o.rollout("random", 8, 40 + i, 3, episode=o.episode, step_in_episode=o.step_in_episode)
        assert_matches_oracle(env.host_state(i), o, f"default order, after a move across classes, env {i}")
    strict = BatchedJssEnv(insts, batch=7, order="by_shape", _backend=backend)
    try:
        strict.assign_instances([int(np.flatnonzero(strict.table_of_env_host == 1)[0])], [0])
        raise AssertionError("an explicit order='by_shape' keeps an env in its class")
    except ValueError:
        pass


def case_rollout_steps(backend, batch=150, steps=6, n_sub=3, seed=31):
    """jss_rollout_steps (n_sub sub-batches on n_sub streams) is the same computation as `steps` calls of
    jss_rollout(n_iter=1): every tensor bit-identical, for the three env -> instance mappings."""
    rng = np.random.default_rng(seed)
    small = [random_instance(rng, 5, 4, max_dur=9) for _ in range(batch)]
    variants = [dict(instances="ta01"),                                         # one shared table (LDS)
                dict(instances=[I.builtin_instance(n) for n in ("ta01", "ta02", "ta03")]),   # env -> instance map
                dict(instances=small)]                                          # one table per env
    for kw in variants:
        a = BatchedJssEnv(batch=batch, seed=seed, env_id_base=9, _backend=backend, **kw)
        b = BatchedJssEnv(batch=batch, seed=seed, env_id_base=9, _backend=backend, **kw)
        a.reset()
        b.reset()
        a.rollout("random", n_iter=7)
        b.rollout("random", n_iter=7)
        a.rollout_steps("random", steps=steps, n_sub=n_sub)
        for _ in range(steps):
            b.rollout("random", n_iter=1)
        a.synchronize()
        for name in BatchedJssEnv._STATE_TENSORS:
            x, y = a.backend.numpy(getattr(a, name)), b.backend.numpy(getattr(b, name))
            assert np.array_equal(x, y), f"rollout_steps differs from rollout x {steps} in {name}"
        assert a.stats()["steps"] > 0
    try:
        a.rollout_steps("random", steps=1, n_sub=17)
        raise AssertionError("n_sub > 16 must be rejected")
    except ValueError:
        pass


def case_dispatching_on_device(backend, inst="ta01", rules=("SPT", "FIFO", "MWR", "CR"), num_episodes=6, seed=5):
    """The two fused entry points of jssenv_amd.dispatching: rule.run_episode(env, device_rng=True) and the batched
    compare_rules -- both against the oracle driven with the same counter RNG (exploration 0.1, dispatching.py:113)."""
    from jssenv_amd import dispatching as D
    instance = I.builtin_instance(inst)

    def oracle_episode_of(rule, sd, env_id, episode):
        o = OracleEnv(instance, strict=True)
        o.reset()
        st, total = 0, 0.0
        while o.nb_legal_actions:
            _, r, _, _, _ = o.step(o.policy(rule, seed=sd, env_id=env_id, episode=episode, step=st, explore=D.EXPLORATION_PROBABILITY))
            total += r
            st += 1
        return total, o.current_time_step, o

    env = JssEnv({"instance_path": inst}, _backend=backend)
    for ep, rule in enumerate(rules[:2], start=1):
        total, makespan = D.get_rule(rule).run_episode(env, device_rng=True, seed=seed + ep)
        want_total, want_makespan, o = oracle_episode_of(rule, seed + ep, 0, ep)
        assert makespan == want_makespan == env.last_time_step and abs(total - want_total) <= 1e-6 * max(1.0, abs(want_total)), (rule, makespan, want_makespan)
        assert (env.solution == o.solution).all() and (np.asarray(env.last_solution) == o.solution).all()
    res = D.compare_rules(env, list(rules), num_episodes=num_episodes, seed=seed)
    for k, rule in enumerate(rules):
        runs = [oracle_episode_of(rule, seed + 7919 * k, i, k + 1)[:2] for i in range(num_episodes)]   # the batch is reset once per rule
        assert abs(res[rule]["avg_makespan"] - sum(m for _, m in runs) / num_episodes) < 1e-9, rule
        assert abs(res[rule]["avg_reward"] - sum(r for r, _ in runs) / num_episodes) < 1e-6 * 50, rule
    np.random.seed(3)
    a = D.compare_rules(env, ["SPT"], num_episodes=3)
    np.random.seed(3)
    assert a == D.compare_rules(env, ["SPT"], num_episodes=3)           # np.random.seed still makes it reproducible
    try:      # 1.2 is no p / q with q a power of two: no exact device comparison, the host loop serves it
        D.CriticalRatio(due_date_factor=1.2).run_episode(env, device_rng=True)
        raise AssertionError("a due-date factor that is not dyadic has no device selector")
    except ValueError:
        pass


def case_compact_equals_full(backend, insts=("ta01", "ta41", "ta51"), batch=7, n_iter=260, seed=23):
    """A shared-instance batch with compact 16-byte job records (the default) and with full 32-byte records (what an
    ABI caller may still pass) are the same simulation: every other tensor bit-identical, the records equal after
    decoding, through rollout, step, advance, partial reset and the trajectory recorder."""
    for inst in insts:
        a = BatchedJssEnv(inst, batch=batch, seed=seed, env_id_base=3, _backend=backend)
        b = BatchedJssEnv(inst, batch=batch, seed=seed, env_id_base=3, compact=False, _backend=backend)
        assert a.compact and not b.compact and a.job_state.shape[-1] == _abi.NFC and b.job_state.shape[-1] == _abi.NF
        n = a.backend.numpy
        for e in (a, b):
            e.reset()
            e.rollout("random", n_iter=n_iter)
            e.rollout_steps("SPT", steps=5, n_sub=2, explore=0.1)
            acts = n(e.policy("FIFO")).astype(np.int32)
            acts[1] = e.jobs_per_env[1]                      # a NOPE forced against the mask
            e.step(acts)
            e.increase_time_step(which=np.arange(batch) % 2)
            e.reset(which=(np.arange(batch) == 2))
            e.step(n(e.policy("random")).astype(np.int32), autoreset=True)
        ta, tb = a.trajectory("MWR", steps=9), b.trajectory("MWR", steps=9)
        for k in ta:
            assert np.array_equal(n(ta[k]), n(tb[k])), f"{inst}: trajectory {k}"
        for name in BatchedJssEnv._STATE_TENSORS:
            if name != "job_state":
                assert np.array_equal(n(getattr(a, name)), n(getattr(b, name))), f"{inst}: {name}"
        for i in range(batch):
            (ja, na, n2a), (jb, nb, n2b) = a.decode_jobs(n(a.job_state)[i], i), b.decode_jobs(n(b.job_state)[i], i)
            assert np.array_equal(ja, jb) and np.array_equal(na, nb) and np.array_equal(n2a, n2b), f"{inst}: records of env {i}"
    try:
        BatchedJssEnv(["ta01", "ta02"], batch=2, compact=True, _backend=backend)
        raise AssertionError("compact records need one shared instance")
    except ValueError:
        pass


def case_compact_limits(backend, shapes=((3, 64), (4, 32), (16, 16), (33, 8)), seed=5):
    """The packed words of the compact record at the limits the library accepts: every duration 65 533-65 535 (so
    `left` and the feature-4 numerator use all 16 bits, total_perform_op_time all 22 on a 64-machine job), one shared
    instance, played to completion in lock step with the oracle; then the same episode with full records, bit-equal."""
    rng = np.random.default_rng(seed)
    for (J, M) in shapes:
        machine = np.stack([rng.permutation(M) for _ in range(J)]).astype(np.int32)
        duration = (65535 - rng.integers(0, 3, size=(J, M))).astype(np.int32)
        duration[0, :] = 65535
        inst = I.Instance(f"limit_{J}x{M}", machine, duration)
        env, orcs = case_batch_lockstep(backend, [inst], batch=3, n_steps=J * M + J, kind="random", seed=seed + J,
                                        nope_every=7, check_every=3)
        assert env.compact
        full = BatchedJssEnv(inst, batch=3, seed=seed, env_id_base=9, compact=False, _backend=backend)
        comp = BatchedJssEnv(inst, batch=3, seed=seed, env_id_base=9, _backend=backend)
        n = comp.backend.numpy
        for e in (full, comp):
            e.reset()
            e.rollout("random", n_iter=J * M - 1, autoreset=False)       # deep into the episode: the largest values
        assert int(n(comp.total_perform_op_time_jobs).max()) > (M - 2) * 65533
        for name in BatchedJssEnv._STATE_TENSORS:
            if name != "job_state":
                assert np.array_equal(n(getattr(full, name)), n(getattr(comp, name))), f"{J}x{M}: {name}"
        for i in range(3):
            (ja, na, n2a), (jb, nb, n2b) = comp.decode_jobs(n(comp.job_state)[i], i), full.decode_jobs(n(full.job_state)[i], i)
            assert np.array_equal(ja, jb) and np.array_equal(na, nb) and np.array_equal(n2a, n2b), f"{J}x{M}: records of env {i}"
        for prop in ("todo_time_step_job", "time_until_finish_current_op_jobs", "total_perform_op_time_jobs",
                     "total_idle_time_jobs", "idle_time_jobs_last_op", "action_illegal_no_op", "needed_machine_jobs"):
            assert np.array_equal(np.asarray(n(getattr(full, prop)))[:, :J], np.asarray(n(getattr(comp, prop)))[:, :J]), f"{J}x{M}: {prop}"


def case_trajectory(backend, instances="ta01", batch=9, steps=40, kind="random", seed=17, explore=0.0, autoreset=True,
                    warm=0, table_of_env=None, order=None):
    """jss_trajectory (K steps per launch, every transition recorded) against K x (jss_policy, jss_step with next-step
    auto-reset) on a twin env: slot by slot the observation and mask the policy saw, the action it took (-2 where
    the env was found done and reset), reward and done; afterwards every state tensor and counter bit-identical to
    both that env and one advanced by jss_rollout(n_iter = K), the call jss_trajectory is defined as.
    (The step-by-step path itself is held to the oracle by case_batch_lockstep.)"""
    # (a ragged list is dealt out by shape class by default: trajectory / rollout(n_iter > 1) then run one launch per class
    #  range with the kernel of the class's shape, policy / step the fused grid; order="interleaved": the padded extents' kernel)
    kw = dict(batch=batch, seed=seed, env_id_base=5, _backend=backend, table_of_env=table_of_env, order=order)
    a = BatchedJssEnv(instances, **kw)
    b = BatchedJssEnv(instances, **kw)
    a.reset()
    b.reset()
    if warm:
        a.rollout(kind, n_iter=warm, explore=explore, autoreset=autoreset)
        b.rollout(kind, n_iter=warm, explore=explore, autoreset=autoreset)
    n = a.backend.numpy
    tr = a.trajectory(kind, steps=steps, explore=explore, autoreset=autoreset)
    tr = {k: n(v) for k, v in tr.items()}
    J = a.jobs_per_env
    n_real = 0
    for k in range(steps):
        obs, mask, done_before = n(b.real_obs), n(b.action_mask), n(b.done).astype(bool)
        for i in range(batch):
            assert np.array_equal(tr["real_obs"][k, i, :J[i]], obs[i, :J[i]]), f"slot {k} env {i}: observation"
            assert not tr["real_obs"][k, i, J[i]:].any()
            assert np.array_equal(tr["action_mask"][k, i], mask[i]), f"slot {k} env {i}: mask"
        act = n(b.policy(kind, explore=explore)).astype(np.int32)
        if autoreset:
            want_a = np.where(done_before, _abi.ACTION_RESET, act)
            b.step(act, autoreset=True)
        else:
            want_a = np.where(done_before, _abi.ACTION_SKIP, act)
            b.step(want_a)
        assert np.array_equal(tr["action"][k], want_a), f"slot {k}: actions {tr['action'][k]} vs {want_a}"
        rew, done = n(b.reward), n(b.done)
        stepped = want_a >= 0
        n_real += int(stepped.sum())
        assert np.array_equal(tr["reward"][k][stepped], rew[stepped]), f"slot {k}: reward"
        assert not tr["reward"][k][~stepped].any()
        want_done = np.where(stepped, done, 0 if autoreset else 1)
        assert np.array_equal(tr["done"][k], want_done), f"slot {k}: done"
    c = BatchedJssEnv(instances, **kw)            # ... and the call it is defined as: jss_rollout(n_iter = steps)
    c.reset()
    if warm:
        c.rollout(kind, n_iter=warm, explore=explore, autoreset=autoreset)
    c.rollout(kind, n_iter=steps, explore=explore, autoreset=autoreset)
    for name in BatchedJssEnv._STATE_TENSORS:
        x, y, z = n(getattr(a, name)), n(getattr(b, name)), n(getattr(c, name))
        assert np.array_equal(x, z), f"trajectory differs from rollout(n_iter={steps}) in {name}"
        if name != "reward":                       # (a reset slot leaves the last real step's reward in `out`, jss_step writes 0)
            assert np.array_equal(x, y), f"trajectory differs from policy + step x {steps} in {name}"
    assert a.stats()["steps"] == b.stats()["steps"] and n_real > 0
    return a


# -----------------------------------------------------------------------------------------
# BASELINE configs 4 and 5: size-independent properties on every env + the oracle on a sample
# -----------------------------------------------------------------------------------------
def one_episode_properties(env, dur, mach, jobs, machines, sum_op, what):
    """Checks on EVERY env of a batch that ran exactly one episode (no auto-restart): all ops scheduled, job
    order respected, makespan = last op end = clock, reward identity sum(reward numerators) =
    2 * sum_op - M * makespan, no error flags; machine exclusivity on a sample.
    dur / mach: (B, Jmax, Mmax) arrays (padding = 0), jobs / machines / sum_op: (B,)."""
    N = env.backend.numpy
    B = env.batch
    assert N(env.done).all(), f"{what}: every episode must have ended"
    assert int(N(env.err).max()) == 0, f"{what}: error flags"
    sol = N(env.solution)
    mk = N(env.makespan).astype(np.int64)
    cnt = N(env.counters)
    jj = np.arange(sol.shape[1])[None, :, None] < jobs[:, None, None]
    mm = np.arange(sol.shape[2])[None, None, :] < machines[:, None, None]
    real = jj & mm
    assert (sol[real] >= 0).all(), f"{what}: unscheduled ops"
    end = np.where(real, sol + dur, 0)
    order_ok = (sol[:, :, 1:] >= end[:, :, :-1]) | ~real[:, :, 1:]
    assert order_ok.all(), f"{what}: ops of a job out of order"
    assert (end.max(axis=(1, 2)) == mk).all() and (mk == N(env.clock)).all(), f"{what}: makespan"
    assert (cnt[:, 1] == 1).all() and (cnt[:, 2] == mk).all(), f"{what}: episode counters"
    assert (cnt[:, 3] == 2 * sum_op.astype(np.int64) - machines.astype(np.int64) * mk).all(), f"{what}: reward identity"
    todo = N(env.todo_time_step_job)
    assert (np.where(jj[:, :, 0], todo, machines[:, None]) == machines[:, None]).all(), f"{what}: todo"
    for b in range(0, B, max(1, B // 257)):                       # machine exclusivity on ~257 envs
        J, M = int(jobs[b]), int(machines[b])
        for m in range(M):
            sel = mach[b, :J, :M] == m
            s0, e0 = sol[b, :J, :M][sel], end[b, :J, :M][sel]
            o = np.argsort(s0)
            assert (s0[o][1:] >= e0[o][:-1]).all(), f"{what}: env {b} machine {m} overlaps"
    return sol, mk, cnt


def oracle_episode(inst, seed, env_id, episode=1):
    orc = OracleEnv(inst, strict=True)
    orc.reset()
    st = 0
    while orc.nb_legal_actions:
        orc.step(orc.policy("random", seed=seed, env_id=env_id, episode=episode, step=st))
        st += 1
    return orc, st


def case_config5_mixed(backend, batch=32768, seed=21, order=None):
    """ta01-ta80 dealt onto the batch (`order`: None = the constructor's default, by shape class; "interleaved" = env i <-
    ta(1 + i % 80)), padded 100x20, ragged J/M; one random episode per env, then the benchmarked auto-restart mode against
    the oracle's rollout."""
    insts = [I.builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
    env = BatchedJssEnv(insts, batch=batch, seed=seed, order=order, _backend=backend)
    assert env.order == (order or "by_shape") and env.steps_by_shape_class == (env.order == "by_shape")
    assert sorted(np.bincount(env.table_of_env_host, minlength=80)) == sorted(np.bincount(np.arange(batch) % 80, minlength=80))
    env.reset()
    env.rollout("random", n_iter=6000, autoreset=False)
    pk, t = env.packed, env.table_of_env_host
    dur, mach = (pk.ops & 0xFFFF)[t], (pk.ops >> 16)[t]
    sol, mk, cnt = one_episode_properties(env, dur, mach, pk.jobs[t], pk.machines[t], pk.sum_op[t], "config 5")
    sample = np.unique(t[::-1], return_index=True)[1]                # one env per instance (the last one that runs it)
    for i in (batch - 1 - sample).tolist():
        orc, st = oracle_episode(insts[t[i]], seed, i)
        J, M = orc.jobs, orc.machines
        assert orc.current_time_step == mk[i] and (orc.solution == sol[i, :J, :M]).all() and st == cnt[i, 0], f"mixed env {i}"
        assert_matches_oracle(env.host_state(i), orc, f"mixed env {i}")
    env.reset()
    env.zero_counters()
    env.rollout("random", n_iter=700, autoreset=True)
    cnt = env.backend.numpy(env.counters)
    for i in sorted({x for x in list(range(0, 80, 9)) + [79, 80 + 70, batch // 2, batch - 1] if x < batch}):
        orc = OracleEnv(insts[t[i]], strict=True)
        orc.reset()
        r = orc.rollout("random", seed, i, 700, episode=2, step_in_episode=0)
        assert_matches_oracle(env.host_state(i), orc, f"mixed env {i} (auto-restart)")
        assert cnt[i, 0] == r["steps"] and cnt[i, 1] == r["episodes"] and cnt[i, 2] == r["makespan_sum"]


def case_config4_synthetic(backend, batch=8192, seed=4, sample=64):
    """Taillard-LCG 50x20 instances, one table per env (time_seed = 1 + 2i, machine_seed = 2 + 2i)."""
    pk = I.synthetic_packed(batch, 50, 20)
    env = BatchedJssEnv(pk, seed=seed, _backend=backend)
    assert env.n_tables == batch
    env.reset()
    env.rollout("random", n_iter=4000, autoreset=False)
    dur, mach = pk.ops & 0xFFFF, pk.ops >> 16
    sol, mk, cnt = one_episode_properties(env, dur, mach, pk.jobs, pk.machines, pk.sum_op, "config 4")
    for i in range(0, batch, max(1, batch // sample)):
        inst = I.Instance(f"syn{i}", mach[i], dur[i])
        assert (inst.packed() == I.synthetic_batch(1, 50, 20, first=i)[0].packed()).all()
        orc, st = oracle_episode(inst, seed, i)
        assert orc.current_time_step == mk[i] and (orc.solution == sol[i]).all() and st == cnt[i, 0], f"synthetic env {i}"
        assert_matches_oracle(env.host_state(i), orc, f"synthetic env {i}")


def case_nope_fuzz(backend, shapes=((2, 2), (3, 3), (4, 2), (5, 4)), batch=12, steps=50, seed=99):
    """Tiny instances with a NOPE forced on half of the envs every other step: drives the rare branches of the
    event jump (a suppressed job left behind by a NOPE; no job that can ever become legal again, with and without
    the reference's IndexError) far more often than the real instances do.  Error flags included."""
    rng = np.random.default_rng(seed)
    for (J, M) in shapes:
        insts = [random_instance(rng, J, M, max_dur=7, permutation=(i % 2 == 0)) for i in range(batch)]
        case_batch_lockstep(backend, insts, batch=batch, n_steps=steps, kind="random", seed=seed + J, nope_every=2, check_every=1)


def case_file_round_trips(backend, tmpdir, seed=12):
    """a packed batch written to disk and read back builds the same env; a checkpoint FILE
    resumes bit-exactly (same state now, same future)."""
    import os
    insts = [I.builtin_instance(n) for n in ("ta01", "ta21", "ta02")]
    pk = I.pack_batch(insts)
    path = os.path.join(str(tmpdir), "batch.npz")
    I.save_batch(path, pk)
    back = I.load_batch(path)
    for f in ("ops", "rem", "inst", "jobs", "machines", "max_time_op", "max_time_jobs", "sum_op"):
        assert np.array_equal(getattr(pk, f), getattr(back, f)), f
    a = BatchedJssEnv(insts, batch=9, seed=seed, env_id_base=3, _backend=backend)
    b = BatchedJssEnv(back, batch=9, seed=seed, env_id_base=3, _backend=backend)       # from the file
    a.reset()
    b.reset()
    a.rollout("random", n_iter=150)
    b.rollout("random", n_iter=150)
    for name in BatchedJssEnv._STATE_TENSORS:
        assert np.array_equal(a.backend.numpy(getattr(a, name)), b.backend.numpy(getattr(b, name))), name
    ck = os.path.join(str(tmpdir), "state.npz")
    a.save_checkpoint(ck)
    a.rollout("random", n_iter=77)                                                      # the future to reproduce
    want = {name: a.backend.numpy(getattr(a, name)) for name in BatchedJssEnv._STATE_TENSORS}
    c = BatchedJssEnv(back, batch=9, seed=seed, env_id_base=3, _backend=backend)       # a fresh object: no reset()
    c.load_checkpoint(ck)
    c.rollout("random", n_iter=77)
    for name, w in want.items():
        assert np.array_equal(c.backend.numpy(getattr(c, name)), w), f"resumed run differs in {name}"
    other = BatchedJssEnv(insts[:2], batch=9, _backend=backend)
    try:
        other.load_checkpoint(ck)
        raise AssertionError("a checkpoint of another batch must be rejected")
    except ValueError:
        pass


def case_hip_equals_twin_full_size(hip_backend, configs=None):
    """EVERY env of the BASELINE-sized batches: the HIP path against the host-core twin (libjss_cpu.so, itself held
    to the oracle by tests/test_cpu_twin.py) after the same fused rollout -- all integer tensors bit-equal, the
    float32 observation and reward bit-equal too (both sides evaluate the same fma sequence)."""
    from jssenv_amd.env import CpuBackend
    cpu = CpuBackend()
    configs = configs or [
        ("config 2: ta01 x 4096, random", dict(instances="ta01", batch=4096), "random", 300),
        ("config 3: ta41 x 16384, SPT", dict(instances="ta41", batch=16384), "SPT", 700),
        ("config 4: synthetic 50x20 x 8192, random", dict(instances=I.synthetic_packed(8192, 50, 20)), "random", 400),
        # (order="interleaved": the kernel of the padded extents, whose resets write every padding row like the twin's do -- the
        #  by-class bodies a ragged list gets by default leave the rows behind their lane group as the allocation made them, so
        #  the tensors agree on every row of a job and not bit for bit; that path is held to the oracle, FULL_SIZE_CONFIGS[7])
        ("config 5: mixed ta01-80 x 32768, random", dict(instances=[I.builtin_instance(f"ta{k:02d}") for k in range(1, 81)], batch=32768,
                                                         order="interleaved"), "random", 300),
        ("headline: ta01 x 65536, random", dict(instances="ta01", batch=65536), "random", 260),
    ]
    for what, kw, policy, n_iter in configs:
        a = BatchedJssEnv(seed=5, env_id_base=123, _backend=hip_backend, **kw)
        # (the same `kernel` on both sides: the default record layout of a batch goes by it, and the tensors are compared as stored)
        b = BatchedJssEnv(seed=5, env_id_base=123, _backend=cpu, kernel=a.kernel, **kw)
        a.reset()
        b.reset()
        a.rollout(policy, n_iter=n_iter)                 # one launch, state in registers
        a.rollout_steps(policy, steps=8, n_sub=3)        # + the benchmarked step-per-launch form
        b.rollout(policy, n_iter=n_iter + 8)
        a.synchronize()
        for name in BatchedJssEnv._STATE_TENSORS:
            x, y = a.backend.numpy(getattr(a, name)), b.backend.numpy(getattr(b, name))
            if x.dtype.kind == "f":
                same = (x.view(np.int32) == y.view(np.int32))
            else:
                same = (x == y)
            assert same.all(), f"{what}: {name} differs from the twin in {int((~same).sum())} of {same.size} elements"
        assert a.stats()["steps"] > 0 and a.stats() == b.stats(), what


FULL_SIZE_CONFIGS = [   # (label, BatchedJssEnv kwargs factory, policy, iterations, explore)
    ("config 2: ta01 x 4096, random", lambda: dict(instances="ta01", batch=4096), "random", 300, 0.0),
    ("config 3: ta41 x 16384, SPT", lambda: dict(instances="ta41", batch=16384), "SPT", 700, 0.0),
    ("config 4: synthetic 50x20 x 8192, random", lambda: dict(instances=I.synthetic_packed(8192, 50, 20)), "random", 400, 0.0),
    # (env i <- ta(1 + i % 80): every env on the padded extents' kernel, <2,5,1>; the default constructor's deal is index 7)
    ("config 5: mixed ta01-80 x 32768, random",
     lambda: dict(instances=[I.builtin_instance(f"ta{k:02d}") for k in range(1, 81)], batch=32768, order="interleaved"), "random", 300, 0.0),
    ("headline: ta01 x 65536, random", lambda: dict(instances="ta01", batch=65536), "random", 280, 0.0),
    # the other two batches bench.py times: per-env 15x15 tables at 65 536 envs (24-byte medium records, <16,5,3>) and ALL
    # of config 4 on one GPU (indices 5, 6: appended, the launch-form table of tests/test_hip_parity.py goes by index)
    ("syn15x15: synthetic 15x15 x 65536, random", lambda: dict(instances=I.synthetic_packed(65536, 15, 15)), "random", 260, 0.0),
    ("config 4 whole: synthetic 50x20 x 65536, random", lambda: dict(instances=I.synthetic_packed(65536, 50, 20)), "random", 150, 0.0),
    # config 5 as the DEFAULT constructor builds it (round 6): a ragged list is dealt out by shape class and stepped by the
    # class-specialised bodies of the fused grid on the padded rows -- what bench.py's c5_mixed_b32768_padded times
    ("config 5 default: mixed ta01-80 x 32768 by shape class, random",
     lambda: dict(instances=[I.builtin_instance(f"ta{k:02d}") for k in range(1, 81)], batch=32768), "random", 300, 0.0),
]


def drive_steps(env, kind, iters, form="fused", explore=0.0, autoreset=True, window=20, n_sub=2):
    """`iters` x (policy + step) on `env` through one of the launch forms bench.py times:
      fused       one jss_rollout(n_iter = iters) launch (state in registers; the kRollout kernels)
      per_launch  iters x jss_rollout(n_iter = 1) on the current stream (the kRollout1 kernels)
      free        windows of `window` steps through bind_rollout_steps(caller_orders_streams=True): n_sub sub-batches
                  on n_sub streams with NO fork / join events, a device-wide synchronize on both sides of every window
                  -- exactly bench.py's timed region (HIP backend; elsewhere the fork-join form)
      fork_join   the same windows with the library's fork / join events (stream-ordered on the caller's stream)
      graph       a captured hipGraph of `window` x jss_rollout(n_iter = 1), replayed iters // window times (+ eager tail)
      steps       the recorded actions of the same policy through jss_steps, `window` steps per launch
      session     the same actions through a step session (state resident on the chip), `window` steps posted per wait;
      session_lockstep: one step per post / wait (mailbox depth 1)
    """
    be = env.backend
    if form == "fused":
        env.rollout(kind, n_iter=iters, autoreset=autoreset, explore=explore)
    elif form == "per_launch":
        for _ in range(iters):
            env.rollout(kind, n_iter=1, autoreset=autoreset, explore=explore)
    elif form in ("free", "fork_join"):
        torch = getattr(be, "torch", None)
        done = 0
        while done < iters:
            n = min(window, iters - done)
            issue = env.bind_rollout_steps(kind, steps=n, n_sub=n_sub, autoreset=autoreset, explore=explore,
                                           caller_orders_streams=(form == "free"))
            if torch is not None:
                torch.cuda.synchronize()
            issue()
            if torch is not None:
                torch.cuda.synchronize()
            done += n
    elif form == "graph":
        torch = be.torch
        dev = be.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        snap = env._arena.clone(), env.solution.clone()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for _ in range(window):
                    env.rollout(kind, n_iter=1, autoreset=autoreset, explore=explore)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        # (capture does not execute; put back what a driver that runs warm-up launches inside capture would have changed)
        env._arena.copy_(snap[0])
        env.solution.copy_(snap[1])
        for _ in range(iters // window):
            graph.replay()
        for _ in range(iters % window):
            env.rollout(kind, n_iter=1, autoreset=autoreset, explore=explore)
        torch.cuda.synchronize()
        del graph
    elif form in ("steps", "session", "session_lockstep"):
        # external actions: the behaviour trajectory of the same policy, recorded from the same state and replayed (slots
        # where an episode had ended carry JSS_ACTION_RESET, so the replay IS the auto-restarting rollout)
        start, sol0 = _state_snapshot(env), be.numpy(env.solution)
        acts = env.trajectory(kind, steps=iters, record=("action",), explore=explore, autoreset=autoreset)["action"]
        env.synchronize()
        _restore(env, start, sol0)
        if form == "steps":
            done = 0
            while done < iters:
                n = min(window, iters - done)
                env.steps(acts[done:done + n])
                done += n
        else:
            depth = 1 if form == "session_lockstep" else window
            with env.session(depth=depth, timeout_ms=3000) as s:
                done = 0
                while done < iters:
                    n = min(depth, iters - done)
                    s.post(acts[done:done + n] if n > 1 else acts[done])
                    s.wait()
                    done += n
    else:
        raise KeyError(form)
    env.synchronize()


def case_every_env_vs_oracle(backend, label, kw, kind, iters, explore=0.0, seed=6, env_id_base=123, autoreset=True,
                             form="fused", n_sub=2):
    """EVERY env of a (full-size) batch against the C oracle itself -- no twin in between: after the same
    rollout from a fresh reset (through launch form `form`, see drive_steps), every integer of the state (clock, the six
    per-job arrays, machine clocks, solution, mask, blocked flags), the RNG position, the four counters and the error
    flags are bit-equal on all envs, the float32 observation within 1e-6 of the oracle's float64 on all envs."""
    env = BatchedJssEnv(seed=seed, env_id_base=env_id_base, _backend=backend, **kw)
    env.reset()
    drive_steps(env, kind, iters, form=form, explore=explore, autoreset=autoreset, n_sub=n_sub)
    assert_batch_equals_oracle(env, kind, seed, iters, f"{label} [{form}]", explore=explore, autoreset=autoreset)
    return env


def assert_batch_equals_oracle(env, kind, seed, iters, label, explore=0.0, autoreset=True):
    """Every env of `env` -- `iters` x (policy + step) after a fresh reset -- against orc_rollout_batch (see
    case_every_env_vs_oracle).  Works for a bucket of a BucketedJssEnv too: explicit global env ids key the RNG."""
    n = env.backend.numpy
    toe = None if env.n_tables == 1 or env.n_tables == env.batch and env._table_of_env is None else env.table_of_env_host
    ids = None if env._env_ids is None else n(env._env_ids)
    want = rollout_batch(env.packed, env.batch, kind, seed, iters, table_of_env=toe, env_id_base=env.env_id_base, env_ids=ids,
                         explore=explore, autoreset=autoreset)
    hdr, js = n(env.env_header), n(env.job_state)
    assert np.array_equal(hdr[:, _abi.H_CLOCK], want["clock"]), f"{label}: clock"
    assert np.array_equal(hdr[:, _abi.H_EPISODE], want["episode"]) and np.array_equal(hdr[:, _abi.H_STEP], want["step_in_episode"]), \
        f"{label}: RNG position"
    assert np.array_equal(hdr[:, _abi.H_STATUS] & 0xFF, want["err"]) and not want["err"].any(), f"{label}: error flags"
    J = env.jobs_per_env
    live = np.arange(env.jmax)[None, :] < J[:, None]                    # rows of real jobs
    need = np.asarray(n(env.needed_machine_jobs))
    got_fields = [n(env.todo_time_step_job), np.where(live, need, 0), n(env.time_until_finish_current_op_jobs),
                  n(env.total_perform_op_time_jobs), n(env.total_idle_time_jobs), n(env.idle_time_jobs_last_op)]   # either record layout
    for f, (name, got) in enumerate(zip(G.JOB_FIELDS, got_fields)):
        bad = np.flatnonzero((got != want["job_fields"][:, f]).any(axis=1))
        assert bad.size == 0, f"{label}: {name} differs on {bad.size} envs, first {bad[:5]}"
    assert np.array_equal(n(env.machine_state), want["tm"]), f"{label}: time_until_available_machine"
    assert np.array_equal(n(env.solution), want["solution"]), f"{label}: solution"
    assert np.array_equal(n(env.action_mask), want["mask"]), f"{label}: action mask"
    assert np.array_equal(n(env.action_illegal_no_op), want["blocked"]), f"{label}: action_illegal_no_op"
    assert np.array_equal(n(env.counters), want["counters"]), f"{label}: counters"
    assert np.array_equal((hdr[:, _abi.H_STATUS] & _abi.STATUS_NOOP) != 0, want["mask"][np.arange(env.batch), J] != 0), f"{label}: NOPE flag"
    err = np.abs(n(env.real_obs).astype(np.float64) - want["obs"]).max()
    assert err <= OBS_TOL, f"{label}: observation max |diff| {err}"
    assert want["counters"][:, 0].min() > 0
    return want


def case_multi_entry_points(backend, steps=40):
    """jss_multi_reset / policy / step / rollout over env sets the caller picked himself: (a) sets that all have a body in
    the fused grid, (b) a mix with a shared-instance set (LDS-staged table: no body in the grid -> one plain launch per set
    on the same stream) -- both bit-identical to the single-set calls on twin objects; argument errors."""
    import ctypes as C
    be = backend
    D, S, O = C.POINTER(_abi.JssDesc), C.POINTER(_abi.JssState), C.POINTER(_abi.JssOut)
    P = C.c_void_p
    groups = {"fused grid": [dict(instances=["ta01", "ta02", "ta03"], batch=9), dict(instances=["ta21", "ta31"], batch=5),
                             dict(instances=["ta51", "ta61"], batch=3), dict(instances=["ta71", "ta52"], batch=4)],   # (last: ragged, 50 and 100 jobs: the grid's two-jobs-per-lane body has no narrow path)
              "fallback": [dict(instances="ta01", batch=6), dict(instances=["ta11", "ta12"], batch=4), dict(instances="ta51", batch=2)]}
    for what, kws in groups.items():
        # (order="interleaved": each object is ONE set here, stepped by the kernel of its own padded extents -- the by-class deal
        #  a ragged list gets by default would make the `b` objects fused grids themselves, and leaves other bytes in padding rows)
        a = [BatchedJssEnv(seed=4, env_id_base=100 * i, order="interleaved", _backend=be, **kw) for i, kw in enumerate(kws)]
        b = [BatchedJssEnv(seed=4, env_id_base=100 * i, order="interleaved", _backend=be, **kw) for i, kw in enumerate(kws)]
        n = len(a)
        sets = ((D * n)(*[C.pointer(e._desc) for e in a]), (S * n)(*[C.pointer(e._state) for e in a]), (O * n)(*[C.pointer(e._out) for e in a]))
        streams = (P * 2)(be.stream(), be.stream())
        assert be.lib.jss_multi_reset(n, *sets, None, be.stream()) == 0
        for e in a:
            e._is_reset = True
        for e in b:
            e.reset()
        assert be.lib.jss_multi_rollout(n, *sets, _abi.POLICY["SPT"], 4, 6554, steps, _abi.ROLLOUT_AUTORESET, 1, streams) == 0
        acts = (P * n)(*[be.ptr(e._actions_out) for e in a])
        for _ in range(5):
            assert be.lib.jss_multi_policy(n, sets[0], sets[1], _abi.POLICY["random"], 4, 0, acts, be.stream()) == 0
            assert be.lib.jss_multi_step(n, sets[0], sets[1], acts, sets[2], _abi.ROLLOUT_AUTORESET, be.stream()) == 0
        # parts on streams (every set cut in two, part i of all sets on streams[i], the library forks / joins): the fused grid
        # and -- round 6 -- the one-launch-per-set fallback alike; n_steps == 0 launches nothing and touches nothing
        two = be.stream_array(2) if hasattr(be, "stream_array") else (P * 2)()
        fj = _abi.ROLLOUT_FORK_JOIN if hasattr(be, "stream_array") else 0
        before = [_state_snapshot(e) for e in a]
        assert be.lib.jss_multi_rollout(n, *sets, _abi.POLICY["random"], 4, 0, 0, _abi.ROLLOUT_AUTORESET | fj, 2, two) == 0
        for e, snap in zip(a, before):
            e.synchronize()
            now = _state_snapshot(e)
            assert all(np.array_equal(now[k], snap[k]) for k in snap), f"jss_multi_rollout(n_steps = 0) ({what}) touched the state"
        assert be.lib.jss_multi_rollout(n, *sets, _abi.POLICY["random"], 4, 0, 3, _abi.ROLLOUT_AUTORESET | fj, 2, two) == 0
        # partial reset: only the sets whose mask says so, only the envs whose byte is set
        which = [be.as_device(np.arange(e.batch) % 2, "uint8") if i != 1 else None for i, e in enumerate(a)]
        keep = list(which)                                                   # (alive until the call has read them)
        assert be.lib.jss_multi_reset(n, *sets, (P * n)(*[be.ptr(w) for w in which]), be.stream()) == 0
        for i, e in enumerate(b):
            e.rollout("SPT", n_iter=steps, seed=4, explore=6554 / 65536)
            for _ in range(5):
                e.step(e.policy("random", seed=4), autoreset=True)
            e.rollout_steps("random", steps=0, seed=4)
            e.rollout_steps("random", steps=3, n_sub=1, seed=4)
            e.reset(which=None if i == 1 else np.arange(e.batch) % 2)
        for x, y in zip(a, b):
            x.synchronize()
            y.synchronize()
            sx, sy = _state_snapshot(x), _state_snapshot(y)
            for name in sx:
                assert np.array_equal(sx[name], sy[name]), f"jss_multi_* ({what}) differs from the single-set calls in {name}"
        del keep
    # argument errors
    assert be.lib.jss_multi_reset(0, *sets, None, be.stream()) == _abi.E_SHAPE and be.lib.jss_multi_reset(17, *sets, None, be.stream()) == _abi.E_SHAPE
    assert be.lib.jss_multi_reset(n, None, sets[1], sets[2], None, be.stream()) == _abi.E_NULL
    assert be.lib.jss_multi_step(n, sets[0], sets[1], None, sets[2], 0, be.stream()) == _abi.E_NULL
    assert be.lib.jss_multi_policy(n, sets[0], sets[1], 99, 0, 0, acts, be.stream()) == _abi.E_KIND
    assert be.lib.jss_multi_rollout(n, *sets, _abi.POLICY["random"], 0, 0, -1, 0, 1, streams) == _abi.E_SHAPE
    assert be.lib.jss_multi_rollout(n, *sets, _abi.POLICY["random"], 0, 0, 1, 0, 0, streams) == _abi.E_SHAPE
    assert be.lib.jss_multi_rollout(n, *sets, _abi.POLICY["random"], 0, 0, 1, 0, 1, None) == _abi.E_NULL
    assert be.lib.jss_multi_rollout(n, *sets, _abi.POLICY_CR_F64, 0, 0, 1, 0, 1, streams) == _abi.E_KIND


def case_by_shape_padded(backend, n_envs=40, iters=60, seed=6, env_id_base=123, taillard=False, tail=4):
    """order='by_shape': a ragged population in ONE set of padded tensors, stepped by the fused grid's class-specialised bodies
    on the padded rows (JssDesc.jclass: packed 16- / 32-lane groups inside rows of jmax, one wavefront per env for the rest).
    Every env equals the oracle -- including instances that FILL their lane group (J = 16, J = 32: the NOPE flag sits on the
    group's edge) and a 64-job instance (which must go with the two-jobs-per-lane class) -- through the fused rollout in one and
    in several parts, the un-fused policy -> step pair, a partial reset, an instance change inside a class, and calls that take
    the plain padded kernel on the same tensors (multi-iteration rollout, trajectory)."""
    rng = np.random.default_rng(seed)
    if taillard:
        insts = [I.builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
    else:
        insts = [I.builtin_instance("ta01"), random_instance(rng, 16, 9, max_dur=40), random_instance(rng, 7, 16, max_dur=30),
                 I.builtin_instance("ta21"), random_instance(rng, 32, 20, max_dur=50), I.builtin_instance("ta41"),
                 I.builtin_instance("ta51"), random_instance(rng, 63, 5, max_dur=20), random_instance(rng, 64, 4, max_dur=20),
                 I.builtin_instance("ta71")]
    env = BatchedJssEnv(insts, batch=n_envs, seed=seed, env_id_base=env_id_base, order="by_shape", _backend=backend)
    cls = env._class_of_table[env.table_of_env_host]
    assert (np.diff(cls) >= 0).all() and env._classes["n"] == 4 and sorted(np.bincount(env.table_of_env_host, minlength=len(insts))) == \
        sorted(np.bincount(np.arange(n_envs) % len(insts), minlength=len(insts)))          # every instance as often as i % n would deal it
    if not taillard:
        assert cls[np.flatnonzero(env.jobs_per_env == 64)].tolist() == [3] * int((env.jobs_per_env == 64).sum())
    env.reset()
    env.rollout_steps("random", steps=iters // 2, n_sub=1)
    env.rollout_steps("random", steps=iters - iters // 2, n_sub=3)
    for _ in range(tail):
        env.step(env.policy("random"), autoreset=True)
    done = iters + tail
    assert_batch_equals_oracle(env, "random", seed, done, f"by_shape x {n_envs}")
    # the plain padded kernel on the same tensors (whatever the class bodies left in the padding rows is nobody's business)
    env.rollout("random", n_iter=7)
    env.trajectory("random", steps=5, record=("action",))
    env.rollout("random", n_iter=1)
    done += 13
    assert_batch_equals_oracle(env, "random", seed, done, f"by_shape x {n_envs}, mixed with the padded kernel")
    # partial reset through the grid: exactly the chosen envs restart
    n = env.backend.numpy
    before = n(env.env_header)[:, _abi.H_EPISODE].copy()
    which = (np.arange(n_envs) % 3 == 0).astype(np.uint8)
    env.reset(which=which)
    after = n(env.env_header)
    assert np.array_equal(after[:, _abi.H_EPISODE], before + which) and (after[which == 1][:, _abi.H_CLOCK] == 0).all()
    # an env may change instance inside its class, not across classes
    if not taillard:
        i16 = int(np.flatnonzero(env.jobs_per_env == 16)[0])
        env.assign_instances([i16], [0])                       # the 16 x 9 env becomes ta01 (class 0 both)
        assert int(env.jobs_per_env[i16]) == 15
        orc = OracleEnv(insts[0], strict=True)
        orc.reset()
        ep = int(n(env.env_header)[i16, _abi.H_EPISODE])
        env.rollout_steps("random", steps=30, n_sub=2)
        orc.rollout("random", seed, env_id_base + i16, 30, episode=ep)
        assert_matches_oracle(env.host_state(i16), orc, "instance change inside a class")
        try:
            env.assign_instances([i16], [len(insts) - 1])
            raise AssertionError("an env must keep its shape class")
        except ValueError:
            pass
    return env


def case_bucketed_every_env_vs_oracle(backend, n_envs=32768, iters=160, seed=6, env_id_base=123, launch="grid", kind="random",
                                      unfused_tail=0):
    """BASELINE config 5 without padding, at the benchmarked size: the mixed ta01-ta80 population as shape classes, stepped by
    ONE grid per step (jss_multi_rollout; launch="streams": one launch per class and step), EVERY env of every class against
    the C oracle.  unfused_tail: that many more steps through jss_multi_policy + jss_multi_step(autoreset), which -- the
    policy being the same counter RNG -- are that many more iterations of the same rollout."""
    from jssenv_amd.bucketed import BucketedJssEnv
    insts = [I.builtin_instance(f"ta{k:02d}") for k in range(1, 81)]
    env = BucketedJssEnv(insts, batch=n_envs, seed=seed, env_id_base=env_id_base, launch=launch, _backend=backend)
    assert sum(b.batch for _, b in env._each()) == n_envs and len(env._each()) == 4
    env.reset()
    env.rollout_steps(kind, steps=iters // 2)
    env.rollout_steps(kind, steps=iters - iters // 2, n_sub=2)          # the pipelined form: two parts per class, a grid per part
    for _ in range(unfused_tail):
        env.step(env.policy(kind), autoreset=True)
    env.synchronize()
    for k, b in env._each():
        assert_batch_equals_oracle(b, kind, seed, iters + unfused_tail, f"mixed ta01-80 x {n_envs}, shape class {k} ({b.batch} envs) [{launch}]")
    return env


def _state_snapshot(env):
    n = env.backend.numpy
    return {k: n(getattr(env, k)) for k in BatchedJssEnv._STATE_TENSORS if k != "machine_state" or not env.no_clocks}


def case_step_graph_replay(backend, inst="ta01", batch=4096, K=40, warm=60, seed=8):
    """The launch form bench.py's `step_only` figure (and config 2) uses: a captured hipGraph of K x jss_step with the
    actions resident in HBM, replayed.  The replay -- twice, from the same starting state -- must leave every state and
    output tensor exactly where K eager jss_step calls leave it, which in turn is where jss_trajectory (the recorder the
    actions came from) left it."""
    torch = backend.torch
    env = BatchedJssEnv(inst, batch=batch, seed=seed, _backend=backend)
    env.reset()
    env.rollout("random", n_iter=warm)
    env.zero_counters()
    snap = env._arena.clone(), env.solution.clone()

    def restore():
        env._arena.copy_(snap[0])
        env.solution.copy_(snap[1])
        torch.cuda.synchronize()

    acts = env.trajectory("random", steps=K, record=("action",))["action"]      # (K, B); -2 = auto-reset slots
    env.synchronize()
    want_traj = _state_snapshot(env)
    restore()
    for k in range(K):
        env.step(acts[k])
    env.synchronize()
    want = _state_snapshot(env)
    for name in want:
        assert np.array_equal(want[name], want_traj[name]), f"K x jss_step differs from jss_trajectory in {name}"
    assert want["counters"][:, 0].sum() > 0
    restore()
    dev = backend.device
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for k in range(K):
                env.step(acts[k])
    torch.cuda.current_stream(dev).wait_stream(side)
    for rep in range(2):
        restore()
        graph.replay()
        torch.cuda.synchronize()
        got = _state_snapshot(env)
        for name in want:
            assert np.array_equal(got[name], want[name]), f"hipGraph replay {rep} of K x jss_step differs from eager in {name}"
    del graph


def case_render_rows_from_device_solution(backend, inst="ta01", rule="SPT", steps=140):
    """N4 on the real engine: the Gantt rows render() hands to plotly, built from the env's device-resident `solution`,
    equal the rows of the oracle's schedule after the same actions (jss_env.py:655-693)."""
    from jssenv_amd.dispatching import get_rule
    from jssenv_amd.render import gantt_frame, gantt_rows
    env = JssEnv({"instance_path": inst}, _backend=backend)
    orc = OracleEnv(I.builtin_instance(inst), strict=True)
    env.reset()
    orc.reset()
    assert gantt_rows(env.solution, env.instance, 0.0) == []
    pick = get_rule(rule)
    np.random.seed(4)
    for _ in range(steps):
        a = pick(env)
        env.step(a)
        orc.step(a)
    rows = gantt_rows(env.solution, env.instance, 1000.0)
    want = gantt_rows(orc.solution, env.instance, 1000.0)
    assert rows == want and len(rows) == int((orc.solution >= 0).sum()) > 0
    frame = gantt_frame(env, size=(320, 200), prefer_plotly=False)
    assert frame.shape == (200, 320, 3) and (frame != 255).any()
    return env


def case_two_streams_two_threads(backend_factory, batch=3000, calls=12, steps=6, seed=2):
    """Two env objects, each driven from its own host thread on its own stream through the library's fork / join form
    (JSS_ROLLOUT_FORK_JOIN): the events belong to the main stream of the call, so neither thread records on the other's.
    Results equal the same calls issued one after the other on one stream."""
    import threading
    be = backend_factory()
    torch = be.torch
    envs = [BatchedJssEnv("ta01", batch=batch, seed=seed + i, env_id_base=1000 * i, _backend=be) for i in range(2)]
    refs = [BatchedJssEnv("ta01", batch=batch, seed=seed + i, env_id_base=1000 * i, _backend=be) for i in range(2)]
    for e in envs + refs:
        e.reset()
    for r in refs:
        for _ in range(calls):
            r.rollout_steps("random", steps=steps, n_sub=3)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=be.device) for _ in range(2)]
    errors = []

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(calls):
                    envs[i].rollout_steps("random", steps=steps, n_sub=3)
            streams[i].synchronize()
        except Exception as exc:            # surfaced in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for e, r in zip(envs, refs):
        a, b = _state_snapshot(e), _state_snapshot(r)
        for name in a:
            assert np.array_equal(a[name], b[name]), f"two threads / two streams: {name} differs from the serial run"


# -----------------------------------------------------------------------------------------
# jss_steps (K x jss_step per launch, actions given up front) and the step session (state resident on the chip)
# -----------------------------------------------------------------------------------------
def recorded_actions(backend, kw, K, kind="random", seed=3, warm=0, explore=0.0, mutate=True):
    """(K, B) int32 action codes of a behaviour trajectory on a fresh batch (-2 where an episode ended: next-step
    auto-reset), optionally salted with what a careless caller sends: skips, out-of-mask jobs, NOPEs against the mask."""
    env = BatchedJssEnv(seed=seed, _backend=backend, **kw)
    env.reset()
    if warm:
        env.rollout(kind, n_iter=warm)
    start = _state_snapshot(env)
    sol0 = env.backend.numpy(env.solution)
    acts = np.array(env.backend.numpy(env.trajectory(kind, steps=K, record=("action",), explore=explore)["action"]), dtype=np.int32)
    if mutate:
        rng = np.random.default_rng(seed)
        B = env.batch
        for k in range(2, K, 5):
            b = rng.integers(0, B, size=max(1, B // 7))
            acts[k, b] = _abi.ACTION_SKIP
        for k in range(3, K, 11):
            b = rng.integers(0, B, size=max(1, B // 9))
            acts[k, b] = rng.integers(0, env.jobs_per_env[b] + 1)              # whatever: job or NOPE, legal or not
        for k in range(7, K, 13):
            acts[k, rng.integers(0, B)] = 1000                                  # out of range
    return env, start, sol0, acts


def _restore(env, start, sol0):
    be = env.backend
    for name, v in start.items():
        if name != "solution":
            be.copy_into(getattr(env, name), v)
    be.copy_into(env.solution, sol0)


def case_steps(backend, kw=None, K=40, kind="random", seed=3, warm=30):
    """jss_steps == K x jss_step: every state and output tensor at the end, and the recorded per-step streams equal
    what each jss_step call leaves in the env's outputs."""
    kw = kw or dict(instances="ta01", batch=70)
    env, start, sol0, acts = recorded_actions(backend, kw, K, kind, seed, warm)
    n = env.backend.numpy
    _restore(env, start, sol0)
    per_step = []
    for k in range(K):
        env.step(acts[k])
        per_step.append({"real_obs": n(env.real_obs), "action_mask": n(env.action_mask), "reward": n(env.reward), "done": n(env.done)})
    want = _state_snapshot(env)
    _restore(env, start, sol0)
    rec = env.steps(acts, record=("real_obs", "action_mask", "reward", "done"))
    env.synchronize()
    got = _state_snapshot(env)
    for name in want:
        assert np.array_equal(got[name], want[name]), f"jss_steps differs from {K} x jss_step in {name}"
    J = env.jobs_per_env
    for k in range(K):
        stepped = acts[k] >= 0
        for i in range(env.batch):                                           # rows < J(env): the padding of a slot is never written
            assert np.array_equal(n(rec["real_obs"][k])[i, :J[i]], per_step[k]["real_obs"][i, :J[i]]), f"step {k} env {i}: recorded obs"
        assert np.array_equal(n(rec["action_mask"][k]), per_step[k]["action_mask"]), f"step {k}: recorded mask"
        r = n(rec["reward"][k])
        assert np.array_equal(r[stepped].view(np.int32), per_step[k]["reward"][stepped].view(np.int32)), f"step {k}: recorded reward"
        assert not r[~stepped].any(), f"step {k}: reward of a skipped / reset slot must be 0"
        assert np.array_equal(n(rec["done"][k])[stepped], per_step[k]["done"][stepped]), f"step {k}: recorded done"
    assert int(np.asarray(n(env.err)).max()) != 0            # the salted trace did raise error flags (and both paths agree on them)
    # no recording: same state
    _restore(env, start, sol0)
    env.steps(acts)
    env.synchronize()
    got = _state_snapshot(env)
    for name in want:
        assert np.array_equal(got[name], want[name]), f"jss_steps (nothing recorded) differs in {name}"
    return env


def case_session(backend, kw=None, K=36, kind="random", seed=5, warm=20, depth=8, slots=0, pattern=(1, 1, 3, 8, 2)):
    """A step session == K x jss_step: after every wait the env's outputs equal what the jss_step calls leave, and
    after close every state and output tensor does; posts run ahead of waits by up to `depth` steps."""
    kw = kw or dict(instances="ta01", batch=150)
    env, start, sol0, acts = recorded_actions(backend, kw, K, kind, seed, warm)
    n = env.backend.numpy
    _restore(env, start, sol0)
    ref_out = []
    for k in range(K):
        env.step(acts[k])
        ref_out.append({"real_obs": n(env.real_obs), "action_mask": n(env.action_mask), "reward": n(env.reward),
                        "done": n(env.done), "makespan": n(env.makespan), "solution": n(env.solution)})
    want = _state_snapshot(env)
    _restore(env, start, sol0)
    dev_acts = env.backend.as_device(acts, "int32") if hasattr(env.backend, "torch") else acts
    with env.session(depth=depth, slots=slots, timeout_ms=3000) as s:
        try:
            env.step(acts[0])
            raise AssertionError("other calls must be refused while a session is open")
        except RuntimeError:
            pass
        k, i = 0, 0
        while k < K:
            m = min(pattern[i % len(pattern)], K - k, depth)
            i += 1
            if m == 1:
                s.step(dev_acts[k])
            else:
                s.post(dev_acts[k:k + m])
                s.wait()
            k += m
            last = ref_out[k - 1]
            if hasattr(env.backend, "torch"):
                # what a learner does: KERNELS on the caller's stream read the outputs behind the wait (through the caches,
                # which still hold the previous steps' lines of these very tensors), not a copy engine
                t = env.backend.torch
                for name in ("real_obs", "action_mask", "reward", "done"):
                    same = t.equal(getattr(env, name), t.as_tensor(last[name], device=env.backend.device))
                    assert same, f"session: a kernel behind wait() sees a stale {name} after step {k - 1}"
            env.backend.sync()
            for name in ("real_obs", "action_mask", "reward", "done", "makespan", "solution"):
                assert np.array_equal(n(getattr(env, name)), last[name]), f"session: {name} after step {k - 1} differs from jss_step"
        try:
            s.post(np.repeat(acts[:1], depth + 1, axis=0))
            raise AssertionError("a post beyond the ring depth must be refused")
        except RuntimeError:
            pass
    st = s.host_status()
    assert st["session_timeouts"] == 0 and st["wait_timeouts"] == 0 and st["wavefronts_exited"] > 0, st
    got = _state_snapshot(env)
    for name in want:
        assert np.array_equal(got[name], want[name]), f"after the session closed: {name} differs from {K} x jss_step"
    env.step(acts[0] * 0 - 1)                                # an ordinary call works again
    return env, st


def case_session_emulator(backend, kw=None, K=14, kind="random", seed=5, warm=20, slots=0, timeout_only=False):
    """The resident kernel under the SIMT emulator, where a launch runs to completion: the mailbox is filled first
    (post, then close), then the session is opened -- the kernel consumes every step and leaves.  Same comparison."""
    import ctypes as C
    kw = kw or dict(instances="ta01", batch=9)
    env, start, sol0, acts = recorded_actions(backend, kw, K, kind, seed, warm)
    be, lib, B = env.backend, env.backend.lib, env.batch
    _restore(env, start, sol0)
    for k in range(K):
        env.step(acts[k])
    want = _state_snapshot(env)
    _restore(env, start, sol0)
    mail, progress, status = np.zeros((K + 1, B), dtype=np.int64), np.zeros(B, dtype=np.int32), np.zeros(4, dtype=np.int32)
    sess = _abi.JssSession(mail.ctypes.data, progress.ctypes.data, status.ctypes.data, K + 1, 5 if timeout_only else 2000, slots, 0)
    d, s, o = env._refs()
    a = np.ascontiguousarray(acts, dtype=np.int32)
    if not timeout_only:
        _abi.check(lib, lib.jss_session_post(d, C.byref(sess), a.ctypes.data, 0, K, 0, 0), "post")
        assert lib.jss_session_post(d, C.byref(sess), a.ctypes.data, K, 2, 0, 0) == _abi.E_SESSION       # ring overrun
        _abi.check(lib, lib.jss_session_close(d, C.byref(sess), K, 0), "close")
    _abi.check(lib, lib.jss_session_open(d, s, o, C.byref(sess), 0), "open")
    _abi.check(lib, lib.jss_session_wait(d, C.byref(sess), 0 if timeout_only else K, 0), "wait")
    if timeout_only:       # nothing was ever posted: every wavefront gives up after the timeout, the state is as it was
        assert status[0] > 0 and status[0] == status[2] and status[1] == 0, status
        got = _state_snapshot(env)
        for name, v in start.items():
            assert np.array_equal(got[name], v), f"a timed-out session changed {name}"
        return status
    assert status[0] == 0 and status[1] == 0 and status[2] > 0 and status[3] in (1, 2, 4, 8), status
    assert (progress[:status[2]] == K).all(), progress
    got = _state_snapshot(env)
    for name in want:
        assert np.array_equal(got[name], want[name]), f"session (emulator, slots {status[3]}): {name} differs from {K} x jss_step"
    return status


def case_cr_any_factor(backend, factors=(1.2, 1.3, 0.7, 1.0 / 3.0, 2.718281828459045), insts=("ta01", "ta31"), batch=5, steps=150, seed=9):
    """CriticalRatio(due_date_factor = any float) on the device: `policy("CR", cr_factor=f)` (JSS_POLICY_CR_F64: the
    reference's float64 expression evaluated by the selector) picks, state by state, the job the reference's rule picks
    (dispatching.py:365-408, the package's rule class running its host loop on the oracle env) -- on a ragged batch, so
    both kernel flavours' selectors are exercised.  The fused rollouts refuse the code (their kernels carry no float64)."""
    from jssenv_amd import dispatching as D
    assert _abi.cr_kind(1.2) is None and _abi.POLICY_CR_F64 == _abi.POLICY["CR"] | (1 << 24)
    real = np.random.random
    np.random.random = lambda *a, **k: 1.0               # no NOPE exploration on the host side
    try:
        instances = [I.builtin_instance(n) for n in insts]
        for factor in factors:
            env = BatchedJssEnv(instances, batch=batch, seed=seed, _backend=backend)      # (a ragged list: dealt out by shape class)
            orcs = [OracleEnv(instances[t], strict=True) for t in env.table_of_env_host]
            rules = [D.CriticalRatio(due_date_factor=factor) for _ in range(batch)]
            env.reset()
            for o in orcs:
                o.reset()
            rng = np.random.default_rng(seed)
            for st in range(steps):
                dev = np.asarray(env.backend.numpy(env.policy("CR", cr_factor=factor)))
                acts = []
                for i, o in enumerate(orcs):
                    if o.nb_legal_actions == 0:
                        acts.append(_abi.ACTION_SKIP)
                        continue
                    want = rules[i](o)
                    assert dev[i] == want, f"factor {factor} step {st} env {i}: device {dev[i]} vs reference rule {want}"
                    if st % 4 == i % 4:                  # every few steps another legal action, so that the envs drift apart
                        want = int(rng.choice(np.flatnonzero(o.legal_actions)))
                    o.step(want)
                    acts.append(want)
                env.step(np.asarray(acts, dtype=np.int32))
        case_cr_f64_unfused(backend)
        # the code is for policy launches: the fused rollouts answer JSS_E_KIND, a factor that is not a positive float is refused
        be = env.backend
        d, s_, o_ = env._refs()
        env._desc.cr_factor = 1.2
        assert be.lib.jss_rollout(d, s_, o_, _abi.POLICY_CR_F64, 0, 0, 1, 0, be.stream()) == _abi.E_KIND
        env._desc.cr_factor = -1.0
        assert be.lib.jss_policy(d, s_, _abi.POLICY_CR_F64, 0, 0, be.ptr(env._actions_out), be.stream()) == _abi.E_KIND
        env._desc.cr_factor = 0.0
        for bad in (0.0, -2.0, float("nan")):
            try:
                env.policy("CR", cr_factor=bad)
                raise AssertionError("a non-positive due-date factor must be refused")
            except ValueError:
                pass
        try:
            env.policy("SPT", cr_factor=1.2)
            raise AssertionError("cr_factor belongs to CR")
        except ValueError:
            pass
    finally:
        np.random.random = real


def case_cr_f64_unfused(backend):
    """A state in which the reference's float64 ratio and a fused multiply-subtract of the same expression disagree
    (dispatching.py:391-398 rounds `length * factor` before it subtracts `now`).  Jobs 0 and 1 are both 100 long; at
    t = 120 both are legal, job 0 with 100 of work left, job 1 with 1.  fl(100 * 1.2) = 120.0, so both ratios are exactly
    0.0 and the strict `<` keeps the lower index: job 0.  With one rounding 100 * 1.2 - 120 = -4.44e-15 and job 1
    (-4.44e-15 / 1 < -4.44e-15 / 100) would win -- what the device code did while hipcc contracted the expression."""
    from jssenv_amd import dispatching as D
    machine = np.array([[2, 0, 1], [1, 0, 2], [2, 0, 1]], dtype=np.int32)
    duration = np.array([[98, 1, 1], [98, 1, 1], [120, 1, 1]], dtype=np.int32)
    inst = I.Instance("cr_unfused_3x3", machine, duration)
    assert 100 * 1.2 - 120 == 0.0                         # the reference's (NumPy / Python float) evaluation
    filler = random_instance(np.random.default_rng(1), 3, 3)   # a second table: the batch runs the per-env-table kernels too
    for insts in ([inst], [inst, filler]):
        env = BatchedJssEnv(insts, batch=len(insts) * 2, seed=1, _backend=backend)
        orc = OracleEnv(inst, strict=True)
        env.reset()
        orc.reset()
        for a in (2, 1, 1):                               # job 2 takes machine 2 until 120; job 1 runs 0-98 and 98-99
            acts = np.full(env.batch, _abi.ACTION_SKIP, dtype=np.int32)
            acts[::len(insts)] = a
            env.step(acts)
            orc.step(a)
        assert orc.current_time_step == 120 and list(np.flatnonzero(orc.legal_actions)) == [0, 1, 2]
        want = D.CriticalRatio(due_date_factor=1.2)(orc)
        assert want == 0
        dev = np.asarray(env.backend.numpy(env.policy("CR", cr_factor=1.2)))
        assert (dev[::len(insts)] == want).all(), f"device picks {dev[::len(insts)]}, the reference's unfused float64 picks {want}"


def case_cr_due_date_factor(backend, factors=(2.0, 0.5, 1.25), inst="ta01", batch=6, steps=260, seed=4):
    """CriticalRatio(due_date_factor = p / q) on the device (the factor travels in the `kind` argument, exact fraction
    comparison) picks, state by state, the job the reference's float expression picks (dispatching.py:365-408, here the
    package's rule class running its host loop on the oracle env)."""
    from jssenv_amd import dispatching as D
    assert _abi.cr_kind(1.5) == _abi.POLICY["CR"] and _abi.cr_kind(1.2) is None and _abi.cr_kind(1000.0) is None
    assert _abi.cr_kind(0.5) == _abi.POLICY["CR"] | (1 << 8) | (2 << 16)
    real = np.random.random
    np.random.random = lambda *a, **k: 1.0               # no NOPE exploration on the host side
    try:
        for factor in factors:
            code = _abi.cr_kind(factor)
            assert code is not None
            env = BatchedJssEnv(inst, batch=batch, seed=seed, _backend=backend)
            orcs = [OracleEnv(I.builtin_instance(inst), strict=True) for _ in range(batch)]
            rules = [D.CriticalRatio(due_date_factor=factor) for _ in range(batch)]
            env.reset()
            for o in orcs:
                o.reset()
            rng = np.random.default_rng(seed)
            for st in range(steps):
                dev = np.asarray(env.backend.numpy(env.policy(code)))
                acts = []
                for i, o in enumerate(orcs):
                    if o.nb_legal_actions == 0:
                        acts.append(_abi.ACTION_SKIP)
                        continue
                    want = rules[i](o)
                    assert dev[i] == want, f"factor {factor} step {st} env {i}: device {dev[i]} vs host {want}"
                    # every few steps another legal action instead, so that the envs drift apart
                    if st % 5 == i % 5:
                        want = int(rng.choice(np.flatnonzero(o.legal_actions)))
                    o.step(want)
                    acts.append(want)
                env.step(np.asarray(acts, dtype=np.int32))
        # the facade: run_episode through the fused device path == the host path of the same rule
        for factor in factors[:1]:
            rule = D.CriticalRatio(due_date_factor=factor)
            env1 = JssEnv({"instance_path": inst}, _backend=backend)
            _, mk_host = rule.run_episode(env1)
            _, mk_dev = D.CriticalRatio(due_date_factor=factor).run_episode(env1, device_rng=False)
            assert mk_host == mk_dev
            if hasattr(env1, "_run_rule"):
                real_explore, D.EXPLORATION_PROBABILITY = D.EXPLORATION_PROBABILITY, 0.0
                try:
                    _, mk_fused = D.CriticalRatio(due_date_factor=factor).run_episode(env1, device_rng=True, seed=1)
                finally:
                    D.EXPLORATION_PROBABILITY = real_explore
                assert mk_fused == mk_host, (factor, mk_fused, mk_host)
    finally:
        np.random.random = real


def case_steps_and_session_edges(backend, emulator=False):
    """Edges of the external-action forms: K = 0 and K = 1, a batch that does not fill its last wavefront, envs that
    were never reset (partial reset: every step-type call leaves them alone, the session too)."""
    import ctypes as C
    env = BatchedJssEnv("ta01", batch=11, seed=2, _backend=backend)
    which = np.ones(11, dtype=np.uint8)
    which[[3, 10]] = 0                                       # envs 3 and 10 are never reset
    env.reset(which=which)
    before = _state_snapshot(env)
    env.steps(np.zeros((0, 11), dtype=np.int32))             # K = 0: nothing happens
    env.synchronize()
    after = _state_snapshot(env)
    for name in before:
        assert np.array_equal(before[name], after[name]), f"jss_steps with K = 0 changed {name}"
    ref = BatchedJssEnv("ta01", batch=11, seed=2, _backend=backend)
    ref.reset(which=which)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 16, size=(7, 11)).astype(np.int32)        # whatever: legal or not, NOPEs included
    for k in range(7):
        ref.step(acts[k])
    want = _state_snapshot(ref)
    env.steps(acts[:1])
    env.steps(acts[1:])
    env.synchronize()
    got = _state_snapshot(env)
    for name in want:
        assert np.array_equal(got[name], want[name]), f"jss_steps (1 + 6 steps, never-reset envs) differs in {name}"
    assert not got["env_header"][[3, 10]].any() and not got["counters"][[3, 10]].any()
    # the same through a session
    env2 = BatchedJssEnv("ta01", batch=11, seed=2, _backend=backend)
    env2.reset(which=which)
    if emulator:
        lib = backend.lib
        mail, progress, status = np.zeros((8, 11), dtype=np.int64), np.zeros(11, dtype=np.int32), np.zeros(4, dtype=np.int32)
        sess = _abi.JssSession(mail.ctypes.data, progress.ctypes.data, status.ctypes.data, 8, 2000, 2, 0)
        d, s, o = env2._refs()
        a = np.ascontiguousarray(acts)
        _abi.check(lib, lib.jss_session_post(d, C.byref(sess), a.ctypes.data, 0, 7, 0, 0), "post")
        _abi.check(lib, lib.jss_session_close(d, C.byref(sess), 7, 0), "close")
        _abi.check(lib, lib.jss_session_open(d, s, o, C.byref(sess), 0), "open")
        assert status[0] == 0 and status[2] > 0
    else:
        with env2.session(depth=4, slots=2, timeout_ms=3000) as s:
            s.post(acts[:3])
            s.wait()
            for k in range(3, 7):
                s.step(acts[k])
    got = _state_snapshot(env2)
    for name in want:
        assert np.array_equal(got[name], want[name]), f"session (never-reset envs, ragged last wavefront) differs in {name}"


def case_medium_equals_full(backend, batch=40, n_iter=500, seed=29):
    """Batches of different instances whose shapes the packed kernels serve keep 24-byte medium records (three 21-bit
    cached ops, no machine clocks); `compact=False` keeps the full 32-byte records.  Same batch, same policy: the decoded
    state is equal after every chunk, and equal to the oracle at the end."""
    rng = np.random.default_rng(seed)
    sets = [[I.builtin_instance(nm) for nm in ("ta01", "ta02", "ta03")],                      # 15 x 15, env -> instance map (G16)
            [I.builtin_instance("ta21"), I.builtin_instance("ta11"), random_instance(rng, 32, 32, max_dur=200)],   # ragged, G32
            [random_instance(rng, 7, 5, max_dur=9) for _ in range(batch)],                     # one table per env
            [I.builtin_instance("ta51"), I.builtin_instance("ta31"), random_instance(rng, 64, 32, max_dur=50)],    # one wavefront per env, ragged
            [I.builtin_instance("ta71"), I.builtin_instance("ta61")]]                          # two jobs per lane (100 x 20), ragged
    for insts in sets:
        a = BatchedJssEnv(insts, batch=batch, seed=seed, env_id_base=7, records="medium", _backend=backend)
        b = BatchedJssEnv(insts, batch=batch, seed=seed, env_id_base=7, records="full", _backend=backend)
        assert a.medium and not b.medium and a.job_state.shape[-1] == _abi.NFM and b.job_state.shape[-1] == _abi.NF
        assert "machine_state" not in a._layout and "machine_state" in b._layout
        a.reset()
        b.reset()
        for chunk in (1, 3, n_iter // 2, n_iter - n_iter // 2 - 4):
            a.rollout("random", n_iter=chunk)
            b.rollout("random", n_iter=chunk)
            for i in range(0, batch, max(1, batch // 6)):
                ha, hb = a.host_state(i), b.host_state(i)
                for key in ("clock", "reward", "done", "err", "noop_flag", "makespan", "episode", "step_in_episode"):
                    assert ha[key] == hb[key], (key, i)
                for key in ("job_state", "next_op", "next2_op", "tm", "mask", "blocked", "obs", "solution", "counters"):
                    assert np.array_equal(ha[key], hb[key]), f"medium vs full records: {key} of env {i}"
            for name in ("todo_time_step_job", "needed_machine_jobs", "time_until_finish_current_op_jobs", "total_perform_op_time_jobs",
                         "total_idle_time_jobs", "idle_time_jobs_last_op", "action_illegal_no_op", "machine_state"):
                x, y = np.asarray(a.backend.numpy(getattr(a, name))), np.asarray(b.backend.numpy(getattr(b, name)))
                live = np.arange(a.jmax)[None, :] < a.jobs_per_env[:, None]
                if x.shape == live.shape:
                    x, y = np.where(live, x, 0), np.where(live, y, 0)
                assert np.array_equal(x, y), f"medium vs full records: attribute {name}"
        # every env against the oracle: clock, RNG position, the six per-job arrays, machine clocks, solution, mask, blocked
        # flags, counters, error flags, the float32 observation
        assert_batch_equals_oracle(a, "random", seed, n_iter, f"medium records, {len(insts)} instances x {batch} envs")
        d = a.state_dict()                                   # checkpoint round trip of the medium layout
        c = BatchedJssEnv(insts, batch=batch, seed=seed, env_id_base=7, records="medium", _backend=backend)
        c.load_state_dict(d)
        c.rollout("random", n_iter=5)
        a.rollout("random", n_iter=5)
        assert np.array_equal(a.backend.numpy(a.job_state), c.backend.numpy(c.job_state))
        try:
            b.load_state_dict(d)
            raise AssertionError("a medium checkpoint must not load into a full-record batch")
        except ValueError:
            pass


def case_medium_limits(backend, seed=5, jobs=32):
    """The packed words of the medium record at the library's limits: 32 jobs x 32 machines, durations 65535 (op =
    31 << 16 | 65535 = all 21 bits set, total_perform_op_time_jobs up to 32 x 65535 < 2^21, todo up to 32), in lock step
    with the oracle to the end of the episode."""
    rng = np.random.default_rng(seed)
    insts = []
    for k, (J, M) in enumerate(((jobs, 32), (jobs, 32), (3, 32))):
        machine = np.stack([rng.permutation(M) for _ in range(J)]).astype(np.int32)
        duration = np.full((J, M), 65535, dtype=np.int32) if k < 2 else (65535 - rng.integers(0, 3, size=(J, M))).astype(np.int32)
        insts.append(I.Instance(f"limit_medium_{k}", machine, duration))
    env, orcs = case_batch_lockstep(backend, insts, batch=6, n_steps=2 * jobs * 32 + 200, kind="random", seed=seed, check_every=64,
                                    records="medium")
    assert env.medium
    h = env.host_state(0)
    assert h["done"] and (h["job_state"][_abi.F_PERF] == 32 * 65535).all() and (h["job_state"][_abi.F_TODO] == 32).all()


def case_policy_step_steps(backend, batch=300, steps=9, seed=13, warm=230):
    """jss_policy_step_steps (the un-fused loop over sub-batches on several streams) == the Python loop
    `env.step(env.policy(kind), autoreset=True)` == the fused rollout: every tensor bit-identical."""
    for kw, kind in ((dict(instances="ta01"), "random"), (dict(instances=[I.builtin_instance(n) for n in ("ta01", "ta31")]), "SPT")):
        a = BatchedJssEnv(batch=batch, seed=seed, env_id_base=3, _backend=backend, **kw)
        b = BatchedJssEnv(batch=batch, seed=seed, env_id_base=3, _backend=backend, **kw)
        for e in (a, b):
            e.reset()
            e.rollout(kind, n_iter=warm if kind == "random" else 5)        # (warm = 230: some envs are about to finish)
        a.policy_step_steps(kind, steps=steps, n_sub=3)
        for _ in range(steps):
            b.step(b.policy(kind), autoreset=True)
        a.synchronize()
        sa, sb = _state_snapshot(a), _state_snapshot(b)
        for name in sa:
            assert np.array_equal(sa[name], sb[name]), f"policy_step_steps differs from the policy / step loop in {name}"
        assert a.stats()["steps"] > 0


def case_integration_level2_stub(lib_path, on_gpu):
    """INTEGRATION.md's Level-2 stub -- the ctypes binding a maintainer of the reference would paste into jss_env.py --
    executed AS PRINTED: the code block is cut out of the document, the library name is given its path, the `...` that
    stands for the reference's own parser (jss_env.py:72-95) is filled with one, and on a box without a GPU "cuda" reads
    "cpu" and the stream is 0.  Nothing else is touched.  A FIFO episode on ta01 through the stub's reset() / step():
    mask, observation, reward, done equal the oracle's at every step, 225 steps, makespan 1486 (golden G3)."""
    import os
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## Level 2"):]
    code = sec[sec.index("```python") + len("```python"):]
    code = code[:code.index("```")]
    assert 'C.CDLL("libjss_hip.so")' in code and code.count("...  ") == 1
    code = code.replace('C.CDLL("libjss_hip.so")', f"C.CDLL({lib_path!r})")
    code = code.replace(code[code.index("..."):code.index("\n", code.index("..."))], "_parse(self, env_config)")
    if not on_gpu:
        code = code.replace('"cuda"', '"cpu"').replace("torch.cuda.current_stream().cuda_stream", "0")
    inst = I.builtin_instance("ta01")

    def _parse(self, env_config):                         # what jss_env.py:72-95 leaves behind
        self.jobs, self.machines = inst.jobs, inst.machines
        self.instance_matrix = np.stack([inst.machine, inst.duration], axis=2).astype(np.int64)   # [j][k] = (machine, duration)
        self.max_time_op = int(inst.duration.max())
        self.jobs_length = inst.duration.sum(axis=1)
        self.max_time_jobs = int(self.jobs_length.max())
        self.sum_op = int(inst.duration.sum())

    ns = {"gym": types.SimpleNamespace(Env=object), "_parse": _parse}
    exec(compile(code, "INTEGRATION.md#level-2", "exec"), ns)
    env = ns["JssEnv"]({"instance_path": "ta01"})
    orc = OracleEnv(inst, strict=True)
    obs, want = env.reset(), orc.reset()
    steps, done = 0, False
    while not done:
        assert (obs["action_mask"] == orc.legal_actions).all(), steps
        assert np.abs(obs["real_obs"].astype(np.float64) - orc.state).max() <= OBS_TOL, steps
        legal = orc.legal_actions[:-1]
        if legal.any():                                   # FIFO (dispatching.py:133-156): longest idle since its last op, first index wins
            a = int(np.argmax(np.where(legal, orc.idle_time_jobs_last_op, -1)))
        else:
            a = orc.jobs
        obs, r, done, trunc, info = env.step(a)
        _, r0, d0 = orc.step(a)[:3]
        assert reward_close(r, r0) and done == bool(d0) and trunc is False and info == {}, steps
        steps += 1
    assert steps == 225 and orc.current_time_step == 1486 and int(env._env[0, 0]) == 1486
    assert (env._sol[0].cpu().numpy() == orc.solution).all()


def case_two_envs_per_wavefront(backend_two, backend_one=None, steps=60, n_envs=7, seed=19, pair_rounds=90):
    """The one-step launches of the one-wavefront-per-env flavour with a wavefront serving TWO envs in turn (round 6:
    jss_kernel_two; `backend_two`'s default kernel carries JSS_KERNEL_TWO_ENVS_PER_WAVE so that
    small batches take the form too).  An ODD number of envs (the last wavefront owns one), ragged 30- / 50-job instances with
    per-env tables: policy + jss_step in lock step with the oracle (forced NOPEs, next-step auto-resets = the kernel's reset
    routing for one env of a pair), jss_rollout(n_iter = 1) and the sub-batch form against orc_rollout, a never-reset env in
    the pair; then the same calls with one env per wavefront give the same bytes."""
    assert _abi.KERNEL[backend_two.default_kernel] & 4
    names = ["ta31", "ta51", "ta41", "ta62"]                      # 30x15, 50x15, 30x20, 50x20: one job per lane, ragged
    case_batch_lockstep(backend_two, names, batch=n_envs, n_steps=steps, kind="random", nope_every=7, check_every=3)
    case_rollout(backend_two, names, batch=n_envs, n_iter=0, chunks=(1,) * 6 + (40, 1, 1), kind="random")
    case_rollout(backend_two, names[:2], batch=5, n_iter=0, chunks=(1,) * 5, kind="SPT", autoreset=False)
    case_rollout_steps(backend_two, batch=2 * 64 + 3, steps=4, n_sub=2, seed=seed)
    # a pair whose second env was never reset: left alone, its neighbour steps
    insts = [I.builtin_instance(n) for n in names]
    env = BatchedJssEnv(insts, batch=4, seed=seed, order="interleaved", _backend=backend_two)
    env.reset(which=np.array([1, 0, 1, 1], dtype=np.uint8))
    before = np.array(env.backend.numpy(env.job_state)[1], copy=True)
    for _ in range(12):
        env.step(env.policy("random"), autoreset=True)
        env.rollout("random", n_iter=1)
    assert np.array_equal(env.backend.numpy(env.job_state)[1], before) and env.backend.numpy(env.env_header)[1, _abi.H_EPISODE] == 0
    assert env.backend.numpy(env.counters)[[0, 2, 3], 0].min() > 0 and env.backend.numpy(env.counters)[1, 0] == 0
    if backend_one is not None:
        # same calls, one env per wavefront: the same bytes -- over many short episodes (tiny instances), so that the restarts
        # inside the fused rollout and jss_step_autoreset's routing of ONE env of a pair through the reset body both happen often
        rng = np.random.default_rng(seed)
        tiny = [random_instance(rng, int(rng.integers(3, 10)), int(rng.integers(2, 6))) for _ in range(5)]
        outs = []
        for be in (backend_two, backend_one):
            e = BatchedJssEnv(tiny, batch=9, seed=seed, env_id_base=9, order="interleaved", _backend=be)
            e.reset()
            for k in range(pair_rounds):
                e.rollout("random", n_iter=1)
                e.step(e.policy("random"), autoreset=True)
            outs.append(_state_snapshot(e))
            assert e.stats()["episodes"] >= pair_rounds // 5
        for name in outs[0]:
            assert np.array_equal(outs[0][name], outs[1][name]), f"two envs per wavefront differs from one in {name}"


def case_fuzz_mixed_calls(backend_factory, rounds=12, seed=2026, max_batch=260, max_iters=140, shapes=None, kernels=None):
    """Random populations through a random MIX of the calls that all advance the same process -- `iters` x (policy + step)
    with auto-restart from a fresh reset: rollout(n), n x rollout(1), rollout_steps (sub-batches / parts on streams),
    trajectory(n), steps replayed from a recorded trajectory is NOT one of them (its actions come from outside), n x
    (policy -> step with next-step auto-reset) -- then EVERY env against the oracle's rollout.  Random shapes (1..128 jobs,
    2..64 machines, durations up to 65 535, machine orders that repeat machines), random batch sizes (odd ones: the last
    wavefront of a two-envs-per-wavefront launch owns one env), every dispatching rule with and without NOPE exploration, all
    deals (the default by shape class, interleaved, explicit by_shape), every kernel form.  `backend_factory(kernel)` -> backend."""
    rng = np.random.default_rng(seed)
    kernels = kernels or ["auto", "wave", "auto-2env", "wave-2env", "auto-1env"]
    kinds = ["random", "FIFO", "SPT", "MWR", "LWR", "MOR", "LOR", "CR"]
    backends = {}
    for r in range(rounds):
        n_inst = int(rng.integers(1, 7))
        insts = []
        for _ in range(n_inst):
            if shapes is not None:
                J, M = shapes[int(rng.integers(len(shapes)))]
            else:
                J = int(rng.choice([rng.integers(1, 17), rng.integers(1, 33), rng.integers(17, 65), rng.integers(60, 129)]))
                M = int(rng.choice([rng.integers(2, 17), rng.integers(2, 33), rng.integers(2, 65)]))
            insts.append(random_instance(rng, J, M, max_dur=int(rng.choice([9, 99, 65535])), permutation=bool(rng.integers(2))))
        batch = int(rng.integers(1, max_batch)) | int(rng.integers(2))
        kernel = kernels[int(rng.integers(len(kernels)))]
        be = backends.setdefault(kernel, backend_factory(kernel))
        ragged = len({(0 if (i.jobs <= 16 and i.machines <= 16) else 1 if (i.jobs <= 32 and i.machines <= 32) else 2) for i in insts}) > 1
        order = [None, "interleaved"][int(rng.integers(2))] if (n_inst == 1 or batch == n_inst) else \
            [None, "interleaved", "by_shape"][int(rng.integers(3))]
        kind = kinds[int(rng.integers(len(kinds)))]
        explore = float(rng.choice([0.0, 0.0, 0.1, 0.35])) if kind != "random" else 0.0
        sd = int(rng.integers(0, 1 << 30))
        base = int(rng.integers(0, 1 << 20))
        label = (f"fuzz round {r}: {[(i.jobs, i.machines) for i in insts]} x {batch}, kernel {kernel}, order {order}, {kind}, "
                 f"explore {explore}, seed {sd}")
        try:
            env = BatchedJssEnv(insts if n_inst > 1 else insts[0], batch=batch, seed=sd, env_id_base=base, order=order, _backend=be)
            env.reset()
            done = 0
            iters = int(rng.integers(1, max_iters))
            while done < iters:
                n = int(min(iters - done, rng.integers(1, 24)))
                how = int(rng.integers(5))
                if how == 0:
                    env.rollout(kind, n_iter=n, explore=explore)
                elif how == 1:
                    for _ in range(n):
                        env.rollout(kind, n_iter=1, explore=explore)
                elif how == 2:
                    env.rollout_steps(kind, steps=n, n_sub=int(rng.integers(1, 4)), explore=explore)
                elif how == 3:
                    env.trajectory(kind, steps=n, explore=explore, record=("action", "reward") if rng.integers(2) else
                                   ("real_obs", "action_mask", "action", "reward", "done"))
                else:
                    for _ in range(n):
                        env.step(env.policy(kind, explore=explore), autoreset=True)
                done += n
            env.synchronize()
            assert_batch_equals_oracle(env, kind, sd, iters, label, explore=explore)
        except AssertionError:
            raise
        except Exception as exc:
            raise AssertionError(f"{label}: {type(exc).__name__}: {exc}") from exc
        del env
    del ragged
