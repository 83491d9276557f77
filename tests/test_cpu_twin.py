"""libjss_cpu.so -- the from-scratch C++/OpenMP twin with the HIP library's C ABI -- against the oracle and the
golden vectors, through the same host layer and the same parity cases as the GPU path (tests/parity_cases.py
is backend-agnostic).  Runs everywhere (no GPU); sizes are the GPU suite's."""
import numpy as np
import pytest

import golden_util as G
import parity_cases as P
from jssenv_amd import instances as I


@pytest.fixture(scope="module")
def cpu():
    from jssenv_amd.env import CpuBackend
    be = CpuBackend()
    assert be.lib.jss_backend().startswith(b"cpu")
    return be


def test_identical_symbols(cpu):
    import ctypes
    from jssenv_amd import _abi
    from jssenv_amd.build import build_extension
    hip = ctypes.CDLL(build_extension())
    for name in _abi.SYMBOLS:
        assert hasattr(hip, name) and hasattr(cpu.lib, name), name
    assert cpu.lib.jss_abi_version() == hip.jss_abi_version() == _abi.ABI_VERSION


@pytest.mark.parametrize("inst", G.PUBLISHED)
def test_published_schedules(cpu, inst):
    P.case_published(cpu, inst)


@pytest.mark.parametrize("inst", G.RANDOM)
def test_random_golden(cpu, inst):
    P.case_random_golden(cpu, inst)


def test_batch_ragged_random_policy(cpu):
    names = ["ta01", "ta11", "ta21", "ta31", "ta41", "ta51", "ta61", "ta71", "dmu16"]
    P.case_batch_lockstep(cpu, names, batch=18, n_steps=300, kind="random", nope_every=9, check_every=7)


@pytest.mark.parametrize("kind", ["FIFO", "SPT", "MWR", "LWR", "MOR", "LOR", "CR"])
def test_batch_rules(cpu, kind):
    P.case_batch_lockstep(cpu, ["ta01", "ta21", "ta72"], batch=3, n_steps=2500, kind=kind, check_every=50)


def test_rollout(cpu):
    P.case_rollout(cpu, ["ta01"], batch=32, n_iter=0, chunks=(300, 1, 1, 555))
    P.case_rollout(cpu, ["ta02", "ta72", "ta45", "dmu17"], batch=8, n_iter=0, chunks=(512, 700), kind="random")
    P.case_rollout(cpu, ["ta02", "ta72"], batch=4, n_iter=0, chunks=(400,), kind="SPT", autoreset=False)


def test_rule_makespans(cpu):
    P.case_rule_makespans(cpu, insts=("ta01", "ta41", "ta80"))


def test_error_semantics(cpu):
    P.case_error_semantics(cpu)
    P.case_facade_errors(cpu)


def test_state_invariants(cpu):
    P.case_state_invariants(cpu, episodes=2)


def test_dispatching_module(cpu):
    P.case_dispatching_seeded(cpu)
    P.case_dispatching_deterministic(cpu, insts=("ta01",))


def test_edge_shapes(cpu):
    P.case_edge_shapes(cpu, steps=200, batch_per_shape=3)


def test_vector_env_and_resampling(cpu):
    P.case_vector_env_features(cpu)
    P.case_vector_facade(cpu)
    P.case_instance_resampling(cpu)


def test_bucketed_and_rollout_steps(cpu):
    P.case_bucketed_equals_padded(cpu, n_envs=48, n_iter=400)


def test_vector_facade_over_shape_classes(cpu):
    P.case_vector_facade_by_shape(cpu)


def test_ragged_population_in_padded_tensors_by_shape_class(cpu):
    P.case_by_shape_padded(cpu, n_envs=200, iters=200)


def test_multi_entry_points_equal_the_single_set_calls(cpu):
    P.case_multi_entry_points(cpu)


def test_bucketed_every_env_equals_the_oracle(cpu):
    """config 5 without padding through the jss_multi_* entry points, every env of every shape class against the oracle"""
    P.case_bucketed_every_env_vs_oracle(cpu, n_envs=400, iters=260, unfused_tail=7)
    P.case_rollout_steps(cpu, batch=150, steps=30, n_sub=3)


def test_every_env_against_the_batch_oracle(cpu):
    """the batch driver of the oracle (OpenMP over envs) as the checker for whole batches, all three table layouts"""
    P.case_every_env_vs_oracle(cpu, "ta01 x 300", dict(instances="ta01", batch=300), "random", 400)
    P.case_every_env_vs_oracle(cpu, "ta41 SPT + exploration", dict(instances="ta41", batch=64), "SPT", 700, explore=0.1)
    P.case_every_env_vs_oracle(cpu, "synthetic 50x20 x 96", dict(instances=I.synthetic_packed(96, 50, 20)), "random", 1200)
    P.case_every_env_vs_oracle(cpu, "mixed x 240", dict(instances=[I.builtin_instance(f"ta{k:02d}") for k in range(1, 81)], batch=240),
                               "random", 500)
    P.case_every_env_vs_oracle(cpu, "frozen", dict(instances="ta01", batch=40), "FIFO", 400, autoreset=False)


def test_trajectory(cpu):
    P.case_trajectory(cpu, "ta01", batch=9, steps=60, kind="random", warm=200)
    P.case_trajectory(cpu, ["ta01", "ta31", "ta71"], batch=7, steps=40, kind="SPT", explore=0.2)
    P.case_trajectory(cpu, ["ta01", "ta31", "ta71"], batch=7, steps=30, kind="SPT", explore=0.2, order="interleaved")
    P.case_trajectory(cpu, "ta01", batch=4, steps=12, kind="FIFO", warm=220, autoreset=False)


def test_config1_ta01_fifo_on_cpu(cpu):
    """BASELINE config 1: ta01 single env on CPU, FIFO dispatching rule to completion -- through the reference's own
    call shape (make + rule(env) + step), no GPU anywhere.  225 steps, makespan 1486 (golden G3)."""
    from jssenv_amd import dispatching as D, make
    env = make("jss-v1", env_config={"instance_path": "ta01"}, device="cpu")
    assert env._b.backend.name == "cpu"
    real = np.random.random
    np.random.random = lambda *a, **k: 1.0        # the golden run disabled the rules' 10 % NOPE exploration this way
    try:
        env.reset()
        rule, steps, done = D.get_rule("FIFO"), 0, False
        while not done:
            _, _, done, _, _ = env.step(rule(env))
            steps += 1
    finally:
        np.random.random = real
    assert (steps, env.current_time_step, env.last_time_step) == (225, 1486, 1486)


def test_thread_count_does_not_change_results():
    from jssenv_amd import BatchedJssEnv
    from jssenv_amd.env import CpuBackend
    outs = []
    for threads in (1, 3):
        env = BatchedJssEnv(["ta01", "ta31"], batch=40, seed=5, _backend=CpuBackend(threads=threads))
        env.reset()
        env.rollout("random", n_iter=500)
        outs.append((env.job_state.copy(), env.counters.copy(), env.real_obs.copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


def test_configs_4_and_5_reduced(cpu):
    """The GPU suite's full-size config 4 / 5 cases at a size the host cores finish in seconds."""
    P.case_config4_synthetic(cpu, batch=192, sample=16)
    P.case_config5_mixed(cpu, batch=640)


def test_nope_fuzz_tiny_instances(cpu):
    P.case_nope_fuzz(cpu, batch=64, steps=200)


def test_critical_ratio_custom_due_date_factor(cpu):
    """The device selector hard-codes the reference's default factor 1.5; any other factor must take the host loop
    (ADVICE r1): the rule then gives the same episode on the package's env as on the oracle (no device selector)."""
    from jssenv_amd import dispatching as D
    from jssenv_amd import instances as I
    from jssenv_amd.env import JssEnv
    from oracle import OracleEnv
    real = np.random.random
    np.random.random = lambda *a, **k: 1.0
    try:
        res = {}
        for factor in (1.5, 0.5):
            rule = D.CriticalRatio(due_date_factor=factor)
            env = JssEnv({"instance_path": "ta01"}, _backend=cpu)
            _, mk_pkg = rule.run_episode(env)
            orc = OracleEnv(I.builtin_instance("ta01"))
            _, mk_orc = D.CriticalRatio(due_date_factor=factor).run_episode(orc)
            assert mk_pkg == mk_orc, (factor, mk_pkg, mk_orc)
            res[factor] = mk_pkg
        assert res[1.5] == 1426 and res[0.5] != 1426      # G3 value for the default; another schedule otherwise
    finally:
        np.random.random = real


def test_config2_every_env_against_the_oracle(cpu):
    """BASELINE config 2 at full size (ta01 x 4096, random masked, auto-restart): EVERY env of the twin's batch against
    its own oracle run -- the link between the oracle and the GPU suite's `every env bit-equal to the twin` test."""
    from jssenv_amd import BatchedJssEnv, builtin_instance
    from oracle import OracleEnv
    inst = builtin_instance("ta01")
    B, seed, iters = 4096, 5, 300
    env = BatchedJssEnv(inst, batch=B, seed=seed, env_id_base=123, _backend=cpu)
    env.reset()
    env.rollout("random", n_iter=iters)
    clock, cnt, sol = env.env_header[:, 0], env.counters, env.solution
    todo = env.todo_time_step_job
    orc = OracleEnv(inst, strict=True)
    for i in range(B):
        orc.reset()
        r = orc.rollout("random", seed, 123 + i, iters, episode=1)
        assert orc.current_time_step == clock[i] and (orc.todo_time_step_job == todo[i]).all(), i
        assert (orc.solution == sol[i]).all(), i
        assert (cnt[i, 0], cnt[i, 1], cnt[i, 2]) == (r["steps"], r["episodes"], r["makespan_sum"]), i
    P.assert_matches_oracle(env.host_state(B - 1), orc, "last env")


def test_dispatching_fused_on_device(cpu):
    P.case_dispatching_on_device(cpu)


def test_rollout_steps_multi_equals_separate_rollouts(cpu):
    """jss_rollout_steps_multi (several env sets in one call) == one jss_rollout per set"""
    import ctypes as C
    from jssenv_amd import _abi
    from jssenv_amd.env import BatchedJssEnv
    kws = [dict(instances="ta01", batch=9), dict(instances=["ta31", "ta02"], batch=5), dict(instances="ta51", batch=3)]
    a = [BatchedJssEnv(seed=4, env_id_base=10 * i, _backend=cpu, **kw) for i, kw in enumerate(kws)]
    b = [BatchedJssEnv(seed=4, env_id_base=10 * i, _backend=cpu, **kw) for i, kw in enumerate(kws)]
    for e in a + b:
        e.reset()
    n = len(a)
    D, S, O = C.POINTER(_abi.JssDesc), C.POINTER(_abi.JssState), C.POINTER(_abi.JssOut)
    rc = cpu.lib.jss_rollout_steps_multi(n, (D * n)(*[C.pointer(e._desc) for e in a]), (S * n)(*[C.pointer(e._state) for e in a]),
                                         (O * n)(*[C.pointer(e._out) for e in a]), _abi.POLICY["random"], 4, 0, 120,
                                         _abi.ROLLOUT_AUTORESET | _abi.ROLLOUT_FORK_JOIN, (C.c_void_p * n)())
    assert rc == 0
    for x, y in zip(a, b):
        y.rollout("random", n_iter=120)
        for name in BatchedJssEnv._STATE_TENSORS:
            assert np.array_equal(getattr(x, name), getattr(y, name)), name
    assert cpu.lib.jss_rollout_steps_multi(0, None, None, None, 0, 0, 0, 1, 0, None) == _abi.E_NULL


def test_compact_records_equal_full_records(cpu):
    P.case_compact_equals_full(cpu, insts=("ta01", "ta41", "ta51", "ta71"))


def test_compact_records_at_the_limits(cpu):
    P.case_compact_limits(cpu)


def test_ragged_batch_with_a_64_job_env(cpu):
    P.case_ragged_j64_nope_flag(cpu, steps=200)


def test_launch_form_driver_and_render_rows(cpu):
    """The launch-form driver of the GPU tests (per-launch and sub-batch forms) and the render rows, on the twin."""
    label, kw, kind, iters, explore = P.FULL_SIZE_CONFIGS[0]
    P.case_every_env_vs_oracle(cpu, label, kw(), kind, 120, explore, form="per_launch")
    P.case_every_env_vs_oracle(cpu, label, kw(), kind, 120, explore, form="fork_join", n_sub=3)
    P.case_render_rows_from_device_solution(cpu)


def test_steps_and_step_session(cpu):
    P.case_steps(cpu, dict(instances="ta01", batch=70), K=60, warm=200)
    P.case_steps(cpu, dict(instances=["ta01", "ta31", "ta71"], batch=11), K=40, kind="SPT", warm=5)
    P.case_steps(cpu, dict(instances=["ta01", "ta31", "ta71"], batch=11, order="interleaved"), K=30, kind="SPT", warm=5)
    P.case_session(cpu, dict(instances="ta01", batch=50), K=36)
    P.case_session(cpu, dict(instances=["ta02", "ta51"], batch=7), K=20, kind="FIFO")


def test_critical_ratio_any_due_date_factor_on_device(cpu):
    P.case_cr_any_factor(cpu)


def test_critical_ratio_due_date_factor_on_device(cpu):
    P.case_cr_due_date_factor(cpu)


def test_steps_and_session_edges(cpu):
    P.case_steps_and_session_edges(cpu)


def test_medium_records_equal_full_records(cpu):
    P.case_medium_equals_full(cpu)


def test_medium_records_at_the_limits(cpu):
    P.case_medium_limits(cpu)


def test_policy_step_steps_equals_the_loop(cpu):
    P.case_policy_step_steps(cpu)


def test_fuzz_mixed_calls_against_the_oracle(cpu):
    P.case_fuzz_mixed_calls(lambda kernel: cpu, rounds=40, max_batch=90, max_iters=120, kernels=["auto"])
