"""Kernel logic under the SIMT emulator (tests/emu): the unmodified
jssenv_amd/csrc/jss_kernels.hip compiled with g++ and executed lane by lane on the CPU,
driven through the same host layer and C ABI as on the GPU.  Sizes are small (the
emulator runs ~250 env-steps/s); the full-size runs live in tests/test_hip_parity.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

import parity_cases as P  # noqa: E402


@pytest.fixture(scope="module", params=["auto", "wave"])
def emu(request):
    """auto: 64/G envs per wavefront where the shape fits (ta01 -> G=16, ta21/ta41/dmu16 -> G=32),
    one wavefront per env otherwise (ta51.., ragged batches up to 100 jobs); wave: force the latter."""
    from emu_backend import EmuBackend
    return EmuBackend(default_kernel=request.param)     # the flavour travels per call in JssDesc.kernel


def test_published_ta01(emu):
    P.case_published(emu, "ta01")


def test_published_ta41_prefix(emu):
    P.case_published(emu, "ta41", max_rows=260)


def test_random_golden_ta01(emu):
    P.case_random_golden(emu, "ta01", max_rows=300)


def test_random_golden_ta80_two_jobs_per_lane(emu):
    P.case_random_golden(emu, "ta80", max_rows=180)


def test_random_golden_dmu16(emu):
    P.case_random_golden(emu, "dmu16", max_rows=150)


def test_batch_ragged_random_policy(emu):
    P.case_batch_lockstep(emu, ["ta01", "ta31", "ta51", "ta71"], batch=6, n_steps=90, kind="random", nope_every=7,
                          check_every=3)


@pytest.mark.parametrize("kind", ["FIFO", "SPT", "MWR", "LWR", "MOR", "LOR", "CR"])
def test_batch_rules(emu, kind):
    P.case_batch_lockstep(emu, ["ta01", "ta21"], batch=3, n_steps=60, kind=kind, check_every=5)


def test_rollout_autoreset(emu):
    P.case_rollout(emu, ["ta01"], batch=5, n_iter=None or 300, chunks=(120, 1, 179))


def test_rollout_ragged(emu):
    P.case_rollout(emu, ["ta02", "ta72"], batch=3, n_iter=0, chunks=(64, 40), kind="SPT")


def test_rule_makespan_ta01(emu):
    P.case_rule_makespans(emu, rules=("FIFO", "SPT", "MOR", "CR"), insts=("ta01",))


def test_error_semantics(emu):
    P.case_error_semantics(emu)
    P.case_facade_errors(emu)


def test_state_invariants(emu):
    P.case_state_invariants(emu, episodes=1)


def test_dispatching_module(emu):
    P.case_dispatching_seeded(emu, keys=["trace_FIFO_ta01_0", "trace_SPT_ta01_1"])
    P.case_dispatching_deterministic(emu, rules=("SPT", "CR"))


def test_edge_shapes(emu):
    P.case_edge_shapes(emu, steps=24, batch_per_shape=2)


def test_ragged_batch_with_a_64_job_env(emu):
    P.case_ragged_j64_nope_flag(emu)


def test_vector_env_features(emu):
    P.case_vector_env_features(emu)


def test_bucketed_equals_padded(emu):
    P.case_bucketed_equals_padded(emu, n_envs=24, n_iter=60)


def test_vector_facade_over_shape_classes(emu):
    P.case_vector_facade_by_shape(emu, steps=40)


def test_ragged_population_in_padded_tensors_by_shape_class(emu):
    P.case_by_shape_padded(emu, n_envs=20, iters=24, tail=2)


def test_multi_entry_points_equal_the_single_set_calls(emu):
    P.case_multi_entry_points(emu, steps=12)


def test_fused_grid_over_the_shape_classes_every_env_equals_the_oracle(emu):
    """jss_multi_kernel<kRollout1 / kPolicy / kStep / kReset>: ta01-ta80 as four shape classes in ONE grid per call"""
    P.case_bucketed_every_env_vs_oracle(emu, n_envs=80, iters=20, unfused_tail=3)


def test_rules_with_exploration(emu):
    """the rules' 10 % NOPE exploration (dispatching.py:113) drawn from the counter RNG on the device"""
    env, orcs = P.case_batch_lockstep(emu, ["ta01", "ta21"], batch=4, n_steps=70, kind="SPT", check_every=9, explore=0.1)
    env, orcs = P.case_batch_lockstep(emu, ["ta71"], batch=2, n_steps=70, kind="FIFO", check_every=9, explore=0.25)


def test_vector_facade(emu):
    P.case_vector_facade(emu)


def test_instance_resampling(emu):
    P.case_instance_resampling(emu)


def test_rollout_steps_equals_rollout(emu):
    P.case_rollout_steps(emu, batch=130, steps=3, n_sub=3)


def test_nope_fuzz_tiny_instances(emu):
    P.case_nope_fuzz(emu, batch=8, steps=30)


def test_trajectory_equals_policy_plus_step(emu):
    P.case_trajectory(emu, "ta01", batch=9, steps=40, kind="random", warm=200)          # crosses episode ends
    P.case_trajectory(emu, ["ta01", "ta31", "ta71"], batch=5, steps=25, kind="SPT", explore=0.2)   # ragged: one launch per shape class
    P.case_trajectory(emu, ["ta01", "ta31", "ta71"], batch=5, steps=14, kind="SPT", explore=0.2, order="interleaved")   # two jobs per lane + narrow body


def test_trajectory_frozen_without_autoreset(emu):
    P.case_trajectory(emu, "ta01", batch=4, steps=12, kind="FIFO", warm=220, autoreset=False)


def test_dispatching_fused_on_device(emu):
    P.case_dispatching_on_device(emu, rules=("SPT", "CR"), num_episodes=3)


def test_compact_records_equal_full_records(emu):
    P.case_compact_equals_full(emu, insts=("ta01", "ta51"), batch=3, n_iter=120)


def test_compact_records_at_the_limits(emu):
    P.case_compact_limits(emu, shapes=((3, 64), (16, 16)))


def test_steps_equal_repeated_step(emu):
    P.case_steps(emu, dict(instances="ta01", batch=9), K=30, warm=200)                           # crosses episode ends
    P.case_steps(emu, dict(instances=["ta01", "ta31", "ta71"], batch=5), K=16, kind="SPT", warm=5)   # ragged: one launch per shape class
    P.case_steps(emu, dict(instances=["ta01", "ta31", "ta71"], batch=5, order="interleaved"), K=10, kind="SPT", warm=5)   # two jobs per lane


def test_step_session_resident_kernel(emu):
    """jss_session_*: every env set in registers (slots 1) and parked in LDS between visits (slots 2, 4)."""
    P.case_session_emulator(emu, dict(instances="ta01", batch=9), K=14, warm=215)                 # shared table, compact records
    P.case_session_emulator(emu, dict(instances="ta01", batch=21), K=10, warm=40, slots=2)
    P.case_session_emulator(emu, dict(instances="ta41", batch=9), K=8, warm=30, slots=4)          # 32-lane groups
    P.case_session_emulator(emu, dict(instances=["ta02", "ta03"], batch=11), K=8, slots=2)        # env -> instance map, full records
    P.case_session_emulator(emu, dict(instances=["ta01", "ta31", "ta71"], batch=7), K=8, kind="SPT", slots=2)   # ragged


def test_step_session_times_out_instead_of_hanging(emu):
    P.case_session_emulator(emu, dict(instances="ta01", batch=9), K=3, timeout_only=True)


def test_critical_ratio_any_due_date_factor_on_device(emu):
    P.case_cr_any_factor(emu, factors=(1.2, 0.7), steps=50, batch=3)


def test_critical_ratio_due_date_factor_on_device(emu):
    P.case_cr_due_date_factor(emu, steps=60, batch=3, factors=(2.0, 0.5))


def test_steps_and_session_edges(emu):
    P.case_steps_and_session_edges(emu, emulator=True)


def test_medium_records_equal_full_records(emu):
    P.case_medium_equals_full(emu, batch=6, n_iter=90)


def test_medium_records_at_the_limits(emu):
    P.case_medium_limits(emu, jobs=4)


def test_policy_step_steps_equals_the_loop(emu):
    P.case_policy_step_steps(emu, batch=66, steps=3, warm=12)


@pytest.mark.parametrize("records", [None, "medium"])
def test_two_envs_per_wavefront(records):
    """jss_kernel_two under the emulator (small batches take the form on request), on full and on medium job records."""
    from emu_backend import EmuBackend
    two, one = EmuBackend(default_kernel="wave-2env"), EmuBackend(default_kernel="wave-1env")
    two.default_records = one.default_records = records
    P.case_two_envs_per_wavefront(two, one, steps=24 if records is None else 10, n_envs=5 if records is None else 3,
                                  pair_rounds=90 if records is None else 30)


def test_fuzz_mixed_calls_against_the_oracle():
    """Random populations, deals, kernel forms and call mixes under the emulator (small: it runs ~250 env-steps/s)."""
    from emu_backend import EmuBackend
    P.case_fuzz_mixed_calls(lambda kernel: EmuBackend(default_kernel=kernel), rounds=5, max_batch=8, max_iters=14, seed=7,
                            shapes=[(3, 3), (9, 5), (17, 4), (20, 20), (40, 6), (66, 3)])
