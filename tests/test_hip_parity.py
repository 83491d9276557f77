"""Parity of the real HIP path (libjss_hip.so on an MI355X) with the oracle and the golden
vectors, through the public host API / C ABI.  Run with -m gpu on the GPU box."""
import numpy as np
import pytest

import golden_util as G
import parity_cases as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "wave"])
def hip(request):
    """auto: packed kernel (64/G envs per wavefront) where the batch shape fits, wave-per-env
    otherwise; wave: force one wavefront per env everywhere."""
    from jssenv_amd.env import HipBackend
    be = HipBackend("cuda:0")
    assert be.name == "hip" and be.lib.jss_backend() == b"hip:gfx950"
    be.default_kernel = request.param                   # the flavour travels per call in JssDesc.kernel
    return be


@pytest.mark.parametrize("inst", G.PUBLISHED)
def test_published_schedules(hip, inst):
    """G1: makespan and machine/job schedules of the reference's tests/test_solutions.py, bit-exact."""
    P.case_published(hip, inst)


@pytest.mark.parametrize("inst", G.RANDOM)
def test_random_golden(hip, inst):
    """G2: random traces incl. NOPEs forced against the mask; ta80 = two jobs per lane."""
    P.case_random_golden(hip, inst)


def test_batch_ragged_random_policy(hip):
    names = ["ta01", "ta11", "ta21", "ta31", "ta41", "ta51", "ta61", "ta71", "dmu16"]
    P.case_batch_lockstep(hip, names, batch=27, n_steps=400, kind="random", nope_every=9, check_every=7)


@pytest.mark.parametrize("kind", ["FIFO", "SPT", "MWR", "LWR", "MOR", "LOR", "CR"])
def test_batch_rules(hip, kind):
    P.case_batch_lockstep(hip, ["ta01", "ta21", "ta72"], batch=6, n_steps=2500, kind=kind, check_every=25)


def test_rollout_autoreset(hip):
    P.case_rollout(hip, ["ta01"], batch=64, n_iter=0, chunks=(300, 1, 1, 555))


def test_rollout_ragged(hip):
    P.case_rollout(hip, ["ta02", "ta72", "ta45", "dmu17"], batch=16, n_iter=0, chunks=(512, 700), kind="random")
    P.case_rollout(hip, ["ta02", "ta72"], batch=4, n_iter=0, chunks=(400,), kind="SPT", autoreset=False)


def test_trajectory_equals_policy_plus_step(hip):
    """jss_trajectory: K steps per launch, every transition recorded, == K x (jss_policy, jss_step)."""
    P.case_trajectory(hip, "ta01", batch=130, steps=300, kind="random", warm=150)       # shared table, crosses episode ends
    P.case_trajectory(hip, ["ta01", "ta31", "ta51", "ta71"], batch=11, steps=120, kind="SPT", explore=0.2)   # ragged: by shape class
    P.case_trajectory(hip, ["ta01", "ta31", "ta51", "ta71"], batch=11, steps=90, kind="random", order="interleaved")   # ragged, padded extents' kernel
    P.case_trajectory(hip, ["ta02", "ta03", "ta04"], batch=9, steps=260, kind="random")  # env -> instance map, global tables
    P.case_trajectory(hip, "ta41", batch=6, steps=40, kind="FIFO", warm=590, autoreset=False)


def test_rule_makespans(hip):
    """G3: ta01 FIFO 1486 / SPT 1462, ta41 SPT 2499 ... as captured from the live reference."""
    P.case_rule_makespans(hip, insts=("ta01", "ta41", "ta80"))


def test_error_semantics(hip):
    P.case_error_semantics(hip)
    P.case_facade_errors(hip)


def test_state_invariants(hip):
    P.case_state_invariants(hip, episodes=5)


def test_config2_ta01_batch4096_random(hip):
    """BASELINE config 2 at full size: size-independent properties on every env + oracle on a sample."""
    from jssenv_amd import BatchedJssEnv, builtin_instance
    from oracle import OracleEnv
    inst = builtin_instance("ta01")
    B, seed = 4096, 9
    env = BatchedJssEnv(inst, batch=B, seed=seed, _backend=hip)
    env.reset()
    env.rollout("random", n_iter=400, autoreset=False)   # every episode ends well before 400 steps
    done = env.done.cpu().numpy()
    assert done.all()
    sol = env.solution.cpu().numpy()
    assert (sol >= 0).all()
    todo = env.todo_time_step_job.cpu().numpy()
    assert (todo == inst.machines).all()
    cnt = env.counters.cpu().numpy()
    mk = env.makespan.cpu().numpy()
    assert (cnt[:, 1] == 1).all() and (cnt[:, 2] == mk).all() and (mk == env.clock.cpu().numpy()).all()
    # reward identity: sum of reward numerators = 2*sum_op - M*makespan
    assert (cnt[:, 3] == 2 * inst.sum_op - inst.machines * mk).all()
    # schedule validity: ops of a job in order, no overlap on a machine
    dur, mach = inst.duration, inst.machine
    end = sol + dur[None]
    assert (sol[:, :, 1:] >= end[:, :, :-1]).all()
    assert (end.max(axis=(1, 2)) == mk).all()
    for b in range(0, B, 97):
        for m in range(inst.machines):
            sel = mach == m
            s, e = sol[b][sel], end[b][sel]
            order = np.argsort(s)
            assert (s[order][1:] >= e[order][:-1]).all()
    assert int(env.err.max().item()) == 0
    # oracle comparison without auto-restart
    for i in range(0, B, 173):
        orc = OracleEnv(inst, strict=True)
        orc.reset()
        st = 0
        while orc.nb_legal_actions:
            orc.step(orc.policy("random", seed=seed, env_id=i, episode=1, step=st))
            st += 1
        assert orc.current_time_step == mk[i] and (orc.solution == sol[i]).all() and st == cnt[i, 0]


def test_config3_ta41_spt_batch16384(hip):
    """BASELINE config 3: ta41, SPT, every env 600 steps and makespan 2499 (G3)."""
    from jssenv_amd import BatchedJssEnv, builtin_instance
    inst = builtin_instance("ta41")
    env = BatchedJssEnv(inst, batch=16384, _backend=hip)
    env.reset()
    env.rollout("SPT", n_iter=1000, autoreset=False)
    assert (env.makespan.cpu().numpy() == 2499).all()
    assert (env.counters.cpu().numpy()[:, 0] == 600).all()
    sol = env.solution.cpu().numpy()
    assert (sol == sol[0]).all()


def test_config5_mixed_ta01_ta80_padded_batch32768(hip):
    """BASELINE config 5 at full size (32 768 envs): properties on every env, the oracle on one env per instance."""
    P.case_config5_mixed(hip, batch=32768)


def test_config4_synthetic_50x20_batch8192(hip):
    """BASELINE config 4, one GPU's share at full size (8 192 envs, one table per env): properties on every env,
    the oracle on a 64-env sample."""
    P.case_config4_synthetic(hip, batch=8192)


def test_dispatching_module(hip):
    P.case_dispatching_seeded(hip)
    P.case_dispatching_deterministic(hip, insts=("ta01", "ta41"))


def test_edge_shapes(hip):
    P.case_edge_shapes(hip, steps=300, batch_per_shape=5)


def test_ragged_batch_with_a_64_job_env(hip):
    P.case_ragged_j64_nope_flag(hip, steps=200)


def test_vector_env_features(hip):
    P.case_vector_env_features(hip)


def test_vector_facade_over_shape_classes(hip):
    P.case_vector_facade_by_shape(hip)


def test_ragged_population_in_padded_tensors_by_shape_class(hip):
    P.case_by_shape_padded(hip, n_envs=700, iters=300)


def test_by_shape_padded_config5_full_size_every_env_equals_the_oracle(hip_auto):
    """BASELINE config 5 in ONE set of padded tensors (100 x 20 rows), envs ordered by shape class, at the benchmarked 32 768:
    class-specialised bodies of the fused grid on the padded rows, every env against the C oracle."""
    P.case_by_shape_padded(hip_auto, n_envs=32768, iters=150, taillard=True)


def test_multi_entry_points_equal_the_single_set_calls(hip):
    P.case_multi_entry_points(hip)


def test_bucketed_equals_padded(hip):
    P.case_bucketed_equals_padded(hip, n_envs=400, n_iter=900)


@pytest.mark.parametrize("launch", ["grid", "streams"])
def test_bucketed_full_size_every_env_equals_the_oracle(hip_auto, launch):
    """BASELINE config 5 without padding at the benchmarked 32 768 envs: ONE grid per step over the four shape classes
    (jss_multi_kernel<kRollout1>), every env of every class against the C oracle; then the un-fused calls (jss_multi_policy,
    jss_multi_step with next-step auto-reset) on top.  launch="streams" = round 3's one-launch-per-class form."""
    P.case_bucketed_every_env_vs_oracle(hip_auto, n_envs=32768, iters=160, launch=launch, unfused_tail=12 if launch == "grid" else 0)


def test_rules_with_exploration(hip):
    """the rules' 10 % NOPE exploration (dispatching.py:113) drawn from the counter RNG on the device"""
    env, orcs = P.case_batch_lockstep(hip, ["ta01", "ta21"], batch=4, n_steps=1200, kind="SPT", check_every=9, explore=0.1)
    env, orcs = P.case_batch_lockstep(hip, ["ta71"], batch=2, n_steps=1200, kind="FIFO", check_every=9, explore=0.25)


def test_largest_shape_per_env_tables(hip):
    """128 jobs x 64 machines, one table per env (read from global memory: the workgroup's LDS holds only the four
    waves' observation images, 4 x 3.6 KB): the largest shape the ABI admits must launch and agree."""
    rng = np.random.default_rng(0)
    from jssenv_amd import BatchedJssEnv
    from oracle import OracleEnv
    insts = [P.random_instance(rng, 128, 64, max_dur=999) for _ in range(3)]
    env = BatchedJssEnv(insts, seed=1, _backend=hip)
    env.reset()
    env.rollout("random", n_iter=400)
    for i, inst in enumerate(insts):
        o = OracleEnv(inst, strict=True)
        o.reset()
        o.rollout("random", 1, i, 400, episode=1)
        P.assert_matches_oracle(env.host_state(i), o, f"128x64 env {i}")


def test_vector_facade(hip):
    P.case_vector_facade(hip)


def test_headline_batch_65536(hip):
    """The benchmarked configuration (ta01, 65 536 envs, random masked policy, one launch per step with
    auto-restart): oracle agreement on a sample and size-independent properties on every env."""
    from jssenv_amd import BatchedJssEnv, builtin_instance
    from oracle import OracleEnv
    inst = builtin_instance("ta01")
    B, seed, iters = 65536, 17, 310
    env = BatchedJssEnv(inst, batch=B, seed=seed, env_id_base=5_000_000_000, _backend=hip)   # ids beyond 32 bits
    env.reset()
    for _ in range(iters):
        env.rollout("random", n_iter=1)
    cnt = env.counters.cpu().numpy()
    hdr = env.env_header.cpu().numpy()
    assert int(env.err.max().item()) == 0
    assert (cnt[:, 0] + hdr[:, 1] - 1 == iters).all()          # every iteration is a step or an auto-reset
    assert (cnt[:, 1] >= 1).all() and (cnt[:, 1] == hdr[:, 1] - 1 + (env.done.cpu().numpy() != 0)).all()
    obs = env.real_obs.cpu().numpy()
    assert obs.min() >= 0.0 and obs.max() <= 1.0 and np.isfinite(obs).all()
    mask = env.action_mask.cpu().numpy()
    assert (obs[:, :, 0] == mask[:, :15]).all()
    for i in list(range(0, B, 4099)) + [B - 1]:
        o = OracleEnv(inst, strict=True)
        o.reset()
        r = o.rollout("random", seed, 5_000_000_000 + i, iters, episode=1)
        P.assert_matches_oracle(env.host_state(i), o, f"headline env {i}")
        assert cnt[i, 0] == r["steps"] and cnt[i, 1] == r["episodes"] and cnt[i, 2] == r["makespan_sum"]


def test_instance_resampling(hip):
    P.case_instance_resampling(hip)


def test_rollout_steps_equals_rollout(hip):
    P.case_rollout_steps(hip, batch=5000, steps=40, n_sub=4)


def test_nope_fuzz_tiny_instances(hip):
    P.case_nope_fuzz(hip, batch=64, steps=200)


def test_checkpoint_file_resume(hip, tmp_path):
    """N3 on the GPU: packed batch file -> env, checkpoint FILE -> bit-exact resume."""
    P.case_file_round_trips(hip, tmp_path)


def test_every_env_at_full_size_equals_the_cpu_twin(hip):
    """Configs 2-5 and the headline batch, every env, every tensor, bit for bit against libjss_cpu.so."""
    P.case_hip_equals_twin_full_size(hip)


@pytest.mark.parametrize("cfg", range(len(P.FULL_SIZE_CONFIGS)), ids=[c[0].split(":")[0] for c in P.FULL_SIZE_CONFIGS])
def test_every_env_at_full_size_equals_the_oracle(hip, cfg):
    """Configs 2-5 and the headline batch: EVERY env against the C oracle's batch driver (no twin in between)."""
    label, kw, kind, iters, explore = P.FULL_SIZE_CONFIGS[cfg]
    P.case_every_env_vs_oracle(hip, label, kw(), kind, iters, explore)


def test_dispatching_fused_on_device(hip):
    P.case_dispatching_on_device(hip, num_episodes=40)


def test_compact_records_equal_full_records(hip):
    P.case_compact_equals_full(hip, insts=("ta01", "ta41", "ta51", "ta71"), batch=70, n_iter=700)   # G16, G32, wave, wave x2


def test_compact_records_at_the_limits(hip):
    P.case_compact_limits(hip)


# ---------------------------------------------------------------------------------------------------------
# the launch forms bench.py times: the kRollout1 instantiations <16,5,2> <32,5,2> <1,5,1> <2,5,1>
# at full size, every env against the oracle, through each benchmarked form
# ---------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip_auto():
    from jssenv_amd.env import HipBackend
    return HipBackend("cuda:0")


_FORMS = [(cfg, "free") for cfg in range(len(P.FULL_SIZE_CONFIGS))] + [(0, "graph"), (2, "graph"), (4, "graph"), (4, "fork_join"),
                                                                      (3, "per_launch")] + \
         [(cfg, "session") for cfg in (0, 1, 2, 4)] + [(0, "session_lockstep"), (2, "session_lockstep")] + \
         [(cfg, "steps") for cfg in range(len(P.FULL_SIZE_CONFIGS))]


@pytest.mark.parametrize("cfg,form", _FORMS, ids=[f"{P.FULL_SIZE_CONFIGS[c][0].split(':')[0]}-{f}" for c, f in _FORMS])
def test_benchmarked_launch_forms_every_env_equals_the_oracle(hip_auto, cfg, form):
    """free = bind_rollout_steps(caller_orders_streams=True) between device-wide synchronizes (bench.py's timed window);
    graph = a captured hipGraph of K x jss_rollout(n_iter = 1), replayed; fork_join = the library-ordered sub-batch
    form; per_launch = K eager launches.  All drive the kRollout1 kernels -- the benchmarked instantiations."""
    label, kw, kind, iters, explore = P.FULL_SIZE_CONFIGS[cfg]
    P.case_every_env_vs_oracle(hip_auto, label, kw(), kind, iters, explore, form=form, n_sub=3 if cfg == 4 else 2)


def test_step_graph_replay_equals_eager_steps(hip_auto):
    P.case_step_graph_replay(hip_auto)
    P.case_step_graph_replay(hip_auto, inst="ta41", batch=2048, K=25, warm=30)


def test_render_rows_from_device_solution(hip_auto):
    P.case_render_rows_from_device_solution(hip_auto)


def test_two_envs_two_threads_two_streams():
    from jssenv_amd.env import HipBackend
    P.case_two_streams_two_threads(lambda: HipBackend("cuda:0"))


def test_steps_equal_repeated_step(hip):
    """jss_steps: K x jss_step per launch with the actions given up front, every step recorded."""
    P.case_steps(hip, dict(instances="ta01", batch=700), K=80, warm=180)                            # crosses episode ends
    P.case_steps(hip, dict(instances=["ta01", "ta31", "ta51", "ta71"], batch=21), K=50, kind="SPT", warm=5)       # one launch per shape class
    P.case_steps(hip, dict(instances=["ta01", "ta31", "ta51", "ta71"], batch=21, order="interleaved"), K=50, kind="SPT", warm=5)
    P.case_steps(hip, dict(instances=["ta02", "ta03", "ta04"], batch=130), K=40)                    # env -> instance map


def test_step_session_equals_repeated_step(hip):
    """jss_session_*: the state resident in a kernel that lives across steps; outputs after every wait and the state
    after close equal K x jss_step.  Env sets in registers (one per wavefront) and parked in LDS (2, 4 per wavefront)."""
    P.case_session(hip, dict(instances="ta01", batch=4096), K=60, warm=200)
    P.case_session(hip, dict(instances="ta01", batch=1000), K=40, warm=20, slots=4, depth=3)
    P.case_session(hip, dict(instances="ta41", batch=700), K=40, kind="SPT", warm=580, slots=2)
    P.case_session(hip, dict(instances=["ta02", "ta03", "ta04"], batch=300), K=30, slots=2)
    P.case_session(hip, dict(instances=["ta01", "ta31", "ta51", "ta71"], batch=40), K=40, kind="FIFO", slots=2)
    P.case_session(hip, dict(instances=["ta61"], batch=100), K=30, depth=1, pattern=(1,))


def test_step_session_fits_the_baseline_batches(hip_auto):
    """Residency: configs 2, 3 and config 4's share open as ONE round of resident workgroups (1, 2, 2 env sets per
    wavefront) with room left for the caller's kernels; the whole headline batch needs 8."""
    from jssenv_amd import BatchedJssEnv
    from jssenv_amd import instances as I
    for kw, want in ((dict(instances="ta01", batch=4096), 1), (dict(instances="ta41", batch=16384), 2),
                     (dict(instances=I.synthetic_packed(8192, 50, 20)), 2), (dict(instances="ta01", batch=65536), 8)):
        env = BatchedJssEnv(_backend=hip_auto, **kw)
        env.reset()
        with env.session(depth=2, timeout_ms=3000) as s:
            s.step(env.backend.torch.zeros(env.batch, dtype=env.backend.torch.int32, device=env.backend.device))
        st = s.host_status()
        assert st["env_sets_per_wavefront"] == want and st["session_timeouts"] == 0 and st["wait_timeouts"] == 0, (kw.get("batch"), st)
    # what does not fit is refused up front (JSS_E_RESIDENT), not started and left to time out: config 5's padded batch --
    # 32 768 envs, one wavefront each, two jobs per lane -- would need more than eight envs per wavefront
    big = BatchedJssEnv([I.builtin_instance(f"ta{k:02d}") for k in range(1, 81)], batch=32768, _backend=hip_auto)
    big.reset()
    with pytest.raises(RuntimeError, match="does not fit"):
        big.session(depth=2)
    assert big._session is None
    big.rollout("random", n_iter=3)                      # the batch is untouched and usable


def test_step_session_gives_up_instead_of_hanging(hip_auto):
    """Nothing is posted: every wavefront of the resident kernel gives up after timeout_ms, stores its (unchanged) state
    and leaves; close() reports it."""
    import time
    from jssenv_amd import BatchedJssEnv
    env = BatchedJssEnv("ta01", batch=2000, _backend=hip_auto)
    env.reset()
    env.rollout("random", n_iter=33)
    env.synchronize()
    before = P._state_snapshot(env)
    s = env.session(depth=2, timeout_ms=50)
    time.sleep(0.3)
    with pytest.raises(RuntimeError, match="timed out"):
        s.close()
    st = s.host_status()
    assert st["session_timeouts"] > 0 and st["session_timeouts"] == st["wavefronts_exited"]
    after = P._state_snapshot(env)
    for name in before:
        assert np.array_equal(before[name], after[name]), name


def test_critical_ratio_any_due_date_factor_on_device(hip):
    P.case_cr_any_factor(hip)


def test_critical_ratio_due_date_factor_on_device(hip):
    P.case_cr_due_date_factor(hip)


def test_steps_and_session_edges(hip):
    P.case_steps_and_session_edges(hip)


def test_medium_records_equal_full_records(hip):
    P.case_medium_equals_full(hip, batch=700, n_iter=900)


def test_medium_records_at_the_limits(hip):
    P.case_medium_limits(hip)


def test_policy_step_steps_equals_the_loop(hip):
    P.case_policy_step_steps(hip, batch=5000, steps=30)
    P.case_policy_step_steps(hip, batch=65536, steps=12, warm=60)       # the benchmarked batch (bench.py policy_then_step_pipelined)


def test_integration_level2_stub_as_printed():
    """INTEGRATION.md's Level-2 ctypes stub, cut out of the document and executed against libjss_hip.so on the GPU."""
    from jssenv_amd import _abi
    P.case_integration_level2_stub(_abi.library_path(), on_gpu=True)


@pytest.mark.parametrize("records", [None, "medium"])
def test_two_envs_per_wavefront_small_batches(records):
    """jss_kernel_two on request (JSS_KERNEL_TWO_ENVS_PER_WAVE) on small, odd, ragged batches, full and medium job records;
    at the benchmarked sizes the form is the default and FULL_SIZE_CONFIGS[2, 6, 7] hold every env of it to the oracle."""
    from jssenv_amd.env import HipBackend
    two, one = HipBackend("cuda:0"), HipBackend("cuda:0")
    two.default_kernel, one.default_kernel = "wave-2env", "wave-1env"
    two.default_records = one.default_records = records
    P.case_two_envs_per_wavefront(two, one, steps=200, n_envs=131)


@pytest.mark.parametrize("cfg", [2, 6, 7])
def test_one_env_per_wavefront_form_at_full_size(hip_auto, cfg):
    """The form the one-wavefront-per-env launches had before round 6 (and still have below 6 144 envs per launch), at the
    benchmarked sizes, every env against the oracle -- the A/B partner of the default."""
    label, kw, kind, iters, explore = P.FULL_SIZE_CONFIGS[cfg]
    P.case_every_env_vs_oracle(hip_auto, label + " [one env per wavefront]", dict(kw(), kernel="auto-1env"), kind, iters, explore=explore, form="free")


def test_fuzz_mixed_calls_against_the_oracle():
    """Random populations (1..128 jobs, 2..64 machines), deals, kernel forms and call mixes on the GPU, every env vs the oracle."""
    from jssenv_amd.env import HipBackend

    def factory(kernel):
        be = HipBackend("cuda:0")
        be.default_kernel = kernel
        return be
    P.case_fuzz_mixed_calls(factory, rounds=150, max_batch=700, max_iters=260)
