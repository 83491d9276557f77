"""CPU-only: the gymnasium registration / spaces path (JSSEnv/__init__.py:6-9, jss_env.py:97,112-119) executed
against a stand-in gymnasium (the package is absent from this image), and the on-disk formats of SURVEY row N3."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import parity_cases as P
from jssenv_amd import instances as I

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_GYM_SCRIPT = textwrap.dedent('''
    import importlib, sys, types
    # a stand-in for the handful of gymnasium names the registration path touches
    gym = types.ModuleType("gymnasium"); spaces = types.ModuleType("gymnasium.spaces")
    envs = types.ModuleType("gymnasium.envs"); registration = types.ModuleType("gymnasium.envs.registration")
    table = {}
    class Discrete:
        def __init__(self, n): self.n = n
    class Box:
        def __init__(self, low, high, shape=None, dtype=None): self.low, self.high, self.shape, self.dtype = low, high, shape, dtype
    class Dict:
        def __init__(self, d): self.spaces = dict(d)
    def register(id, entry_point=None, **kw): table[id] = entry_point
    def make(id, **kw):
        mod, cls = table[id].split(":")
        return getattr(importlib.import_module(mod), cls)(**kw)
    class Env:                      # gymnasium.Env / gymnasium.vector.VectorEnv: the package subclasses them when present
        pass
    class VectorEnv:
        pass
    vector = types.ModuleType("gymnasium.vector"); vector.VectorEnv = VectorEnv
    gym.spaces, gym.envs, gym.make, gym.Env, gym.vector = spaces, envs, make, Env, vector
    spaces.Discrete, spaces.Box, spaces.Dict = Discrete, Box, Dict
    registration.register = register; envs.registration = registration
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs,
                        "gymnasium.envs.registration": registration, "gymnasium.vector": vector})
    sys.path.insert(0, %r)
    import gymnasium
    import jssenv_amd
    assert table == {"jss-v1": "jssenv_amd.facade:JssEnv"}, table
    env = gymnasium.make("jss-v1", env_config={"instance_path": "ta01"}, device="cpu")
    assert type(env) is jssenv_amd.JssEnv
    assert isinstance(env, gymnasium.Env)          # the reference: class JssEnv(gym.Env), jss_env.py:14
    assert isinstance(env.action_space, Discrete) and env.action_space.n == 16
    d = env.observation_space.spaces
    assert set(d) == {"action_mask", "real_obs"}
    assert d["action_mask"].shape == (16,) and (d["action_mask"].low, d["action_mask"].high) == (0, 1)
    assert d["real_obs"].shape == (15, 7) and (d["real_obs"].low, d["real_obs"].high) == (0.0, 1.0)
    obs = env.reset()
    assert obs["real_obs"].shape == (15, 7) and obs["action_mask"].shape == (16,)
    obs, reward, done, truncated, info = env.step(3)
    assert truncated is False and info == {} and not done and obs["action_mask"][3] == False
    # default instance (env_config=None) is ta80, as at jss_env.py:35-38
    assert gymnasium.make("jss-v1", device="cpu").jobs == 100
    # the vector facade gets its spaces from the same package
    from jssenv_amd.vector import JssVectorEnv
    v = JssVectorEnv("ta01", num_envs=3, device="cpu")
    assert v.single_action_space.n == 16 and v.single_observation_space.spaces["real_obs"].shape == (15, 7)
    assert isinstance(v, gymnasium.vector.VectorEnv)
    env.close(); v.close()
    print("GYM-PATH-OK")
''')


def test_gymnasium_registration_and_spaces():
    out = subprocess.run([sys.executable, "-c", _GYM_SCRIPT % ROOT], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "GYM-PATH-OK" in out.stdout, out.stdout + out.stderr


def test_real_gymnasium_in_the_loop():
    """With the real package (the reference's CI pins gymnasium==0.29.1): gym.make, isinstance, spaces, an episode
    through the registered id.  Skipped where gymnasium is not installed (this image)."""
    gymnasium = pytest.importorskip("gymnasium")
    if not hasattr(gymnasium, "__version__"):
        pytest.skip("a stand-in gymnasium module is installed, not the real package")
    import jssenv_amd
    env = gymnasium.make("jss-v1", env_config={"instance_path": "ta01"}, device="cpu", disable_env_checker=True)
    base = env.unwrapped
    assert isinstance(base, jssenv_amd.JssEnv) and isinstance(base, gymnasium.Env)
    assert base.action_space.n == 16 and base.observation_space["real_obs"].shape == (15, 7)
    obs = base.reset()
    done, steps = False, 0
    while not done:
        a = int(np.flatnonzero(obs["action_mask"])[0])
        obs, reward, done, truncated, info = base.step(a)
        steps += 1
    assert steps >= 225 and base.last_time_step == base.current_time_step
    from jssenv_amd.vector import JssVectorEnv
    v = JssVectorEnv("ta01", num_envs=3, device="cpu")
    assert isinstance(v, gymnasium.vector.VectorEnv) and v.action_space is not None
    v.close()


def test_packed_batch_and_checkpoint_files(tmp_path):
    from jssenv_amd.env import CpuBackend
    P.case_file_round_trips(CpuBackend(), tmp_path)


def test_load_batch_rejects_garbage(tmp_path):
    pk = I.pack_batch([I.builtin_instance("ta01")])
    p = tmp_path / "b.npz"
    I.save_batch(p, pk)
    bad = pk.inst.copy()
    bad[0, 4] += 1                                # sum_op no longer matches the op table
    with open(tmp_path / "bad.npz", "wb") as fh:
        np.savez(fh, format=np.int32(1), ops=pk.ops, inst=bad)
    with pytest.raises(ValueError):
        I.load_batch(tmp_path / "bad.npz")
    with open(tmp_path / "v2.npz", "wb") as fh:
        np.savez(fh, format=np.int32(2), ops=pk.ops, inst=pk.inst)
    with pytest.raises(ValueError):
        I.load_batch(tmp_path / "v2.npz")
    assert np.array_equal(I.load_batch(p).ops, pk.ops)


def test_instance_text_file_round_trip(tmp_path):
    """The reference's only on-disk format (jss_env.py:72-88): write, re-read, and run from a path."""
    from jssenv_amd import make
    ta = I.builtin_instance("ta03")
    p = tmp_path / "my_instance"
    p.write_text(ta.to_text())
    again = I.load_instance_file(p)
    assert (again.packed() == ta.packed()).all()
    env = make("jss-v1", env_config={"instance_path": str(p)}, device="cpu")
    assert (env.jobs, env.machines, env.max_time_op) == (15, 15, ta.max_time_op)


def test_render_gantt_end_to_end():
    """render() (jss_env.py:655-693) through plotly: None before anything is scheduled, then a Figure with one bar
    per scheduled operation, machines as the colour index, start/finish = solution + duration."""
    pytest.importorskip("plotly")
    pytest.importorskip("pandas")
    from jssenv_amd import make
    from jssenv_amd.dispatching import get_rule
    from jssenv_amd.render import gantt_rows
    env = make("jss-v1", env_config={"instance_path": "ta01"}, device="cpu")
    env.reset()
    assert env.render() is None
    rule = get_rule("SPT")
    np.random.seed(0)
    for _ in range(40):
        env.step(rule(env))
    fig = env.render()
    sol = env.solution
    n_scheduled = int((sol >= 0).sum())
    assert n_scheduled > 0 and type(fig).__name__ == "Figure"
    rows = gantt_rows(sol, env.instance, env.start_timestamp)
    assert len(rows) == n_scheduled
    # plotly's create_gantt draws one filled scatter trace per machine: 5 points per bar, the bars of a trace
    # separated by sharing the closing point (5 n - 1 points for n bars)
    bars = [t for t in fig.data if getattr(t, "fill", None) == "toself"]
    assert len(bars) == len({r["Resource"] for r in rows})
    assert sum((len(t.x) + 1) // 5 for t in bars) == n_scheduled
    assert fig.layout.yaxis.autorange == "reversed"
    for r in rows[:5]:
        job = int(r["Task"].split()[1])
        k = [x for x in rows if x["Task"] == r["Task"]].index(r)
        assert (r["Finish"] - r["Start"]).total_seconds() == env.instance.duration[job][k]
        assert r["Resource"] == f"Machine {int(env.instance.machine[job][k])}"


def test_schedule_gif_is_written_without_imageio(tmp_path):
    """The README's GIF recipe (README.md:158-197) as one call: frames of the growing schedule, written by imageio when
    it exists and by Pillow otherwise; the plotly raster is used when kaleido exists, a Pillow drawing of the same
    rows otherwise."""
    pytest.importorskip("PIL")
    from PIL import Image
    from jssenv_amd import make
    from jssenv_amd.dispatching import get_rule
    from jssenv_amd.render import gantt_frame, record_episode_gif
    env = make("jss-v1", env_config={"instance_path": "ta01"}, device="cpu")
    env.reset()
    assert gantt_frame(env) is None
    rule = get_rule("MOR")
    np.random.seed(1)
    out = tmp_path / "ta01.gif"
    n, makespan = record_episode_gif(env, rule, out, every=25, size=(480, 270))
    assert makespan == env.current_time_step and n >= 225 // 25
    with Image.open(out) as im:
        assert im.format == "GIF" and getattr(im, "n_frames", 1) == n and im.size == (480, 270)
    frame = gantt_frame(env, size=(480, 270), prefer_plotly=False)
    assert frame.shape == (270, 480, 3) and frame.dtype == np.uint8 and (frame != 255).any()


@pytest.mark.refcheck
def test_render_rows_equal_the_reference():
    """Build container only: the rows handed to create_gantt equal the reference's render() dataframe for the same
    schedule (jss_env.py:666-677)."""
    tools = os.path.join(ROOT, "tools")
    sys.path.insert(0, tools)
    import refload
    if not refload.reference_available():
        pytest.skip("reference tree absent")
    pytest.importorskip("plotly")
    from jssenv_amd import make
    from jssenv_amd.render import gantt_rows
    Ref, _ = refload.load_reference()
    ref = Ref({"instance_path": refload.reference_instance_path("ta01")})
    env = make("jss-v1", env_config={"instance_path": "ta01"}, device="cpu")
    ref.reset()
    env.reset()
    rng = np.random.default_rng(1)
    for _ in range(60):
        a = int(rng.choice(np.flatnonzero(ref.legal_actions)))
        ref.step(a)
        env.step(a)
    fig = ref.render()
    mine = gantt_rows(env.solution, env.instance, ref.start_timestamp)
    import datetime
    want = []
    for job in range(ref.jobs):
        for k in range(ref.machines):
            if ref.solution[job][k] == -1:
                break
            s0 = ref.start_timestamp + ref.solution[job][k]
            want.append({"Task": f"Job {job}", "Start": datetime.datetime.fromtimestamp(s0),
                         "Finish": datetime.datetime.fromtimestamp(s0 + ref.instance_matrix[job][k][1]),
                         "Resource": f"Machine {ref.instance_matrix[job][k][0]}"})
    assert mine == want and fig is not None
    assert len(env.render().data) == len(fig.data)


def _facade_attribute_walk(ref, env, rng, episodes=2, with_advance=False):
    """Random masked episodes (forced NOPEs included), the facade in lock step with `ref` (anything with the reference's
    attributes): next_jobs / next_time_step equal after every call."""
    for _ in range(episodes):
        ref.reset()
        env.reset()
        assert env.next_jobs == [] and list(ref.next_jobs) == []
        done, n = False, 0
        while not done:
            legal = np.flatnonzero(ref.legal_actions)
            a = int(rng.choice(legal))
            if n % 11 == 5 and len(ref.next_time_step) > 0 and ref.legal_actions[:-1].sum() > 0:
                a = ref.jobs                                   # a NOPE whatever the mask says (queues nothing, pops events)
            _, _, done, _, _ = ref.step(a)
            env.step(a)
            n += 1
            assert env.next_time_step == [int(t) for t in ref.next_time_step], n
            assert env.next_jobs == [int(j) for j in ref.next_jobs], (n, env.next_jobs, list(ref.next_jobs))
            if with_advance and n % 17 == 3 and len(ref.next_time_step) > 0:
                assert env.increase_time_step() == ref.increase_time_step()
                assert env.next_jobs == [int(j) for j in ref.next_jobs]
                done = ref.nb_legal_actions == 0 if hasattr(ref, "nb_legal_actions") else done
                if done:
                    break


@pytest.mark.parametrize("name", ["ta01", "ta31", "dmu16"])
def test_facade_next_jobs_colors_start_timestamp(name):
    """The three public attributes of the reference's env that no test of its reads (jss_env.py:56 / :70 / :99): next_jobs
    -- the job that queued each pending event -- against the NumPy restatement (pinned to the reference's golden traces,
    and it keeps the list the reference's way, :453 / :518) on random episodes; colors and start_timestamp as render() uses them."""
    import time
    from jssenv_amd import builtin_instance, make
    from oracle.np_restatement import NumpyJssEnv
    t0 = time.time()
    env = make("jss-v1", env_config={"instance_path": name}, device="cpu")
    assert t0 - 1 <= env.start_timestamp <= time.time() + 1
    assert len(env.colors) == env.machines and all(len(c) == 3 and all(0.0 <= x <= 1.0 for x in c) for c in env.colors)
    _facade_attribute_walk(NumpyJssEnv(builtin_instance(name)), env, np.random.default_rng(5), with_advance=True)
    # a caller may set them like on the reference's env: render() takes its time origin and colours from there
    env.start_timestamp, env.colors = 1000.0, [(0.1, 0.2, 0.3)] * env.machines
    from jssenv_amd.render import gantt_rows
    import datetime
    rows = gantt_rows(env.solution, env.instance, env.start_timestamp)
    assert rows and min(r["Start"] for r in rows) == datetime.datetime.fromtimestamp(1000.0)
    # after a device-side episode (the rule runs on the device: no per-call log) the list is still well formed
    from jssenv_amd.dispatching import get_rule
    get_rule("SPT").run_episode(env, device_rng=True, seed=1)
    assert env.next_jobs == [] and env.next_time_step == []


@pytest.mark.refcheck
def test_facade_next_jobs_equal_the_reference():
    """Build container only: the same walk against the live reference's own lists."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import refload
    if not refload.reference_available():
        pytest.skip("reference tree absent")
    from jssenv_amd import make
    Ref, _ = refload.load_reference()
    for name in ("ta01", "ta41"):
        ref = Ref({"instance_path": refload.reference_instance_path(name)})
        env = make("jss-v1", env_config={"instance_path": name}, device="cpu")
        _facade_attribute_walk(ref, env, np.random.default_rng(9), episodes=2)
        assert hasattr(ref, "colors") and hasattr(ref, "start_timestamp") and len(env.colors) == len(ref.colors)
