"""Drop-in check (build container only, marker `refcheck`): the reference's OWN test files --
/root/reference/tests/test_state.py, test_solutions.py, test_dispatching.py, read from the reference tree at run
time, unmodified, nothing copied -- executed against THIS package: `gymnasium.make('jss-v1', ...)` resolves to
`jssenv_amd.JssEnv` on the host-core twin (no GPU here) and `import JSSEnv` / `from JSSEnv.dispatching import ...` to
`jssenv_amd` / `jssenv_amd.dispatching`.  The instance paths those tests build point into the reference tree and are
parsed by jssenv_amd.instances.  test_rendering.py (Gantt/GIF) is not part of the hot path and is
not run."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/tests"

_SCRIPT = textwrap.dedent('''
    import importlib, importlib.util, sys, types, unittest
    sys.path.insert(0, %(root)r)
    sys.dont_write_bytecode = True                      # never write into the read-only reference tree
    # gymnasium stand-in (the package is absent from this image): the names the registration path touches;
    # make() looks the id up in the table jssenv_amd's own register() call filled, and adds device="cpu"
    gym = types.ModuleType("gymnasium"); spaces = types.ModuleType("gymnasium.spaces")
    envs = types.ModuleType("gymnasium.envs"); registration = types.ModuleType("gymnasium.envs.registration")
    table = {}
    class _Space:
        def __init__(self, *a, **k): self.args, self.kwargs = a, k
    def register(id, entry_point=None, **kw): table[id] = entry_point
    def make(id, **kw):
        mod, cls = table[id].split(":")
        return getattr(importlib.import_module(mod), cls)(device="cpu", **kw)
    gym.spaces, gym.envs, gym.make = spaces, envs, make
    spaces.Discrete = spaces.Box = spaces.Dict = _Space
    registration.register = register; envs.registration = registration
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": spaces, "gymnasium.envs": envs,
                        "gymnasium.envs.registration": registration})
    import jssenv_amd, jssenv_amd.dispatching
    assert table == {"jss-v1": "jssenv_amd.facade:JssEnv"}, table
    # the reference's package name resolves to this package
    sys.modules["JSSEnv"] = jssenv_amd
    sys.modules["JSSEnv.dispatching"] = jssenv_amd.dispatching
    suite = unittest.TestSuite()
    for name in %(files)r:
        spec = importlib.util.spec_from_file_location("ref_" + name, %(ref)r + "/" + name + ".py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        suite.addTests(unittest.defaultTestLoader.loadTestsFromModule(mod))
    res = unittest.TextTestRunner(verbosity=1, stream=sys.stdout).run(suite)
    print("REF-SUITE ran=%%d failures=%%d errors=%%d" %% (res.testsRun, len(res.failures), len(res.errors)))
    sys.exit(0 if res.wasSuccessful() else 1)
''')


@pytest.mark.refcheck
def test_reference_test_files_pass_against_this_package():
    files = ["test_state", "test_solutions", "test_dispatching"]
    for f in files:
        assert os.path.isfile(os.path.join(REF_TESTS, f + ".py"))
    script = _SCRIPT % {"root": ROOT, "files": files, "ref": REF_TESTS}
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "REF-SUITE ran=" in out.stdout and "failures=0 errors=0" in out.stdout, tail
    ran = int(out.stdout.split("REF-SUITE ran=")[1].split()[0])
    assert ran >= 20, tail                               # 1 state + 12 published schedules + 7 dispatching tests
