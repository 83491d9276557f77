"""Tooling (build container only): make /root/reference importable.

The reference env subclasses ``gymnasium.Env`` and registers itself with
gymnasium (JSSEnv/__init__.py:3-9, JSSEnv/envs/jss_env.py:8,14).  gymnasium is
not installed in this image, and the simulator arithmetic never touches it, so
an in-memory stand-in for the handful of names the import needs is enough.

Nothing here ships to the GPU box and nothing from the reference is copied:
this module is only used by ``tools/make_golden.py`` and by the
``refcheck`` tests that validate ``oracle/`` against the live reference when
``/root/reference`` happens to exist.
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("JSS_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "JSSEnv", "envs", "jss_env.py"))


def _install_gymnasium_stand_in():
    if "gymnasium" in sys.modules:
        return
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")
    envs = types.ModuleType("gymnasium.envs")
    registration = types.ModuleType("gymnasium.envs.registration")
    table = {}

    class Env:  # base class only; no behaviour is inherited by the reference
        pass

    class _Space:
        def __init__(self, *args, **kwargs):
            self.args, self.kwargs = args, kwargs

    def register(id, entry_point=None, **kwargs):  # noqa: A002
        table[id] = entry_point

    def make(id, **kwargs):  # noqa: A002
        module_name, class_name = table[id].split(":")
        return getattr(importlib.import_module(module_name), class_name)(**kwargs)

    gym.Env, gym.spaces, gym.envs, gym.make = Env, spaces, envs, make
    spaces.Discrete = spaces.Box = spaces.Dict = _Space
    registration.register = register
    envs.registration = registration
    sys.modules.update({
        "gymnasium": gym,
        "gymnasium.spaces": spaces,
        "gymnasium.envs": envs,
        "gymnasium.envs.registration": registration,
    })


def load_reference():
    """Return (JssEnv class, dispatching module) of the live reference."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    _install_gymnasium_stand_in()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write into the read-only reference tree
    env_mod = importlib.import_module("JSSEnv.envs.jss_env")
    disp_mod = importlib.import_module("JSSEnv.dispatching")
    return env_mod.JssEnv, disp_mod


def reference_instance_path(name: str) -> str:
    return os.path.join(REFERENCE_ROOT, "JSSEnv", "envs", "instances", name)
