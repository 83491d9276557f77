"""jssenv_amd -- MI355X-native batched Job-Shop-Scheduling environment.

Drop-in for the hot path of prosysscience/JSSEnv (``JssEnv.reset()/step()`` and
what they call): ``make('jss-v1', env_config=...)`` / ``JssEnv`` keep the
reference's single-env API, ``BatchedJssEnv`` runs thousands of envs per GPU
with hand-written HIP kernels (``csrc/jss_kernels.hip``) behind a C ABI
(``include/jss_hip.h``).  ``device="cpu"`` selects -- explicitly, never as a
fallback -- the host-core twin ``libjss_cpu.so`` with the same C ABI.
"""
from .instances import (Instance, PackedBatch, available_instances, builtin_instance, load_batch,  # noqa: F401
                        load_instance_file, pack_batch, parse_instance_text, save_batch, synthetic_batch,
                        synthetic_packed, taillard_instance)
from .backends import CpuBackend, HipBackend, make_backend  # noqa: F401
from .env import BatchedJssEnv  # noqa: F401
from .facade import JssEnv, make  # noqa: F401
from .bucketed import BucketedJssEnv  # noqa: F401
from .vector import JssVectorEnv  # noqa: F401

__version__ = "0.1.0"


def _register_with_gymnasium():
    """``gym.make('jss-v1', env_config=...)`` as in JSSEnv/__init__.py:6-9, when gymnasium exists."""
    try:
        from gymnasium.envs.registration import register
        register(id="jss-v1", entry_point="jssenv_amd.facade:JssEnv")
    except Exception:
        pass


_register_with_gymnasium()
