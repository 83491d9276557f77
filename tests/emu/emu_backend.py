"""TEST INFRASTRUCTURE: a NumPy 'device' for jssenv_amd.BatchedJssEnv backed by
tests/emu/libjss_emu.so (the unmodified kernel source compiled against the SIMT
emulator).  Lets the host layer and the kernel logic run in a container without
a GPU.  Never imported by the package."""
import ctypes as C
import os
import subprocess

import numpy as np

from jssenv_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libjss_emu.so")
_CSRC = os.path.join(_HERE, "..", "..", "jssenv_amd", "csrc")
_SRC = [os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.join(_HERE, "..", "..", "include", "jss_hip.h")] + \
       [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)]


def build(force=False):
    stale = not os.path.isfile(_LIB) or any(os.path.getmtime(p) > os.path.getmtime(_LIB) for p in _SRC)
    if force or stale:
        subprocess.check_call([os.path.join(_HERE, "build_emu.sh")])
    return _LIB


class EmuBackend:
    name = "emu"
    device = "emu"

    def __init__(self, default_kernel="auto"):
        self.lib = _abi.bind(C.CDLL(build()))
        self.default_kernel = default_kernel
        self._keep = []

    def zeros(self, shape, dtype):
        return np.zeros(shape, dtype=getattr(np, dtype))

    def from_numpy(self, a):
        return np.ascontiguousarray(a).copy()

    def ptr(self, x):
        if x is None:
            return 0
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data

    def numpy(self, x):
        return np.array(x, copy=True)

    def stream(self):
        return 0

    def sync(self):
        pass

    def on_device(self):
        from jssenv_amd.env import _NULL_CTX
        return _NULL_CTX

    def with_streams(self, n, fn, events=None):
        return fn((C.c_void_p * n)())

    def as_device(self, x, dtype):
        a = np.ascontiguousarray(np.asarray(x).astype(getattr(np, dtype)))
        self._keep = [a]
        return a

    def select_into(self, out, flags, a, b):
        np.copyto(out, b)
        out[flags != 0] = a

    def copy_into(self, dst, src):
        dst[...] = src

    def close(self):
        pass
