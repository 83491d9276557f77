"""Builds the two native libraries of the package, in-tree:

* ``libjss_hip.so``  -- the MI355X kernels + C ABI (hipcc, gfx950 only);
* ``libjss_cpu.so``  -- the host-core twin with the identical C ABI (g++, OpenMP).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "jss_kernels.hip")
OUT = os.path.join(_HERE, "libjss_hip.so")
# (the counters are bumped by one lane per env: the compiler's wave-aggregation scaffolding around every atomic --
#  mbcnt, compare, exec save / restore, popcount, multiply -- is pure overhead there)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(_ROOT, "include"),
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
CPU_SRC = os.path.join(_HERE, "csrc", "jss_cpu.cpp")
CPU_OUT = os.path.join(_HERE, "libjss_cpu.so")
CPU_FLAGS = ["-O3", "-std=c++17", "-fopenmp", "-fPIC", "-shared", "-Wall", "-I" + os.path.join(_ROOT, "include")]
_HEADER = os.path.join(_ROOT, "include", "jss_hip.h")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _fresh(out, deps):
    return os.path.isfile(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps)


def build_extension(force: bool = False, extra=(), out: str = OUT) -> str:
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp"))] + [_HEADER]
    if not force and _fresh(out, deps):
        return out
    subprocess.check_call([hipcc(), *FLAGS, *extra, SRC, "-o", out])
    return out


def build_cpu_twin(force: bool = False) -> str:
    if not force and _fresh(CPU_OUT, [CPU_SRC, _HEADER]):
        return CPU_OUT
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("g++ not found")
    tmp = CPU_OUT + f".tmp{os.getpid()}"
    subprocess.check_call([cxx, *CPU_FLAGS, CPU_SRC, "-o", tmp])
    os.replace(tmp, CPU_OUT)     # atomic: several processes (ranks, xdist workers) may build at once
    return CPU_OUT
