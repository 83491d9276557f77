"""Builds jssenv_amd/libjss_hip.so (hipcc, gfx950 only)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "jss_kernels.hip")
OUT = os.path.join(_HERE, "libjss_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(_ROOT, "include")]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_extension(force: bool = False, extra=()) -> str:
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(_ROOT, "include", "jss_hip.h")]
    if not force and os.path.isfile(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    subprocess.check_call([hipcc(), *FLAGS, *extra, SRC, "-o", OUT])
    return OUT
