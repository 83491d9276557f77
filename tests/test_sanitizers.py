"""SURVEY.md section 5 (sanitizers): the host-core twin compiled with AddressSanitizer + UndefinedBehaviorSanitizer and
driven through the C ABI by tests/san/twin_driver.cpp (rollouts on ta01- and ta80-shaped instances, edge shapes up to
the ABI's 128 x 64 limit, forced NOPEs and hostile actions, partial resets, the trajectory recorder).  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_twin_under_asan_and_ubsan(tmp_path):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "twin_san")
    cmd = [cxx, "-O1", "-g", "-std=c++17", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "jssenv_amd", "csrc", "jss_cpu.cpp"), os.path.join(ROOT, "tests", "san", "twin_driver.cpp"), "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "SANITIZED-TWIN-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
