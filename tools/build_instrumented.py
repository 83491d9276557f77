"""Profiling aid: build the two instrumented variants of the packed kernel used by
tools/gpu_wave_timeline.py (s_memtime stamps at phase boundaries of three waves -> variants/timing.so) and
tools/gpu_dispatch_timeline.py (start/end of every workgroup on the 100 MHz real-time counter ->
variants/timeline.so).  The product sources are copied to /tmp and patched there; nothing instrumented is
committed or shipped."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tag, patch):
    d = f"/tmp/var_{tag}"
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "jssenv_amd", "csrc"), d)
    patch(d)
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(d, "jss_kernels.hip"),
                           "-o", os.path.join(ROOT, "variants", f"{tag}.so")])
    print("built variants/%s.so" % tag)


def _sub(path, old, new, count=-1):
    s = open(path).read()
    assert old in s, (path, old[:60])
    open(path, "w").write(s.replace(old, new, count))


def patch_timeline(d):
    f = os.path.join(d, "jss_packed_env.hpp")
    _sub(f, "namespace jss {\n", "namespace jss {\n__device__ unsigned long long jss_dbg_tl[8192][2];\n", 1)
    _sub(f, "    HIP_DYNAMIC_SHARED(int32_t, lds)\n    constexpr int E = kWave / G;                      // envs per wave\n"
            "    constexpr int EB = E * kWavesPerBlock;            // envs per workgroup",
         "    HIP_DYNAMIC_SHARED(int32_t, lds)\n    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();\n"
         "    constexpr int E = kWave / G;                      // envs per wave\n"
         "    constexpr int EB = E * kWavesPerBlock;            // envs per workgroup", 1)
    _sub(f, "    if (!(p.ablate & JSS_ABLATE_OBS)) p_store_obs(e, c, p, scratch, first_env, wave_whole);\n}\n\n"
            "// ---------------------------------------------------------------------------------------\n// persistent variant",
         "    if (!(p.ablate & JSS_ABLATE_OBS)) p_store_obs(e, c, p, scratch, first_env, wave_whole);\n"
         "    if (MODE == kRollout1 && threadIdx.x == 0 && blockIdx.x < 8192) {\n"
         "        jss_dbg_tl[blockIdx.x][0] = t_start;\n"
         "        jss_dbg_tl[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();\n    }\n}\n\n"
         "// ---------------------------------------------------------------------------------------\n// persistent variant")
    _sub(os.path.join(d, "jss_kernels.hip"), 'extern "C" {\n',
         'extern "C" {\nint jss_debug_timeline(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, '
         'HIP_SYMBOL(jss::jss_dbg_tl), sizeof(unsigned long long) * 8192 * 2); }\n', 1)


if __name__ == "__main__":
    _build("timeline", patch_timeline)
