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


def patch_timing(d):
    f = os.path.join(d, "jss_packed_env.hpp")
    _sub(f, "namespace jss {\n",
         "namespace jss {\n__device__ unsigned long long jss_dbg_t[4][16];\n"
         "#define TS(i) do { __builtin_amdgcn_sched_barrier(0); ts[i] = __builtin_amdgcn_s_memtime(); "
         "__builtin_amdgcn_sched_barrier(0); } while (0)\n", 1)
    _sub(f, "const Params &p, int a, int32_t *mvtab) {", "const Params &p, int a, int32_t *mvtab, unsigned long long *ts) {")
    _sub(f, "    const bool stepping = alloc || is_nope;\n    for (;;) {", "    const bool stepping = alloc || is_nope;\n    TS(4);\n    for (;;) {")
    _sub(f, "    if (!(p.ablate & JSS_ABLATE_PRIORITIZE)) p_prioritize(e, c, stepping);      // :432 / :471",
         "    TS(5);\n    if (!(p.ablate & JSS_ABLATE_PRIORITIZE)) p_prioritize(e, c, stepping);\n    TS(6);")
    _sub(f, "    if (!(p.ablate & JSS_ABLATE_CHECK_NO_OP)) p_check_no_op(e, c, stepping, mvtab);  // :433 / :472",
         "    if (!(p.ablate & JSS_ABLATE_CHECK_NO_OP)) p_check_no_op(e, c, stepping, mvtab);\n    TS(7);")
    _sub(f, "p_step(e, c, p, a_in, mvtab);", "p_step(e, c, p, a_in, mvtab, ts);")
    _sub(f, "const int rn = p_step(e, c, p, a, mvtab);", "TS(3);\n            const int rn = p_step(e, c, p, a, mvtab, ts);")
    _sub(f, "int a_in, bool selected,\n                                       int32_t *mvtab) {",
         "int a_in, bool selected,\n                                       int32_t *mvtab, unsigned long long *ts) {")
    _sub(f, "p_body<G, MODE>(e, hd, c, p, a_in, selected, mvtab);", "p_body<G, MODE>(e, hd, c, p, a_in, selected, mvtab, ts);")
    _sub(f, "        p_body<G, MODE>(e, hd, c, p, a_in, true, mvtab);",
         "        unsigned long long ts[16];\n        p_body<G, MODE>(e, hd, c, p, a_in, true, mvtab, ts);")
    _sub(f, "    HIP_DYNAMIC_SHARED(int32_t, lds)\n    constexpr int E = kWave / G;                      // envs per wave\n"
            "    constexpr int EB = E * kWavesPerBlock;            // envs per workgroup",
         "    HIP_DYNAMIC_SHARED(int32_t, lds)\n    unsigned long long ts[16];\n    for (int i = 0; i < 16; ++i) ts[i] = 0;\n    TS(0);\n"
         "    constexpr int E = kWave / G;                      // envs per wave\n"
         "    constexpr int EB = E * kWavesPerBlock;            // envs per workgroup", 1)
    _sub(f, "    __syncthreads();\n\n    PEnv<G> e;\n    PHeader hd = p_unpack(e, c, raw);\n"
            "    p_body<G, MODE>(e, hd, c, p, a_in, selected, mvtab, ts);\n    if (MODE == kPolicy) return;\n    p_store(e, c, p, hd);\n"
            "    if (!(p.ablate & JSS_ABLATE_OBS)) p_store_obs(e, c, p, scratch, first_env, wave_whole);\n}",
         "    __syncthreads();\n    TS(1);\n    PEnv<G> e;\n    PHeader hd = p_unpack(e, c, raw);\n"
         "    { int dummy = e.t + e.todo + e.tm; asm volatile(\"\" :: \"v\"(dummy)); }\n    TS(2);\n"
         "    p_body<G, MODE>(e, hd, c, p, a_in, selected, mvtab, ts);\n    if (MODE == kPolicy) return;\n    p_store(e, c, p, hd);\n    TS(8);\n"
         "    if (!(p.ablate & JSS_ABLATE_OBS)) p_store_obs(e, c, p, scratch, first_env, wave_whole);\n    TS(9);\n"
         "    if (MODE == kRollout1 && lane == 0 && wave == 0) {\n"
         "        int slot = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1));\n"
         "        if (slot >= 0) for (int i = 0; i < 16; ++i) jss_dbg_t[slot][i] = ts[i];\n    }\n}")
    _sub(os.path.join(d, "jss_kernels.hip"), 'extern "C" {\n',
         'extern "C" {\nint jss_debug_times(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, '
         'HIP_SYMBOL(jss::jss_dbg_t), sizeof(unsigned long long) * 64); }\n', 1)


if __name__ == "__main__":
    _build("timeline", patch_timeline)
    _build("timing", patch_timing)
