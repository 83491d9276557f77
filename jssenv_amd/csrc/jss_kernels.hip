// jssenv_amd/csrc/jss_kernels.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI of the
// batched Job-Shop-Scheduling simulator.  Interface and layout: include/jss_hip.h.
//
//   jss_common.hpp      parameters, wave helpers, counter RNG
//   jss_wave_env.hpp    one wavefront per env        (any J <= 128, M <= 64)
//   jss_packed_env.hpp  64/G envs per wavefront      (J, M <= G, G = 16 or 32)
//
// No MFMA anywhere: the path is integer indexing, there is no dense contraction.
#include "jss_common.hpp"
#include "jss_packed_env.hpp"
#include "jss_wave_env.hpp"

namespace {
using namespace jss;

// ---------------------------------------------------------------------------------------
// host side of the C ABI
// ---------------------------------------------------------------------------------------
int check_args(const JssDesc *d, const JssState *s, const JssOut *o, bool need_out) {
    if (!d || !s) return JSS_E_NULL;
    if (!d->ops || !d->jobs || !d->machines || !d->max_time_op || !d->max_time_jobs || !d->sum_op) return JSS_E_NULL;
    if (!s->env || !s->job || !s->machine || !s->solution) return JSS_E_NULL;
    if (need_out && (!o || !o->real_obs || !o->action_mask || !o->reward || !o->done || !o->makespan)) return JSS_E_NULL;
    if (d->batch < 0 || d->jmax < 1 || d->jmax > JSS_MAX_JOBS || d->mmax < 2 || d->mmax > JSS_MAX_MACHINES ||
        d->n_tables < 1)
        return JSS_E_SHAPE;
    if (!d->table_of_env && d->n_tables != 1 && d->n_tables != d->batch) return JSS_E_SHAPE;
    return 0;
}

int g_kernel_choice = JSS_KERNEL_AUTO;
int g_ablate = 0;
int g_lds_pad = 0;
int g_persist = 0;   // waves per SIMD of the persistent kernel; 0 = off (default: measured slower, profiles/README.md)
int g_cu_count = 0;  // 0 = ask the runtime

int compute_units() {
    if (g_cu_count > 0) return g_cu_count;
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cached = n;
        else
            cached = 256;
    }
    return cached;
}

// Kernel flavour for a batch shape: the packed kernel needs every env's jobs AND machines to fit
// a 16- or 32-lane group.
int packed_group(const JssDesc &d) {
    if (g_kernel_choice == JSS_KERNEL_WAVE) return 0;
    if (d.jmax <= 16 && d.mmax <= 16) return 16;
    if (d.jmax <= 32 && d.mmax <= 32) return 32;
    return 0;
}

template <int MODE>
int launch(Params &p, void *stream) {
    if (p.d.batch == 0) return 0;
    p.stride = p.d.mmax;
    p.region_ints = p.d.jmax * p.stride;
    p.shared_table = p.d.n_tables == 1 ? 1 : 0;
    p.ablate = g_ablate;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int G = packed_group(p.d);
    if (G) {
        const int envs_per_block = (kWave / G) * kWavesPerBlock;
        const int n_regions = p.shared_table ? 1 : envs_per_block;
        p.obs_off_ints = (n_regions * p.region_ints + 3) & ~3;
        p.obs_wave_floats = ((kWave / G) * p.d.jmax * 7 + 3) & ~3;
        p.mv_off_ints = p.obs_off_ints + kWavesPerBlock * p.obs_wave_floats;
        const size_t shmem = sizeof(int32_t) * ((size_t)p.mv_off_ints + kBlock) + g_lds_pad;
        const int blocks = (p.d.batch + envs_per_block - 1) / envs_per_block;
        // persistent variant: shared instance, step-per-launch modes, more env sets than resident waves
        if ((MODE == kStep || MODE == kRollout1) && p.shared_table && g_persist > 0) {
            const int pblocks = compute_units() * g_persist;              // one wave per SIMD per workgroup
            if (blocks > pblocks) {
                constexpr int PM = (MODE == kStep) ? kStep : kRollout1;   // only these two are instantiated
                if (G == 16)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(jss_packed_persistent<16, PM>), dim3(pblocks), dim3(kBlock), shmem, st, p);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(jss_packed_persistent<32, PM>), dim3(pblocks), dim3(kBlock), shmem, st, p);
                return (int)hipGetLastError();
            }
        }
        if (G == 16)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(jss_packed_kernel<16, MODE>), dim3(blocks), dim3(kBlock), shmem, st, p);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(jss_packed_kernel<32, MODE>), dim3(blocks), dim3(kBlock), shmem, st, p);
        return (int)hipGetLastError();
    }
    const int n_regions = p.shared_table ? 1 : kWavesPerBlock;
    const size_t shmem = sizeof(int32_t) * (size_t)n_regions * p.region_ints + sizeof(float) * kWavesPerBlock * p.d.jmax * 7;
    const int blocks = (p.d.batch + kWavesPerBlock - 1) / kWavesPerBlock;
    if (p.d.jmax <= kWave)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(jss_kernel<1, MODE>), dim3(blocks), dim3(kBlock), shmem, st, p);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(jss_kernel<2, MODE>), dim3(blocks), dim3(kBlock), shmem, st, p);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int jss_abi_version(void) { return JSS_ABI_VERSION; }

int jss_set_option(int option, int value) {
    if (option == JSS_OPT_KERNEL && value >= JSS_KERNEL_AUTO && value <= JSS_KERNEL_WAVE) {
        g_kernel_choice = value;
        return 0;
    }
    if (option == JSS_OPT_LDS_PAD && value >= 0 && value <= 150000) {
        g_lds_pad = value;
        return 0;
    }
    if (option == JSS_OPT_PERSIST && value >= 0 && value <= 8) {
        g_persist = value;
        return 0;
    }
    if (option == JSS_OPT_CU_COUNT && value >= 0) {
        g_cu_count = value;
        return 0;
    }
    if (option == JSS_OPT_ABLATE) {
        g_ablate = value;
        return 0;
    }
    return JSS_E_KIND;
}

const char *jss_error_string(int code) {
    switch (code) {
    case 0: return "ok";
    case JSS_E_NULL: return "null pointer in JssDesc/JssState/JssOut or arguments";
    case JSS_E_SHAPE: return "bad shape (batch/jmax/mmax/n_tables)";
    case JSS_E_KIND: return "unknown policy kind";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int jss_reset(const JssDesc *desc, const JssState *state, const JssOut *out, const uint8_t *which, void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.which = which;
    return launch<kReset>(p, stream);
}

int jss_step(const JssDesc *desc, const JssState *state, const int32_t *actions, const JssOut *out, void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (!actions) return JSS_E_NULL;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.actions = actions;
    return launch<kStep>(p, stream);
}

int jss_advance(const JssDesc *desc, const JssState *state, const uint8_t *which, int32_t *hole, const JssOut *out,
                void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.which = which; p.hole = hole;
    return launch<kAdvance>(p, stream);
}

int jss_policy(const JssDesc *desc, const JssState *state, int kind, uint64_t seed, uint32_t explore_q16,
               int32_t *actions, void *stream) {
    int rc = check_args(desc, state, nullptr, false);
    if (rc) return rc;
    if (!actions) return JSS_E_NULL;
    if (kind < 0 || kind >= JSS_N_POLICIES) return JSS_E_KIND;
    Params p = {};
    p.d = *desc; p.s = *state; p.actions_out = actions; p.kind = kind; p.seed = seed; p.explore_q16 = explore_q16;
    return launch<kPolicy>(p, stream);
}

int jss_rollout(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                uint32_t explore_q16, int32_t n_iter, int32_t flags, void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (kind < 0 || kind >= JSS_N_POLICIES) return JSS_E_KIND;
    if (n_iter < 0) return JSS_E_SHAPE;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.kind = kind; p.seed = seed; p.explore_q16 = explore_q16;
    p.n_iter = n_iter; p.flags = flags;
    return n_iter == 1 ? launch<kRollout1>(p, stream) : launch<kRollout>(p, stream);
}

}  // extern "C"
