// jssenv_amd/csrc/jss_kernels.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI of the
// batched Job-Shop-Scheduling simulator.  Interface and layout: include/jss_hip.h.
//
//   jss_common.hpp      parameters, wave helpers, counter RNG
//   jss_wave_env.hpp    one wavefront per env        (any J <= 128, M <= 64)
//   jss_packed_env.hpp  64/G envs per wavefront      (J, M <= G, G = 16 or 32)
//
// No MFMA anywhere: the path is integer indexing, there is no dense contraction.
#include <mutex>
#include <unordered_map>

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
    if (!d->ops || !d->inst) return JSS_E_NULL;
    if (!s->env || !s->env_const || !s->job || !s->solution) return JSS_E_NULL;
    if (!s->machine && d->record_ints != JSS_NFC && d->record_ints != JSS_NFM) return JSS_E_NULL;   // compact / medium batches keep no machine clocks
    if (need_out && (!o || !o->real_obs || !o->action_mask || !o->reward || !o->done || !o->makespan)) return JSS_E_NULL;
    if (d->batch < 0 || d->jmax < 1 || d->jmax > JSS_MAX_JOBS || d->mmax < 2 || d->mmax > JSS_MAX_MACHINES ||
        d->n_tables < 1)
        return JSS_E_SHAPE;
    if (!d->table_of_env && d->n_tables != 1 && d->n_tables != d->batch) return JSS_E_SHAPE;
    if (d->kernel & ~(JSS_KERNEL_WAVE | JSS_KERNEL_ONE_ENV_PER_WAVE | JSS_KERNEL_TWO_ENVS_PER_WAVE)) return JSS_E_KIND;
    if (d->record_ints != 0 && d->record_ints != JSS_NF && d->record_ints != JSS_NFC && d->record_ints != JSS_NFM) return JSS_E_SHAPE;
    if (d->record_ints == JSS_NFC && d->n_tables != 1) return JSS_E_SHAPE;   // compact records need the ONE table in LDS
    // medium records: 21-bit ops (machines <= 32, whatever the number of jobs), per-env tables
    if (d->record_ints == JSS_NFM && (d->mmax > 32 || d->n_tables == 1)) return JSS_E_SHAPE;
    return 0;
}

int record_ints_of(const JssDesc &d) { return d.record_ints == JSS_NFC ? JSS_NFC : d.record_ints == JSS_NFM ? JSS_NFM : JSS_NF; }

// f64_ok: the call's policy is a launch of its own (the kernels that carry JSS_POLICY_CR_F64's float64 selector)
int check_kind(const JssDesc *d, int kind_arg, bool f64_ok = false) {
    const int kind = kind_arg & 0xFF, fp = (kind_arg >> 8) & 0xFF, fq = (kind_arg >> 16) & 0xFF;
    if (kind_arg < 0 || (kind_arg >> 25) || kind >= JSS_N_POLICIES) return JSS_E_KIND;
    if ((kind_arg >> 24) & 1) {                      // JSS_POLICY_CR_F64: the factor is JssDesc.cr_factor
        if (!f64_ok || kind != JSS_POLICY_CR || fp || fq || !(d->cr_factor > 0.0) || !(d->cr_factor < 1e300)) return JSS_E_KIND;
    }
    if (fp || fq) {                                  // a due-date factor p / q: CriticalRatio only, q a power of two <= 64
        if (kind != JSS_POLICY_CR || fp < 1 || fq < 1 || fq > 64 || (fq & (fq - 1))) return JSS_E_KIND;
    }
    if ((kind == JSS_POLICY_MWR || kind == JSS_POLICY_LWR || kind == JSS_POLICY_CR) && !d->rem) return JSS_E_NULL;
    return 0;
}

// Events of JSS_ROLLOUT_FORK_JOIN: one set per main stream (streams[0]), created on first use on the device that is
// current then -- the device the caller launches on -- and kept for the life of the process.  Two env objects on two
// devices, or two host threads driving two streams, never share an event; the registry itself is guarded by a mutex.
struct ForkJoinEvents {
    hipEvent_t fork = nullptr;
    hipEvent_t join[16] = {};
};
std::mutex g_events_mutex;
std::unordered_map<void *, ForkJoinEvents> g_events;

int events_for(void *main_stream, ForkJoinEvents **out) {
    std::lock_guard<std::mutex> lock(g_events_mutex);
    ForkJoinEvents &ev = g_events[main_stream];
    if (!ev.fork) {
        if (hipEventCreateWithFlags(&ev.fork, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
        for (int i = 0; i < 16; ++i)
            if (hipEventCreateWithFlags(&ev.join[i], hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
    }
    *out = &ev;                        // (unordered_map never moves its elements)
    return 0;
}

// streams[1..n) start behind everything queued on streams[0] so far ...
int fork_streams(const ForkJoinEvents &ev, void *const *streams, int n) {
    if (hipEventRecord(ev.fork, reinterpret_cast<hipStream_t>(streams[0])) != hipSuccess) return (int)hipGetLastError();
    for (int i = 1; i < n; ++i)
        if (hipStreamWaitEvent(reinterpret_cast<hipStream_t>(streams[i]), ev.fork, 0) != hipSuccess) return (int)hipGetLastError();
    return 0;
}
// ... and streams[0] continues behind all of them (every stream is joined even if one of the calls fails: a launch
// error in the middle of a window must not leave side streams running free of the caller's stream)
int join_streams(const ForkJoinEvents &ev, void *const *streams, int n) {
    int rc = 0;
    for (int i = 1; i < n; ++i) {
        if (hipEventRecord(ev.join[i], reinterpret_cast<hipStream_t>(streams[i])) != hipSuccess ||
            hipStreamWaitEvent(reinterpret_cast<hipStream_t>(streams[0]), ev.join[i], 0) != hipSuccess) {
            const int e = (int)hipGetLastError();
            if (!rc) rc = e;
        }
    }
    return rc;
}

#ifdef JSS_PROFILING
int g_ablate = 0;
int g_lds_pad = 0;
unsigned long long *g_stamps = nullptr;
#endif

// Kernel flavour for a batch shape: the packed kernel needs every env's jobs AND machines to fit
// a 16- or 32-lane group.
// by_class: the fused multi-set grid, which honours JssDesc.jclass / mclass (a shape class inside wider padded tensors)
int class_jobs(const JssDesc &d, bool by_class) { return by_class && d.jclass > 0 ? d.jclass : d.jmax; }
int class_machines(const JssDesc &d, bool by_class) { return by_class && d.mclass > 0 ? d.mclass : d.mmax; }
int packed_group(const JssDesc &d, bool by_class = false) {
    if (d.kernel & JSS_KERNEL_WAVE) return 0;
    const int j = class_jobs(d, by_class), m = class_machines(d, by_class);
    if (j <= 16 && m <= 16) return 16;
    if (j <= 32 && m <= 32) return 32;
    return 0;
}

using KernelFn = void (*)(Params);

// Jobs per lane of the one-wavefront-per-env flavour for a plain (single-set) launch: by the padded extent -- unless the call
// names a shape class inside wider rows (JssDesc.jclass: every env of the call has J <= jclass) that fits one job per lane.
// (64 jobs inside rows wider than 64 stay with two jobs per lane: the NOPE flag of such an env is byte 64 of its mask row.)
int wave_jpl(const JssDesc &d) {
    if (d.jmax <= kWave) return 1;
    return (d.jclass > 0 && d.jclass < kWave) ? 1 : 2;
}

// Two envs per wavefront, one after the other (jss_wave_env.hpp, wave_block2): the one-step modes of the one-wavefront-per-env
// flavour with one job per lane and per-env tables (full or medium records).  Measured (profiles/r06_misc/two_per_wave_ab.txt,
// wave_timeline_two_per_wave.txt): with half the wavefronts a launch of 8 192 envs is 15-18 % SLOWER (4 wavefronts per SIMD are
// latency-bound: a wavefront's two steps take 26.7 k cycles where one took 16.3 k), 16 384 envs 9-10 % slower, and only from
// about three rounds of resident wavefronts per launch on does the form come out ahead (65 536 envs in two or three
// sub-batches: +6-7 %) -- there the second env's state arrives under the first env's step instead of at the head of a new
// wavefront's life.  Hence the threshold; JssDesc.kernel's JSS_KERNEL_ONE / TWO_ENVS_PER_WAVE bits override it per call.
#ifndef JSS_TWO_PER_WAVE_MIN_BATCH
#define JSS_TWO_PER_WAVE_MIN_BATCH 20480
#endif
template <int MODE>
bool two_per_wave(const JssDesc &d, int G, int class_j) {
    if (MODE != kRollout1 && MODE != kStep) return false;
    if (G || class_j > kWave || d.n_tables == 1) return false;
    if (d.kernel & JSS_KERNEL_ONE_ENV_PER_WAVE) return false;
    return (d.kernel & JSS_KERNEL_TWO_ENVS_PER_WAVE) || d.batch >= JSS_TWO_PER_WAVE_MIN_BATCH;
}

template <int MODE, int TAB>
KernelFn pick_tab(int G, int jpl) {
    if (G == 16) return jss_packed_kernel<16, MODE, TAB>;
    if (G == 32) return jss_packed_kernel<32, MODE, TAB>;
    if (jpl == 1) return jss_kernel<1, MODE, TAB>;
    return jss_kernel<2, MODE, TAB>;
}
template <int MODE>
KernelFn pick(int G, int jpl, bool shared, int record_ints) {
    if (!shared) return record_ints == JSS_NFM ? pick_tab<MODE, kTabGlobalM>(G, jpl) : pick_tab<MODE, kTabGlobal>(G, jpl);
    return record_ints == JSS_NFC ? pick_tab<MODE, kTabLdsC>(G, jpl) : pick_tab<MODE, kTabLds>(G, jpl);
}

template <int MODE>
KernelFn pick_two(int record_ints) {       // (instantiated for the one-step modes only)
    if constexpr (MODE == kRollout1 || MODE == kStep)
        return record_ints == JSS_NFM ? jss_kernel_two<MODE, kTabGlobalM> : jss_kernel_two<MODE, kTabGlobal>;
    else return nullptr;
}

struct LaunchPlan {
    KernelFn fn;
    int envs_per_block;
    size_t shmem;
    bool two;               // one-wavefront-per-env flavour, two envs per wavefront
};

constexpr size_t kMaxDynamicLds = 64 * 1024;   // available to a workgroup without raising the function attribute

// Fills the launch-derived fields of `p` (LDS layout) from the whole batch's description.
template <int MODE>
int plan(Params &p, LaunchPlan &lp, bool by_class = false) {
    const bool shared = p.d.n_tables == 1;
    p.region_ints = p.d.jmax * p.d.mmax;
    p.table_lds_ints = shared ? ((p.region_ints + 3) & ~3) : 0;
    const int G = packed_group(p.d, by_class);
    lp.two = false;
    if (G) {
        lp.envs_per_block = (kWave / G) * kWavesPerBlock;
        p.obs_wave_floats = ((kWave / G) * (p.d.jmax < G ? p.d.jmax : G) * 7 + 3) & ~3;      // (jmax > G: a class inside padded rows)
        p.mv_off_ints = p.table_lds_ints + kWavesPerBlock * p.obs_wave_floats;
        p.norm_off_ints = p.mv_off_ints + kBlock;                       // one int per lane (six used per group), kTabGlobal
        lp.shmem = sizeof(int32_t) * ((size_t)p.norm_off_ints + (shared ? 0 : kBlock));
    } else {
        lp.two = !by_class && two_per_wave<MODE>(p.d, G, p.d.jmax);      // (the fused grid keeps one env per wavefront: its classes are parts of a batch)
        lp.envs_per_block = kWavesPerBlock * (lp.two ? 2 : 1);
        p.obs_wave_floats = (p.d.jmax * 7 + 3 + 3) & ~3;               // + up to 3 floats of alignment shift (store_obs)
        if (p.obs_wave_floats < kWave) p.obs_wave_floats = kWave;      // unpack_env borrows it: one int per machine
        p.mv_off_ints = 0;
        p.norm_off_ints = 0;
        lp.shmem = sizeof(int32_t) * ((size_t)p.table_lds_ints + kWavesPerBlock * p.obs_wave_floats);
    }
#ifdef JSS_PROFILING
    p.ablate = g_ablate;
    p.stamps = g_stamps;
    lp.shmem += g_lds_pad;
#endif
    if (lp.shmem > kMaxDynamicLds) return JSS_E_LDS;
    lp.fn = by_class ? nullptr : lp.two ? pick_two<MODE>(p.d.record_ints) : pick<MODE>(G, wave_jpl(p.d), shared, p.d.record_ints);   // (the grid has its own kernel)
    return 0;
}

// ---- step session ---------------------------------------------------------------------------------------
template <int TAB>
KernelFn pick_session_tab(int G, int jpl) {
    if (G == 16) return jss_packed_session_kernel<16, TAB>;
    if (G == 32) return jss_packed_session_kernel<32, TAB>;
    if (jpl == 1) return jss_session_kernel<1, TAB>;
    return jss_session_kernel<2, TAB>;
}
KernelFn pick_session(int G, int jpl, bool shared, int record_ints) {
    if (!shared) return record_ints == JSS_NFM ? pick_session_tab<kTabGlobalM>(G, jpl) : pick_session_tab<kTabGlobal>(G, jpl);
    return record_ints == JSS_NFC ? pick_session_tab<kTabLdsC>(G, jpl) : pick_session_tab<kTabLds>(G, jpl);
}

struct SessionInfo {          // what jss_session_wait needs to know about an open session (keyed by its progress pointer)
    int active_waves;         // wavefronts that own at least one env set: the progress words the waiter sweeps
    int slots;
    long long timeout_ticks;
};
std::mutex g_sessions_mutex;
std::unordered_map<const void *, SessionInfo> g_sessions;

constexpr size_t kMaxSessionLds = 96 * 1024;       // per workgroup
constexpr size_t kSessionLdsPerCu = 96 * 1024;     // of a CU's 160 KB, all resident workgroups of the session together
constexpr long long kTicksPerMs = 100000;        // the wall clock of the device counts at 100 MHz

// LDS layout, grid and residency of a session over the batch `p` describes with `slots` env sets per wavefront.
// Fits = every workgroup is resident at once AND the chip keeps room for the caller's own kernels (post, wait, the
// policy network): see the two budgets at the end.
int plan_session(Params &p, LaunchPlan &lp, int slots, int *blocks_out, int *active_out) {
    const bool shared = p.d.n_tables == 1, compact = p.d.record_ints == JSS_NFC;
    const int G = packed_group(p.d);
    const int jpl = p.d.jmax <= kWave ? 1 : 2;
    p.region_ints = p.d.jmax * p.d.mmax;
    p.table_lds_ints = shared ? ((p.region_ints + 3) & ~3) : 0;
    p.slots = slots;
    int envs_per_wave, park4;                        // park4: int4 per parked env set of one wavefront
    if (G) {
        envs_per_wave = kWave / G;
        p.obs_wave_floats = ((kWave / G) * p.d.jmax * 7 + 3) & ~3;
        p.mv_off_ints = p.table_lds_ints + kWavesPerBlock * p.obs_wave_floats;
        p.norm_off_ints = p.mv_off_ints + kBlock;
        p.norm_slot_ints = shared ? 0 : kBlock;
        p.park_off_ints = p.norm_off_ints + (shared ? 0 : slots * kBlock);
        park4 = (compact ? 1 : p.d.record_ints == JSS_NFM ? 2 : 3) * kWave + 8;
    } else {
        envs_per_wave = 1;
        p.obs_wave_floats = (p.d.jmax * 7 + 3 + 3) & ~3;
        if (p.obs_wave_floats < kWave) p.obs_wave_floats = kWave;
        p.mv_off_ints = p.norm_off_ints = p.norm_slot_ints = 0;
        p.park_off_ints = p.table_lds_ints + kWavesPerBlock * p.obs_wave_floats;
        park4 = (compact ? jpl : 2 * jpl) * kWave + kWave / 4 + 1;   // (medium records: lo + hi rows like full ones)
    }
    lp.shmem = sizeof(int32_t) * ((size_t)p.park_off_ints + (slots > 1 ? (size_t)kWavesPerBlock * slots * park4 * 4 : 0));
    if (lp.shmem > kMaxSessionLds) return JSS_E_RESIDENT;
    lp.fn = pick_session(G, jpl, shared, p.d.record_ints);
    lp.envs_per_block = envs_per_wave * kWavesPerBlock * slots;
    const int sets = (p.d.batch + envs_per_wave - 1) / envs_per_wave;
    const int waves = (sets + slots - 1) / slots;
    const int blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    if (lp.shmem > kMaxDynamicLds &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(lp.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp.shmem) != hipSuccess) {
        (void)hipGetLastError();
        return JSS_E_LDS;
    }
    int dev = 0, n_cu = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(lp.fn), kBlock, 0) != hipSuccess)
        return (int)hipGetLastError();
    // Registers / wave slots: the occupancy query overstates what the hardware admits when a kernel uses more than 80
    // SGPRs (these do: 6 workgroups per CU at most, MI355X_MICROARCH.md "Residency"); two workgroup slots per CU stay
    // free.  LDS: the session takes at most kSessionLdsPerCu of a CU's 160 KB, so that a kernel of the caller that
    // asks for up to 64 KB always finds room next to it.
    if (occ > 6) occ = 6;
    int usable = occ - 2;
    const int by_lds = (int)(kSessionLdsPerCu / (lp.shmem ? lp.shmem : 1));
    if (by_lds < usable) usable = by_lds;
    if (usable < 1 || blocks > usable * n_cu) return JSS_E_RESIDENT;
    *blocks_out = blocks;
    *active_out = blocks * kWavesPerBlock < sets ? blocks * kWavesPerBlock : sets;
    return 0;
}

// mail[(step % depth) * B + i] = (step + 1) << 32 | action, steps [first, first + n): one thread per env
__global__ void jss_session_post_kernel(unsigned long long *mail, const int32_t *actions, int batch, int depth, int first, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= batch) return;
    for (int k = 0; k < n; ++k) {
        const int step = first + k;
        const int a = actions ? actions[(size_t)k * batch + i] : JSS_ACTION_CLOSE;
        wt_store(mail + (size_t)(step % depth) * batch + i, ((unsigned long long)(unsigned)(step + 1) << 32) | (unsigned)a);
    }
}

// returns once every active wavefront has published `steps_done` steps (one workgroup sweeps the progress words);
// bounded: gives up -- status[1] += 1 -- after the session's timeout or as soon as a wavefront of the session has
__device__ __forceinline__ void session_wait(const int32_t *progress, int32_t *status, int active, int steps_done, long long timeout_ticks) {
    __shared__ int behind;     // 1: somebody has not published `steps_done` yet; 2: give up (decided by thread 0 for the workgroup)
    long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        if (threadIdx.x == 0) behind = 0;
        __syncthreads();
        int mine = 0;
        for (int i = (int)threadIdx.x; i < active; i += (int)blockDim.x) mine |= fresh_load(progress + i) < steps_done ? 1 : 0;
        if (mine) behind = 1;
        __syncthreads();
        if (threadIdx.x == 0 && behind) {
            // ONE thread reads the status word and the clock and decides for everybody: wavefronts that looked for themselves
            // (each at its own instant, with its own t0) could disagree, and one of them would leave the others at the barrier
            bool give_up = fresh_load(status + 0) != 0;                      // a wavefront of the session timed out: it is dead
            if ((spins & 63u) == 63u) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > timeout_ticks) give_up = true;
            }
            if (give_up) {
                atomicAdd(status + 1, 1);
                behind = 2;
            }
        }
        __syncthreads();
        const int b = behind;
        __syncthreads();                                                     // (everybody has read it before thread 0 clears it)
        if (b != 1) return;
        __builtin_amdgcn_s_sleep(1);
    }
}
__global__ void jss_session_wait_kernel(const int32_t *progress, int32_t *status, int active, int steps_done, long long timeout_ticks) {
    session_wait(progress, status, active, steps_done, timeout_ticks);
}
// post of one step, and the first workgroup waits for it
__global__ void jss_session_step_kernel(unsigned long long *mail, const int32_t *actions, int batch, int depth, int step,
                                        const int32_t *progress, int32_t *status, int active, long long timeout_ticks) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < batch)
        wt_store(mail + (size_t)(step % depth) * batch + i, ((unsigned long long)(unsigned)(step + 1) << 32) | (unsigned)actions[i]);
    if (blockIdx.x == 0) session_wait(progress, status, active, step + 1, timeout_ticks);
}

int fire(const Params &p, const LaunchPlan &lp, void *stream) {
    if (p.d.batch == 0) return 0;
    const int blocks = (p.d.batch + lp.envs_per_block - 1) / lp.envs_per_block;
    hipLaunchKernelGGL(lp.fn, dim3(blocks), dim3(kBlock), lp.shmem, reinterpret_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}

template <int MODE>
int launch(Params &p, void *stream) {
    LaunchPlan lp;
    const int rc = plan<MODE>(p, lp);
    return rc ? rc : fire(p, lp, stream);
}

// The description of envs [start, start + count) of the batch `p` describes, for the step-type modes: every per-env
// pointer moves; the instance tables stay (an env's header names its table by its index in the WHOLE batch's
// tables, so n_tables keeps describing those).  Not a description a reset may be launched with.
Params sub_batch(const Params &p, int start, int count) {
    Params q = p;
    const size_t s0 = (size_t)start, jm = (size_t)p.d.jmax, mm = (size_t)p.d.mmax;
    q.d.batch = count;
    if (p.d.table_of_env) q.d.table_of_env = p.d.table_of_env + s0;
    if (p.d.env_ids) q.d.env_ids = p.d.env_ids + s0;
    q.d.env_id_base = p.d.env_id_base + start;
    q.s.env = p.s.env + s0 * JSS_NH;
    q.s.env_const = p.s.env_const + s0 * JSS_NC;
    q.s.job = p.s.job + s0 * jm * record_ints_of(p.d);
    q.s.machine = p.s.machine ? p.s.machine + s0 * mm : nullptr;
    q.s.solution = p.s.solution + s0 * jm * mm;
    if (p.s.counters) q.s.counters = p.s.counters + s0 * 4;
    q.o.real_obs = p.o.real_obs + s0 * jm * 7;
    q.o.action_mask = p.o.action_mask + s0 * (jm + 1);
    q.o.reward = p.o.reward + s0;
    q.o.done = p.o.done + s0;
    q.o.makespan = p.o.makespan + s0;
    return q;
}

// ---- several independent env sets in ONE grid (jss_multi_*) ------------------------------------------------------
// The shape classes of a ragged population (jssenv_amd.BucketedJssEnv: 16-lane groups, 32-lane groups, one wavefront per
// env, two jobs per lane) are compact batches of their own.  Launched one by one they need one launch per class and step
// and -- to overlap -- one stream each, at the mercy of how HIP deals streams onto hardware queues (round 4: 0.37-0.55 of
// the roofline for the same work, depending on the box).  Here ONE grid covers them all: a workgroup finds its env set by
// its index (block ranges, scalar compares on kernel arguments) and runs that set's body -- the same device functions
// the plain kernels are made of -- on its own Params.  The sets with the longest-lived wavefronts come first in the
// grid, so that the short ones fill the tail.  Per-env-table layouts only (an LDS-staged shared table would add four
// more bodies); other sets make the entry points fall back to one launch per set on the same stream.
constexpr int kMultiMaxSets = 6;
enum MultiFlavour { kMfW2G = 0, kMfW1G = 1, kMfP32G = 2, kMfP32M = 3, kMfP16G = 4, kMfP16M = 5, kMfNone = 6 };   // grid order
struct MultiParams {
    Params p[kMultiMaxSets];
    int32_t block_end[kMultiMaxSets];   // first workgroup index behind set i
    int32_t flavour[kMultiMaxSets];
    int32_t n_sets;
};

#ifndef JSS_MULTI_STEP_MIN_BLOCKS
#define JSS_MULTI_STEP_MIN_BLOCKS 5
#endif
constexpr int multi_min_blocks(int mode) { return mode == kStep ? JSS_MULTI_STEP_MIN_BLOCKS : mode == kRollout1 ? 7 : mode == kReset ? 6 : 8; }

template <int MODE>
__global__ __launch_bounds__(kBlock, multi_min_blocks(MODE)) void jss_multi_kernel(MultiParams mp) {
    HIP_DYNAMIC_SHARED(int32_t, lds)
    const int blk = (int)blockIdx.x;
    int k = 0;
#pragma unroll
    for (int i = 0; i + 1 < kMultiMaxSets; ++i)
        if (i + 1 < mp.n_sets && blk >= mp.block_end[i]) k = i + 1;
    const int block = blk - (k ? mp.block_end[k - 1] : 0);
    // The set's Params copied up front -- every field loaded in the entry block, behind the state loads, like a kernel that
    // takes them by value -- rather than read in place at their uses (-DJSS_MULTI_PARAMS_IN_PLACE: no spilled SGPRs instead of
    // 30, and 3 % slower: profiles/r05_misc/bucketed_grid_vs_streams.txt; the same trade as JSS_PARAMS_OF in jss_common.hpp)
#ifdef JSS_MULTI_PARAMS_IN_PLACE
    const Params &p = mp.p[k];
#else
    const Params p = mp.p[k];
#endif
#ifdef JSS_EXP_MULTI_ONLY
    switch (JSS_EXP_MULTI_ONLY) {
#else
    switch (mp.flavour[k]) {
#endif
    case kMfW2G: wave_block<2, MODE, kTabGlobal, false>(p, block, lds); break;   // (one body: 66 VGPRs, no spills at 7 waves / SIMD)
    case kMfW1G: wave_block<1, MODE, kTabGlobal>(p, block, lds); break;
    case kMfP32G: packed_block<32, MODE, kTabGlobal, true>(p, block, lds); break;   // (full records: also classes inside padded rows)
    case kMfP32M: packed_block<32, MODE, kTabGlobalM>(p, block, lds); break;
    case kMfP16G: packed_block<16, MODE, kTabGlobal, true>(p, block, lds); break;
    default: packed_block<16, MODE, kTabGlobalM>(p, block, lds); break;
    }
}

int multi_flavour(const JssDesc &d) {
    if (d.n_tables == 1) return kMfNone;                                  // shared table: LDS-staged bodies are not in the grid
    const int G = packed_group(d, true);
    const bool medium = d.record_ints == JSS_NFM;
    if (G && d.jmax > G && (medium || d.jclass > d.jmax || d.mclass > d.mmax)) return kMfNone;   // a class inside padded rows: full records
    if (d.jclass > d.jmax || d.mclass > d.mmax) return kMfNone;
    if (G == 16) return medium ? kMfP16M : kMfP16G;
    if (G == 32) return medium ? kMfP32M : kMfP32G;
    if (medium) return kMfNone;
    // one job per lane: J <= 64 -- but a 64-job env inside rows wider than 64 keeps its NOPE flag at byte 64 of the mask row,
    // which only the two-jobs-per-lane body writes
    return class_jobs(d, true) <= (d.jmax > kWave ? kWave - 1 : kWave) ? kMfW1G : kMfW2G;
}

// `ps[0..n)`: fully filled Params of the sets (everything but the launch-derived LDS fields).  One fused launch per step when
// every set has a body in the grid, otherwise one plain launch per set and step.  n_sub > 1: every set is cut into n_sub
// contiguous parts (boundaries at multiples of 64 envs) and part i of ALL sets is one grid on streams[i] -- step s of a part
// depends only on its own step s - 1, so one part's drain overlaps another's fill (what jss_rollout_steps does for one set).
// Step-type modes only (sub_batch); fork_join as in jss_rollout_steps.
template <int MODE>
int launch_multi(Params *ps, int n, int n_steps, int n_sub, void *const *streams, bool fork_join) {
    bool fused = n >= 2 && n <= kMultiMaxSets;
    for (int i = 0; i < n && fused; ++i) fused = multi_flavour(ps[i].d) != kMfNone;
    constexpr int kMaxParts = 4;
    if (n_sub > kMaxParts) n_sub = kMaxParts;
    if (!fused) {
        // No fused body for this combination (a shared-table set, medium records on the one-wavefront-per-env shapes, more
        // than kMultiMaxSets sets): one plain launch per set, part and step -- parts and streams exactly as in the fused form
        // (part i of every set on streams[i], JSS_ROLLOUT_FORK_JOIN honoured), so that a caller who asked for overlap gets it.
        // (The reset / policy modes come here with n_sub == 1: sub_batch describes a part for the step-type modes only.)
        struct Item { int set; Params p; };
        static Item items[kMaxParts][16];                                 // (0.5 KB each: not on the stack of every caller)
        static std::mutex items_mutex;
        std::lock_guard<std::mutex> lock(items_mutex);
        LaunchPlan lps[16];
        int count[kMaxParts] = {}, parts = 0;
        for (int i = 0; i < n; ++i) {
            const int rc = plan<MODE>(ps[i], lps[i]);                    // (the plan of a set serves its parts: fire() sizes the grid)
            if (rc) return rc;
        }
        for (int part = 0; part < n_sub; ++part) {
            int k = 0;
            for (int i = 0; i < n; ++i) {
                const Params &whole = ps[i];
                if (n_sub == 1) { items[parts][k++] = Item{i, whole}; continue; }
                const int chunk = (((whole.d.batch + n_sub - 1) / n_sub) + 63) & ~63;
                const int start = part * chunk;
                if (start >= whole.d.batch) continue;
                items[parts][k++] = Item{i, sub_batch(whole, start, whole.d.batch - start < chunk ? whole.d.batch - start : chunk)};
            }
            if (k) count[parts++] = k;
        }
        if (parts == 0) return 0;
        ForkJoinEvents *ev = nullptr;
        int rc = 0;
        fork_join = fork_join && parts > 1;
        if (fork_join && ((rc = events_for(streams[0], &ev)) || (rc = fork_streams(*ev, streams, parts)))) return rc;
        for (int s = 0; s < n_steps && !rc; ++s)
            for (int part = 0; part < parts && !rc; ++part)
                for (int k = 0; k < count[part] && !rc; ++k) rc = fire(items[part][k].p, lps[items[part][k].set], streams[part]);
        const int jrc = fork_join ? join_streams(*ev, streams, parts) : 0;
        return rc ? rc : jrc;
    }
    int order[kMultiMaxSets];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i)                                          // insertion sort by flavour (= grid order), stable
        for (int j = i; j > 0 && multi_flavour(ps[order[j]].d) < multi_flavour(ps[order[j - 1]].d); --j) {
            const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
    static MultiParams mp[kMaxParts];                                    // (2 KB each: not on the stack of every caller)
    static std::mutex mp_mutex;
    std::lock_guard<std::mutex> lock(mp_mutex);
    size_t shmem = 0;
    int blocks[kMaxParts] = {}, parts = 0;
    for (int part = 0; part < n_sub; ++part) {
        MultiParams &m = mp[parts];
        int nb = 0, k = 0;
        for (int q = 0; q < n; ++q) {
            Params &whole = ps[order[q]];
            const int chunk = n_sub == 1 ? whole.d.batch : ((((whole.d.batch + n_sub - 1) / n_sub) + 63) & ~63);
            const int start = part * chunk;
            if (start >= whole.d.batch) continue;
            LaunchPlan lp;
            Params p = n_sub == 1 ? whole : sub_batch(whole, start, whole.d.batch - start < chunk ? whole.d.batch - start : chunk);
            const int rc = plan<MODE>(p, lp, true);                      // (fills the LDS layout fields; class-aware; sized for THIS part)
            if (rc) return rc;
            nb += (p.d.batch + lp.envs_per_block - 1) / lp.envs_per_block;
            m.p[k] = p;
            m.block_end[k] = nb;
            m.flavour[k] = multi_flavour(p.d);

            if (lp.shmem > shmem) shmem = lp.shmem;
            ++k;
        }
        if (k == 0) continue;
        m.n_sets = k;
        blocks[parts++] = nb;
    }
    if (parts == 0) return 0;
    ForkJoinEvents *ev = nullptr;
    int rc = 0;
    fork_join = fork_join && parts > 1;
    if (fork_join && ((rc = events_for(streams[0], &ev)) || (rc = fork_streams(*ev, streams, parts)))) return rc;
    for (int s = 0; s < n_steps && !rc; ++s)                             // (the arguments are copied at every launch)
        for (int i = 0; i < parts && !rc; ++i) {
            hipLaunchKernelGGL(jss_multi_kernel<MODE>, dim3(blocks[i]), dim3(kBlock), shmem, reinterpret_cast<hipStream_t>(streams[i]), mp[i]);
            rc = (int)hipGetLastError();
        }
    const int jrc = fork_join ? join_streams(*ev, streams, parts) : 0;
    return rc ? rc : jrc;
}

int check_multi(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const JssOut *const *outs, bool need_out) {
    if (!descs || !states || (need_out && !outs)) return JSS_E_NULL;
    if (n_sets < 1 || n_sets > 16) return JSS_E_SHAPE;
    for (int i = 0; i < n_sets; ++i) {
        const int rc = check_args(descs[i], states[i], need_out ? outs[i] : nullptr, need_out);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" {

int jss_abi_version(void) { return JSS_ABI_VERSION; }

const char *jss_backend(void) { return "hip:gfx950"; }

#ifdef JSS_PROFILING
int jss_profiling_set(int option, int value) {
    if (option == JSS_PROF_LDS_PAD && value >= 0 && value <= 150000) {
        g_lds_pad = value;
        return 0;
    }
    if (option == JSS_PROF_ABLATE) {
        g_ablate = value;
        return 0;
    }
    return JSS_E_KIND;
}
int jss_profiling_stamps(void *device_buffer) {      // [B][16] uint64 (NULL: off)
    g_stamps = static_cast<unsigned long long *>(device_buffer);
    return 0;
}
#endif

const char *jss_error_string(int code) {
    switch (code) {
    case 0: return "ok";
    case JSS_E_NULL: return "null pointer in JssDesc/JssState/JssOut or arguments";
    case JSS_E_SHAPE: return "bad shape (batch/jmax/mmax/n_tables/n_sub)";
    case JSS_E_KIND: return "unknown policy kind or kernel flavour";
    case JSS_E_LDS: return "batch shape needs more LDS per workgroup than the device provides";
    case JSS_E_RESIDENT: return "the batch does not fit the chip as one round of resident workgroups (step session)";
    case JSS_E_SESSION: return "step session: bad step range (mailbox ring overrun, or the session was never opened)";
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

int jss_step_autoreset(const JssDesc *desc, const JssState *state, const int32_t *actions, const JssOut *out, void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (!actions) return JSS_E_NULL;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.actions = actions; p.flags = JSS_ROLLOUT_AUTORESET;
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
    if ((rc = check_kind(desc, kind, true))) return rc;
    Params p = {};
    p.d = *desc; p.s = *state; p.actions_out = actions; p.kind = kind; p.seed = seed; p.explore_q16 = explore_q16;
    return launch<kPolicy>(p, stream);
}

int jss_rollout(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                uint32_t explore_q16, int32_t n_iter, int32_t flags, void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if ((rc = check_kind(desc, kind))) return rc;
    if (n_iter < 0) return JSS_E_SHAPE;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.kind = kind; p.seed = seed; p.explore_q16 = explore_q16;
    p.n_iter = n_iter; p.flags = flags;
    return n_iter == 1 ? launch<kRollout1>(p, stream) : launch<kRollout>(p, stream);
}

int jss_trajectory(const JssDesc *desc, const JssState *state, const JssOut *out, const JssTraj *traj, int kind,
                   uint64_t seed, uint32_t explore_q16, int32_t n_steps, int32_t flags, void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (!traj) return JSS_E_NULL;
    if ((rc = check_kind(desc, kind))) return rc;
    if (n_steps < 0) return JSS_E_SHAPE;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.t = *traj; p.kind = kind; p.seed = seed; p.explore_q16 = explore_q16;
    p.n_iter = n_steps; p.flags = flags;
    return launch<kTraj>(p, stream);
}

int jss_steps(const JssDesc *desc, const JssState *state, const JssOut *out, const JssTraj *traj, const int32_t *actions,
              int32_t n_steps, void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (n_steps < 0) return JSS_E_SHAPE;
    if (n_steps == 0) return 0;                       // nothing to do (an empty action buffer has no address)
    if (!actions) return JSS_E_NULL;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.actions = actions; p.n_iter = n_steps;
    if (traj) p.t = *traj;
    p.t.action = nullptr;
    return launch<kSteps>(p, stream);
}

int jss_session_open(const JssDesc *desc, const JssState *state, const JssOut *out, const JssSession *session, void *stream) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (!session || !session->mail || !session->progress || !session->status) return JSS_E_NULL;
    if (session->depth < 1 || session->timeout_ms < 0 || desc->batch < 1) return JSS_E_SHAPE;
    const int want = session->slots;
    if (want != 0 && want != 1 && want != 2 && want != 4 && want != 8) return JSS_E_SHAPE;
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out;
    p.mail = reinterpret_cast<const unsigned long long *>(session->mail);
    p.progress = session->progress;
    p.status = session->status;
    p.depth = session->depth;
    p.timeout_ticks = (long long)(session->timeout_ms ? session->timeout_ms : 2000) * kTicksPerMs;
    LaunchPlan lp;
    int blocks = 0, active = 0;
    rc = JSS_E_RESIDENT;
    for (int slots = want ? want : 1; slots <= (want ? want : 8); slots *= 2) {
        rc = plan_session(p, lp, slots, &blocks, &active);
        if (rc != JSS_E_RESIDENT) break;
    }
    if (rc) return rc;
    hipLaunchKernelGGL(lp.fn, dim3(blocks), dim3(kBlock), lp.shmem, reinterpret_cast<hipStream_t>(stream), p);
    const int lrc = (int)hipGetLastError();
    if (lrc) return lrc;
    {   // registered only once the resident kernel is really on its way: wait / step of a session that never opened -> JSS_E_SESSION
        std::lock_guard<std::mutex> lock(g_sessions_mutex);
        g_sessions[session->progress] = SessionInfo{active, p.slots, p.timeout_ticks};
    }
    return 0;
}

int jss_session_post(const JssDesc *desc, const JssSession *session, const int32_t *actions, int32_t first_step,
                     int32_t n_steps, int32_t waited, void *stream) {
    if (!desc || !session || !session->mail || !actions) return JSS_E_NULL;
    if (first_step < 0 || n_steps < 1 || waited < 0 || waited > first_step || first_step + n_steps - waited > session->depth)
        return JSS_E_SESSION;
    hipLaunchKernelGGL(jss_session_post_kernel, dim3((desc->batch + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<unsigned long long *>(session->mail), actions,
                       desc->batch, session->depth, first_step, n_steps);
    return (int)hipGetLastError();
}

int jss_session_wait(const JssDesc *desc, const JssSession *session, int32_t steps_done, void *stream) {
    if (!desc || !session || !session->progress || !session->status) return JSS_E_NULL;
    if (steps_done < 0) return JSS_E_SESSION;
    SessionInfo info;
    {
        std::lock_guard<std::mutex> lock(g_sessions_mutex);
        const auto it = g_sessions.find(session->progress);
        if (it == g_sessions.end()) return JSS_E_SESSION;                    // never opened
        info = it->second;
    }
    hipLaunchKernelGGL(jss_session_wait_kernel, dim3(1), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream),
                       session->progress, session->status, info.active_waves, steps_done, info.timeout_ticks);
    return (int)hipGetLastError();
}

int jss_session_step(const JssDesc *desc, const JssSession *session, const int32_t *actions, int32_t step, void *stream) {
    if (!desc || !session || !session->mail || !session->progress || !session->status || !actions) return JSS_E_NULL;
    if (step < 0) return JSS_E_SESSION;
    SessionInfo info;
    {
        std::lock_guard<std::mutex> lock(g_sessions_mutex);
        const auto it = g_sessions.find(session->progress);
        if (it == g_sessions.end()) return JSS_E_SESSION;
        info = it->second;
    }
    hipLaunchKernelGGL(jss_session_step_kernel, dim3((desc->batch + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<unsigned long long *>(session->mail), actions,
                       desc->batch, session->depth, step, session->progress, session->status, info.active_waves, info.timeout_ticks);
    return (int)hipGetLastError();
}

int jss_session_close(const JssDesc *desc, const JssSession *session, int32_t next_step, void *stream) {
    if (!desc || !session || !session->mail) return JSS_E_NULL;
    if (next_step < 0) return JSS_E_SESSION;
    // (the caller has waited for every step it posted: the slot of next_step is free)
    {   // the session is over for the host: wait / step on it answer JSS_E_SESSION from here on (the progress / status buffers
        // may be freed by the caller once its stream has drained)
        std::lock_guard<std::mutex> lock(g_sessions_mutex);
        g_sessions.erase(session->progress);
    }
    hipLaunchKernelGGL(jss_session_post_kernel, dim3((desc->batch + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<unsigned long long *>(session->mail),
                       static_cast<const int32_t *>(nullptr), desc->batch, session->depth, next_step, 1);
    return (int)hipGetLastError();
}

int jss_sync_check(void *stream) {
    const hipError_t rc = hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));
    const hipError_t sticky = hipGetLastError();
    return (int)(rc != hipSuccess ? rc : sticky);
}

int jss_rollout_steps(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                      uint32_t explore_q16, int32_t n_steps, int32_t flags, int32_t n_sub, void *const *streams) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if ((rc = check_kind(desc, kind))) return rc;
    if (n_steps < 0 || n_sub < 1 || n_sub > 16) return JSS_E_SHAPE;
    if (!streams) return JSS_E_NULL;
    if (n_steps == 0) return 0;                       // no step: nothing is launched, nothing is touched (both libraries)
    Params p = {};
    p.d = *desc; p.s = *state; p.o = *out; p.kind = kind; p.seed = seed; p.explore_q16 = explore_q16;
    p.n_iter = 1; p.flags = flags & ~JSS_ROLLOUT_FORK_JOIN;
    LaunchPlan lp;
    if ((rc = plan<kRollout1>(p, lp))) return rc;
    // contiguous sub-batches with boundaries at multiples of 64 envs (whole workgroups, 16-byte aligned rows)
    const int chunk = (((desc->batch + n_sub - 1) / n_sub) + 63) & ~63;
    Params sub[16];
    int n = 0;
    for (int i = 0; i < n_sub; ++i) {
        const int start = i * chunk;
        if (start >= desc->batch) break;
        sub[n++] = sub_batch(p, start, desc->batch - start < chunk ? desc->batch - start : chunk);
    }
    if (n > 1 && (rc = plan<kRollout1>(sub[0], lp))) return rc;         // the kernel form goes by the size of a LAUNCH (two_per_wave)
    const bool fork_join = (flags & JSS_ROLLOUT_FORK_JOIN) != 0 && n > 1;
    ForkJoinEvents *ev = nullptr;
    if (fork_join && ((rc = events_for(streams[0], &ev)) || (rc = fork_streams(*ev, streams, n)))) return rc;
    for (int s = 0; s < n_steps && !rc; ++s)
        for (int i = 0; i < n && !rc; ++i) rc = fire(sub[i], lp, streams[i]);
    const int jrc = fork_join ? join_streams(*ev, streams, n) : 0;
    return rc ? rc : jrc;
}

int jss_policy_step_steps(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                          uint32_t explore_q16, int32_t *actions, int32_t n_steps, int32_t flags, int32_t n_sub,
                          void *const *streams) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if ((rc = check_kind(desc, kind, true))) return rc;
    if (n_steps < 0 || n_sub < 1 || n_sub > 16) return JSS_E_SHAPE;
    if (!streams || !actions) return JSS_E_NULL;
    if (n_steps == 0) return 0;
    Params pp = {}, ps = {};
    pp.d = *desc; pp.s = *state; pp.actions_out = actions; pp.kind = kind; pp.seed = seed; pp.explore_q16 = explore_q16;
    ps.d = *desc; ps.s = *state; ps.o = *out; ps.actions = actions; ps.flags = flags & JSS_ROLLOUT_AUTORESET;
    LaunchPlan lpp, lps;
    if ((rc = plan<kPolicy>(pp, lpp)) || (rc = plan<kStep>(ps, lps))) return rc;
    const int chunk = (((desc->batch + n_sub - 1) / n_sub) + 63) & ~63;       // whole workgroups, 16-byte aligned rows
    Params subp[16], subs[16];
    int n = 0;
    for (int i = 0; i < n_sub; ++i) {
        const int start = i * chunk;
        if (start >= desc->batch) break;
        const int count = desc->batch - start < chunk ? desc->batch - start : chunk;
        subp[n] = sub_batch(pp, start, count);
        subp[n].actions_out = actions + start;
        subs[n] = sub_batch(ps, start, count);
        subs[n].actions = actions + start;
        ++n;
    }
    if (n > 1 && (rc = plan<kStep>(subs[0], lps))) return rc;           // the kernel form goes by the size of a LAUNCH (two_per_wave)
    const bool fork_join = (flags & JSS_ROLLOUT_FORK_JOIN) != 0 && n > 1;
    ForkJoinEvents *ev = nullptr;
    if (fork_join && ((rc = events_for(streams[0], &ev)) || (rc = fork_streams(*ev, streams, n)))) return rc;
    for (int s = 0; s < n_steps && !rc; ++s)
        for (int i = 0; i < n && !rc; ++i) {
            rc = fire(subp[i], lpp, streams[i]);
            if (!rc) rc = fire(subs[i], lps, streams[i]);
        }
    const int jrc = fork_join ? join_streams(*ev, streams, n) : 0;
    return rc ? rc : jrc;
}

int jss_multi_reset(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const JssOut *const *outs,
                    const uint8_t *const *which, void *stream) {
    int rc = check_multi(n_sets, descs, states, outs, true);
    if (rc) return rc;
    Params ps[16];
    for (int i = 0; i < n_sets; ++i) {
        ps[i] = {};
        ps[i].d = *descs[i]; ps[i].s = *states[i]; ps[i].o = *outs[i]; ps[i].which = which ? which[i] : nullptr;
    }
    return launch_multi<kReset>(ps, n_sets, 1, 1, &stream, false);
}

int jss_multi_step(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const int32_t *const *actions,
                   const JssOut *const *outs, int32_t flags, void *stream) {
    int rc = check_multi(n_sets, descs, states, outs, true);
    if (rc) return rc;
    if (!actions) return JSS_E_NULL;
    Params ps[16];
    for (int i = 0; i < n_sets; ++i) {
        if (!actions[i]) return JSS_E_NULL;
        ps[i] = {};
        ps[i].d = *descs[i]; ps[i].s = *states[i]; ps[i].o = *outs[i]; ps[i].actions = actions[i];
        ps[i].flags = flags & JSS_ROLLOUT_AUTORESET;
    }
    return launch_multi<kStep>(ps, n_sets, 1, 1, &stream, false);
}

int jss_multi_policy(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, int kind, uint64_t seed,
                     uint32_t explore_q16, int32_t *const *actions, void *stream) {
    int rc = check_multi(n_sets, descs, states, nullptr, false);
    if (rc) return rc;
    if (!actions) return JSS_E_NULL;
    Params ps[16];
    for (int i = 0; i < n_sets; ++i) {
        if (!actions[i]) return JSS_E_NULL;
        if ((rc = check_kind(descs[i], kind, true))) return rc;
        ps[i] = {};
        ps[i].d = *descs[i]; ps[i].s = *states[i]; ps[i].actions_out = actions[i]; ps[i].kind = kind; ps[i].seed = seed;
        ps[i].explore_q16 = explore_q16;
    }
    return launch_multi<kPolicy>(ps, n_sets, 1, 1, &stream, false);
}

int jss_multi_rollout(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const JssOut *const *outs,
                      int kind, uint64_t seed, uint32_t explore_q16, int32_t n_steps, int32_t flags, int32_t n_sub,
                      void *const *streams) {
    int rc = check_multi(n_sets, descs, states, outs, true);
    if (rc) return rc;
    if (n_steps < 0 || n_sub < 1 || n_sub > 16) return JSS_E_SHAPE;
    if (!streams) return JSS_E_NULL;
    for (int i = 0; i < n_sets; ++i)
        if ((rc = check_kind(descs[i], kind))) return rc;
    if (n_steps == 0) return 0;                       // no step: nothing is launched, nothing is touched (both libraries)
    Params ps[16];
    for (int i = 0; i < n_sets; ++i) {
        ps[i] = {};
        ps[i].d = *descs[i]; ps[i].s = *states[i]; ps[i].o = *outs[i]; ps[i].kind = kind; ps[i].seed = seed;
        ps[i].explore_q16 = explore_q16; ps[i].n_iter = 1; ps[i].flags = flags & JSS_ROLLOUT_AUTORESET;
    }
    return launch_multi<kRollout1>(ps, n_sets, n_steps, n_sub, streams, (flags & JSS_ROLLOUT_FORK_JOIN) != 0);
}

int jss_rollout_steps_multi(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states,
                            const JssOut *const *outs, int kind, uint64_t seed, uint32_t explore_q16, int32_t n_steps,
                            int32_t flags, void *const *streams) {
    if (!descs || !states || !outs || !streams) return JSS_E_NULL;
    if (n_sets < 1 || n_sets > 16 || n_steps < 0) return JSS_E_SHAPE;
    Params ps[16];
    LaunchPlan lps[16];
    const bool nothing = n_steps == 0;                // (arguments are still checked)
    for (int i = 0; i < n_sets; ++i) {
        int rc = check_args(descs[i], states[i], outs[i], true);
        if (rc) return rc;
        if ((rc = check_kind(descs[i], kind))) return rc;
        Params &p = ps[i];
        p = {};
        p.d = *descs[i]; p.s = *states[i]; p.o = *outs[i]; p.kind = kind; p.seed = seed; p.explore_q16 = explore_q16;
        p.n_iter = 1; p.flags = flags & ~JSS_ROLLOUT_FORK_JOIN;
        if ((rc = plan<kRollout1>(p, lps[i]))) return rc;
    }
    if (nothing) return 0;
    const bool fork_join = (flags & JSS_ROLLOUT_FORK_JOIN) != 0 && n_sets > 1;
    ForkJoinEvents *ev = nullptr;
    int rc = 0;
    if (fork_join && ((rc = events_for(streams[0], &ev)) || (rc = fork_streams(*ev, streams, n_sets)))) return rc;
    for (int s = 0; s < n_steps && !rc; ++s)
        for (int i = 0; i < n_sets && !rc; ++i) rc = fire(ps[i], lps[i], streams[i]);
    const int jrc = fork_join ? join_streams(*ev, streams, n_sets) : 0;
    return rc ? rc : jrc;
}

}  // extern "C"
