// jssenv_amd/csrc/jss_common.hpp -- parameters, wave helpers, counter RNG shared by the two
// kernel flavours of the MI355X (gfx950 / CDNA4) batched Job-Shop-Scheduling simulator.
// Interface and layout: include/jss_hip.h.
//
// Execution model
//   * wave flavour (jss_wave_env.hpp): one 64-lane wavefront simulates one env; job j sits on
//     lane j%64 (slot j/64, JPL = 1 or 2 slots), machine m on lane m.  Packed flavour
//     (jss_packed_env.hpp): 64/G envs per wavefront, G = 16 or 32 lanes per env.
//   * everything a step needs to know about its env sits in the env's 16-byte header (clock, episode, step,
//     status) and its 48-byte constants record (J, M, the observation's normalisers, the op table index:
//     written by reset, JssState.env_const): no env -> instance -> shape chain of dependent loads in front
//     of the state, no second fetch behind it.
//   * the env's whole state lives in registers for the duration of the call: 8 int32 per job,
//     which include the job's next THREE ops (current, next, and the one after it in the spare
//     bits of word 0), so that a step touches the op table only when a job moves on to a new op
//     or a look-ahead walk goes further than three ops.
//   * step()'s `while nothing is legal: increase_time_step()` loops are one closed-form jump to
//     the first time a job becomes legal (p_jump / jump); unchanged halves of a job record and
//     unchanged machine clocks are not written back; the observation leaves with streaming stores.
//   * the op table (machine << 16 | duration) of a batch that shares ONE instance is staged in
//     LDS once per workgroup (kTabLds); batches with one instance per env or an env -> instance
//     map read the few entries they need straight from global memory (kTabGlobal): staging a
//     2-8 KB table per env per step was 1.4-1.8x the algorithmic traffic (profiles/README.md).
//   * addresses are wave-uniform 64-bit bases (SGPR pairs) + 32-bit lane offsets, so every
//     access is a `global_load/store ... v_off, s[base:base+1]`.
//   * cross-lane work: __ballot for every "for job in range(J)" predicate of the reference,
//     DPP row reductions for the next event time, readlane / ds_bpermute for the (<= 4) legal
//     jobs the order-dependent pass of _check_no_op walks through.
//   * no MFMA: the path is integer indexing, there is no dense contraction.
//
// Semantics follow the reference JSSEnv/envs/jss_env.py (cited per function) in the
// queue-free form: the reference's sorted event list `next_time_step` always equals
// the distinct values {t + tm[m] : tm[m] > 0}, its M x J `illegal_actions` matrix
// equals blocked[j] && need[j] == m, and nb_legal_actions / nb_machine_legal /
// machine_legal are functions of the legal set (tests/ checks all of this against
// the oracle step by step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jss_hip.h"

namespace jss {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kBig = 0x3fffffff;
constexpr int kDurMask = 0xffff;

// kRollout1 = kRollout with n_iter == 1 compiled loop-free (fewer live registers: the benchmarked
// one-launch-per-env-step path)
// kTraj = kRollout that also records every iteration's transition (JssTraj)
// kSteps = n_iter x kStep per launch with the actions given up front ([K][B]), optionally recording every step (JssTraj)
// kSession = kStep's semantics inside the resident step-session kernel (outputs write-through, counters tallied in
//            registers until the session closes); only the session kernels are instantiated with it
enum Mode { kReset = 0, kStep = 1, kAdvance = 2, kPolicy = 3, kRollout = 4, kRollout1 = 5, kTraj = 6, kSteps = 7, kSession = 8 };
// where the op table lives: LDS (one instance shared by the batch) or global memory; kTabLdsC = LDS + compact 16-byte
// job records (the three cached ops are re-read from the LDS table, the machine clocks rebuilt from the records)
// kTabGlobalM = global memory + 24-byte medium records (packed kernels only: jobs, machines <= 32; the three cached ops in 21
// bits each, machine clocks rebuilt from the records)
enum Tab { kTabLds = 0, kTabGlobal = 1, kTabLdsC = 2, kTabGlobalM = 3 };
constexpr bool tab_in_lds(int tab) { return tab == kTabLds || tab == kTabLdsC; }
constexpr bool tab_global(int tab) { return !tab_in_lds(tab); }
constexpr bool tab_compact(int tab) { return tab == kTabLdsC; }
constexpr bool tab_medium(int tab) { return tab == kTabGlobalM; }
constexpr bool tab_no_clocks(int tab) { return tab_compact(tab) || tab_medium(tab); }   // machine clocks rebuilt from the job records
constexpr int tab_record_ints(int tab) { return tab_compact(tab) ? JSS_NFC : tab_medium(tab) ? JSS_NFM : JSS_NF; }

struct Params {
    JssDesc d;
    JssState s;
    JssOut o;
    JssTraj t;               // kTraj
    const int32_t *actions;  // kStep
    int32_t *actions_out;    // kPolicy
    const uint8_t *which;    // kReset / kAdvance
    int32_t *hole;           // kAdvance
    uint64_t seed;
    uint32_t explore_q16;
    int32_t kind;
    int32_t n_iter;
    int32_t flags;
    int32_t region_ints;     // ints of one op table (jmax * mmax)
    int32_t table_lds_ints;  // LDS ints reserved for the shared op table (0 with kTabGlobal), multiple of 4
    int32_t obs_wave_floats; // floats of one wave's observation image in LDS (multiple of 4)
    int32_t mv_off_ints;     // packed kernel: LDS offset (ints) of the per-lane max_horizon_machine table
    int32_t norm_off_ints;   // packed kernel, kTabGlobal: LDS offset (ints) of the per-group observation normalisers
    int32_t ablate;          // JSS_PROFILING builds: JSS_PROF_ABLATE mask; 0 otherwise
    // step session (jss_session_open)
    const unsigned long long *mail;   // [depth][B] action granules: (step + 1) << 32 | action
    int32_t *progress;                // one word per wavefront: steps published
    int32_t *status;                  // [4] JssSession.status
    long long timeout_ticks;          // bound of a mail wait, in ticks of the 100 MHz wall clock
    int32_t depth;                    // ring depth in steps
    int32_t slots;                    // env sets per wavefront (1, 2, 4, 8)
    int32_t park_off_ints;            // LDS offset (ints) of the parked env sets: [wave][slot][kParkInt4][64 lanes] int4
    int32_t norm_slot_ints;           // packed kernel, kTabGlobal: ints between two slots' normaliser tables
#ifdef JSS_PROFILING
    unsigned long long *stamps;       // instrumented builds: [B][16] shader-clock stamps of the one-wavefront-per-env kernels (JSS_STAMP)
#endif
};

// Where a kernel reads its Params from.  They arrive by value (the kernel-argument segment); handed on by reference, a
// by-value struct is first copied into a private object, which the optimiser turns into "every field loaded in the entry
// block": some 60 scalar registers live from the first instruction to each field's last use.  That -- not the algorithm --
// is what makes the one-wavefront-per-env kernels overrun their SGPR budget (16-58 SGPR values parked in VGPR lanes in the
// step / rollout kernels, up to 236 and some scratch in the trajectory ones).  Read IN PLACE, through a pointer to the segment,
// a field is a scalar load next to its use and the spills are gone: jss_kernel<2, kRollout1, kTabGlobal> 54 -> 0 spilled
// SGPRs, <1, kStep, kTabGlobal> 16 -> 0, every kRollout1 / kStep instantiation 0 at unchanged occupancy (tools/kernel_resources.py
// on a -DJSS_PARAMS_ALL_IN_PLACE build).  It does not pay: the entry-block loads wait once, behind the state loads that
// every wave waits for anyway, while an in-place load waits in the middle of the dependent chain.  Same-box A/B, round 5
// (profiles/r05_misc/ab_params_in_place.txt): config 5 padded rollout +-0 %, its jss_step -3 %, config 4's share -2 %,
// config 3's packed kernel -10 %, headline +-0.  A spilled SGPR is two VALU-lane moves among ~800 instructions; the wait
// is what costs.  So the one-step kernels keep the by-value form.  The exception are the one-wavefront-per-env kernels that
// LOOP over steps with the state in registers (kTraj, kRollout, kSteps): by value every field stays live through the whole
// loop (the recorder's two-jobs-per-lane form even keeps 14-29 VGPRs in scratch memory), and there reading in place wins --
// trajectory mode +6.5 % on config 4's share, +10 % on config 5 padded, the 64-iteration rollout +6 % / +3 %, jss_steps +2.5 %
// (profiles/r05_misc/ab_params_in_place.txt).  The packed flavour's looping kernels: see jss_packed_kernel.  The fused multi-set grid (jss_multi_kernel) copies its set's Params up front for the
// same reason the one-step kernels take them by value (+3 % over in place).
// The explicit arguments start at offset 0 of the segment.  (Host pass and the test emulator: the argument itself.
// -DJSS_PARAMS_ALL_IN_PLACE: A/B builds.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JSS_PARAMS_BY_VALUE)
#ifdef JSS_PARAMS_ALL_IN_PLACE
#define JSS_PARAMS_OF(p, arg, in_place) \
    const Params &p = *reinterpret_cast<const Params *>(__builtin_amdgcn_kernarg_segment_ptr())
#else
#define JSS_PARAMS_OF(p, arg, in_place) \
    const Params &p = (in_place) ? *reinterpret_cast<const Params *>(__builtin_amdgcn_kernarg_segment_ptr()) : (arg)
#endif
#else
#define JSS_PARAMS_OF(p, arg, in_place) const Params &p = (arg)
#endif

// Instrumented builds: JSS_STAMP(p, env, k, dep) = lane 0 of env's wavefront records the shader clock in slot k once `dep` (a
// value the phase before it produced) is available -- a phase timeline of the wave's life (tools/gpu_wave_timeline.py).
#if defined(JSS_PROFILING) && defined(__HIP_DEVICE_COMPILE__)
#define JSS_STAMP(p, env, k, dep)                                                                  \
    do {                                                                                           \
        if ((p).stamps) {                                                                          \
            unsigned long long jss_t_;                                                             \
            asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(jss_t_) : "v"(dep) : "memory"); \
            if ((threadIdx.x & 63) == 0) (p).stamps[(size_t)(env) * 16 + (k)] = jss_t_;            \
        }                                                                                          \
    } while (0)
#else
#define JSS_STAMP(p, env, k, dep) do {} while (0)
#endif

#ifdef JSS_PROFILING
#define JSS_ABLATED(p, bit) (((p).ablate & (bit)) != 0)
#else
#define JSS_ABLATED(p, bit) false
#define JSS_ABLATE_CHECK_NO_OP 1
#define JSS_ABLATE_PRIORITIZE 2
#define JSS_ABLATE_OBS 4
#define JSS_ABLATE_SELECT 8
#define JSS_ABLATE_ADVANCE 16
#endif

// envs between slot k and slot k + 1 of the step-major [K][B] buffers (JssTraj.stride: a call on a range of a larger batch)
__device__ __forceinline__ size_t traj_stride(const Params &p) { return p.t.stride ? (size_t)p.t.stride : (size_t)p.d.batch; }

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// a / b for small non-negative integers as float32, given rb = fl(1 / b) (correctly rounded, from the
// instance record): quotient estimate and one residual correction -- the core of the IEEE division
// sequence without its range scaling (operands here are far from denormal/overflow).  |error| <= 1 ulp,
// i.e. < 1.2e-7 on values in [0, 1]: inside the 1e-6 budget of the float observation.
__device__ __forceinline__ float div_by(float a, float b, float rb) {
    const float q = a * rb;
    return __builtin_fmaf(__builtin_fmaf(-q, b, a), rb, q);
}
__device__ __forceinline__ float as_float(int bits) {
    union { int i; float f; } u;
    u.i = bits;
    return u.f;
}
__device__ __forceinline__ int as_int(float x) {
    union { int i; float f; } u;
    u.f = x;
    return u.i;
}

// A wave-uniform value that only vector instructions consume (the observation's normalisers): moved out of the scalar
// register file once, so that it does not sit in SGPRs -- the scarce resource of the one-wavefront-per-env kernels --
// from the first load to the last store.  (Host pass and the test emulator: the value itself.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ int in_vgpr(int x) {
    int r;
    asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(x));
    return r;
}
#else
__device__ __forceinline__ int in_vgpr(int x) { return x; }
#endif

// Every control used below (quad_perm, row_half_mirror, row_mirror) reads a valid lane of the row, so bound_ctrl never
// takes effect -- but only with it set does the compiler fold the move into its consumer (v_min_i32_dpp ... instead of
// v_mov_b32_dpp + v_min_i32): a row reduction is 4 instructions instead of 8.
#define JSS_DPP(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xF, 0xF, true)

// min / max / or over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ int row_min(int v) {
    v = imin(v, JSS_DPP(v, 0xB1));   // quad_perm [1,0,3,2]   lane ^ 1
    v = imin(v, JSS_DPP(v, 0x4E));   // quad_perm [2,3,0,1]   lane ^ 2
    v = imin(v, JSS_DPP(v, 0x141));  // row_half_mirror       quads 0<->1, 2<->3 (quads are uniform by now)
    v = imin(v, JSS_DPP(v, 0x140));  // row_mirror            halves of the row
    return v;
}
__device__ __forceinline__ int row_max(int v) {
    v = imax(v, JSS_DPP(v, 0xB1));
    v = imax(v, JSS_DPP(v, 0x4E));
    v = imax(v, JSS_DPP(v, 0x141));
    v = imax(v, JSS_DPP(v, 0x140));
    return v;
}
__device__ __forceinline__ int row_or(int v) {
    v |= JSS_DPP(v, 0xB1);
    v |= JSS_DPP(v, 0x4E);
    v |= JSS_DPP(v, 0x141);
    v |= JSS_DPP(v, 0x140);
    return v;
}

__device__ __forceinline__ int row_sum(int v) {
    v += JSS_DPP(v, 0xB1);
    v += JSS_DPP(v, 0x4E);
    v += JSS_DPP(v, 0x141);
    v += JSS_DPP(v, 0x140);
    return v;
}

// wave-wide: four row results combined on the scalar unit
__device__ __forceinline__ int wave_min(int v) {
    v = row_min(v);
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return imin(imin(a, b), imin(c, d));
}
__device__ __forceinline__ int wave_max(int v) {
    v = row_max(v);
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return imax(imax(a, b), imax(c, d));
}

__device__ __forceinline__ int wave_sum(int v) {
    v = row_sum(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// the batch's one shared op table -> LDS, whole workgroup (dwordx4 when the table is 16-byte sized)
__device__ __forceinline__ void stage_shared_table(int32_t *dst, const int32_t *src, int n, int tid) {
    if ((n & 3) == 0) {
        const int4 *s4 = reinterpret_cast<const int4 *>(src);
        int4 *d4 = reinterpret_cast<int4 *>(dst);
        for (int i = tid; i < (n >> 2); i += kBlock) d4[i] = s4[i];
    } else {
        for (int i = tid; i < n; i += kBlock) dst[i] = src[i];
    }
}

// counters[4] += (steps, episodes, makespans, reward numerators) as result-less atomics: a plain `+=` is
// a load the wave has to wait for (~600 cycles at the end of every wave); these are fire-and-forget.
__device__ __forceinline__ void add_counters(int64_t *cn, int steps, int episodes, int makespan_sum, int reward_num) {
#ifdef JSS_EXP_NO_COUNTERS   // A/B builds only: what the per-env counters cost (read the answer off ms_per_step: the rates need them)
    return;
#endif
    unsigned long long *u = reinterpret_cast<unsigned long long *>(cn);
    if (steps) atomicAdd(u + 0, (unsigned long long)(long long)steps);
    if (episodes) atomicAdd(u + 1, (unsigned long long)(long long)episodes);
    if (makespan_sum) atomicAdd(u + 2, (unsigned long long)(long long)makespan_sum);
    if (reward_num) atomicAdd(u + 3, (unsigned long long)(long long)reward_num);   // two's complement: negative adds wrap correctly
}

// Write-through ("sc1") stores: the store leaves the XCD's L2 for memory as it is executed, so that a kernel of ANOTHER
// launch (or another XCD's L2) finds it without this kernel ending or fencing -- how the resident step-session kernel
// publishes a step's outputs (MI355X_MICROARCH.md, inter-workgroup visibility: 16-byte sc1 stores cost what plain ones
// do).  <= 8 bytes: a relaxed agent-scope atomic store IS an sc1 store; 16 bytes: the instruction itself.  The wave
// orders them before its progress word with wt_drain().
// (16 bytes go out as a buffer store with the sc1 bit -- a builtin, so that the compiler's s_waitcnt bookkeeping counts
//  it: an inline-asm store is invisible to it, and every later wait for a LOAD then also drained the newest stores, one
//  memory round trip per step in the session kernels)
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned jss_wt_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wt_store16(void *uniform_base, unsigned byte_off, float4 v) {
    const jss_wt_v4u x = {(unsigned)as_int(v.x), (unsigned)as_int(v.y), (unsigned)as_int(v.z), (unsigned)as_int(v.w)};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_base, (short)0, -1, 0x00020000);   // raw, unbounded
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, (int)byte_off, 0, 16);                                             // aux 16 = sc1
}
__device__ __forceinline__ void wt_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void wt_store16(void *uniform_base, unsigned byte_off, float4 v) {
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(uniform_base) + byte_off) = v;
}
__device__ __forceinline__ void wt_drain() {}
#endif
template <class T>
__device__ __forceinline__ void wt_store(T *p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a word another launch writes while this kernel runs (mailbox granules, progress words): never from L1, never cached in a register
template <class T>
__device__ __forceinline__ T fresh_load(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS writes of one wave consumed by other lanes of the same wave
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Counter RNG, identical to oracle/jss_oracle.c orc_rng_u32: the key words are combined with odd
// multipliers (independent multiplies, issued back to back) and pushed through two rounds of a 32-bit
// finaliser.  32-bit and short on purpose: the packed kernel evaluates it per lane on the VALU and the
// dependent chain of the policy sits on every wave's critical path (profiles/README.md).
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t step) {
    const uint32_t a = (uint32_t)seed + (uint32_t)env_id * 0x9E3779B9u + episode * 0x85EBCA6Bu + step * 0xC2B2AE35u;
    const uint32_t b = (uint32_t)(seed >> 32) ^ ((uint32_t)(env_id >> 32) * 0x27D4EB2Fu);
    return fmix32(fmix32(a) ^ b);
}
// CriticalRatio (dispatching.py:365-408): ratio = (factor * job_length - now) / remaining_work, smallest first, lowest
// job index on ties; factor = p / q, by default 3 / 2.  Compared exactly as the fraction (p * job_length - q * now) / remaining_work by cross
// multiplication in int64: two different such fractions differ by > 1e-13 relative, far above a double's 1.1e-16, so
// the order (and the ties) are the ones the reference's float comparison sees.
struct CrKey {
    int num, den, idx;
};
// due-date factor p / q carried in the `kind` argument (include/jss_hip.h JSS_POLICY_CR_FACTOR); none = the default 3 / 2
__device__ __forceinline__ int cr_p(const Params &p) { return ((p.kind >> 8) & 0xFF) ? ((p.kind >> 8) & 0xFF) : 3; }
__device__ __forceinline__ int cr_q(const Params &p) { return ((p.kind >> 8) & 0xFF) ? ((p.kind >> 16) & 0xFF) : 2; }
__device__ __forceinline__ bool cr_better(const CrKey &a, const CrKey &b) {
    const long long l = (long long)a.num * b.den, r = (long long)b.num * a.den;
    return l < r || (l == r && a.idx < b.idx);
}
template <int WIDTH>
__device__ __forceinline__ CrKey cr_argmin(CrKey k) {   // butterfly inside aligned groups of WIDTH lanes
#pragma unroll
    for (int off = WIDTH / 2; off > 0; off >>= 1) {
        CrKey o;
        o.num = __shfl_xor(k.num, off);
        o.den = __shfl_xor(k.den, off);
        o.idx = __shfl_xor(k.idx, off);
        if (cr_better(o, k)) k = o;
    }
    return k;
}
constexpr int kCrNone = 1 << 20;

// JSS_POLICY_CR_F64: the reference's float64 expression itself (dispatching.py:351-363, :391-398) -- due date = length *
// factor, ratio = (due date - now) / remaining -- every operation rounded on its own (no fused multiply-add), so that the
// doubles, their order and their ties are the reference's.  Smallest ratio first, lowest job index on ties (strict `<`, :399).
// (ROCm's __dmul_rn / __dsub_rn are plain `*` / `-`, and hipcc's default -ffp-contract=fast-honor-pragmas fuses them into one
//  v_fma_f64 -- a single rounding, one ulp away from the reference whenever `now` is close to the due date: length 100,
//  factor 1.2, now 120 gives 0.0 unfused and -4.4e-15 fused.  The pragma has to sit in THIS body: the operations are inlined
//  from here, a pragma at the call site does not reach them.  parity_cases.case_cr_f64_unfused holds that state.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double cr_ratio_f64(int length, double factor, int now, int remaining) {
#pragma clang fp contract(off)
    const double due = (double)length * factor;
    const double left = due - (double)now;
    return left / (double)remaining;
}
#else
__device__ __forceinline__ double cr_ratio_f64(int length, double factor, int now, int remaining) {
    volatile double due = (double)length * factor;      // (volatile: no contraction on the host pass either)
    volatile double left = due - (double)now;
    return left / (double)remaining;
}
#endif
struct CrKeyF {
    double ratio;
    int idx;
};
__device__ __forceinline__ bool cr_better_f64(const CrKeyF &a, const CrKeyF &b) {
    return a.ratio < b.ratio || (a.ratio == b.ratio && a.idx < b.idx);
}
template <int WIDTH>
__device__ __forceinline__ CrKeyF cr_argmin_f64(CrKeyF k) {   // butterfly inside aligned groups of WIDTH lanes
#pragma unroll
    for (int off = WIDTH / 2; off > 0; off >>= 1) {
        CrKeyF o;
        union { double d; int w[2]; } mine, theirs;
        mine.d = k.ratio;
        theirs.w[0] = __shfl_xor(mine.w[0], off);
        theirs.w[1] = __shfl_xor(mine.w[1], off);
        o.ratio = theirs.d;
        o.idx = __shfl_xor(k.idx, off);
        if (cr_better_f64(o, k)) k = o;
    }
    return k;
}
constexpr double kCrInf = __builtin_huge_val();

constexpr uint64_t kExploreSeedXor = 0x5851F42D4C957F2DULL;

}  // namespace jss
