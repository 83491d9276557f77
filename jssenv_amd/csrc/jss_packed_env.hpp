#pragma once
// Packed kernel: 64/G envs per wavefront, G = 16 or 32 lanes per env (J <= G, M <= G).
//
// On ta01-shaped instances (15 x 15) a wave-per-env kernel keeps 15 of 64 lanes busy and is
// instruction-issue bound (profiles/r01_wave_per_env).  Here every env owns an aligned group of
// G lanes: job j on group lane j, machine m on group lane m, and everything the wave kernel
// keeps wave-uniform in SGPRs (clock, legal set, loop conditions) becomes group-uniform values
// replicated in VGPRs.  Cross-lane traffic stays inside a group:
//   * reductions (next event time, arg-best of a rule) are DPP row operations -- a 16-lane DPP
//     row is exactly one env at G = 16; G = 32 adds one ds_swizzle (lane ^ 16);
//   * "for job in range(J)" predicates are one __ballot, each lane then shifts out its group's
//     bits;
//   * reading job a's / machine m's register is one ds_bpermute inside the group.
// Data-dependent loops (`while nb_machine_legal == 0: increase_time_step()`) run while ANY group
// of the wave still needs them, with the state updates predicated per group.
//
// Semantics and citations are the same as jss_wave_env.hpp (reference JSSEnv/envs/jss_env.py).
#include "jss_common.hpp"

namespace jss {

template <class T>
__device__ __forceinline__ T ld_off(const void *base, unsigned byte_off) {   // uniform base + 32-bit lane offset
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void st_off(void *base, unsigned byte_off, T v) {
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
// Streaming ("nt") store for the observation, the one large output stream that is written once per step in whole
// cache lines and never read back by a kernel: it should not displace state lines in L2.  Measured on the headline
// (profiles/README.md): 20.2 -> 17.7 us per step.  The same hint on the small outputs (1- and 4-byte stores of mask,
// reward, done) or on the state records makes things WORSE (partial lines go out uncombined; 17.7 -> 18.8 / 19.5 us),
// and nt state loads cost 25 %: those stay ordinary accesses.
typedef float jss_v4f __attribute__((vector_size(16)));
__device__ __forceinline__ void st_nt(void *base, unsigned byte_off, float4 v) {
    const jss_v4f x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<jss_v4f *>(reinterpret_cast<char *>(base) + byte_off));
}

// an output store: plain, or write-through in the resident step-session kernel (jss_common.hpp wt_store)
template <bool WT, class T>
__device__ __forceinline__ void st_out(void *base, unsigned byte_off, T v) {
    if (WT) wt_store(reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off), v);
    else st_off<T>(base, byte_off, v);
}

template <int G, int TAB>
struct PCtx {                 // per-lane view of "my env"
    int lane, gl, gbase;      // gl = lane within the group, gbase = first lane of the group
    unsigned rel;             // my env's index within the wave's env set (clamped like b)
    int first_env;            // wave-uniform: first env of the wave's set
    bool alive;               // my env exists (first_env + lane / G < batch)
    bool jvalid, mvalid;      // gl < J, gl < M
    int tid;                  // instance of my env
    int J, M, max_time_op;
    // the observation's normalisers: wave-uniform registers with kTabLds (one instance for the whole batch); with
    // kTabGlobal they differ per group and wait in LDS (`norm`: 6 ints per group, written when the kernel starts)
    int max_time_jobs, sum_op;
    float r_op, r_jobs, r_sum, r_m;
    int32_t *norm;
    const int32_t *lds_row;   // kTabLds: op table row of my job in LDS
    // op table row of my job.  With kTabGlobal the 64-bit address is rebuilt from `tid` at each of its (few) uses
    // rather than carried in two VGPRs through the whole kernel (these kernels are the register-hungry ones).
    __device__ __forceinline__ const int32_t *row(const Params &p) const {
        if (tab_in_lds(TAB)) return lds_row;
        return p.d.ops + (size_t)tid * p.region_ints + (gl < p.d.jmax ? gl : 0) * p.d.mmax;
    }
};

template <int G>
struct PEnv {
    int t;                    // group-uniform
    int todo, cur, nxt, nxt2, left, perf, idle, idle_last, f4;  // job gl (cur / nxt / nxt2: the job's next three ops, -1 = none)
    int tm;                   // machine gl
    bool legal, blocked;      // job gl
    int noop, err;            // group-uniform
};

template <int G>
__device__ __forceinline__ uint32_t grp_ballot(bool p, int gbase) {
    const uint64_t w = __ballot(p);
    const uint32_t x = (uint32_t)(w >> gbase);
    return G == 32 ? x : (x & 0xFFFFu);
}

template <int G>
__device__ __forceinline__ bool grp_any(bool p, int gbase) {
    return grp_ballot<G>(p, gbase) != 0;
}

// value of `v` on group lane `idx` (idx group-uniform or not, any lane may ask for any lane)
template <int G>
__device__ __forceinline__ int grp_read(int v, int idx, int gbase) {
#ifdef JSS_EXP_NO_GRPREAD   // A/B builds only (WRONG results): every cross-lane read answers the lane's own value -- what a pass
    return v;               // would cost if ALL of them were free (profiles/r05_misc/ablate_cross_lane.txt)
#endif
    return __builtin_amdgcn_ds_bpermute((gbase + (idx & (G - 1))) << 2, v);
}

template <int G>
__device__ __forceinline__ int grp_min(int v) {
    v = row_min(v);                                                       // a DPP row is one env at G = 16
    if (G == 32) v = imin(v, __builtin_amdgcn_ds_swizzle(v, 0x401F));     // lane ^ 16
    return v;
}

template <int G>
__device__ __forceinline__ int grp_max(int v) {
    v = row_max(v);
    if (G == 32) v = imax(v, __builtin_amdgcn_ds_swizzle(v, 0x401F));
    return v;
}

template <int G>
__device__ __forceinline__ int grp_sum(int v) {
    v = row_sum(v);
    if (G == 32) v += __builtin_amdgcn_ds_swizzle(v, 0x401F);
    return v;
}

// ---------------------------------------------------------------------------------------
// reset(): jss_env.py:145-181 (registers only; `on` = groups being reset)
// ---------------------------------------------------------------------------------------
template <int G, int TAB, bool WT = false>
__device__ __forceinline__ void p_reset(PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, bool on) {
    if (on) {
        e.t = 0;                                                         // :154
        e.tm = 0;                                                        // :164
        e.noop = 0;                                                      // :161
        e.err = 0;
        e.todo = 0;                                                      // :166
        e.cur = c.jvalid ? c.row(p)[0] : -1;                                // :174-176
        e.nxt = (c.jvalid && 1 < c.M) ? c.row(p)[1] : -1;
        e.nxt2 = (c.jvalid && 2 < c.M) ? c.row(p)[2] : -1;
        e.left = e.perf = e.idle = e.idle_last = 0;                      // :165-170
        e.f4 = 0;                                                        // :180
        e.legal = c.jvalid;                                              // :160
        e.blocked = false;                                               // :171-172
        if (c.alive) {                                                   // solution = -1 (:163), the whole padded block
            const unsigned n = (unsigned)(p.d.jmax * p.d.mmax);         // wave-uniform base + 32-bit lane offset
            int32_t *sol = p.s.solution + (size_t)c.first_env * n;
            for (unsigned i = c.gl; i < n; i += G) st_out<WT, int>(sol, (c.rel * n + i) * 4u, -1);
        }
    }
}

// ---------------------------------------------------------------------------------------
// increase_time_step(): jss_env.py:495-637 for the groups with `act`; returns hole_planning.
// ---------------------------------------------------------------------------------------
// time to the next event of my env = earliest machine release (:517-522; the reference's queue is
// {t + tm[m] : tm[m] > 0}); kBig when no machine is busy (empty queue)
template <int G>
__device__ __forceinline__ int p_next_event(const PEnv<G> &e) {
    return grp_min<G>(e.tm > 0 ? e.tm : kBig);
}

template <int G, int TAB>
__device__ __forceinline__ int p_advance(PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, bool act, int d) {
    const int idle_machines = __popc(grp_ballot<G>(c.mvalid && e.tm < d, c.gbase));
    const int hole = d * idle_machines;                                  // :606-608 (tm < d only when tm == 0)
    bool fin = false;
    if (act) {
        e.t += d;
        const int was = e.left;
        if (was > 0) {                                                   // :529 running
            const int nl = imax(0, was - d);                             // :534
            e.perf += imin(d, was);                                      // :531,:544
            e.left = nl;
            if (nl == 0) {                                               // :550 op finished
                e.idle += d - was;                                       // :552
                e.idle_last = d - was;                                   // :554
                e.todo += 1;                                             // :558
                fin = true;
            }
        } else if (c.jvalid && e.todo < c.M) {                           // :594 waiting
            e.idle += d;                                                 // :596
            e.idle_last += d;                                            // :597
        }
        e.tm = imax(0, e.tm - d);                                        // :611
        if (fin) {                                                       // :562-566 / :581: the job moves on to its next op,
            e.cur = e.nxt;                                               // which the record carries (cur <- nxt <- nxt2); the
            e.nxt = e.nxt2;                                              // op two further on is the step's only op table read
            e.nxt2 = (e.todo + 2 < c.M) ? c.row(p)[e.todo + 2] : -1;        // (not needed before a long look-ahead walk)
        }
    }
    // time left on the machine my job needs (after the update): feature-4 numerator
    // max(0, tm_old[need] - d) (:569-578) and the "machine is free" test of :616 in one read
    const int tm_need = grp_read<G>(e.tm, e.cur >> 16, c.gbase);
    if (fin) e.f4 = e.cur >= 0 ? tm_need : JSS_F4_ONE;                   // :586
    // re-legalisation :616-634: need[j] free, not legal, not blocked (a finished job has cur = -1)
    if (act && c.jvalid && e.cur >= 0 && tm_need == 0 && !e.blocked) e.legal = true;
    return hole;
}

// ---------------------------------------------------------------------------------------
// `while nb_legal_actions == 0 (and a machine is busy): increase_time_step()` (jss_env.py:429-430 / :469-470) in ONE
// jump for the groups with `want` (no legal job): nothing is allocated inside that loop, so every clock runs down
// linearly and the first time T at which a job becomes legal is known up front --
//   a running job with a next op becomes legal at max(its finish time, release time of its next machine),
//   a waiting, unblocked job when its machine is released,
// both of which are event times; T is their minimum.  All the events up to T are then applied at once:
//   job finishing at f = left <= T: perf += f, then waits: idle += T - f, idle_last = T - f (it was reset to 0 at the
//   finish, :554, where d - left == 0 because a machine and the job on it run down together), feature-4 numerator =
//   time its next machine still needed at f (:569-578); job still running: perf += T, left -= T; waiting job:
//   idle / idle_last += T (:596-597); machines: tm = max(0, tm - T), hole_planning = sum of max(0, T - tm) (:606-608).
// Two rare cases (one wave-uniform branch): a waiting job whose machine is already free ("orphan": suppressed by
// _prioritization_non_final, then left behind by a NOPE) is re-legalised by the reference at the very next event,
// so T = min(T, first event); and when no job can ever become legal again (the tail of an episode) the loop runs
// until every machine is idle, T = last event.  With nothing busy there is no event at all: a NOPE then is the
// reference's IndexError (:517) and is flagged.
// ---------------------------------------------------------------------------------------
template <int G, int TAB>
__device__ __forceinline__ void p_jump(PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, bool want, bool is_nope, int &rn) {
    const bool running = e.left > 0;
    const bool waiting = c.jvalid && !running && e.cur >= 0;             // cur >= 0  <=>  todo < M
    const int tmx = grp_read<G>(e.tm, (running ? e.nxt : e.cur) >> 16, c.gbase);   // release time of the machine I need (next)
    int cand = kBig;
    bool orphan = false;
    if (running) {
        if (e.nxt >= 0) cand = imax(e.left, tmx);
    } else if (waiting && !e.blocked) {
        if (tmx > 0) cand = tmx;
        else orphan = true;
    }
    int T = grp_min<G>(cand);
    bool fast = want;
    if (__ballot(want && (orphan || T >= kBig)) != 0) {                  // rare; collectives below run on every lane
        const int first = grp_min<G>(e.tm > 0 ? e.tm : kBig);            // first event (kBig: nothing busy)
        const int last = grp_max<G>(e.tm);                               // last event
        const bool any_orphan = grp_any<G>(orphan, c.gbase);
        if (any_orphan) T = imin(T, first);
        else if (T >= kBig) {                                            // nobody will ever be legal again: run out of
            T = last > 0 ? last : kBig;                                  // events; a NOPE then pops the reference's empty
            if (want && is_nope) e.err |= JSS_ERR_NOPE_IDLE;             // event list (:517)
        }
        if (T >= kBig) {                                                 // no event to advance to at all
            if (want && is_nope) e.err |= JSS_ERR_NOPE_IDLE;
            fast = false;
        }
    }
    const int hole = grp_sum<G>((fast && c.mvalid) ? imax(0, T - e.tm) : 0);      // :606-608 summed over the events
    if (fast) {
        rn -= hole;
        e.t += T;
        e.tm = imax(0, e.tm - T);                                        // :611
        if (running) {
            if (e.left <= T) {                                           // finishes at f = left (:550)
                const int f = e.left;
                e.perf += f;                                             // :531, :544
                e.left = 0;
                e.todo += 1;                                             // :558
                e.cur = e.nxt;                                           // :562-566 / :581
                e.nxt = e.nxt2;
                e.nxt2 = (e.todo + 2 < c.M) ? c.row(p)[e.todo + 2] : -1;
                const bool more = e.cur >= 0;
                e.idle += more ? T - f : 0;                              // :552 (+0 at the finish), then :596 per later event
                e.idle_last = more ? T - f : 0;                          // :554, then :597
                e.f4 = more ? imax(0, tmx - f) : JSS_F4_ONE;             // :569-586
                if (more && tmx <= T && !e.blocked) e.legal = true;      // :616-634 at T
            } else {
                e.perf += T;
                e.left -= T;
            }
        } else if (waiting) {
            e.idle += T;                                                 // :596
            e.idle_last += T;                                            // :597
            if (!e.blocked && tmx <= T) e.legal = true;                  // :616-634 at T
        }
    }
}

// ---------------------------------------------------------------------------------------
// _prioritization_non_final(): jss_env.py:183-254 for the groups with `on`
// ---------------------------------------------------------------------------------------
template <int G, int TAB>
__device__ __forceinline__ void p_prioritize(PEnv<G> &e, const PCtx<G, TAB> &c, bool on) {
    const bool fin = on && e.legal && e.todo == c.M - 1;                 // :217 final ops among legal jobs
    uint32_t bits = grp_ballot<G>(fin, c.gbase);
    if (__ballot(bits != 0) == 0) return;                                // nothing to suppress anywhere in the wave
    bool nf = false;
    int tm_next = 1;
    {
        const bool cand = on && e.legal && e.todo < c.M - 1;             // :219
        const int next_m = cand ? (e.nxt >> 16) : 0;                     // :227
        tm_next = grp_read<G>(e.tm, next_m, c.gbase);
        nf = cand && tm_next == 0;                                       // :234 next machine idle
    }
    const int my_m = e.cur >> 16, my_d = e.cur & kDurMask;
    while (__ballot(bits != 0) != 0) {                                   // :244 each final job of each group
        const int l = bits ? __ffs(bits) - 1 : 0;
        const int cf = grp_read<G>(e.cur, l, c.gbase);
        const int mf = cf >> 16, df = cf & kDurMask;
        // a non-final job on the same machine, strictly shorter: df > min_non_final (:252)
        const bool hit = grp_any<G>(bits != 0 && nf && my_m == mf && my_d < df, c.gbase);
        if (bits != 0 && hit && c.gl == l) e.legal = false;              // :253-254
        bits &= bits - 1;
    }
}

// ---------------------------------------------------------------------------------------
// _check_no_op(): jss_env.py:256-401 for the groups with `on`.
//
// Pass 1 of the reference walks the (<= 4) legal jobs in ascending index and keeps, per machine,
// max_horizon_machine[m] = min(t + max_time_op, end of every legal job on m seen so far), and
// max_horizon = max over the jobs of that running value at the job's own turn (order dependent).
// Here each lane keeps the piece it owns: a legal job lane its own running-prefix value `h`
// (min over the legal jobs with a lower or equal index on its machine), a machine lane its
// machine's final minimum `mv`.  One round per legal job broadcasts that job's (machine, end).
// Pass 2 looks max_horizon_machine up in a per-group LDS table (`mvtab`, one int per lane) because
// the walk is divergent, and collects the covered machines as a bit mask (M <= G <= 32).  The walk's
// first ops are the three the job record carries (current, next, the one after it); only a walk that goes
// further reads the op table (2-3 % of the walking lanes, tools/walk_depth.py).
// ---------------------------------------------------------------------------------------
template <int G, int TAB>
__device__ __forceinline__ void p_check_no_op(PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, bool on, int32_t *mvtab) {
    if (on) e.noop = 0;                                                  // :278
    const uint32_t lm = grp_ballot<G>(e.legal, c.gbase);
    const int nl = __popc(lm);
    const int d_next = p_next_event(e);                                  // also next_time_step[0] - t of :293
    const bool busy = d_next < kBig;                                     // :285 len(next_time_step) > 0
    bool gate = on && nl >= 1 && nl <= 4 && busy;                        // :284-288 (nb_machine_legal checked below)
    if (__ballot(gate) == 0) return;
    const int nxt = e.t + d_next;                                        // :293
    const int cap = e.t + c.max_time_op;                                 // :300-302
    const int my_m = e.cur >> 16;
    const int my_end = e.t + (e.cur & kDurMask);                         // :310
    // PASS 1 (:305-321)
    int h = imin(cap, my_end);   // legal job lane: max_horizon_machine[my_m] right after my own turn
    int mv = cap;                // machine lane: max_horizon_machine[gl] after the whole pass
    bool m_legal = false;        // machine lane: machine_legal[gl]
    // the (<= 4) legal jobs' current ops: four independent cross-lane reads in flight at once
    int cf[4];
    bool has[4];
    {
        uint32_t bits = lm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            has[r] = gate && bits != 0;
            const int l = bits ? __ffs(bits) - 1 : 0;
            cf[r] = grp_read<G>(e.cur, l, c.gbase);
            bits &= bits - 1;
        }
    }
    // time left on the machine my own job needs (:376), issued here so its latency hides behind pass 1
    const int tm_need = grp_read<G>(e.tm, my_m, c.gbase);
    {
        uint32_t bits = lm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = bits ? __ffs(bits) - 1 : 0;
            const int m_r = cf[r] >> 16, end_r = e.t + (cf[r] & kDurMask);
            if (has[r] && e.legal && l < c.gl && m_r == my_m) h = imin(h, end_r);   // :318 earlier job on my machine
            if (has[r] && c.gl == m_r) {                                            // :318 / machine_legal
                mv = imin(mv, end_r);
                m_legal = true;
            }
            bits &= bits - 1;
        }
    }
    const uint32_t legal_machines = grp_ballot<G>(m_legal, c.gbase);
    gate = gate && __popc(legal_machines) <= 3;                          // :286 nb_machine_legal <= 3
    const bool ends_early = grp_any<G>(e.legal && my_end < nxt, c.gbase);  // collective: evaluate on every lane
    gate = gate && !ends_early;                                          // :314-315 some legal job ends before the next event
    const int mh = grp_max<G>(e.legal ? h : e.t);                        // :296, :321 max_horizon
    // max_horizon_machine of the legal machines for the divergent walk; others never qualify (:348)
    mvtab[c.lane] = m_legal ? mv : -kBig;
    wave_lds_sync();
    // PASS 2 (:324-401): every illegal job walks its future ops
    const bool caseA = c.jvalid && !e.legal && e.left > 0 && e.todo + 1 < c.M;      // :327-330
    const bool caseB = c.jvalid && !e.legal && !caseA && !e.blocked && e.todo < c.M; // :366-369
    int k = caseA ? e.todo + 1 : e.todo;                                              // :332 / :370
    int tn = caseA ? e.t + e.left : e.t + tm_need;                                    // :334-337 / :374-377
    uint32_t u = 0;                                                                   // machine_next as a bit mask
    const int last = c.M - 1;
    const int32_t *tab = mvtab + c.gbase;
    // the loop of :340-363 / :380-401, its first iterations on the ops the record carries
    bool go = gate && (caseA || caseB) && k < last && mh > tn;
    if (go && caseB) {                                                                // op k == todo: the current op
        const int m = e.cur >> 16;
        if (tab[m] > tn) u |= 1u << m;                                                // :346-351
        tn += e.cur & kDurMask;                                                       // :362
        ++k;
        go = k < last && mh > tn;
    }
    if (go) {                                                                         // op k == todo + 1: the next op
        const int m = e.nxt >> 16;
        if (tab[m] > tn) u |= 1u << m;
        tn += e.nxt & kDurMask;
        ++k;
        go = k < last && mh > tn;
    }
    if (go) {                                                                         // op k == todo + 2: the one after it
        const int m = e.nxt2 >> 16;
        if (tab[m] > tn) u |= 1u << m;
        tn += e.nxt2 & kDurMask;
        ++k;
        go = k < last && mh > tn;
    }
    if (go) {                                                                         // further: the op table, two entries per trip
        do {
            const int32_t *row = c.row(p);
            const int op0 = row[k], op1 = row[k + 1];                             // k + 1 <= M - 1: inside the row
            int m = op0 >> 16;
            if (tab[m] > tn) u |= 1u << m;
            tn += op0 & kDurMask;
            ++k;
            if (k < last && mh > tn) {
                m = op1 >> 16;
                if (tab[m] > tn) u |= 1u << m;
                tn += op1 & kDurMask;
                ++k;
            }
        } while (k < last && mh > tn);
    }
    int covered = row_or((int)u);                                                     // union over the group
    if (G == 32) covered |= __builtin_amdgcn_ds_swizzle(covered, 0x401F);
    if (gate && (uint32_t)covered == legal_machines) e.noop = 1;                      // :357-359 / :395-397
    wave_lds_sync();                                                                  // mvtab is reused by the next call
}

// ---------------------------------------------------------------------------------------
// step(): jss_env.py:403-481.  `a` is group-uniform.  Returns the reward numerator.
// ---------------------------------------------------------------------------------------
template <int G, int TAB, bool WT = false>
__device__ __forceinline__ int p_step(PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, int a, int32_t *mvtab) {
    const bool is_nope = c.alive && a == c.J;                            // :419
    const bool is_job = c.alive && a >= 0 && a < c.J;
    if (c.alive && (a < JSS_ACTION_RESET || a > c.J)) e.err |= JSS_ERR_BAD_ACTION;   // SKIP / RESET never reach a job or NOPE branch
    const bool mine = c.gl == a;
    const bool a_legal = grp_any<G>(mine && e.legal, c.gbase);
    if (is_job && !a_legal) e.err |= JSS_ERR_ILLEGAL_ACTION;             // outside the mask: ignored + flagged
    const bool alloc = is_job && a_legal;
    const int ca = grp_read<G>(e.cur, a, c.gbase);
    const int m = ca >> 16, d = ca & kDurMask;                           // :443-444
    int rn = 0;
    if (alloc) {                                                         // :441 allocate job a
        rn = d;                                                          // :445
        if (c.gl == m) e.tm = d;                                         // :446
        if (mine) {
            e.left = d;                                                  // :447
            st_out<WT, int>(p.s.solution + (size_t)c.first_env * p.d.jmax * p.d.mmax,               // :454
                            (((unsigned)c.rel * p.d.jmax + a) * p.d.mmax + e.todo) * 4u, e.t);
        }
        if (e.cur >= 0 && (e.cur >> 16) == m) {
            e.legal = false;                                             // :455-463
            e.blocked = false;                                           // :464-467
        }
    }
    if (is_nope) {                                                       // :422-428
        e.blocked = e.blocked || e.legal;
        e.legal = false;
    }
    const bool stepping = alloc || is_nope;
    {   // :429-430 / :469-470: `while nb_legal_actions == 0: increase_time_step()` as one jump
        const bool none_legal = !grp_any<G>(e.legal, c.gbase);
        if (__ballot(stepping && none_legal) != 0 && !JSS_ABLATED(p, JSS_ABLATE_ADVANCE))
            p_jump(e, c, p, stepping && none_legal, is_nope, rn);
    }
    if (!JSS_ABLATED(p, JSS_ABLATE_PRIORITIZE)) p_prioritize(e, c, stepping);        // :432 / :471
    if (!JSS_ABLATED(p, JSS_ABLATE_CHECK_NO_OP)) p_check_no_op(e, c, p, stepping, mvtab);  // :433 / :472
    return rn;
}

// ---------------------------------------------------------------------------------------
// action selectors (group-uniform result; -1 when nothing is legal)
// ---------------------------------------------------------------------------------------
// F64: the instantiation also carries JSS_POLICY_CR_F64's float64 selector (the policy kernels only)
template <int G, int TAB, bool F64 = false>
__device__ __forceinline__ int p_select(const PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, uint64_t env_id,
                                        uint32_t episode, uint32_t step) {
    const int kind = p.kind & 0xFF;
    const uint32_t lm = grp_ballot<G>(e.legal, c.gbase);
    const int nl = __popc(lm);
    const int n = nl + (e.noop ? 1 : 0);
    int a;
    if (kind == JSS_POLICY_RANDOM) {
        const uint32_t r = rng_u32(p.seed, env_id, episode, step);
        const int pick = (int)__umulhi(r, (uint32_t)n);
        const int below = __popc(lm & ((1u << c.gl) - 1u));
        const uint32_t hit = grp_ballot<G>(e.legal && below == pick, c.gbase);
        a = hit ? __ffs(hit) - 1 : c.J;                                  // pick >= nl: NOPE (it is legal then)
    } else {
        // remaining-work table row of my job: rem[k] = durations of ops k..M-1 (MWR / LWR / CR)
        const int32_t *rem = p.d.rem + (size_t)c.tid * p.region_ints + c.gl * p.d.mmax;
        if (F64 && kind == JSS_POLICY_CR && ((p.kind >> 24) & 1)) {
            CrKeyF key;
            key.ratio = e.legal ? cr_ratio_f64(rem[0], p.d.cr_factor, e.t, rem[e.todo]) : kCrInf;
            key.idx = e.legal ? c.gl : kCrNone;
            key = cr_argmin_f64<G>(key);
            a = key.idx < kCrNone ? key.idx : c.J;                       // no job legal: NOPE
        } else if (kind == JSS_POLICY_CR) {
            const int total = e.legal ? rem[0] : 0;                      // dispatching.py:373 job length
            const int remaining = e.legal ? rem[e.todo] : 1;             // :391
            CrKey key;
            key.num = e.legal ? cr_p(p) * total - cr_q(p) * e.t : 0x3fffffff;
            key.den = remaining;
            key.idx = e.legal ? c.gl : kCrNone;
            key = cr_argmin<G>(key);
            a = key.idx < kCrNone ? key.idx : c.J;                       // no job legal: NOPE
        } else {
            const bool larger = (kind == JSS_POLICY_FIFO || kind == JSS_POLICY_MWR || kind == JSS_POLICY_MOR);
            int v;
            if (kind == JSS_POLICY_FIFO) v = e.idle_last;                // dispatching.py:146
            else if (kind == JSS_POLICY_SPT) v = e.cur & kDurMask;       // :105-106
            else if (kind == JSS_POLICY_MOR || kind == JSS_POLICY_LOR) v = c.M - e.todo;  // :273 / :314
            else v = e.legal ? rem[e.todo] : 0;                          // MWR / LWR :187-189 / :230-232
            const int key = e.legal ? (larger ? v : -v) : -kBig;
            const int best = grp_max<G>(key);
            const uint32_t hit = grp_ballot<G>(e.legal && key == best, c.gbase);
            a = hit ? __ffs(hit) - 1 : c.J;                              // no job legal: NOPE
        }
        if (e.noop && p.explore_q16 != 0) {
            const uint32_t r = rng_u32(p.seed ^ kExploreSeedXor, env_id, episode, step);
            if ((r >> 16) < p.explore_q16) a = c.J;
        }
    }
    return n == 0 ? -1 : a;
}

// ---------------------------------------------------------------------------------------
// HBM <-> registers.  One 32-byte record per job (two dwordx4 per lane), one int4 header per env.
// Every address is the wave's uniform base + a 32-bit lane offset and none depends on another load,
// so everything is in flight at once.
// ---------------------------------------------------------------------------------------
struct PHeader {
    int episode, step;
};

template <int G>
struct PRaw {  // loads issued before the op table is staged; unpacked after the barrier
    int4 h, lo, hi;
    int tm;
#ifdef JSS_COUNTERS_PLAIN
    int cw;    // word gl (< 8) of my env's four 64-bit counters: see p_bump_counters
#endif
};

#ifdef JSS_COUNTERS_PLAIN
// A/B build: the env's counter row -- four int64 = eight words -- is loaded with the state, one word
// per group lane, and the words that change are stored with it, instead of 2-4 result-less atomics per env step.  Nobody else
// touches an env's counters while a launch that owns the env runs.  Word 2k / 2k + 1 = low / high half of counter k; the
// carry (and the sign extension of a negative reward numerator) crosses from the even lane to the odd one by one DPP move.
template <int G, int TAB>
__device__ __forceinline__ void p_bump_counters(const PCtx<G, TAB> &c, const Params &p, int cw, int steps, int episodes, int makespan_sum, int reward_num) {
    const int k = c.gl >> 1;
    const int add = k == 0 ? steps : k == 1 ? episodes : k == 2 ? makespan_sum : reward_num;
    const unsigned lo_new = (unsigned)cw + (unsigned)add;
    const int hi_delta = (add >> 31) + (lo_new < (unsigned)cw ? 1 : 0);        // (meaningful on even lanes)
    const int from_lo = JSS_DPP(hi_delta, 0xB1);                               // lane ^ 1
    const int now = (c.gl & 1) ? cw + from_lo : (int)lo_new;
    if (c.alive && c.gl < 8 && now != cw)
        st_off<int>(reinterpret_cast<int32_t *>(p.s.counters) + (size_t)c.first_env * 8, (c.rel * 8u + (unsigned)c.gl) * 4u, now);
}
#endif

template <int G, int TAB>
__device__ __forceinline__ PRaw<G> p_issue_loads(const PCtx<G, TAB> &c, const Params &p) {
    PRaw<G> r;
    const unsigned jm = (unsigned)p.d.jmax, mm = (unsigned)p.d.mmax;
    const unsigned jc = (unsigned)c.gl < jm ? c.gl : 0;
    const unsigned mc = (unsigned)c.gl < mm ? c.gl : 0;
    const size_t fe = (size_t)c.first_env;
    r.h = ld_off<int4>(p.s.env + fe * JSS_NH, c.rel * (JSS_NH * 4u));
    if (tab_compact(TAB)) {          // 16-byte records (JSS_FC_*): one access per job
        r.lo = ld_off<int4>(p.s.job + fe * jm * JSS_NFC, (c.rel * jm + jc) * (JSS_NFC * 4u));
        r.hi = make_int4(0, 0, 0, 0);
    } else if (tab_medium(TAB)) {    // 24-byte records (JSS_FM_*): three 8-byte accesses (a record is 8-byte aligned)
        const int32_t *jb = p.s.job + fe * jm * JSS_NFM;
        const unsigned jo = (c.rel * jm + jc) * (JSS_NFM * 4u);
        const int2 a = ld_off<int2>(jb, jo), b = ld_off<int2>(jb, jo + 8u), d = ld_off<int2>(jb, jo + 16u);
        r.lo = make_int4(a.x, a.y, b.x, b.y);
        r.hi = make_int4(d.x, d.y, 0, 0);
    } else {
        const int32_t *jb = p.s.job + fe * jm * JSS_NF;
        const unsigned jo = (c.rel * jm + jc) * 32u;
        r.lo = ld_off<int4>(jb, jo);
        r.hi = ld_off<int4>(jb, jo + 16u);
    }
    // compact batches keep no machine clocks in memory: a machine is busy for as long as the job on it (p_unpack)
    r.tm = tab_no_clocks(TAB) ? 0 : ld_off<int>(p.s.machine + fe * mm, (c.rel * mm + mc) * 4u);
#ifdef JSS_COUNTERS_PLAIN
    r.cw = (p.s.counters && c.gl < 8) ? ld_off<int>(reinterpret_cast<const int32_t *>(p.s.counters) + fe * 8, (c.rel * 8u + (unsigned)c.gl) * 4u) : 0;
#endif
    return r;
}

template <int G, int TAB>
__device__ __forceinline__ PHeader p_unpack(PEnv<G> &e, const PCtx<G, TAB> &c, const PRaw<G> &r, int32_t *mvtab) {
    e.t = r.h.x;
    e.err = r.h.w & 0xFF;
    e.noop = (r.h.w & JSS_STATUS_NOOP) ? 1 : 0;
    e.tm = c.mvalid ? r.tm : 0;
    const bool v = c.jvalid;
    if (tab_compact(TAB)) {          // the job's next three ops are what the LDS table says (staged before this runs)
        const unsigned w0 = (unsigned)r.lo.x, w1 = (unsigned)r.lo.y;
        e.todo = v ? (int)(w0 & JSS_FC_TODO_MASK) : 0;
        e.left = v ? (int)(w1 & 0xffffu) : 0;
        e.perf = v ? (int)(w0 >> JSS_FC_PERF_SHIFT) : 0;
        e.idle = v ? r.lo.z : 0;
        e.idle_last = v ? r.lo.w : 0;
        e.f4 = v ? ((w0 & JSS_FC_FLAG_F4_ONE) ? JSS_F4_ONE : (int)(w1 >> 16)) : 0;
        e.cur = (v && e.todo < c.M) ? c.lds_row[e.todo] : -1;
        e.nxt = (v && e.todo + 1 < c.M) ? c.lds_row[e.todo + 1] : -1;
        e.nxt2 = (v && e.todo + 2 < c.M) ? c.lds_row[e.todo + 2] : -1;
        e.legal = v && (w0 & JSS_FC_FLAG_LEGAL);
        e.blocked = v && (w0 & JSS_FC_FLAG_BLOCKED);
    } else if (tab_medium(TAB)) {    // the three cached ops travel in the record, 21 bits each (0 = none)
        const unsigned w0 = (unsigned)r.lo.x, w1 = (unsigned)r.lo.y, w2 = (unsigned)r.lo.z, w3 = (unsigned)r.lo.w;
        const unsigned cur = (w0 >> JSS_FM_CUR_SHIFT) & JSS_FM_OP_MASK;
        const unsigned nxt = (w2 >> 21) | ((w3 & 0x3FFu) << 11), nxt2 = (w3 >> 10) & JSS_FM_OP_MASK;
        e.todo = v ? (int)(w0 & JSS_FM_TODO_MASK) : 0;
        e.left = v ? (int)(w1 & 0xffffu) : 0;
        e.perf = v ? (int)(w2 & JSS_FM_OP_MASK) : 0;
        e.idle = v ? r.hi.x : 0;
        e.idle_last = v ? r.hi.y : 0;
        e.f4 = v ? ((w0 & JSS_FM_FLAG_F4_ONE) ? JSS_F4_ONE : (int)(w1 >> 16)) : 0;
        e.cur = (v && cur) ? (int)cur : -1;
        e.nxt = (v && nxt) ? (int)nxt : -1;
        e.nxt2 = (v && nxt2) ? (int)nxt2 : -1;
        e.legal = v && (w0 & JSS_FM_FLAG_LEGAL);
        e.blocked = v && (w0 & JSS_FM_FLAG_BLOCKED);
    } else {
        e.todo = v ? (r.lo.x & JSS_TODO_MASK) : 0;
        e.cur = v ? r.lo.y : -1;
        e.left = v ? r.lo.z : 0;
        e.perf = v ? r.lo.w : 0;
        e.idle = v ? r.hi.x : 0;
        e.idle_last = v ? r.hi.y : 0;
        e.f4 = v ? r.hi.z : 0;
        e.nxt = v ? r.hi.w : -1;
        e.nxt2 = (v && ((unsigned)r.lo.x >> JSS_NEXT2_SHIFT)) ? (int)((unsigned)r.lo.x >> JSS_NEXT2_SHIFT) : -1;
        e.legal = v && (r.lo.x & JSS_FLAG_LEGAL);
        e.blocked = v && (r.lo.x & JSS_FLAG_BLOCKED);
    }
    if (tab_no_clocks(TAB)) {
        // time_until_available_machine[m] == time_until_finish_current_op_jobs[the job running on m] (both are set to
        // the op's duration at :446-449 and count down together at :521-530), 0 for an idle machine
        mvtab[c.lane] = 0;
        wave_lds_sync();
        if (e.left > 0 && e.cur >= 0) mvtab[c.gbase + ((e.cur >> 16) & (G - 1))] = e.left;   // (cur < 0: a reset call reading stale memory)
        wave_lds_sync();
        e.tm = c.mvalid ? mvtab[c.lane] : 0;
        wave_lds_sync();
    }
    PHeader hd;
    hd.episode = r.h.y;
    hd.step = r.h.z;
    return hd;
}

// action mask rows of jmax + 1 bytes at `mk` (first env of the wave): legal jobs, the NOPE flag at index J, zeros behind it
// PADDED (here and below): the rows of the tensors may be wider than the lane group -- a class of small instances inside a batch
// that is padded to a larger one (jmax > G; only the fused multi-set grid instantiates it).  Strides come from the layout (jmax,
// mmax), extents from the group: the group touches rows / bytes < G only, what lies behind them was zero-filled by the allocation
// and is nobody's to change as long as the env keeps its class.
template <int G, int TAB, bool WT = false, bool PADDED = false>
__device__ __forceinline__ void p_store_mask(const PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, uint8_t *mk) {
    if (!c.alive) return;
    const unsigned jm = (unsigned)p.d.jmax;
    const unsigned mo = c.rel * (jm + 1);
    if ((unsigned)c.gl <= jm)
        st_out<WT, uint8_t>(mk, mo + c.gl, (uint8_t)(c.jvalid ? (e.legal ? 1 : 0) : (c.gl == c.J ? e.noop : 0)));
    if (PADDED && jm > (unsigned)G) {
        if (c.J == G && c.gl == 0) st_out<WT, uint8_t>(mk, mo + G, (uint8_t)e.noop);      // the NOPE flag of a group-filling instance
    } else if (jm == (unsigned)G && c.gl == 0) st_out<WT, uint8_t>(mk, mo + jm, (uint8_t)(c.J == G ? e.noop : 0));
}

// the observation's normalisers of my env (see PCtx)
struct PNorm {
    int max_time_jobs, sum_op;
    float r_op, r_jobs, r_sum, r_m;
};
template <int G, int TAB>
__device__ __forceinline__ PNorm p_norm(const PCtx<G, TAB> &c) {
    PNorm n;
    if (tab_in_lds(TAB)) {
        n.max_time_jobs = c.max_time_jobs; n.sum_op = c.sum_op;
        n.r_op = c.r_op; n.r_jobs = c.r_jobs; n.r_sum = c.r_sum; n.r_m = c.r_m;
    } else {
        n.max_time_jobs = c.norm[0]; n.sum_op = c.norm[1];
        n.r_op = as_float(c.norm[2]); n.r_jobs = as_float(c.norm[3]); n.r_sum = as_float(c.norm[4]); n.r_m = as_float(c.norm[5]);
    }
    return n;
}

// The record words of my job as they are stored (the inverse of p_unpack; PRaw.h = the env's header)
template <int G, int TAB>
__device__ __forceinline__ PRaw<G> p_pack(const PEnv<G> &e, const PCtx<G, TAB> &c, const PHeader &hd) {
    PRaw<G> r;
    r.h = make_int4(e.t, hd.episode, hd.step, (e.err & 0xFF) | (e.noop ? JSS_STATUS_NOOP : 0));
    r.tm = e.tm;
    if (tab_compact(TAB)) {
        const bool one = e.f4 == JSS_F4_ONE;
        r.lo = make_int4((int)((unsigned)e.todo | (e.legal ? JSS_FC_FLAG_LEGAL : 0u) | (e.blocked ? JSS_FC_FLAG_BLOCKED : 0u) |
                               (one ? JSS_FC_FLAG_F4_ONE : 0u) | ((unsigned)e.perf << JSS_FC_PERF_SHIFT)),
                         (int)((unsigned)e.left | ((unsigned)(one ? 0 : e.f4) << 16)), e.idle, e.idle_last);
        r.hi = make_int4(0, 0, 0, 0);
    } else if (tab_medium(TAB)) {
        const bool one = e.f4 == JSS_F4_ONE;
        const unsigned cur = e.cur >= 0 ? (unsigned)e.cur : 0u, nxt = e.nxt >= 0 ? (unsigned)e.nxt : 0u, nxt2 = e.nxt2 >= 0 ? (unsigned)e.nxt2 : 0u;
        r.lo = make_int4((int)((unsigned)e.todo | (e.legal ? JSS_FM_FLAG_LEGAL : 0u) | (e.blocked ? JSS_FM_FLAG_BLOCKED : 0u) |
                               (one ? JSS_FM_FLAG_F4_ONE : 0u) | (cur << JSS_FM_CUR_SHIFT)),
                         (int)((unsigned)e.left | ((unsigned)(one ? 0 : e.f4) << 16)),
                         (int)((unsigned)e.perf | (nxt << 21)), (int)((nxt >> 11) | (nxt2 << 10)));
        r.hi = make_int4(e.idle, e.idle_last, 0, 0);
    } else {
        r.lo = make_int4(e.todo | (e.legal ? JSS_FLAG_LEGAL : 0) | (e.blocked ? JSS_FLAG_BLOCKED : 0) |
                             (e.nxt2 >= 0 ? (int)((unsigned)e.nxt2 << JSS_NEXT2_SHIFT) : 0), e.cur, e.left, e.perf);
        r.hi = make_int4(e.idle, e.idle_last, e.f4, e.nxt);
    }
    return r;
}

// State back to HBM.  fresh = my env was (re)initialised by this call: every row of its padded block is written (rows
// behind J(env) as "no job") together with the instance constants in its header; otherwise rows < J(env), and of
// those only the halves that changed.
// DIFF = false (the modes that loop over steps with the state in registers: one store per K steps): every row of a job is
// written without comparing it with what was loaded, so that `raw` -- 9 VGPRs -- is dead from the unpack on instead of live
// through the whole loop.
// DIFF: which records of a stepped env go back to memory.
//   kStoreAll      every row of a job, nothing compared (the modes that loop over steps: `raw` is dead after the unpack);
//   kStoreCompare  the halves / thirds of a record that differ from what was loaded;
//   kStoreCause    (the one-step modes) by what the step did instead of by comparing eight words per job: word 0 -- flags, todo --
//                  is compared; the rest of a record changes only for the job that was scheduled (time left) and, when the clock
//                  moved, for every job that is not complete already (a complete job: no current op, word 0 unchanged).  A
//                  superset of the compared set by a few records per episode; of `raw` only word 0 of each record stays live
//                  through the step -- 7 VGPRs per job slot less (`moved` = the clock differs from the one loaded, or the env
//                  was restarted in the kernel; `a_sched` = the job the step scheduled, -1 if none).
enum { kStoreAll = 0, kStoreCompare = 1, kStoreCause = 2 };
#ifndef JSS_ONE_STEP_STORES
#define JSS_ONE_STEP_STORES kStoreCause      // (-DJSS_ONE_STEP_STORES=kStoreCompare: the A/B partner)
#endif
template <int G, int TAB, int DIFF = kStoreCompare>
__device__ __forceinline__ void p_store(const PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, const PHeader &hd,
                                        const PRaw<G> &raw, bool fresh, bool moved = false, int a_sched = -1) {
    if (!c.alive) return;
    const bool adv = DIFF == kStoreCause && moved;
    const unsigned jm = (unsigned)p.d.jmax, mm = (unsigned)p.d.mmax;
    const size_t fe = (size_t)c.first_env;
    const PRaw<G> now = p_pack(e, c, hd);
    if (c.gl == 0) {
        st_off(p.s.env + fe * JSS_NH, c.rel * (JSS_NH * 4u), now.h);
        if (fresh) {   // the instance constants of the env travel with it from here on (include/jss_hip.h JSS_C_*)
            const PNorm n = p_norm(c);
            int32_t *cp = p.s.env_const + fe * JSS_NC;
            const unsigned co = c.rel * (JSS_NC * 4u);
            st_off(cp, co, make_int4(c.J, c.M, c.max_time_op, c.tid));
            st_off(cp, co + 16u, make_int4(n.max_time_jobs, n.sum_op, as_int(n.r_op), as_int(n.r_jobs)));
            st_off(cp, co + 32u, make_int4(as_int(n.r_sum), as_int(n.r_m), 0, 0));
        }
    }
    if (tab_no_clocks(TAB)) {
        // no machine clocks in memory (p_unpack)
    } else if (fresh ? (unsigned)c.gl < mm : (c.mvalid && (DIFF == kStoreAll || e.tm != raw.tm)))      // idle machines stay 0
        st_off(p.s.machine + fe * mm, (c.rel * mm + c.gl) * 4u, e.tm);
    // kStoreCause: the part of a record that holds word 0 / the time left, and the parts only a clock move touches
    const bool w0_changed = now.lo.x != raw.lo.x;
    const bool head_dirty = w0_changed || c.gl == a_sched || (adv && e.cur >= 0);
    const bool rest_dirty = adv && (e.cur >= 0 || w0_changed);
    if (tab_medium(TAB)) {
        if (c.jvalid || (fresh && (unsigned)c.gl < jm)) {   // the thirds of the record that changed
            int32_t *jb = p.s.job + fe * jm * JSS_NFM;
            const unsigned jo = (c.rel * jm + c.gl) * (JSS_NFM * 4u);
            const int4 lo = now.lo, hi = now.hi;
            if (DIFF == kStoreCause && !fresh) {
                if (head_dirty) st_off(jb, jo, make_int2(lo.x, lo.y));
                if (rest_dirty) {
                    st_off(jb, jo + 8u, make_int2(lo.z, lo.w));
                    st_off(jb, jo + 16u, make_int2(hi.x, hi.y));
                }
            } else {
                if (DIFF == kStoreAll || fresh || lo.x != raw.lo.x || lo.y != raw.lo.y) st_off(jb, jo, make_int2(lo.x, lo.y));
                if (DIFF == kStoreAll || fresh || lo.z != raw.lo.z || lo.w != raw.lo.w) st_off(jb, jo + 8u, make_int2(lo.z, lo.w));
                if (DIFF == kStoreAll || fresh || hi.x != raw.hi.x || hi.y != raw.hi.y) st_off(jb, jo + 16u, make_int2(hi.x, hi.y));
            }
        }
    } else if (tab_compact(TAB)) {
        if (c.jvalid || (fresh && (unsigned)c.gl < jm)) {
            const int4 lo = now.lo;
            // an unchanged record is not rewritten (steps without a time advance touch few jobs)
            bool dirty;
            if (DIFF == kStoreCause) dirty = head_dirty || rest_dirty;
            else dirty = DIFF == kStoreAll || lo.x != raw.lo.x || lo.y != raw.lo.y || lo.z != raw.lo.z || lo.w != raw.lo.w;
            if (fresh || dirty) st_off(p.s.job + fe * jm * JSS_NFC, (c.rel * jm + c.gl) * (JSS_NFC * 4u), lo);
        }
    } else if (c.jvalid || (fresh && (unsigned)c.gl < jm)) {
        int32_t *jb = p.s.job + fe * jm * JSS_NF;
        const unsigned jo = (c.rel * jm + c.gl) * 32u;
        const int4 lo = now.lo, hi = now.hi;
        // unchanged halves of the record are not rewritten (steps without a time advance touch few jobs)
        if (DIFF == kStoreCause && !fresh) {
            if (head_dirty) st_off(jb, jo, lo);
            if (rest_dirty) st_off(jb, jo + 16u, hi);
        } else {
            if (DIFF == kStoreAll || fresh || lo.x != raw.lo.x || lo.y != raw.lo.y || lo.z != raw.lo.z || lo.w != raw.lo.w) st_off(jb, jo, lo);
            if (DIFF == kStoreAll || fresh || hi.x != raw.hi.x || hi.y != raw.hi.y || hi.z != raw.hi.z || hi.w != raw.hi.w) st_off(jb, jo + 16u, hi);
        }
        // A class of small instances inside wider padded rows (jmax > G: the fused grid's PADDED bodies): a reset leaves the rows
        // behind the lane group as "no job" records too, like the reset of the padded extents' kernel, the session's write-back
        // and the host twin do -- whichever path (re)initialised an env, its padded block holds the same bytes (a few KB per
        // EPISODE; the steps never touch those rows).
        if (fresh)
            for (unsigned r = (unsigned)c.gl + G; r < jm; r += G) {
                st_off(jb, (c.rel * jm + r) * 32u, make_int4(0, -1, 0, 0));
                st_off(jb, (c.rel * jm + r) * 32u + 16u, make_int4(0, 0, 0, -1));
            }
    }
}

// (J,7) float32 observation (jss_env.py:102-111).  Each lane writes its job's row into an LDS
// image of the wave's E consecutive envs ([E][jmax][7], padding rows zero), which then goes out as
// one linear copy -- dwordx4 per lane when the wave's block is whole and 16-byte sized.
template <int G, int TAB, bool WT = false, bool PADDED = false>
__device__ __forceinline__ void p_store_obs(const PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, float *dst,
                                            float *scratch, bool wave_whole) {   // dst: block of the wave's first env
    constexpr int E = kWave / G;
    if (PADDED && p.d.jmax > G) {
        // rows of jmax * 7 floats in memory, an image of G rows per env in LDS: every env's first G rows leave on their own,
        // 64 bytes per group and store (the rows behind J(env) as zeros; the rows behind G are never touched)
        const PNorm nr = p_norm(c);
        const float f_op = (float)c.max_time_op, f_jobs = (float)nr.max_time_jobs, f_sum = (float)nr.sum_op, f_m = (float)c.M;
        float *mine = scratch + (c.gbase / G) * (G * 7);
        float *row = mine + c.gl * 7;
        row[0] = e.legal ? 1.0f : 0.0f;
        row[1] = div_by((float)e.left, f_op, nr.r_op);
        row[2] = div_by((float)e.todo, f_m, nr.r_m);
        row[3] = div_by((float)e.perf, f_jobs, nr.r_jobs);
        row[4] = e.f4 == JSS_F4_ONE ? 1.0f : div_by((float)e.f4, f_op, nr.r_op);
        row[5] = div_by((float)e.idle_last, f_sum, nr.r_sum);
        row[6] = div_by((float)e.idle, f_sum, nr.r_sum);
        wave_lds_sync();
        if (c.alive)
            for (int i = c.gl; i < G * 7; i += G) st_out<WT, float>(dst, (c.rel * (unsigned)(p.d.jmax * 7) + i) * 4u, mine[i]);
        wave_lds_sync();
        return;
    }
    const int row_floats = p.d.jmax * 7;
    const PNorm nr = p_norm(c);
    const float f_op = (float)c.max_time_op, f_jobs = (float)nr.max_time_jobs, f_sum = (float)nr.sum_op;
    const float f_m = (float)c.M;
    const float r_op = nr.r_op, r_jobs = nr.r_jobs, r_sum = nr.r_sum, r_m = nr.r_m;
    float *mine = scratch + (c.gbase / G) * row_floats;   // image slot = physical group (dead groups share c.rel with a live one)
    if (c.gl < p.d.jmax) {
        float *row = mine + c.gl * 7;
        row[0] = e.legal ? 1.0f : 0.0f;                                                  // :130
        row[1] = div_by((float)e.left, f_op, r_op);                                      // :448, :539
        row[2] = div_by((float)e.todo, f_m, r_m);                                        // :559
        row[3] = div_by((float)e.perf, f_jobs, r_jobs);                                  // :545
        row[4] = e.f4 == JSS_F4_ONE ? 1.0f : div_by((float)e.f4, f_op, r_op);            // :569-586
        row[5] = div_by((float)e.idle_last, f_sum, r_sum);                               // :555, :600
        row[6] = div_by((float)e.idle, f_sum, r_sum);                                    // :553, :601
    }
    wave_lds_sync();
    const int n = E * row_floats;
    if (wave_whole && (n & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        for (int i = c.lane; i < (n >> 2); i += kWave) {
            if (WT) wt_store16(dst, (unsigned)i * 16u, reinterpret_cast<const float4 *>(scratch)[i]);
            else st_nt(dst, (unsigned)i * 16u, reinterpret_cast<const float4 *>(scratch)[i]);
        }
    } else if (c.alive) {
        // (the last wavefront of a ragged batch only: left to itself the compiler unrolls this 16 times with a 64-bit address
        // per store -- the register peak of the fused grid's step kernel)
#pragma unroll 1
        for (int i = c.gl; i < row_floats; i += G) st_out<WT, float>(dst, (c.rel * row_floats + i) * 4u, mine[i]);
    }
    wave_lds_sync();
}

// ---------------------------------------------------------------------------------------
// mode bodies: everything between "state unpacked into registers" and "state stored".  Returns "my env was
// (re)initialised by this call".
// ---------------------------------------------------------------------------------------
// my env is being reset and may have been given another instance since (table_of_env): take its constants from the
// instance record (kTabGlobal; with kTabLds the whole batch shares one instance and nothing changes)
template <int G, int TAB>
__device__ __forceinline__ void p_reload_instance(PCtx<G, TAB> &c, const Params &p, bool on) {
    if (tab_in_lds(TAB)) return;
    if (__ballot(on) == 0) return;
    const size_t fe = (size_t)c.first_env;
    int tid = c.tid;
    if (on && p.d.table_of_env) tid = ld_off<int>(p.d.table_of_env + fe, c.rel * 4u);
    const int32_t *ir = p.d.inst + (size_t)tid * JSS_NI;
    if (on) {
        c.tid = tid;
        c.J = ir[JSS_I_JOBS];
        c.M = ir[JSS_I_MACHINES];
        c.max_time_op = ir[JSS_I_MAX_TIME_OP];
        c.jvalid = c.gl < c.J;
        c.mvalid = c.gl < c.M;
        if (c.gl < 6) c.norm[c.gl] = ir[JSS_I_MAX_TIME_JOBS + c.gl];      // record words 3..8 = the six normalisers
    }
    wave_lds_sync();
}

// One jss_step call on the registers, in two halves: the computation -- the JSS_ACTION_RESET restart, step(), the
// header's step count -- and the per-env scalar outputs: reward / done / makespan / counters (a skipped env keeps them).
// WT = the stores are write-through (step session, which puts its progress word between the two halves: see there).
// rn = the reward numerator, called = the env was stepped (not skipped, not restarted).  Returns "re-initialised".
struct PStepResult {
    int rn;
    bool called, restart, done;
};
template <int G, int TAB, bool WT>
__device__ __forceinline__ PStepResult p_step_compute(PEnv<G> &e, PHeader &hd, PCtx<G, TAB> &c, const Params &p, int a_in, int32_t *mvtab) {
    PStepResult r;
    r.restart = c.alive && a_in == JSS_ACTION_RESET;                    // reset() this env instead of stepping it
    p_reload_instance(c, p, r.restart);
    p_reset<G, TAB, WT>(e, c, p, r.restart);
    if (r.restart) {
        hd.episode += 1;
        hd.step = 0;
    }
    r.rn = p_step<G, TAB, WT>(e, c, p, a_in, mvtab);
    r.called = a_in != JSS_ACTION_SKIP && !r.restart;
    r.done = !grp_any<G>(e.legal, c.gbase);
    if (r.called) hd.step += 1;
    return r;
}
template <int G, int TAB, bool WT>
__device__ __forceinline__ void p_step_outputs(const PEnv<G> &e, const PCtx<G, TAB> &c, const Params &p, const PStepResult &r) {
    const size_t fe = (size_t)c.first_env;
    if (!c.alive || c.gl != 0) return;
    if (r.restart) {
        st_out<WT, float>(p.o.reward + fe, c.rel * 4u, 0.f);
        st_out<WT, uint8_t>(p.o.done + fe, c.rel, (uint8_t)0);
    }
    if (r.called) {                                   // a skipped env keeps its reward / done / makespan
        st_out<WT, float>(p.o.reward + fe, c.rel * 4u, div_by((float)r.rn, (float)c.max_time_op, p_norm(c).r_op));   // :483-493
        st_out<WT, uint8_t>(p.o.done + fe, c.rel, (uint8_t)(r.done ? 1 : 0));                      // :639-653
        if (r.done) st_out<WT, int>(p.o.makespan + fe, c.rel * 4u, e.t);                           // :650
        if (p.s.counters) add_counters(p.s.counters + (fe + c.rel) * 4, 1, r.done ? 1 : 0, r.done ? e.t : 0, r.rn);
    }
}
template <int G, int TAB, bool WT>
__device__ __forceinline__ bool p_step_call(PEnv<G> &e, PHeader &hd, PCtx<G, TAB> &c, const Params &p, int a_in,
                                            int32_t *mvtab, int &rn, bool &called) {
    const PStepResult r = p_step_compute<G, TAB, WT>(e, hd, c, p, a_in, mvtab);
    p_step_outputs<G, TAB, WT>(e, c, p, r);
    rn = r.rn;
    called = r.called;
    return r.restart;
}

// (a_sched / restarted: what the one step of kStep / kRollout1 did to my env -- p_store, kStoreCause)
template <int G, int MODE, int TAB>
__device__ __forceinline__ bool p_body(PEnv<G> &e, PHeader &hd, PCtx<G, TAB> &c, const Params &p, int a_in, bool selected,
                                       int32_t *mvtab, float *scratch, bool wave_whole, int &a_sched, bool &restarted, int cw = 0) {
    const size_t fe = (size_t)c.first_env;
    bool fresh = false;
    a_sched = -1;
    restarted = false;
    if (MODE == kReset) {
        const bool on = c.alive && selected;          // untouched groups are written back unchanged
        p_reset(e, c, p, on);
        if (on) {
            fresh = true;
            hd.episode += 1;
            hd.step = 0;
            if (c.gl == 0) {
                st_off<float>(p.o.reward + fe, c.rel * 4u, 0.f);
                st_off<uint8_t>(p.o.done + fe, c.rel, 0);
            }
        }
    } else if (MODE == kStep) {
        int rn;
        bool called;
        fresh = p_step_call<G, TAB, false>(e, hd, c, p, a_in, mvtab, rn, called);
        a_sched = a_in;
    } else if (MODE == kSteps) {
        // n_iter x jss_step with the actions given up front: the state stays in registers, every step optionally recorded
        for (int it = 0; it < p.n_iter; ++it) {
            const size_t slot0 = (size_t)it * traj_stride(p) + fe;       // [it][first env of the wave]
            const int a = ld_off<int>(p.actions + slot0, c.rel * 4u);
            int rn;
            bool called;
            fresh |= p_step_call<G, TAB, false>(e, hd, c, p, a, mvtab, rn, called);
            if (p.t.real_obs) p_store_obs<G, TAB>(e, c, p, p.t.real_obs + slot0 * p.d.jmax * 7, scratch, wave_whole);
            if (p.t.action_mask) p_store_mask<G, TAB>(e, c, p, p.t.action_mask + slot0 * (p.d.jmax + 1));
            const bool done = !grp_any<G>(e.legal, c.gbase);
            if (c.alive && c.gl == 0) {
                if (p.t.reward) p.t.reward[slot0 + c.rel] = called ? div_by((float)rn, (float)c.max_time_op, p_norm(c).r_op) : 0.f;
                if (p.t.done) p.t.done[slot0 + c.rel] = done ? 1 : 0;
            }
        }
    } else if (MODE == kAdvance) {
        const bool on = c.alive && selected;
        const int d = p_next_event(e);
        const bool busy = d < kBig;
        if (on && !busy) e.err |= JSS_ERR_NOPE_IDLE;                     // reference: IndexError (:517)
        const int hole = p_advance(e, c, p, on && busy, d);
        if (on && c.gl == 0 && p.hole) st_off<int>(p.hole + fe, c.rel * 4u, busy ? hole : 0);
    } else if (MODE == kPolicy) {
        const uint64_t env_id = (uint64_t)(p.d.env_ids ? ld_off<int64_t>(p.d.env_ids + fe, c.rel * 8u)
                                                       : p.d.env_id_base + (int64_t)(fe + c.rel));
        const int a = p_select<G, TAB, true>(e, c, p, env_id, (uint32_t)hd.episode, (uint32_t)hd.step);
        if (c.alive && c.gl == 0) st_off<int>(p.actions_out + fe, c.rel * 4u, a);
    } else {  // kRollout / kRollout1 / kTraj
        const uint64_t env_id = (uint64_t)(p.d.env_ids ? ld_off<int64_t>(p.d.env_ids + fe, c.rel * 8u)
                                                       : p.d.env_id_base + (int64_t)(fe + c.rel));
        int n_steps = 0, n_done = 0, last_rn = 0, last_makespan = -1, sum_makespan = 0, sum_rn = 0;
        const bool autoreset = (p.flags & JSS_ROLLOUT_AUTORESET) != 0;
        const int n_iter = MODE == kRollout1 ? 1 : p.n_iter;
        for (int it = 0; it < n_iter; ++it) {
            const size_t slot0 = (size_t)it * traj_stride(p) + fe;       // kTraj: slot [it][first env of the wave]
            if (MODE == kTraj) {                                         // what the policy sees in this slot
                if (p.t.real_obs) p_store_obs<G, TAB>(e, c, p, p.t.real_obs + slot0 * p.d.jmax * 7, scratch, wave_whole);
                if (p.t.action_mask) p_store_mask<G, TAB>(e, c, p, p.t.action_mask + slot0 * (p.d.jmax + 1));
            }
            const bool done0 = !grp_any<G>(e.legal, c.gbase);            // :639-653
            const bool do_reset = c.alive && done0 && autoreset;
            const bool do_step = c.alive && !done0;
            if (MODE != kRollout1 && MODE != kTraj && __ballot(do_reset || do_step) == 0) break;  // every env frozen
            p_reset(e, c, p, do_reset);
            if (do_reset) {
                hd.episode += 1;
                hd.step = 0;
            }
            int a = JSS_ABLATED(p, JSS_ABLATE_SELECT) ? __ffs(grp_ballot<G>(e.legal, c.gbase)) - 1
                                                      : p_select(e, c, p, env_id, (uint32_t)hd.episode, (uint32_t)hd.step);
            if (!do_step) a = JSS_ACTION_SKIP;
            if (MODE == kRollout1) {
                a_sched = a;
                restarted = do_reset;
            }
            const int rn = p_step(e, c, p, a, mvtab);
            const bool done1 = !grp_any<G>(e.legal, c.gbase);            // collective: outside the divergent branch
            if (do_step) {
                last_rn = rn;
                hd.step += 1;
                n_steps += 1;
                sum_rn += rn;
                if (done1) {
                    n_done += 1;
                    sum_makespan += e.t;
                    last_makespan = e.t;
                }
            }
            if (MODE == kTraj && c.alive && c.gl == 0) {
                const size_t slot = slot0 + c.rel;
                if (p.t.action) p.t.action[slot] = do_step ? a : (do_reset ? JSS_ACTION_RESET : JSS_ACTION_SKIP);
                if (p.t.reward) p.t.reward[slot] = do_step ? div_by((float)rn, (float)c.max_time_op, p_norm(c).r_op) : 0.f;
                if (p.t.done) p.t.done[slot] = do_step ? (done1 ? 1 : 0) : (do_reset ? 0 : 1);
            }
        }
        const bool done = !grp_any<G>(e.legal, c.gbase);
#ifdef JSS_COUNTERS_PLAIN
        if (p.s.counters) p_bump_counters(c, p, cw, n_steps, n_done, sum_makespan, sum_rn);
#endif
        if (c.alive && c.gl == 0) {
            if (n_steps) st_off<float>(p.o.reward + fe, c.rel * 4u, div_by((float)last_rn, (float)c.max_time_op, p_norm(c).r_op));
            st_off<uint8_t>(p.o.done + fe, c.rel, done ? 1 : 0);
            if (last_makespan >= 0) st_off<int>(p.o.makespan + fe, c.rel * 4u, last_makespan);
#ifndef JSS_COUNTERS_PLAIN
            if (p.s.counters) add_counters(p.s.counters + (fe + c.rel) * 4, n_steps, n_done, sum_makespan, sum_rn);
#endif
        }
    }
    return fresh;
}

// ---------------------------------------------------------------------------------------
// the packed kernel, one env set (E = 64/G envs) per wave
// ---------------------------------------------------------------------------------------
// launch bounds: the largest occupancy each mode reaches without spilling (8 waves per SIMD = 64 VGPRs)
// With kTabGlobal the instance (tid, J, M, max_time_op) is group-uniform data in VGPRs instead of SGPRs: the step
// kernels need 66-70 VGPRs.  7 waves/SIMD (72 VGPRs) beats 8 with two VGPRs in scratch: 19.4 vs 21.5 us per step
// on synthetic 15x15, B = 65 536 (profiles/README.md).  The medium-record one-step rollout needs 62: 8 waves per SIMD.
// The recorders (kTraj / kSteps) on a shared table need 94-97 VGPRs since the ragged observation store is no longer unrolled
// (p_store_obs): 5 waves per SIMD (96), +7.5 % on the headline's trajectory against 4; so do the medium-record ones with their
// arguments read in place (93-95); full records with per-env tables need 101-102 (2 VGPRs in scratch at 5): 4.
#ifndef JSS_PTRAJ_LDS_MIN_BLOCKS
#define JSS_PTRAJ_LDS_MIN_BLOCKS 5
#endif
#ifndef JSS_PTRAJ_GLOBAL_MIN_BLOCKS
#define JSS_PTRAJ_GLOBAL_MIN_BLOCKS 4
#endif
#ifndef JSS_PACKED_GLOBAL_MIN_BLOCKS
#define JSS_PACKED_GLOBAL_MIN_BLOCKS 7
#endif
#ifndef JSS_PACKED_MEDIUM_MIN_BLOCKS
#define JSS_PACKED_MEDIUM_MIN_BLOCKS 8
#endif
// One workgroup's share of a packed launch: `block` = its index among the workgroups of THIS env set (blockIdx.x of a
// plain launch; a workgroup of the fused multi-set grid -- jss_multi_kernel -- passes its index within its own set).
template <int G, int MODE, int TAB, bool PADDED = false>
__device__ __forceinline__ void packed_block(const Params &p, int block, int32_t *lds) {
    static_assert(!PADDED || (MODE != kTraj && MODE != kSteps && MODE != kSession), "the recorders write whole rows");
    constexpr int E = kWave / G;                      // envs per wave
    constexpr int EB = E * kWavesPerBlock;            // envs per workgroup
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // obs image of this wave: E * jmax * 7 floats, 16-byte aligned (table_lds_ints is a multiple of 4)
    float *scratch = reinterpret_cast<float *>(lds + p.table_lds_ints) + wave * p.obs_wave_floats;
    int32_t *mvtab = lds + p.mv_off_ints + wave * kWave;                 // one int per lane, see p_check_no_op

    PCtx<G, TAB> c;
    c.lane = lane;
    c.gl = lane & (G - 1);
    c.gbase = lane & ~(G - 1);
    c.norm = lds + p.norm_off_ints + wave * kWave + c.gbase;             // kTabGlobal: my group's six normalisers
    c.first_env = block * EB + wave * E;                                 // wave-uniform
    const bool wave_dead = c.first_env >= p.d.batch;                     // only in the last workgroup
    const int e_in_wave = lane / G;
    c.alive = c.first_env + e_in_wave < p.d.batch;
    c.rel = (unsigned)(c.alive ? e_in_wave : (wave_dead ? 0 : p.d.batch - 1 - c.first_env));
    const bool wave_whole = c.first_env + E <= p.d.batch;
    const size_t fe = (size_t)c.first_env;
    // 1. state loads first: they depend on nothing but the env index.  With kTabGlobal the env's shape and op table
    //    index come from its constants record (JssState.env_const, written by reset) and its six observation
    //    normalisers (words 4-9 of that record) are parked in LDS, one word per lane, until the observation is
    //    written; a reset call takes both from the instance record instead (env -> instance -> record).
    PRaw<G> raw;
    int a_in = JSS_ACTION_SKIP;
    bool selected = true;
    int4 hx = make_int4(0, 0, 0, 0);
    c.tid = 0;
    if (!wave_dead) {
        raw = p_issue_loads<G, TAB>(c, p);
        if (MODE == kStep) {
            a_in = ld_off<int>(p.actions + fe, c.rel * 4u);
            // jss_step_autoreset: an env that reported done on the previous call is reset instead of stepped
            if ((p.flags & JSS_ROLLOUT_AUTORESET) && ld_off<uint8_t>(p.o.done + fe, c.rel) != 0) a_in = JSS_ACTION_RESET;
        }
        if ((MODE == kReset || MODE == kAdvance) && p.which) selected = ld_off<uint8_t>(p.which + fe, c.rel) != 0;
        if (tab_global(TAB)) {
            if (MODE == kReset) {
                c.tid = p.d.table_of_env ? ld_off<int>(p.d.table_of_env + fe, c.rel * 4u) : (int)(fe + c.rel);
            } else {
                hx = ld_off<int4>(p.s.env_const + fe * JSS_NC, c.rel * (JSS_NC * 4u));
                if (c.gl < 6) c.norm[c.gl] = ld_off<int>(p.s.env_const + fe * JSS_NC, c.rel * (JSS_NC * 4u) + 16u + (unsigned)c.gl * 4u);
            }
        }
    }
    // 2. instance constants; the shared op table -> LDS
    if (tab_in_lds(TAB)) {
        stage_shared_table(lds, p.d.ops, p.d.jmax * p.d.mmax, (int)threadIdx.x);   // one instance: jmax rows are its J rows
        __syncthreads();
    }
    if (wave_dead) return;
    if (tab_in_lds(TAB)) {                            // one instance for the whole batch: scalar loads, issued up here
        const int32_t *ir = p.d.inst;
        c.J = ir[JSS_I_JOBS];
        c.M = ir[JSS_I_MACHINES];
        c.max_time_op = ir[JSS_I_MAX_TIME_OP];
        c.max_time_jobs = ir[JSS_I_MAX_TIME_JOBS];
        c.sum_op = ir[JSS_I_SUM_OP];
        c.r_op = as_float(ir[JSS_I_RCP_MAX_TIME_OP]);
        c.r_jobs = as_float(ir[JSS_I_RCP_MAX_TIME_JOBS]);
        c.r_sum = as_float(ir[JSS_I_RCP_SUM_OP]);
        c.r_m = as_float(ir[JSS_I_RCP_MACHINES]);
    } else if (MODE == kReset) {
        const int32_t *ir = p.d.inst + (size_t)c.tid * JSS_NI;
        c.J = ir[JSS_I_JOBS];
        c.M = ir[JSS_I_MACHINES];
        c.max_time_op = ir[JSS_I_MAX_TIME_OP];
        if (c.gl < 6) c.norm[c.gl] = ir[JSS_I_MAX_TIME_JOBS + c.gl];
        wave_lds_sync();
    } else {
        c.J = hx.x;
        c.M = hx.y;
        c.max_time_op = hx.z;
        c.tid = hx.w;
        wave_lds_sync();
    }
    c.jvalid = c.gl < c.J;
    c.mvalid = c.gl < c.M;
    c.lds_row = lds + (c.gl < p.d.jmax ? c.gl : 0) * p.d.mmax;

    PEnv<G> e;
    PHeader hd = p_unpack(e, c, raw, mvtab);
    // an env that was never reset (episode counter 0: every reset bumps it) is left alone by the step-type calls, like
    // the one-wavefront-per-env kernel and the host twin do (J == 0 in its constants record): no stores, no counters
    if (MODE != kReset && hd.episode == 0) c.alive = false;
    int a_sched;
    bool restarted;
#ifdef JSS_COUNTERS_PLAIN
    const bool fresh = p_body<G, MODE, TAB>(e, hd, c, p, a_in, selected, mvtab, scratch, wave_whole, a_sched, restarted, raw.cw);
#else
    const bool fresh = p_body<G, MODE, TAB>(e, hd, c, p, a_in, selected, mvtab, scratch, wave_whole, a_sched, restarted);
#endif
    if (MODE == kPolicy) return;
    // (by cause with per-env tables only: +1.6 % on 15x15 x 65 536, full records 67 -> 61 VGPRs = 8 wavefronts per SIMD; on a
    //  shared table -- 16-byte records, four words to compare -- it measured 0..2 % SLOWER: profiles/r06_misc/stores_by_cause_ab.txt)
    constexpr int kDiffStores = (MODE == kRollout || MODE == kTraj || MODE == kSteps) ? kStoreAll
                                : ((MODE == kStep || MODE == kRollout1) && tab_global(TAB)) ? JSS_ONE_STEP_STORES : kStoreCompare;
    p_store<G, TAB, kDiffStores>(e, c, p, hd, raw, fresh, restarted || e.t != raw.h.x, a_sched);
    p_store_mask<G, TAB, false, PADDED>(e, c, p, p.o.action_mask + fe * (p.d.jmax + 1));
    if (!JSS_ABLATED(p, JSS_ABLATE_OBS))
        p_store_obs<G, TAB, false, PADDED>(e, c, p, p.o.real_obs + fe * p.d.jmax * 7, scratch, wave_whole);
}

template <int G, int MODE, int TAB>
__global__ __launch_bounds__(kBlock, (MODE == kTraj || MODE == kSteps) ? (tab_global(TAB) && !tab_medium(TAB) ? JSS_PTRAJ_GLOBAL_MIN_BLOCKS : JSS_PTRAJ_LDS_MIN_BLOCKS)
                                     : MODE == kRollout ? (tab_global(TAB) ? 4 : 5)
                                     : (tab_medium(TAB) && MODE == kRollout1) ? JSS_PACKED_MEDIUM_MIN_BLOCKS
                                     : ((tab_global(TAB) && (MODE == kStep || MODE == kRollout1)) ? JSS_PACKED_GLOBAL_MIN_BLOCKS : 8))
void jss_packed_kernel(Params p_arg) {
    HIP_DYNAMIC_SHARED(int32_t, lds)
    // By value for the one-step kernels: they fit their SGPR budget, and with every argument loaded up front -- behind the state
    // loads -- no scalar load waits in the middle of the dependent chain (JSS_PARAMS_OF in jss_common.hpp).  The kernels that
    // LOOP over steps read them in place where that measured faster (round 6, same-box A/B, profiles/r06_misc/
    // packed_params_in_place.txt): with per-env tables (23-31 fewer SGPR values parked in VGPR lanes; the medium-record recorders
    // drop from 100-107 to 93-95 VGPRs, i.e. from 4 to 5 wavefronts per SIMD: trajectory +8 %, jss_steps +3 %, the rollout +2 %
    // on 15x15 x 65 536) and the shared-table recorder (+3 %); the shared-table rollout / jss_steps stay by value (+1 % at
    // 65 536 envs, -2..-3 % at 4 096).
    JSS_PARAMS_OF(p, p_arg, (MODE == kTraj || MODE == kSteps || MODE == kRollout) && (tab_global(TAB) || MODE == kTraj));
    packed_block<G, MODE, TAB>(p, (int)blockIdx.x, lds);
}


// ---------------------------------------------------------------------------------------
// The resident step-session kernel (include/jss_hip.h, jss_session_*), packed flavour.
//
// A wavefront owns `slots` env sets (E = 64/G envs each): set = wave index + slot * waves in the grid.  With one set
// the state lives in registers from open to close; with several, each set is parked in LDS between its visits as the
// records it would be stored as (p_pack / p_unpack: 2 int4 per lane with compact records, 5 with full ones).  A step:
// wait for the set's action granules (prefetched one (step, set) pair ahead), jss_step's semantics on the registers
// (p_step_call), outputs write-through, and -- once per step, as late as possible -- the wavefront's progress word
// behind a drain of those stores.  Nothing of the state touches memory between the loads up front and the stores at
// the end.  Every wait is bounded by the wall clock.
// ---------------------------------------------------------------------------------------
// LDS footprint of one parked env set, in int4: per lane the job record (compact: 1 int4; full: lo, hi and the machine
// clock), then per env (<= 4 of them) the header and -- per-env tables -- the instance constants
template <int TAB>
constexpr int park_lane_int4() { return tab_compact(TAB) ? 1 : tab_medium(TAB) ? 2 : 3; }
template <int TAB>
constexpr int park_int4() { return park_lane_int4<TAB>() * kWave + 8; }

template <int G, int TAB>
struct PSlot {            // what distinguishes one env set of the wavefront from another
    int first_env;
    bool live, whole;
};

template <int G, int TAB>
__device__ __forceinline__ PSlot<G, TAB> p_enter_slot(PCtx<G, TAB> &c, const Params &p, int32_t *lds, int wave, int gw,
                                                      int n_waves, int slot) {
    constexpr int E = kWave / G;
    PSlot<G, TAB> sl;
    sl.first_env = (gw + slot * n_waves) * E;                            // wave-uniform
    sl.live = sl.first_env < p.d.batch;
    sl.whole = sl.first_env + E <= p.d.batch;
    c.first_env = sl.live ? sl.first_env : 0;
    const int e_in_wave = c.lane / G;
    c.alive = sl.live && sl.first_env + e_in_wave < p.d.batch;
    c.rel = (unsigned)(c.alive ? e_in_wave : (sl.live ? p.d.batch - 1 - sl.first_env : 0));
    c.norm = lds + p.norm_off_ints + slot * p.norm_slot_ints + wave * kWave + c.gbase;
    return sl;
}

template <int G, int TAB>
__device__ __forceinline__ void p_park(int4 *park, int slot, int lane, const PRaw<G> &r, const PCtx<G, TAB> &c) {
    int4 *q = park + (size_t)slot * park_int4<TAB>();
    q[lane] = r.lo;
    if (!tab_compact(TAB)) q[kWave + lane] = r.hi;
    if (!tab_no_clocks(TAB)) q[2 * kWave + lane] = make_int4(r.tm, 0, 0, 0);
    if (c.gl == 0) {                                   // per env: its header, its instance constants
        int4 *g = q + park_lane_int4<TAB>() * kWave + 2 * (lane / G);
        g[0] = r.h;
        if (tab_global(TAB)) g[1] = make_int4(c.J, c.M, c.max_time_op, c.tid);
    }
    wave_lds_sync();                                   // the group's lanes read what its lane 0 wrote
}
template <int G, int TAB>
__device__ __forceinline__ PRaw<G> p_unpark(const int4 *park, int slot, int lane, PCtx<G, TAB> &c) {
    const int4 *q = park + (size_t)slot * park_int4<TAB>();
    const int4 *g = q + park_lane_int4<TAB>() * kWave + 2 * (lane / G);
    PRaw<G> r;
    r.h = g[0];
    r.lo = q[lane];
    r.hi = make_int4(0, 0, 0, 0);
    r.tm = 0;
    if (!tab_compact(TAB)) r.hi = q[kWave + lane];
    if (!tab_no_clocks(TAB)) r.tm = q[2 * kWave + lane].x;
    if (tab_global(TAB)) {                             // the set's instance constants (an env's restart may change them)
        const int4 k = g[1];
        c.J = k.x;
        c.M = k.y;
        c.max_time_op = k.z;
        c.tid = k.w;
        c.jvalid = c.gl < c.J;
        c.mvalid = c.gl < c.M;
    }
    return r;
}

template <int G, int TAB>
__global__ __launch_bounds__(kBlock, tab_global(TAB) ? 5 : 6) void jss_packed_session_kernel(Params p_arg) {   // (per-env tables: 90 VGPRs, no scratch at 5)
    HIP_DYNAMIC_SHARED(int32_t, lds)
    JSS_PARAMS_OF(p, p_arg, false);
    constexpr int E = kWave / G;
    constexpr int MODE = kSession;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *scratch = reinterpret_cast<float *>(lds + p.table_lds_ints) + wave * p.obs_wave_floats;
    int32_t *mvtab = lds + p.mv_off_ints + wave * kWave;
    const int slots = p.slots;
    int4 *park = reinterpret_cast<int4 *>(lds + p.park_off_ints) + (size_t)wave * slots * park_int4<TAB>();
    const int n_waves = (int)gridDim.x * kWavesPerBlock;
    const int gw = (int)blockIdx.x * kWavesPerBlock + wave;

    PCtx<G, TAB> c;
    c.lane = lane;
    c.gl = lane & (G - 1);
    c.gbase = lane & ~(G - 1);
    c.tid = 0;
    c.lds_row = lds + (c.gl < p.d.jmax ? c.gl : 0) * p.d.mmax;
    if (tab_in_lds(TAB)) {
        stage_shared_table(lds, p.d.ops, p.d.jmax * p.d.mmax, (int)threadIdx.x);
        __syncthreads();
        const int32_t *ir = p.d.inst;
        c.J = ir[JSS_I_JOBS];
        c.M = ir[JSS_I_MACHINES];
        c.max_time_op = ir[JSS_I_MAX_TIME_OP];
        c.max_time_jobs = ir[JSS_I_MAX_TIME_JOBS];
        c.sum_op = ir[JSS_I_SUM_OP];
        c.r_op = as_float(ir[JSS_I_RCP_MAX_TIME_OP]);
        c.r_jobs = as_float(ir[JSS_I_RCP_MAX_TIME_JOBS]);
        c.r_sum = as_float(ir[JSS_I_RCP_SUM_OP]);
        c.r_m = as_float(ir[JSS_I_RCP_MACHINES]);
        c.jvalid = c.gl < c.J;
        c.mvalid = c.gl < c.M;
    }
    if (gw * E >= p.d.batch) return;                 // a wavefront without a single env set (only in the last workgroup)
    if (gw == 0 && lane == 0) wt_store(p.status + 3, slots);

    PEnv<G> e;
    PHeader hd;
    bool never = false;                               // my env was never reset: the session leaves it alone
    // ---- the state of every env set of this wavefront: loaded once ----
    for (int slot = 0; slot < slots; ++slot) {
        const PSlot<G, TAB> sl = p_enter_slot(c, p, lds, wave, gw, n_waves, slot);
        if (!sl.live) break;
        const size_t fe = (size_t)c.first_env;
        PRaw<G> raw = p_issue_loads<G, TAB>(c, p);
        if (tab_global(TAB)) {
            const int4 hx = ld_off<int4>(p.s.env_const + fe * JSS_NC, c.rel * (JSS_NC * 4u));
            if (c.gl < 6) c.norm[c.gl] = ld_off<int>(p.s.env_const + fe * JSS_NC, c.rel * (JSS_NC * 4u) + 16u + (unsigned)c.gl * 4u);
            c.J = hx.x;
            c.M = hx.y;
            c.max_time_op = hx.z;
            c.tid = hx.w;
            c.jvalid = c.gl < c.J;
            c.mvalid = c.gl < c.M;
            wave_lds_sync();
        }
        if (slots > 1) p_park(park, slot, lane, raw, c);
        else {
            hd = p_unpack(e, c, raw, mvtab);
            never = hd.episode == 0;
        }
    }

    // ---- step after step ----
    const unsigned long long *mail = p.mail;
    const size_t B = (size_t)p.d.batch;
    auto granule = [&](int step, int slot) -> const unsigned long long * {     // my env's granule of (step, slot)
        const int fe = (gw + slot * n_waves) * E;
        const int live = fe < p.d.batch;
        int env = fe + lane / G;
        if (!live || env >= p.d.batch) env = live ? p.d.batch - 1 : 0;          // dead lanes poll a neighbour's word
        return mail + (size_t)(step % p.depth) * B + env;
    };
    int my_slots = 0;
    for (int slot = 0; slot < slots; ++slot) my_slots += (gw + slot * n_waves) * E < p.d.batch ? 1 : 0;
    int step = 0;
    int pending = 0;                                  // steps finished whose progress word is not out yet (value to publish)
    bool closing = false, timed_out = false;
    unsigned long long x = fresh_load(granule(0, 0));
    while (!closing) {
        for (int slot = 0; slot < my_slots; ++slot) {
            // the next (step, set) pair's granule is requested before this pair is waited for and computed
            const int nslot = slot + 1 < my_slots ? slot + 1 : 0;
            const int nstep = step + (nslot == 0 ? 1 : 0);
            unsigned long long xn = fresh_load(granule(nstep, nslot));
            const unsigned want = (unsigned)step + 1u;
            long long t0 = 0;
            unsigned spins = 0;
            while (__ballot((unsigned)(x >> 32) != want) != 0) {
                if (pending) {                        // nothing to do anyway: publish the finished step now
                    wt_drain();
                    if (lane == 0) wt_store(p.progress + gw, pending);
                    pending = 0;
                }
                if (spins < 32u) __builtin_amdgcn_s_sleep(2);       // a few quick looks, then back off: thousands of wavefronts
                else if (spins < 1024u) __builtin_amdgcn_s_sleep(16);   // polling flat out would flood the fabric; after ~0.5 ms
                else __builtin_amdgcn_s_sleep(127);                 // of silence (the caller is busy elsewhere) one look per ~3 us
                if ((++spins & 63u) == 0) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > p.timeout_ticks) {
                        timed_out = true;
                        break;
                    }
                }
                x = fresh_load(granule(step, slot));
            }
            const int a = (int)(unsigned)x;           // group-uniform: every lane of a group reads its env's granule
            if (timed_out || __ballot(a == JSS_ACTION_CLOSE) != 0) {
                closing = true;
                break;
            }
            const PSlot<G, TAB> sl = p_enter_slot(c, p, lds, wave, gw, n_waves, slot);
            if (slots > 1) {
                const PRaw<G> raw = p_unpark(park, slot, lane, c);
                hd = p_unpack(e, c, raw, mvtab);
                never = hd.episode == 0;
            }
            if (never) c.alive = false;
            const PStepResult res = p_step_compute<G, TAB, true>(e, hd, c, p, a, mvtab);
            if (pending) {                            // the previous step's stores have had this step's compute to drain: the
                wt_drain();                           // progress word goes out BEFORE any output store of this step is issued
                if (lane == 0) wt_store(p.progress + gw, pending);
                pending = 0;
            }
            p_step_outputs<G, TAB, true>(e, c, p, res);
            const size_t fe = (size_t)c.first_env;
            p_store_mask<G, TAB, true>(e, c, p, p.o.action_mask + fe * (p.d.jmax + 1));
            p_store_obs<G, TAB, true>(e, c, p, p.o.real_obs + fe * p.d.jmax * 7, scratch, sl.whole);
            if (slots > 1) p_park(park, slot, lane, p_pack(e, c, hd), c);
            x = xn;
        }
        if (!closing) {
            ++step;
            pending = step;
        }
    }
    if (pending) {
        wt_drain();
        if (lane == 0) wt_store(p.progress + gw, pending);
    }
    // ---- the state goes back to memory (every row: nothing was loaded to compare with) ----
    for (int slot = 0; slot < my_slots; ++slot) {
        p_enter_slot(c, p, lds, wave, gw, n_waves, slot);
        PRaw<G> raw;
        raw.h = raw.lo = raw.hi = make_int4(0, 0, 0, 0);
        raw.tm = 0;
        if (slots > 1) {
            raw = p_unpark(park, slot, lane, c);
            hd = p_unpack(e, c, raw, mvtab);
            never = hd.episode == 0;
        }
        if (never) c.alive = false;
        p_store(e, c, p, hd, raw, true);
    }
    if (lane == 0) {
        if (timed_out) atomicAdd(p.status + 0, 1);
        atomicAdd(p.status + 2, 1);
    }
    (void)MODE;
}

}  // namespace jss
