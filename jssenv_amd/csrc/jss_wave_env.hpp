#pragma once
// Wave-per-env kernel: one 64-lane wavefront simulates one env (J <= 128, M <= 64).
// Job j on lane j % 64, slot j / 64 (JPL = 1 or 2 slots); machine m on lane m.  The legal job set is a
// wave-uniform 64-bit mask (an SGPR pair per slot): nb_legal_actions is one s_bcnt1, "any legal" one s_cmp,
// and the data-dependent while-loops of step() are scalar branches.  The blocked flag
// (action_illegal_no_op) is only ever needed by its own job: it stays a per-lane bit.
//
// Memory round trips of a step-type call (the wave is latency-bound at the batch sizes this flavour
// serves -- profiles/README.md): ONE mandatory trip -- the env's header and constants record (scalar loads: clock,
// J, M, the observation's normalisers, the op table index) together with the job records and machine
// clocks, none of whose addresses depends on another load (ragged batches: header first, then the rows
// < J(env)) -- plus, when a job moves on to a new op, the 4-byte op table entry that refills its record,
// which is issued inside jump() and not waited for before the state is packed for the store (kPending).
#include "jss_common.hpp"
#include "jss_packed_env.hpp"   // ld_off / st_off / st_nt

namespace jss {

// Per-env constants, all wave-uniform.
struct Ctx {
    int b;
    int J, M;
    int max_time_op;
    int tid;
    const int32_t *tab;  // op table of my env (LDS with kTabLds, global with kTabGlobal), row stride `stride`
    int stride;
    int lane;
    // the observation's normalisers (jss_env.py:102-111) and their float32 reciprocals
    int max_time_jobs, sum_op;
    float r_op, r_jobs, r_sum, r_m;
};

// nxt2 of a job that has just moved on to a new op while the op table entry that refills it is still
// in flight (Env::fill receives it): see jump() / settle()
constexpr int kPending = -2;

template <int JPL>
struct Env {
    int t;                                                               // current_time_step
    int todo[JPL], cur[JPL], nxt[JPL], nxt2[JPL], left[JPL], perf[JPL], idle[JPL], idle_last[JPL], f4[JPL];
    int fill[JPL];                                                       // the load behind a kPending nxt2
    uint64_t legal[JPL];                                                 // legal job set, wave-uniform
    bool blocked[JPL];                                                   // action_illegal_no_op of my job
    int tm;                                                              // lane m: time_until_available_machine[m]
    int noop;                                                            // legal_actions[J]
    int err;
};

template <int JPL>
__device__ __forceinline__ int nb_legal(const Env<JPL> &e) {
    int n = 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) n += __popcll(e.legal[s]);
    return n;
}

template <int JPL>
__device__ __forceinline__ bool any_legal(const Env<JPL> &e) {
    uint64_t m = 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) m |= e.legal[s];
    return m != 0;
}

// value of a per-job register of job `a` (wave-uniform a)
template <int JPL>
__device__ __forceinline__ int job_value(const int (&v)[JPL], int a) {
    int x = v[0];
    if (JPL > 1 && (a >> 6)) x = v[JPL - 1];
    return __builtin_amdgcn_readlane(x, a & 63);
}

// the refill of a record's third cached op has landed: from here on nxt2 is a plain value again
template <int JPL>
__device__ __forceinline__ void settle(Env<JPL> &e) {
#pragma unroll
    for (int s = 0; s < JPL; ++s)
        if (e.nxt2[s] == kPending) e.nxt2[s] = e.fill[s];
}

// ---------------------------------------------------------------------------------------
// reset(): jss_env.py:145-181
// ---------------------------------------------------------------------------------------
template <int JPL, bool WT = false>
__device__ __forceinline__ void reset_env(Env<JPL> &e, const Ctx &c, const Params &p) {
    e.t = 0;                                                             // :154
    e.tm = 0;                                                            // :164
    e.noop = 0;                                                          // :161
    e.err = 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        const bool v = j < c.J;
        e.todo[s] = 0;                                                   // :166
        e.cur[s] = v ? c.tab[j * c.stride] : -1;                         // :174-176 needed machine = op 0
        e.nxt[s] = (v && 1 < c.M) ? c.tab[j * c.stride + 1] : -1;
        e.nxt2[s] = (v && 2 < c.M) ? c.tab[j * c.stride + 2] : -1;
        e.fill[s] = -1;
        e.left[s] = e.perf[s] = e.idle[s] = e.idle_last[s] = 0;          // :165-170
        e.f4[s] = 0;                                                     // :180 state zeros
        e.legal[s] = __ballot(v);                                        // :160
        e.blocked[s] = false;                                            // :171-172
    }
    // solution = -1 (:163): the whole padded [jmax][mmax] block of the env, coalesced (rows behind J(env) too, so
    // that nothing of a previous, larger instance of this env survives a reset)
    int32_t *sol = p.s.solution + (size_t)c.b * p.d.jmax * p.d.mmax;
    const int n = p.d.jmax * p.d.mmax;
    for (int i = c.lane; i < n; i += kWave) st_out<WT, int>(sol, (unsigned)i * 4u, -1);
}

// ---------------------------------------------------------------------------------------
// increase_time_step(): jss_env.py:495-637.  Returns hole_planning.
// Caller guarantees a busy machine exists (the reference pops an empty list otherwise).
// ---------------------------------------------------------------------------------------
template <int JPL>
__device__ __forceinline__ int advance(Env<JPL> &e, const Ctx &c) {
    // next event = earliest machine release  (:517-522; the queue is {t + tm[m] : tm[m] > 0})
    const int d = wave_min(e.tm > 0 ? e.tm : kBig);
    e.t += d;
    bool fin[JPL];
#pragma unroll
    for (int s = 0; s < JPL; ++s) {                                      // job loop :525-601
        const int was = e.left[s];
        const bool v = s * kWave + c.lane < c.J;
        fin[s] = false;
        if (was > 0) {                                                   // :529 running
            const int nl = imax(0, was - d);                             // :534
            e.perf[s] += imin(d, was);                                   // :531,:544
            e.left[s] = nl;
            if (nl == 0) {                                               // :550 op finished
                e.idle[s] += d - was;                                    // :552
                e.idle_last[s] = d - was;                                // :554
                e.todo[s] += 1;                                          // :558
                fin[s] = true;
            }
        } else if (v && e.todo[s] < c.M) {                               // :594 waiting
            e.idle[s] += d;                                              // :596
            e.idle_last[s] += d;                                         // :597
        }
    }
    // machines :604-613.  tm < d only for idle machines (d is the smallest positive tm),
    // so sum(d - tm) over them is d * count.
    const uint64_t idle_m = __ballot(c.lane < c.M && e.tm < d);
    const int hole = d * __popcll(idle_m);                               // :606-608
    e.tm = imax(0, e.tm - d);                                            // :611
    const uint64_t free_m = __ballot(c.lane < c.M && e.tm == 0);         // :616
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        if (fin[s]) {                                                    // :562-566 / :581: the job moves on to the op its
            e.cur[s] = e.nxt[s];                                         // record carries (cur <- nxt <- nxt2); the op two
            e.nxt[s] = e.nxt2[s];                                        // further on is the only op table read of the step
            e.nxt2[s] = (e.todo[s] + 2 < c.M) ? c.tab[j * c.stride + e.todo[s] + 2] : -1;
        }
        const int ncur = e.cur[s];
        // feature 4 numerator: max(0, tm_old[need] - d) (:569-578) == tm_new[need]
        const int tm_need = __shfl(e.tm, (ncur >> 16) & 63);
        if (fin[s]) e.f4[s] = ncur >= 0 ? tm_need : JSS_F4_ONE;          // :586 "1.0" when the job is complete
        // re-legalisation :616-634: need[j] on a free machine, not legal, not blocked.
        // (a job that just completed has cur = -1 and is never legal, :589-591)
        const bool can = j < c.J && ncur >= 0 && ((free_m >> ((ncur >> 16) & 63)) & 1);
        e.legal[s] |= __ballot(can && !e.blocked[s]);
    }
    return hole;
}

// ---------------------------------------------------------------------------------------
// `while nb_legal_actions == 0 (and a machine is busy): increase_time_step()` in one jump to the first time T at which
// a job becomes legal (derivation and the two rare cases: p_jump in jss_packed_env.hpp).  Caller guarantees no legal job.
// A job that finishes inside the jump moves on (cur <- nxt <- nxt2) and the op table entry that refills nxt2 is
// requested here but NOT waited for: nxt2 = kPending until settle() (end of step(); earlier only if a look-ahead
// walk of _check_no_op needs that very entry), so the request's latency hides behind the rest of the step.
// ---------------------------------------------------------------------------------------
template <int JPL>
__device__ __forceinline__ void jump(Env<JPL> &e, const Ctx &c, bool is_nope, int &rn) {
    int tmx[JPL];
    int cand = kBig;
    bool orphan = false;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const bool v = s * kWave + c.lane < c.J;
        const bool running = e.left[s] > 0;
        const bool waiting = v && !running && e.cur[s] >= 0;
        const bool bl = e.blocked[s];
        tmx[s] = __shfl(e.tm, ((running ? e.nxt[s] : e.cur[s]) >> 16) & 63);
        if (running) {
            if (e.nxt[s] >= 0) cand = imin(cand, imax(e.left[s], tmx[s]));
        } else if (waiting && !bl) {
            if (tmx[s] > 0) cand = imin(cand, tmx[s]);
            else orphan = true;
        }
    }
    int T = wave_min(cand);
    const bool any_orphan = __ballot(orphan) != 0;
    if (any_orphan || T >= kBig) {                                       // rare (wave-uniform)
        if (any_orphan) T = imin(T, wave_min(e.tm > 0 ? e.tm : kBig));   // re-legalised at the very next event
        else {
            const int last = wave_max(e.tm);                             // nobody will ever be legal again: run out of events
            T = last > 0 ? last : kBig;
            if (is_nope) e.err |= JSS_ERR_NOPE_IDLE;                     // the reference pops its empty event list (:517)
        }
        if (T >= kBig) {                                                 // nothing busy: no event to advance to
            if (is_nope) e.err |= JSS_ERR_NOPE_IDLE;
            return;
        }
    }
    rn -= wave_sum(c.lane < c.M ? imax(0, T - e.tm) : 0);                // :606-608 summed over the events
    e.t += T;
    e.tm = imax(0, e.tm - T);                                            // :611
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        const bool v = j < c.J;
        const bool running = e.left[s] > 0;
        const bool waiting = v && !running && e.cur[s] >= 0;
        bool can = false;
        if (running) {
            if (e.left[s] <= T) {                                        // finishes at f = left (:550)
                const int f = e.left[s];
                e.perf[s] += f;
                e.left[s] = 0;
                e.todo[s] += 1;
                e.cur[s] = e.nxt[s];
                e.nxt[s] = e.nxt2[s];
                if (e.todo[s] + 2 < c.M) {
                    e.fill[s] = c.tab[j * c.stride + e.todo[s] + 2];     // in flight until settle()
                    e.nxt2[s] = kPending;
                } else {
                    e.nxt2[s] = -1;
                }
                const bool more = e.cur[s] >= 0;
                e.idle[s] += more ? T - f : 0;
                e.idle_last[s] = more ? T - f : 0;
                e.f4[s] = more ? imax(0, tmx[s] - f) : JSS_F4_ONE;
                can = more && tmx[s] <= T;
            } else {
                e.perf[s] += T;
                e.left[s] -= T;
            }
        } else if (waiting) {
            e.idle[s] += T;
            e.idle_last[s] += T;
            can = tmx[s] <= T;
        }
        e.legal[s] |= __ballot(can && !e.blocked[s]);                    // :616-634 at T
    }
}

// ---------------------------------------------------------------------------------------
// _prioritization_non_final(): jss_env.py:183-254
// ---------------------------------------------------------------------------------------
template <int JPL>
__device__ __forceinline__ void prioritize(Env<JPL> &e, const Ctx &c) {
    uint64_t fin_legal[JPL];
    uint64_t any_fin = 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        fin_legal[s] = e.legal[s] & __ballot(e.todo[s] == c.M - 1);      // :217 final ops among legal jobs
        any_fin |= fin_legal[s];
    }
    if (any_fin == 0) return;  // no final op is legal: nothing can be suppressed
    const uint64_t free_m = __ballot(c.lane < c.M && e.tm == 0);
    bool nf[JPL];
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const bool lg = (e.legal[s] >> c.lane) & 1;
        nf[s] = false;
        if (lg && e.todo[s] < c.M - 1) {                                 // :219-239 non-final, next machine idle
            const int next_m = e.nxt[s] >> 16;                           // :227
            nf[s] = (free_m >> next_m) & 1;                              // :234
        }
    }
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        uint64_t todo_bits = fin_legal[s];
        while (todo_bits) {                                              // :244 each final job (wave-uniform loop)
            const int l = __ffsll((unsigned long long)todo_bits) - 1;
            todo_bits &= todo_bits - 1;
            const int cf = __builtin_amdgcn_readlane(e.cur[s], l);
            const int mf = cf >> 16, df = cf & kDurMask;
            uint64_t hit = 0;
#pragma unroll
            for (int q = 0; q < JPL; ++q)  // a non-final job on the same machine strictly shorter: df > min_non_final (:252)
                hit |= __ballot(nf[q] && (e.cur[q] >> 16) == mf && (e.cur[q] & kDurMask) < df);
            if (hit) e.legal[s] &= ~(1ULL << l);                         // :253-254
        }
    }
}

// ---------------------------------------------------------------------------------------
// _check_no_op(): jss_env.py:256-401
// ---------------------------------------------------------------------------------------
struct Horizon {   // pass-1 result: the <= 3 legal machines, their max_horizon_machine, max_horizon
    int mm0, mm1, mm2, mv0, mv1, mv2, mh;
};
__device__ __forceinline__ int walk_op(const Horizon &hz, int op, int tn, int &u) {   // :346-351, :362
    const int m = op >> 16;
    if (m == hz.mm0 && hz.mv0 > tn) u |= 1;
    if (m == hz.mm1 && hz.mv1 > tn) u |= 2;
    if (m == hz.mm2 && hz.mv2 > tn) u |= 4;
    return tn + (op & kDurMask);
}

template <int JPL>
__device__ __forceinline__ void check_no_op(Env<JPL> &e, const Ctx &c) {
    e.noop = 0;                                                          // :278
    const int nl = nb_legal(e);
    if (nl > 4 || nl == 0) return;                                       // :287 (nl == 0: no legal machine, U can never match)
    const uint64_t busy = __ballot(e.tm > 0);
    if (busy == 0) return;                                               // :285 len(next_time_step) > 0
    // PASS 1 (:305-321): sequential in ascending job index over the <= 4 legal jobs; max_horizon
    // sees the running prefix minimum of max_horizon_machine, so the order matters.
    Horizon hz;
    hz.mm0 = hz.mm1 = hz.mm2 = -1;           // the <= 3 legal machines ...
    int n_ml = 0;                            // nb_machine_legal
    int cf[4];                               // packed current op of the i-th legal job (ascending job index)
    {
        uint64_t b0 = e.legal[0];
        uint64_t b1 = JPL > 1 ? e.legal[JPL - 1] : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cf[i] = -1;
            if (i < nl) {
                if (JPL == 1 || b0) {                                    // (one slot: i < nl says a bit is left)
                    const int l = __ffsll((unsigned long long)b0) - 1;
                    b0 &= b0 - 1;
                    cf[i] = __builtin_amdgcn_readlane(e.cur[0], l);
                } else {
                    const int l = __ffsll((unsigned long long)b1) - 1;
                    b1 &= b1 - 1;
                    cf[i] = __builtin_amdgcn_readlane(e.cur[JPL - 1], l);
                }
            }
        }
    }
    // nb_machine_legal = distinct machines of the legal jobs (the :286 gate needs it before pass 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nl) {
            const int m = cf[i] >> 16;
            if (m != hz.mm0 && m != hz.mm1 && m != hz.mm2) {
                if (n_ml == 0) hz.mm0 = m; else if (n_ml == 1) hz.mm1 = m; else if (n_ml == 2) hz.mm2 = m;
                ++n_ml;
            }
        }
    }
    if (n_ml > 3) return;                                                // :286
    const int nxt = e.t + wave_min(e.tm > 0 ? e.tm : kBig);              // :293 next_time_step[0]
    hz.mh = e.t;                                                         // :296
    hz.mv0 = hz.mv1 = hz.mv2 = e.t + c.max_time_op;                      // :300-302
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nl) {
            const int m = cf[i] >> 16;
            const int end = e.t + (cf[i] & kDurMask);                    // :310
            if (end < nxt) return;                                       // :314-315
            int h;
            if (m == hz.mm0) { hz.mv0 = imin(hz.mv0, end); h = hz.mv0; } // :318
            else if (m == hz.mm1) { hz.mv1 = imin(hz.mv1, end); h = hz.mv1; }
            else { hz.mv2 = imin(hz.mv2, end); h = hz.mv2; }
            hz.mh = imax(hz.mh, h);                                      // :321
        }
    }
    // PASS 2 (:324-401): every illegal job walks its future ops; order-free, so one lane per job.  The first
    // ops of the walk are the ones the job record carries (current, next); only a longer walk reads the op table.
    int u = 0;  // bit i: legal machine i "would be better used by waiting"
    const int last = c.M - 1;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        const bool v = j < c.J;
        const bool lg = (e.legal[s] >> c.lane) & 1;
        const bool bl = e.blocked[s];
        const bool caseA = v && !lg && e.left[s] > 0 && e.todo[s] + 1 < c.M;      // :327-330
        const bool caseB = v && !lg && !caseA && !bl && e.todo[s] < c.M;          // :366-369
        const int tm_need = __shfl(e.tm, (e.cur[s] >> 16) & 63);                   // :376
        int k = caseA ? e.todo[s] + 1 : e.todo[s];                                // :332 / :370
        int tn = caseA ? e.t + e.left[s] : e.t + tm_need;                         // :334-337 / :374-377
        bool go = (caseA || caseB) && k < last && hz.mh > tn;                     // :340-342 / :380-382
        if (go && caseB) {                                                        // op k == todo: the current op
            tn = walk_op(hz, e.cur[s], tn, u);
            ++k;
            go = k < last && hz.mh > tn;
        }
        if (go) {                                                                 // op k == todo + 1: the next op
            tn = walk_op(hz, e.nxt[s], tn, u);
            ++k;
            go = k < last && hz.mh > tn;
        }
        if (__ballot(go && e.nxt2[s] == kPending) != 0) settle(e);                // rare: the walk needs the entry being refilled
        if (go) {                                                                 // op k == todo + 2: the one after it
            tn = walk_op(hz, e.nxt2[s], tn, u);
            ++k;
            go = k < last && hz.mh > tn;
        }
#ifndef JSS_EXP_NO_DEEP_WALK   // A/B builds only (wrong results): what the table reads of the look-ahead cost
        if (go) {                                                                 // further: the op table, two entries per trip
            const int32_t *row = c.tab + j * c.stride;
            do {
                const int op0 = row[k], op1 = row[k + 1];                         // k + 1 <= M - 1: inside the row
                tn = walk_op(hz, op0, tn, u);
                ++k;
                if (k < last && hz.mh > tn) {
                    tn = walk_op(hz, op1, tn, u);
                    ++k;
                }
            } while (k < last && hz.mh > tn);
        }
#endif
    }
    const int covered = (__ballot(u & 1) != 0) + (__ballot(u & 2) != 0) + (__ballot(u & 4) != 0);
    e.noop = (covered == n_ml) ? 1 : 0;                                  // :357-359 / :395-397
}

// ---------------------------------------------------------------------------------------
// step(): jss_env.py:403-481.  `a` is wave-uniform.  Returns the reward numerator
// (reward * max_time_op, an exact integer: scheduled duration minus idle-machine time).
// ---------------------------------------------------------------------------------------
template <int JPL, bool WT = false>
__device__ __forceinline__ int step_env(Env<JPL> &e, const Ctx &c, const Params &p, int a) {
    if (a == JSS_ACTION_SKIP || a == JSS_ACTION_RESET) return 0;         // RESET is handled by the caller
    if (a < 0 || a > c.J) {
        e.err |= JSS_ERR_BAD_ACTION;
        return 0;
    }
    int rn = 0;
    if (a == c.J) {                                                      // :419 NOPE
#pragma unroll
        for (int s = 0; s < JPL; ++s) {                                  // :422-428
            e.blocked[s] = e.blocked[s] || ((e.legal[s] >> c.lane) & 1);
            e.legal[s] = 0;
        }
        if (!JSS_ABLATED(p, JSS_ABLATE_ADVANCE)) jump(e, c, true, rn);   // :429-430 in one jump
    } else {                                                             // :441 allocate job a
        const int sa = a >> 6, la = a & 63;
        uint64_t lg = e.legal[0];
        if (JPL > 1 && sa) lg = e.legal[JPL - 1];
        if (!((lg >> la) & 1)) {  // outside the mask: the reference corrupts its counters; we ignore + flag
            e.err |= JSS_ERR_ILLEGAL_ACTION;
            return 0;
        }
        const int ca = job_value<JPL>(e.cur, a);
        const int k = job_value<JPL>(e.todo, a);                         // :442
        const int m = ca >> 16;                                          // :443
        const int d = ca & kDurMask;                                     // :444
        rn = d;                                                          // :445
        if (c.lane == m) e.tm = d;                                       // :446
#pragma unroll
        for (int s = 0; s < JPL; ++s)
            if (s == sa && c.lane == la) e.left[s] = d;                  // :447
        if (c.lane == 0) st_out<WT, int>(p.s.solution + ((size_t)c.b * p.d.jmax + a) * p.d.mmax + k, 0u, e.t);  // :454
#pragma unroll
        for (int s = 0; s < JPL; ++s) {
            const uint64_t same = __ballot(e.cur[s] >= 0 && (e.cur[s] >> 16) == m);   // padding lanes hold cur = -1
            const bool mine = e.cur[s] >= 0 && (e.cur[s] >> 16) == m;
            e.legal[s] &= ~same;                                         // :455-463
            if (mine) e.blocked[s] = false;                              // :464-467
        }
        if (!any_legal(e) && !JSS_ABLATED(p, JSS_ABLATE_ADVANCE)) jump(e, c, false, rn);   // :469-470 in one jump
    }
    JSS_STAMP(p, c.b, 7, e.left[0] + e.t);
    if (!JSS_ABLATED(p, JSS_ABLATE_PRIORITIZE)) prioritize(e, c);        // :432 / :471
    JSS_STAMP(p, c.b, 8, (int)e.legal[0]);
    if (!JSS_ABLATED(p, JSS_ABLATE_CHECK_NO_OP)) check_no_op(e, c);      // :433 / :472
    JSS_STAMP(p, c.b, 9, e.noop);
    settle(e);                                                           // the refill jump() requested has had the rest of the step to land
    return rn;
}

// ---------------------------------------------------------------------------------------
// action selectors.  Returns a wave-uniform action, -1 when nothing is legal.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int nth_set_bit(uint64_t mask, int n, int lane) {
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    const uint64_t hit = __ballot(((mask >> lane) & 1) && (int)below == n);
    return __ffsll((unsigned long long)hit) - 1;
}

// F64: the instantiation also carries JSS_POLICY_CR_F64's float64 selector (the policy kernels only)
template <int JPL, bool F64 = false>
__device__ __forceinline__ int select_action(const Env<JPL> &e, const Ctx &c, const Params &p, uint64_t env_id,
                                             uint32_t episode, uint32_t step) {
    const int kind = p.kind & 0xFF;
    const int nl = nb_legal(e);
    const int n = nl + (e.noop ? 1 : 0);
    if (n == 0) return -1;
    if (kind == JSS_POLICY_RANDOM) {
        // README.md:58-60: uniform over the set bits of the mask, NOPE included
        const uint32_t r = rng_u32(p.seed, env_id, episode, step);
        int pick = (int)__umulhi(r, (uint32_t)n);
        int a = c.J;
        bool found = false;
#pragma unroll
        for (int s = 0; s < JPL; ++s) {
            const int cnt = __popcll(e.legal[s]);
            const int idx = nth_set_bit(e.legal[s], pick, c.lane);
            if (!found && pick < cnt) {
                a = s * kWave + idx;
                found = true;
            }
            pick -= cnt;
        }
        return a;
    }
    if (nl == 0) return c.J;  // only NOPE is legal (dispatching.py:96-97)
    int a = -1;
    // remaining-work table of my env: rem[j][k] = durations of ops k..M-1 of job j (MWR / LWR / CR)
    const int32_t *rem = p.d.rem + (size_t)c.tid * p.region_ints;
    if (F64 && kind == JSS_POLICY_CR && ((p.kind >> 24) & 1)) {
        CrKeyF best;
        best.ratio = kCrInf;
        best.idx = kCrNone;
#pragma unroll
        for (int s = 0; s < JPL; ++s) {
            const int j = s * kWave + c.lane;
            const bool lg = (e.legal[s] >> c.lane) & 1;
            CrKeyF key;
            key.ratio = lg ? cr_ratio_f64(rem[j * c.stride], p.d.cr_factor, e.t, rem[j * c.stride + e.todo[s]]) : kCrInf;
            key.idx = lg ? j : kCrNone;
            if (cr_better_f64(key, best)) best = key;
        }
        best = cr_argmin_f64<kWave>(best);
        a = __builtin_amdgcn_readfirstlane(best.idx);
    } else if (kind == JSS_POLICY_CR) {                                  // dispatching.py:365-408
        CrKey best;
        best.num = 0x3fffffff;
        best.den = 1;
        best.idx = kCrNone;
#pragma unroll
        for (int s = 0; s < JPL; ++s) {
            const int j = s * kWave + c.lane;
            const bool lg = (e.legal[s] >> c.lane) & 1;
            CrKey key;
            key.num = lg ? cr_p(p) * rem[j * c.stride] - cr_q(p) * e.t : 0x3fffffff; // :373 job length
            key.den = lg ? rem[j * c.stride + e.todo[s]] : 1;            // :391 remaining work
            key.idx = lg ? j : kCrNone;
            if (cr_better(key, best)) best = key;
        }
        best = cr_argmin<kWave>(best);
        a = __builtin_amdgcn_readfirstlane(best.idx);
    } else {
        int key[JPL];
        const bool larger = (kind == JSS_POLICY_FIFO || kind == JSS_POLICY_MWR || kind == JSS_POLICY_MOR);
#pragma unroll
        for (int s = 0; s < JPL; ++s) {
            const int j = s * kWave + c.lane;
            const bool lg = (e.legal[s] >> c.lane) & 1;
            int v;
            if (kind == JSS_POLICY_FIFO) v = e.idle_last[s];                 // dispatching.py:146
            else if (kind == JSS_POLICY_SPT) v = e.cur[s] & kDurMask;        // :105-106
            else if (kind == JSS_POLICY_MOR || kind == JSS_POLICY_LOR) v = c.M - e.todo[s];  // :273 / :314
            else v = lg ? rem[j * c.stride + e.todo[s]] : 0;                 // MWR / LWR :187-189 / :230-232
            key[s] = lg ? (larger ? v : -v) : -kBig;
        }
        int best = key[0];
#pragma unroll
        for (int s = 1; s < JPL; ++s) best = imax(best, key[s]);
        best = wave_max(best);
#pragma unroll
        for (int s = 0; s < JPL; ++s) {  // strict comparisons in the reference: the first index wins ties
            const uint64_t hit = __ballot(key[s] == best) & e.legal[s];
            if (a < 0 && hit) a = s * kWave + __ffsll((unsigned long long)hit) - 1;
        }
    }
    if (e.noop && p.explore_q16 != 0) {                                  // dispatching.py:113: 10 % NOPE when NOPE is legal
        const uint32_t r = rng_u32(p.seed ^ kExploreSeedXor, env_id, episode, step);
        if ((r >> 16) < p.explore_q16) a = c.J;
    }
    return a;
}

// ---------------------------------------------------------------------------------------
// HBM <-> registers.  One 32-byte record per job (two dwordx4 per lane), a 16-byte header + 48 bytes of constants per env.
// The env index is wave-uniform, so every base is an SGPR pair and the lane offset 32 bits.
// ---------------------------------------------------------------------------------------
struct Header {
    int episode, step;
};

template <int JPL>
struct RawEnv {  // what the state loads returned: unchanged halves of a record are not stored back
    int4 lo[JPL], hi[JPL];
    int tm;
};

// Rows j < jlimit of the env's job records + its machine clocks.  Lanes behind the limit issue no request and hold
// the record of a job that does not exist (todo 0, no op, nothing running) -- which is also what reset() leaves in
// the rows between J(env) and jmax, so nothing has to be masked after the load.
template <int JPL, int TAB>
__device__ __forceinline__ RawEnv<JPL> issue_loads(int b, int lane, const Params &p, int jlimit) {
    RawEnv<JPL> r;
    const int32_t *jb = p.s.job + (size_t)b * p.d.jmax * tab_record_ints(TAB);
    // compact / medium batches keep no machine clocks in memory: a machine is busy for as long as the job on it (unpack_env)
    r.tm = tab_no_clocks(TAB) ? 0 : ld_off<int>(p.s.machine + (size_t)b * p.d.mmax, (unsigned)(lane < p.d.mmax ? lane : 0) * 4u);
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + lane;
        if (tab_compact(TAB)) {          // 16-byte records (JSS_FC_*): one access per job
            r.lo[s] = make_int4(0, 0, 0, 0);
            r.hi[s] = make_int4(0, 0, 0, 0);
            if (j < jlimit) r.lo[s] = ld_off<int4>(jb, (unsigned)j * (JSS_NFC * 4u));
        } else if (tab_medium(TAB)) {    // 24-byte records (JSS_FM_*): three 8-byte accesses; "no job" is all zeros
            r.lo[s] = make_int4(0, 0, 0, 0);
            r.hi[s] = make_int4(0, 0, 0, 0);
            if (j < jlimit) {
                const unsigned jo = (unsigned)j * (JSS_NFM * 4u);
                const int2 a = ld_off<int2>(jb, jo), bb = ld_off<int2>(jb, jo + 8u), d = ld_off<int2>(jb, jo + 16u);
                r.lo[s] = make_int4(a.x, a.y, bb.x, bb.y);
                r.hi[s] = make_int4(d.x, d.y, 0, 0);
            }
        } else {
            r.lo[s] = make_int4(0, -1, 0, 0);
            r.hi[s] = make_int4(0, 0, 0, -1);
            if (j < jlimit) {
                r.lo[s] = ld_off<int4>(jb, (unsigned)j * 32u);
                r.hi[s] = ld_off<int4>(jb, (unsigned)j * 32u + 16u);
            }
        }
    }
    return r;
}

template <int JPL, int TAB>
__device__ __forceinline__ RawEnv<JPL> blank_raw() {                      // reset: nothing is read, everything is written
    RawEnv<JPL> r;
    r.tm = 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        r.lo[s] = tab_no_clocks(TAB) ? make_int4(0, 0, 0, 0) : make_int4(0, -1, 0, 0);   // (compact / medium: "no job" is all zeros)
        r.hi[s] = tab_no_clocks(TAB) ? make_int4(0, 0, 0, 0) : make_int4(0, 0, 0, -1);
    }
    return r;
}

template <int JPL, int TAB>
__device__ __forceinline__ void unpack_env(Env<JPL> &e, const Ctx &c, const RawEnv<JPL> &r, int clock, int status, int32_t *scr) {
    e.t = clock;
    e.err = status & 0xFF;
    e.noop = (status & JSS_STATUS_NOOP) ? 1 : 0;
    e.tm = c.lane < c.M ? r.tm : 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int4 lo = r.lo[s], hi = r.hi[s];
        if (tab_compact(TAB)) {      // the job's next three ops are what the LDS table says
            const unsigned w0 = (unsigned)lo.x, w1 = (unsigned)lo.y;
            e.todo[s] = (int)(w0 & JSS_FC_TODO_MASK);
            const int j = s * kWave + c.lane, k = e.todo[s];
            const bool v = j < c.J;
            e.left[s] = (int)(w1 & 0xffffu);
            e.perf[s] = (int)(w0 >> JSS_FC_PERF_SHIFT);
            e.idle[s] = lo.z;
            e.idle_last[s] = lo.w;
            e.f4[s] = (w0 & JSS_FC_FLAG_F4_ONE) ? JSS_F4_ONE : (int)(w1 >> 16);
            e.legal[s] = __ballot((w0 & JSS_FC_FLAG_LEGAL) != 0);
            e.blocked[s] = (w0 & JSS_FC_FLAG_BLOCKED) != 0;
            e.cur[s] = (v && k < c.M) ? c.tab[j * c.stride + k] : -1;
            e.nxt[s] = (v && k + 1 < c.M) ? c.tab[j * c.stride + k + 1] : -1;
            e.nxt2[s] = (v && k + 2 < c.M) ? c.tab[j * c.stride + k + 2] : -1;
        } else if (tab_medium(TAB)) {    // the three cached ops travel in the record, 21 bits each (0 = none)
            const unsigned w0 = (unsigned)lo.x, w1 = (unsigned)lo.y, w2 = (unsigned)lo.z, w3 = (unsigned)lo.w;
            const unsigned cur = (w0 >> JSS_FM_CUR_SHIFT) & JSS_FM_OP_MASK;
            const unsigned nxt = (w2 >> 21) | ((w3 & 0x3FFu) << 11), nxt2 = (w3 >> 10) & JSS_FM_OP_MASK;
            e.todo[s] = (int)(w0 & JSS_FM_TODO_MASK);
            e.left[s] = (int)(w1 & 0xffffu);
            e.perf[s] = (int)(w2 & JSS_FM_OP_MASK);
            e.idle[s] = hi.x;
            e.idle_last[s] = hi.y;
            e.f4[s] = (w0 & JSS_FM_FLAG_F4_ONE) ? JSS_F4_ONE : (int)(w1 >> 16);
            e.legal[s] = __ballot((w0 & JSS_FM_FLAG_LEGAL) != 0);
            e.blocked[s] = (w0 & JSS_FM_FLAG_BLOCKED) != 0;
            e.cur[s] = cur ? (int)cur : -1;
            e.nxt[s] = nxt ? (int)nxt : -1;
            e.nxt2[s] = nxt2 ? (int)nxt2 : -1;
        } else {
            e.todo[s] = lo.x & JSS_TODO_MASK;
            e.legal[s] = __ballot((lo.x & JSS_FLAG_LEGAL) != 0);
            e.blocked[s] = (lo.x & JSS_FLAG_BLOCKED) != 0;
            e.cur[s] = lo.y;
            e.left[s] = lo.z;
            e.perf[s] = lo.w;
            e.idle[s] = hi.x;
            e.idle_last[s] = hi.y;
            e.f4[s] = hi.z;
            e.nxt[s] = hi.w;
            const int n2 = (int)((unsigned)lo.x >> JSS_NEXT2_SHIFT);
            e.nxt2[s] = n2 ? n2 : -1;
        }
        e.fill[s] = -1;
    }
    if (tab_no_clocks(TAB)) {
        // time_until_available_machine[m] == time_until_finish_current_op_jobs[the job running on m] (both are set to
        // the op's duration at :446-449 and count down together at :521-530), 0 for an idle machine.  scr: kWave ints
        scr[c.lane] = 0;
        wave_lds_sync();
#pragma unroll
        for (int s = 0; s < JPL; ++s)
            if (e.left[s] > 0 && e.cur[s] >= 0) scr[(e.cur[s] >> 16) & (kWave - 1)] = e.left[s];
        wave_lds_sync();
        e.tm = c.lane < c.M ? scr[c.lane] : 0;
        wave_lds_sync();                                                 // scr is the observation image later on
    }
}

// action mask row: legal jobs, the NOPE flag at index J, zeros behind it
template <int JPL, bool WT = false>
__device__ __forceinline__ void store_mask(const Env<JPL> &e, const Ctx &c, uint8_t *mk, int jm) {
    if (c.lane == 0) st_out<WT, uint8_t>(mk, (unsigned)jm, (uint8_t)(c.J == jm ? e.noop : 0));   // last byte of the row
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        const int lg = (int)((e.legal[s] >> c.lane) & 1);
        if (j < jm) st_out<WT, uint8_t>(mk, (unsigned)j, (uint8_t)(j < c.J ? lg : (j == c.J ? e.noop : 0)));
    }
}

// The record words of my jobs as they are stored (the inverse of unpack_env)
template <int JPL, int TAB>
__device__ __forceinline__ RawEnv<JPL> pack_env(const Env<JPL> &e, const Ctx &c) {
    RawEnv<JPL> r;
    r.tm = e.tm;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int lg = (int)((e.legal[s] >> c.lane) & 1), bl = e.blocked[s] ? 1 : 0;
        if (tab_compact(TAB)) {
            const bool one = e.f4[s] == JSS_F4_ONE;
            r.lo[s] = make_int4((int)((unsigned)e.todo[s] | (lg ? JSS_FC_FLAG_LEGAL : 0u) | (bl ? JSS_FC_FLAG_BLOCKED : 0u) |
                                      (one ? JSS_FC_FLAG_F4_ONE : 0u) | ((unsigned)e.perf[s] << JSS_FC_PERF_SHIFT)),
                                (int)((unsigned)e.left[s] | ((unsigned)(one ? 0 : e.f4[s]) << 16)), e.idle[s], e.idle_last[s]);
            r.hi[s] = make_int4(0, 0, 0, 0);
        } else if (tab_medium(TAB)) {
            const bool one = e.f4[s] == JSS_F4_ONE;
            const unsigned cur = e.cur[s] >= 0 ? (unsigned)e.cur[s] : 0u, nxt = e.nxt[s] >= 0 ? (unsigned)e.nxt[s] : 0u;
            const unsigned nxt2 = e.nxt2[s] >= 0 ? (unsigned)e.nxt2[s] : 0u;
            r.lo[s] = make_int4((int)((unsigned)e.todo[s] | (lg ? JSS_FM_FLAG_LEGAL : 0u) | (bl ? JSS_FM_FLAG_BLOCKED : 0u) |
                                      (one ? JSS_FM_FLAG_F4_ONE : 0u) | (cur << JSS_FM_CUR_SHIFT)),
                                (int)((unsigned)e.left[s] | ((unsigned)(one ? 0 : e.f4[s]) << 16)),
                                (int)((unsigned)e.perf[s] | (nxt << 21)), (int)((nxt >> 11) | (nxt2 << 10)));
            r.hi[s] = make_int4(e.idle[s], e.idle_last[s], 0, 0);
        } else {
            r.lo[s] = make_int4(e.todo[s] | (lg ? JSS_FLAG_LEGAL : 0) | (bl ? JSS_FLAG_BLOCKED : 0) |
                                    (e.nxt2[s] >= 0 ? (int)((unsigned)e.nxt2[s] << JSS_NEXT2_SHIFT) : 0), e.cur[s], e.left[s], e.perf[s]);
            r.hi[s] = make_int4(e.idle[s], e.idle_last[s], e.f4[s], e.nxt[s]);
        }
    }
    return r;
}

// State back to HBM.  all_rows = the env was (re)initialised: every row of the padded block is written (rows behind
// J(env) as "no job"); otherwise rows < J(env), and of those only the halves that changed.
// DIFF = false (the modes that loop over steps with the state in registers: one store per K steps): every row of a job is
// written without comparing it with what was loaded -- `raw` is dead from the unpack on instead of live through the whole
// loop (9 VGPRs per job slot: what kept the two-jobs-per-lane recorder in scratch memory).
// DIFF: kStoreAll / kStoreCompare / kStoreCause (jss_packed_env.hpp, p_store)
template <int JPL, int TAB, int DIFF = kStoreCompare>
__device__ __forceinline__ void store_env(const Env<JPL> &e, const Ctx &c, const Params &p, const Header &hd,
                                          const RawEnv<JPL> &raw, bool all_rows, bool moved = false, int a_sched = -1) {
    const bool adv = DIFF == kStoreCause && moved;
    const int jm = p.d.jmax;
    int32_t *jb = p.s.job + (size_t)c.b * jm * tab_record_ints(TAB);
    if (c.lane == 0) {
        *reinterpret_cast<int4 *>(p.s.env + (size_t)c.b * JSS_NH) =
            make_int4(e.t, hd.episode, hd.step, (e.err & 0xFF) | (e.noop ? JSS_STATUS_NOOP : 0));
        if (all_rows) {   // the instance constants of the env travel with it from here on (include/jss_hip.h JSS_C_*)
            int32_t *cp = p.s.env_const + (size_t)c.b * JSS_NC;
            *reinterpret_cast<int4 *>(cp) = make_int4(c.J, c.M, c.max_time_op, c.tid);
            *reinterpret_cast<int4 *>(cp + 4) = make_int4(c.max_time_jobs, c.sum_op, as_int(c.r_op), as_int(c.r_jobs));
            *reinterpret_cast<int4 *>(cp + 8) = make_int4(as_int(c.r_sum), as_int(c.r_m), 0, 0);
        }
    }
    if (tab_no_clocks(TAB)) {
        // no machine clocks in memory (unpack_env)
    } else if (all_rows) {
        if (c.lane < p.d.mmax) st_off<int>(p.s.machine + (size_t)c.b * p.d.mmax, (unsigned)c.lane * 4u, e.tm);
    } else if (c.lane < c.M && (DIFF == kStoreAll || e.tm != raw.tm)) {  // idle machines stay 0
        st_off<int>(p.s.machine + (size_t)c.b * p.d.mmax, (unsigned)c.lane * 4u, e.tm);
    }
    const RawEnv<JPL> now = pack_env<JPL, TAB>(e, c);
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        const int4 lo = now.lo[s], hi = now.hi[s];
        // kStoreCause: the part of a record that holds word 0 / the time left, and the parts only a clock move touches
        const bool w0_changed = lo.x != raw.lo[s].x;
        const bool head_dirty = w0_changed || j == a_sched || (adv && e.cur[s] >= 0);
        const bool rest_dirty = adv && (e.cur[s] >= 0 || w0_changed);
        if (tab_medium(TAB)) {           // the thirds of the record that changed
            const unsigned jo = (unsigned)j * (JSS_NFM * 4u);
            if (all_rows ? j < jm : j < c.J) {
                if (DIFF == kStoreCause && !all_rows) {
                    if (head_dirty) st_off<int2>(jb, jo, make_int2(lo.x, lo.y));
                    if (rest_dirty) {
                        st_off<int2>(jb, jo + 8u, make_int2(lo.z, lo.w));
                        st_off<int2>(jb, jo + 16u, make_int2(hi.x, hi.y));
                    }
                } else {
                    const int4 lo0 = raw.lo[s], hi0 = raw.hi[s];
                    if (DIFF == kStoreAll || all_rows || lo.x != lo0.x || lo.y != lo0.y) st_off<int2>(jb, jo, make_int2(lo.x, lo.y));
                    if (DIFF == kStoreAll || all_rows || lo.z != lo0.z || lo.w != lo0.w) st_off<int2>(jb, jo + 8u, make_int2(lo.z, lo.w));
                    if (DIFF == kStoreAll || all_rows || hi.x != hi0.x || hi.y != hi0.y) st_off<int2>(jb, jo + 16u, make_int2(hi.x, hi.y));
                }
            }
            continue;
        }
        if (tab_compact(TAB)) {
            const unsigned jo = (unsigned)j * (JSS_NFC * 4u);
            bool dirty;
            if (DIFF == kStoreCause) dirty = head_dirty || rest_dirty;
            else {
                const int4 lo0 = raw.lo[s];
                dirty = DIFF == kStoreAll || lo.x != lo0.x || lo.y != lo0.y || lo.z != lo0.z || lo.w != lo0.w;
            }
            if (all_rows ? j < jm : (j < c.J && dirty)) st_off<int4>(jb, jo, lo);
            continue;
        }
        if (all_rows) {
            if (j < jm) {
                st_off<int4>(jb, (unsigned)j * 32u, lo);
                st_off<int4>(jb, (unsigned)j * 32u + 16u, hi);
            }
            // a one-job-per-lane body inside rows wider than its 64 lanes (a class of the fused grid on padded tensors): the rows
            // behind the lanes as "no job" records too, like the full-width reset writes them (same bytes whichever path resets)
            if (s == JPL - 1)
                for (int r = JPL * kWave + c.lane; r < jm; r += kWave) {
                    st_off<int4>(jb, (unsigned)r * 32u, make_int4(0, -1, 0, 0));
                    st_off<int4>(jb, (unsigned)r * 32u + 16u, make_int4(0, 0, 0, -1));
                }
        } else if (j < c.J) {   // steps without a time advance touch few jobs
            if (DIFF == kStoreCause) {
                if (head_dirty) st_off<int4>(jb, (unsigned)j * 32u, lo);
                if (rest_dirty) st_off<int4>(jb, (unsigned)j * 32u + 16u, hi);
            } else {
                const int4 lo0 = raw.lo[s], hi0 = raw.hi[s];
                if (DIFF == kStoreAll || lo.x != lo0.x || lo.y != lo0.y || lo.z != lo0.z || lo.w != lo0.w) st_off<int4>(jb, (unsigned)j * 32u, lo);
                if (DIFF == kStoreAll || hi.x != hi0.x || hi.y != hi0.y || hi.z != hi0.z || hi.w != hi0.w) st_off<int4>(jb, (unsigned)j * 32u + 16u, hi);
            }
        }
    }
}

// The (J,7) observation of jss_env.py:102-111, float32, to `dst` (first float of the env's [jmax][7] block).  Every
// column is a function of the integer state (column 4 of its own stored numerator: the reference writes it only
// when an op finishes).  Transposed through LDS so the HBM write is rows*7 contiguous floats; the LDS image is
// shifted by the block's misalignment so that 16-byte lines of the image are 16-byte lines of the destination
// (a 50-job block is 1400 bytes: every other env starts 8 bytes past a line) and everything but the first and
// last line leaves as streaming dwordx4 stores.
template <int JPL, bool WT = false>
__device__ __forceinline__ void store_obs(const Env<JPL> &e, const Ctx &c, float *dst, float *scratch, int rows) {
    // rows = jmax when the env is (re)initialised by a reset call, J(env) otherwise: the rows behind J are zeros
    // from that reset on and nothing ever changes them, so a step does not rewrite them (ragged, padded batches)
    const float f_op = (float)c.max_time_op, f_m = (float)c.M, f_jobs = (float)c.max_time_jobs, f_sum = (float)c.sum_op;
    const int sh = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);   // floats past a 16-byte boundary (wave-uniform)
    float *img = scratch + sh;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        if (j < rows) {
            const bool v = j < c.J;  // padding rows are written as zeros
            float *row = img + j * 7;
            row[0] = v ? (float)((e.legal[s] >> c.lane) & 1) : 0.f;                       // :130
            row[1] = div_by((float)e.left[s], f_op, c.r_op);                             // :448, :539
            row[2] = div_by((float)e.todo[s], f_m, c.r_m);                               // :559
            row[3] = div_by((float)e.perf[s], f_jobs, c.r_jobs);                       // :545
            row[4] = e.f4[s] == JSS_F4_ONE ? 1.0f : div_by((float)e.f4[s], f_op, c.r_op);  // :569-586
            row[5] = div_by((float)e.idle_last[s], f_sum, c.r_sum);                    // :555, :600
            row[6] = div_by((float)e.idle[s], f_sum, c.r_sum);                         // :553, :601
        }
    }
    wave_lds_sync();
    float *dst0 = dst - sh;                                              // 16-byte aligned
    const int end = sh + rows * 7;                                       // image floats [sh, end) are the block
    // whole 16-byte lines [i0, i1) as streaming dwordx4 stores (whole lines, never read back: see st_nt); the <= 3
    // floats in front of the first line and behind the last one as single dwords
    const int i0 = (sh + 3) >> 2, i1 = end >> 2;
    for (int i = i0 + c.lane; i < i1; i += kWave) {
        if (WT) wt_store16(dst0, (unsigned)i * 16u, reinterpret_cast<const float4 *>(scratch)[i]);
        else st_nt(dst0, (unsigned)i * 16u, reinterpret_cast<const float4 *>(scratch)[i]);
    }
    const int head_end = imin(i0 << 2, end), tail_begin = imax(i1 << 2, head_end);
    if (sh + c.lane < head_end) st_out<WT, float>(dst0, (unsigned)(sh + c.lane) * 4u, scratch[sh + c.lane]);
    if (tail_begin + c.lane < end) st_out<WT, float>(dst0, (unsigned)(tail_begin + c.lane) * 4u, scratch[tail_begin + c.lane]);
    wave_lds_sync();
}

// _reward_scaler (jss_env.py:483-493): the integer numerator over max_time_op, with the record's reciprocal and one
// residual correction like the observation (<= 1 ulp; the host-core twin evaluates the same sequence)
__device__ __forceinline__ float reward_of(int rn, const Ctx &c) {
    return div_by((float)rn, (float)c.max_time_op, c.r_op);
}

// The env's instance constants from the instance record (reset paths; step-type calls take them from the header)
__device__ __forceinline__ void ctx_from_instance(Ctx &c, const Params &p, int tid) {
    const int32_t *ir = p.d.inst + (size_t)tid * JSS_NI;
    c.tid = tid;
    c.J = __builtin_amdgcn_readfirstlane(ir[JSS_I_JOBS]);
    c.M = __builtin_amdgcn_readfirstlane(ir[JSS_I_MACHINES]);
    c.max_time_op = __builtin_amdgcn_readfirstlane(ir[JSS_I_MAX_TIME_OP]);
    c.max_time_jobs = __builtin_amdgcn_readfirstlane(ir[JSS_I_MAX_TIME_JOBS]);
    c.sum_op = __builtin_amdgcn_readfirstlane(ir[JSS_I_SUM_OP]);
    c.r_op = as_float(__builtin_amdgcn_readfirstlane(ir[JSS_I_RCP_MAX_TIME_OP]));
    c.r_jobs = as_float(__builtin_amdgcn_readfirstlane(ir[JSS_I_RCP_MAX_TIME_JOBS]));
    c.r_sum = as_float(__builtin_amdgcn_readfirstlane(ir[JSS_I_RCP_SUM_OP]));
    c.r_m = as_float(__builtin_amdgcn_readfirstlane(ir[JSS_I_RCP_MACHINES]));
}

template <int TAB>
__device__ __forceinline__ void ctx_table(Ctx &c, const Params &p, const int32_t *lds) {
    c.stride = p.d.mmax;
    c.tab = tab_in_lds(TAB) ? lds : p.d.ops + (size_t)c.tid * p.region_ints;
}

// Header and constants record of env b as wave-uniform values (scalar loads: nothing in this kernel has written them yet)
struct HeaderWords {
    int clock, episode, step, status, J, M, max_time_op, tid;
    int max_time_jobs, sum_op, r_op, r_jobs, r_sum, r_m;
};
__device__ __forceinline__ HeaderWords load_header(const Params &p, int b) {
    const int32_t *hp = p.s.env + (size_t)b * JSS_NH;
    const int32_t *cp = p.s.env_const + (size_t)b * JSS_NC;
    HeaderWords h;
    h.clock = hp[JSS_H_CLOCK];
    h.episode = hp[JSS_H_EPISODE];
    h.step = hp[JSS_H_STEP];
    h.status = hp[JSS_H_STATUS];
    h.J = cp[JSS_C_JOBS];
    h.M = cp[JSS_C_MACHINES];
    h.max_time_op = cp[JSS_C_MAX_TIME_OP];
    h.tid = cp[JSS_C_TABLE];
    h.max_time_jobs = cp[JSS_C_MAX_TIME_JOBS];
    h.sum_op = cp[JSS_C_SUM_OP];
    h.r_op = cp[JSS_C_RCP_MAX_TIME_OP];
    h.r_jobs = cp[JSS_C_RCP_MAX_TIME_JOBS];
    h.r_sum = cp[JSS_C_RCP_SUM_OP];
    h.r_m = cp[JSS_C_RCP_MACHINES];
    return h;
}
__device__ __forceinline__ void ctx_from_header(Ctx &c, const HeaderWords &h) {
    c.J = __builtin_amdgcn_readfirstlane(h.J);
    c.M = __builtin_amdgcn_readfirstlane(h.M);
    c.max_time_op = __builtin_amdgcn_readfirstlane(h.max_time_op);
    c.tid = __builtin_amdgcn_readfirstlane(h.tid);
    c.max_time_jobs = in_vgpr(__builtin_amdgcn_readfirstlane(h.max_time_jobs));     // consumed by the observation's VALU code only
    c.sum_op = in_vgpr(__builtin_amdgcn_readfirstlane(h.sum_op));
    c.r_op = as_float(in_vgpr(__builtin_amdgcn_readfirstlane(h.r_op)));
    c.r_jobs = as_float(in_vgpr(__builtin_amdgcn_readfirstlane(h.r_jobs)));
    c.r_sum = as_float(in_vgpr(__builtin_amdgcn_readfirstlane(h.r_sum)));
    c.r_m = as_float(in_vgpr(__builtin_amdgcn_readfirstlane(h.r_m)));
}

// One jss_step call on the registers, in two halves: the computation -- the JSS_ACTION_RESET restart, step(), the
// header's step count -- and the env's scalar outputs: reward / done / makespan / counters (a skipped env keeps them).
// WT = the stores are write-through (step session, which puts its progress word between the two halves).
struct StepResult {
    int rn;
    bool called, restart, done;
};
// RESTART = false: the caller has routed JSS_ACTION_RESET elsewhere (jss_step sends such envs through the reset body:
// the step path then carries none of the restart's live values -- SGPRs are what the one-wavefront-per-env kernels run out of)
template <int JPL, int TAB, bool WT, bool RESTART = true>
__device__ __forceinline__ StepResult step_compute(Env<JPL> &e, Header &hd, Ctx &c, const Params &p, const int32_t *lds, int a_in) {
    StepResult r;
    r.restart = RESTART && a_in == JSS_ACTION_RESET;                     // reset() this env instead of stepping it
    if (RESTART && r.restart) {
        // the env may have been given another instance since its last reset (table_of_env)
        const int tid = tab_in_lds(TAB) ? 0 : __builtin_amdgcn_readfirstlane(p.d.table_of_env ? p.d.table_of_env[c.b] : c.tid);
        ctx_from_instance(c, p, tid);
        ctx_table<TAB>(c, p, lds);
        hd.episode += 1;
        hd.step = 0;
        reset_env<JPL, WT>(e, c, p);
    }
    r.rn = step_env<JPL, WT>(e, c, p, a_in);
    r.called = a_in != JSS_ACTION_SKIP && !r.restart;
    r.done = !any_legal(e);
    if (r.called) hd.step += 1;
    return r;
}
template <int JPL, bool WT>
__device__ __forceinline__ void step_outputs(const Env<JPL> &e, const Ctx &c, const Params &p, const StepResult &r) {
    const int b = c.b;
    if (c.lane != 0) return;
    if (r.restart) {
        st_out<WT, float>(p.o.reward + b, 0u, 0.f);
        st_out<WT, uint8_t>(p.o.done + b, 0u, (uint8_t)0);
    }
    if (r.called) {                                                      // a skipped env keeps its reward / done / makespan
        st_out<WT, float>(p.o.reward + b, 0u, reward_of(r.rn, c));       // :483-493 (0 for ignored actions)
        st_out<WT, uint8_t>(p.o.done + b, 0u, (uint8_t)(r.done ? 1 : 0));   // :639-653
        if (r.done) st_out<WT, int>(p.o.makespan + b, 0u, e.t);          // last_time_step :650
        if (p.s.counters) add_counters(p.s.counters + (size_t)b * 4, 1, r.done ? 1 : 0, r.done ? e.t : 0, r.rn);
    }
}
template <int JPL, int TAB, bool WT>
__device__ __forceinline__ bool step_call(Env<JPL> &e, Header &hd, Ctx &c, const Params &p, const int32_t *lds, int a_in,
                                          int &rn, bool &called) {
    const StepResult r = step_compute<JPL, TAB, WT>(e, hd, c, p, lds, a_in);
    step_outputs<JPL, WT>(e, c, p, r);
    rn = r.rn;
    called = r.called;
    return r.restart;
}

// ---------------------------------------------------------------------------------------
// one env, one mode: everything behind "the header words are on their way", in two halves -- wave_issue (the state loads of a
// step-type call: nothing is waited for but, on ragged batches, the header word that says how many rows there are) and
// wave_finish (everything else).  A wavefront that serves two envs (wave_block, EPW = 2) issues both envs' loads before it
// finishes the first.
// ---------------------------------------------------------------------------------------
template <int JPL, int TAB>
__device__ __forceinline__ RawEnv<JPL> wave_issue(const Params &p, int b, int lane, const HeaderWords &h, bool ragged) {
    // the addresses depend on nothing but the env index -- unless the batch is ragged (jmin < jmax): then the rows behind
    // J(env), known from the header, are never requested
    return issue_loads<JPL, TAB>(b, lane, p, ragged ? __builtin_amdgcn_readfirstlane(h.J) : p.d.jmax);
}

// "These loads have landed": an empty asm that reads every register of `r`, so that the s_waitcnt for them is placed HERE.
// vmcnt counts loads and stores alike on this architecture and retires in order: a wait for a load that is placed behind
// stores of unknown number (the store loops of the observation) has to wait for every one of them -- vmcnt(0).  A wavefront
// that serves two envs therefore claims its second env's records after the first env's compute and BEFORE the first env's
// stores are issued: nothing but loads is outstanding then, and they have had the whole first step to arrive.
template <int JPL>
__device__ __forceinline__ void loads_landed(const RawEnv<JPL> &r) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("; jss loads_landed tm %0" ::"v"(r.tm));
#pragma unroll
    for (int s = 0; s < JPL; ++s)
        asm volatile("; jss loads_landed %0 %1 %2 %3 %4 %5 %6 %7" ::"v"(r.lo[s].x), "v"(r.lo[s].y), "v"(r.lo[s].z), "v"(r.lo[s].w), "v"(r.hi[s].x), "v"(r.hi[s].y), "v"(r.hi[s].z), "v"(r.hi[s].w));
#else
    (void)r;
#endif
}

// `next`: the records of the env this wavefront serves after this one (wave_block2), claimed before this env's epilogue stores
// (else NULL).  The claim sits at ONE point that every path to the next env runs through -- compute first, under `live`, then
// the claim, then the stores, under `live` again: with an early return in front of it the waits of the two paths merge at the
// join and the second env starts with an s_waitcnt for the first env's stores after all.
// (NEXT = false, one env per wavefront: `live` false returns on the spot -- the form, and the code, these kernels always had)
template <int JPL, int MODE, int TAB, bool NEXT = false>
__device__ __forceinline__ void wave_finish(const Params &p, Ctx &c, const HeaderWords &h, RawEnv<JPL> raw, int a_in,
                                            const int32_t *lds, float *scratch, const RawEnv<JPL> *next = nullptr) {
    const int b = c.b, lane = c.lane;
    Header hd = {0, 0};
    Env<JPL> e = {};
    bool fresh = false;                                                  // the env was (re)initialised by this call
    bool live = true;                                                    // false: never reset -- nothing to step, nothing to store
    if (MODE == kReset) {
        // nothing of the old state is needed but the episode counter
        raw = blank_raw<JPL, TAB>();
        hd.episode = __builtin_amdgcn_readfirstlane(h.episode) + 1;
        hd.step = 0;
        ctx_table<TAB>(c, p, lds);
        reset_env(e, c, p);
        fresh = true;
        if (lane == 0) {
            p.o.reward[b] = 0.f;
            p.o.done[b] = 0;
        }
    } else {
        ctx_from_header(c, h);
        live = c.J != 0;
        if (!NEXT && !live) return;
        if (live) {
            ctx_table<TAB>(c, p, lds);
            hd.episode = __builtin_amdgcn_readfirstlane(h.episode);
            hd.step = __builtin_amdgcn_readfirstlane(h.step);
            if (MODE == kStep || MODE == kAdvance) {                     // no policy keys an RNG with them here: only the header
                hd.episode = in_vgpr(hd.episode);                        // store at the very end reads them -- out of the scalar
                hd.step = in_vgpr(hd.step);                              // register file, which these kernels run out of
            }
            JSS_STAMP(p, b, 1, c.J);
            unpack_env<JPL, TAB>(e, c, raw, __builtin_amdgcn_readfirstlane(h.clock), __builtin_amdgcn_readfirstlane(h.status),
                                 reinterpret_cast<int32_t *>(scratch));
            JSS_STAMP(p, b, 2, e.left[0] + e.idle[0] + e.tm);
        }
    }

    StepResult sr = {0, false, false, false};                            // kStep
    int n_steps = 0, n_done = 0, sum_makespan = 0, sum_rn = 0, last_rn = 0, last_makespan = -1;   // the rollouts
    int a_sched = -1;                                                    // the job the (one) step scheduled: store_env, kStoreCause
    bool restarted = false;                                              // ... and "reset inside the rollout": every record changes
    if (!live) {
        // nothing
    } else if (MODE == kStep) {                                          // (JSS_ACTION_RESET never gets here: jss_kernel)
        sr = step_compute<JPL, TAB, false, false>(e, hd, c, p, lds, a_in);
        a_sched = a_in;
    } else if (MODE == kSteps) {
        // n_iter x jss_step with the actions given up front: the state stays in registers, every step optionally recorded
        for (int it = 0; it < p.n_iter; ++it) {
            const size_t slot = (size_t)it * traj_stride(p) + b;         // [it][b]
            const int a = __builtin_amdgcn_readfirstlane(p.actions[slot]);
            int rn;
            bool called;
            fresh |= step_call<JPL, TAB, false>(e, hd, c, p, lds, a, rn, called);
            if (p.t.real_obs) store_obs(e, c, p.t.real_obs + slot * p.d.jmax * 7, scratch, c.J);
            if (p.t.action_mask) store_mask(e, c, p.t.action_mask + slot * (p.d.jmax + 1), p.d.jmax);
            if (lane == 0) {
                if (p.t.reward) p.t.reward[slot] = called ? reward_of(rn, c) : 0.f;
                if (p.t.done) p.t.done[slot] = any_legal(e) ? 0 : 1;
            }
        }
    } else if (MODE == kAdvance) {
        int hole = 0;
        if (__ballot(e.tm > 0) == 0) e.err |= JSS_ERR_NOPE_IDLE;        // reference: IndexError (:517)
        else hole = advance(e, c);
        if (lane == 0 && p.hole) p.hole[b] = hole;
    } else if (MODE == kPolicy) {
        const int a = select_action<JPL, true>(e, c, p, (uint64_t)(p.d.env_ids ? p.d.env_ids[b] : p.d.env_id_base + b),
                                    (uint32_t)hd.episode, (uint32_t)hd.step);
        if (lane == 0) p.actions_out[b] = a;
        return;
    } else if (MODE == kRollout || MODE == kRollout1 || MODE == kTraj) {
        // n_iter x (policy + step), state stays in registers; kTraj also records every iteration (JssTraj)
        const uint64_t env_id = (uint64_t)(p.d.env_ids ? p.d.env_ids[b] : p.d.env_id_base + b);
        const int n_iter = MODE == kRollout1 ? 1 : p.n_iter;
        for (int it = 0; it < n_iter; ++it) {
            const size_t slot = (size_t)it * traj_stride(p) + b;         // kTraj: [it][b]
            if (MODE == kTraj) {                                         // what the policy sees in this slot
                if (p.t.real_obs) store_obs(e, c, p.t.real_obs + slot * p.d.jmax * 7, scratch, c.J);
                if (p.t.action_mask) store_mask(e, c, p.t.action_mask + slot * (p.d.jmax + 1), p.d.jmax);
            }
            if (!any_legal(e)) {                                         // done (:639-653)
                const bool autoreset = (p.flags & JSS_ROLLOUT_AUTORESET) != 0;
                if (MODE == kTraj && lane == 0) {
                    if (p.t.action) p.t.action[slot] = autoreset ? JSS_ACTION_RESET : JSS_ACTION_SKIP;
                    if (p.t.reward) p.t.reward[slot] = 0.f;
                    if (p.t.done) p.t.done[slot] = autoreset ? 0 : 1;
                }
                if (!autoreset) {
                    if (MODE == kTraj) continue;                         // frozen: every remaining slot says so
                    break;
                }
                reset_env(e, c, p);
                hd.episode += 1;
                hd.step = 0;
                restarted = true;
                continue;
            }
            const int a = select_action(e, c, p, env_id, (uint32_t)hd.episode, (uint32_t)hd.step);
            JSS_STAMP(p, b, 3, a);
            if (MODE == kRollout1) a_sched = a;
            last_rn = step_env(e, c, p, a);
            JSS_STAMP(p, b, 4, last_rn + e.fill[0]);
            hd.step += 1;
            n_steps += 1;
            sum_rn += last_rn;
            const bool done = !any_legal(e);
            if (done) {
                n_done += 1;
                sum_makespan += e.t;
                last_makespan = e.t;
            }
            if (MODE == kTraj && lane == 0) {
                if (p.t.action) p.t.action[slot] = a;
                if (p.t.reward) p.t.reward[slot] = reward_of(last_rn, c);
                if (p.t.done) p.t.done[slot] = done ? 1 : 0;
            }
        }
    }
    if (NEXT) loads_landed(*next);
    if (!live) return;
    // ---- the epilogue: the env's scalar outputs, its state, mask and observation ----
    if (MODE == kStep) {
        step_outputs<JPL, false>(e, c, p, sr);
    } else if (MODE == kRollout || MODE == kRollout1 || MODE == kTraj) {
        if (lane == 0) {
            if (n_steps) p.o.reward[b] = reward_of(last_rn, c);
            p.o.done[b] = any_legal(e) ? 0 : 1;
            if (last_makespan >= 0) p.o.makespan[b] = last_makespan;
            if (p.s.counters) add_counters(p.s.counters + (size_t)b * 4, n_steps, n_done, sum_makespan, sum_rn);
        }
    }
    // (the one-job-per-lane recorder keeps comparing: without `raw` it needs 61 VGPRs instead of 78, runs 8 wavefronts per
    //  SIMD instead of 6 and is 5 % SLOWER on 8 192 envs -- one round of wavefronts in lock step instead of two that overlap;
    //  profiles/r06_misc/looping_stores_ab.txt)
    // (the one-step modes with per-env tables store by cause: config 5 by shape class +3 %, config 4's share +1.6 %, its jss_step
    //  +3 %; on a shared table -- four words per record to compare -- by cause measured 1-2 % slower:
    //  profiles/r06_misc/stores_by_cause_ab.txt)
    constexpr int kDiffStores = (MODE == kRollout || MODE == kSteps || (MODE == kTraj && JPL == 2)) ? kStoreAll
                                : ((MODE == kStep || MODE == kRollout1) && tab_global(TAB)) ? JSS_ONE_STEP_STORES : kStoreCompare;
    store_env<JPL, TAB, kDiffStores>(e, c, p, hd, raw, fresh, restarted || e.t != __builtin_amdgcn_readfirstlane(h.clock), a_sched);
    store_mask(e, c, p.o.action_mask + (size_t)b * (p.d.jmax + 1), p.d.jmax);
    JSS_STAMP(p, b, 5, e.t);
    if (!JSS_ABLATED(p, JSS_ABLATE_OBS))
        store_obs(e, c, p.o.real_obs + (size_t)b * p.d.jmax * 7, scratch, fresh ? imin(p.d.jmax, JPL * kWave) : c.J);   // (a one-job-per-lane
    JSS_STAMP(p, b, 6, e.t);                                     //  body inside wider rows owns the first 64: the fused grid's classes)
}

template <int JPL, int MODE, int TAB>
__device__ __forceinline__ void wave_main(const Params &p, Ctx &c, const HeaderWords &h, bool ragged, int a_in,
                                          const int32_t *lds, float *scratch) {
    RawEnv<JPL> raw;
    if (MODE != kReset) raw = wave_issue<JPL, TAB>(p, c.b, c.lane, h, ragged);   // (a reset reads nothing)
    wave_finish<JPL, MODE, TAB>(p, c, h, raw, a_in, lds, scratch);
}

// ---------------------------------------------------------------------------------------
// the kernel: one mode per instantiation
// ---------------------------------------------------------------------------------------
// Occupancy bound: 8 waves per SIMD keeps 8 192 envs (BASELINE config 4's share of one GPU) in ONE round of resident
// waves; the price is an SGPR budget of 80, which the step/rollout modes overrun by values that live in spare
// VGPR lanes (v_writelane / v_readlane, no scratch).  JSS_WAVE_MIN_BLOCKS = 7 lifts the budget to 102 (A/B builds).
// Two jobs per lane (J > 64) runs at 7 (5 for the multi-iteration rollout): more would spill VGPRs to scratch.
// A batch padded to more than 64 jobs is compiled with two jobs per lane, but every wave whose own env has
// J <= 64 (70 of 80 Taillard instances in the mixed ta01-ta80 batch) runs the one-job-per-lane body.
#ifndef JSS_WAVE_MIN_BLOCKS
#define JSS_WAVE_MIN_BLOCKS 8
#endif
#ifndef JSS_WAVE2_MIN_BLOCKS
#define JSS_WAVE2_MIN_BLOCKS 7
#endif
#ifndef JSS_TRAJ1_MIN_BLOCKS
#define JSS_TRAJ1_MIN_BLOCKS 6
#endif
#ifndef JSS_TRAJ2_MIN_BLOCKS
#define JSS_TRAJ2_MIN_BLOCKS 5
#endif
// waves per SIMD each instantiation is compiled for: the largest occupancy it reaches without scratch memory (round 6: NO
// instantiation uses scratch -- tests/test_abi_and_host.py reads the code object's notes; the kernels that loop over steps
// got there by storing the state without comparing it with what was loaded, store_env<.., DIFF = false>, the two-jobs-per-lane
// recorder in addition by one occupancy step, 6 -> 5, the two-jobs-per-lane one-step rollout on 24-byte records 7 -> 6)
constexpr int wave_min_blocks(int jpl, int mode, int tab) {
    return mode == kTraj ? (jpl == 2 ? JSS_TRAJ2_MIN_BLOCKS : JSS_TRAJ1_MIN_BLOCKS)
         : mode == kRollout ? (jpl == 2 ? 5 : 7)
         : mode == kStep ? (jpl == 2 ? 5 : 8)
         : mode == kSteps ? (jpl == 2 ? 4 : 6)
         : mode == kRollout1 ? (jpl == 2 ? (tab_medium(tab) ? 6 : JSS_WAVE2_MIN_BLOCKS) : JSS_WAVE_MIN_BLOCKS)
         : (jpl == 2 ? 7 : 8);
}
// One workgroup's share of a launch (`block`: see packed_block).  NARROW: a two-jobs-per-lane instantiation also holds the
// one-job-per-lane body for the envs of a ragged batch that fit it (below); the fused multi-set grid, whose two-jobs-per-lane
// class holds nothing but J > 64 instances, leaves it out.
template <int JPL, int MODE, int TAB, bool NARROW = true>
__device__ __forceinline__ void wave_block(const Params &p, int block, int32_t *lds) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // obs image of this wave, 16-byte aligned (table_lds_ints and obs_wave_floats are multiples of 4)
    float *scratch = reinterpret_cast<float *>(lds + p.table_lds_ints) + wave * p.obs_wave_floats;

    const int b_raw = block * kWavesPerBlock + wave;                      // one env per wave
    const bool alive = b_raw < p.d.batch;
    const int b = alive ? b_raw : p.d.batch - 1;
    Ctx c;
    c.b = b;
    c.lane = lane;
    JSS_STAMP(p, b, 0, lane);
    const HeaderWords h = load_header(p, b);
    int a_in = JSS_ACTION_SKIP;
    if (MODE == kStep) {
        a_in = __builtin_amdgcn_readfirstlane(p.actions[b]);
        // jss_step_autoreset: an env that reported done on the previous call is reset instead of stepped
        if ((p.flags & JSS_ROLLOUT_AUTORESET) && __builtin_amdgcn_readfirstlane((int)p.o.done[b]) != 0) a_in = JSS_ACTION_RESET;
    }
    bool selected = true;
    if ((MODE == kReset || MODE == kAdvance) && p.which) selected = __builtin_amdgcn_readfirstlane((int)p.which[b]) != 0;
    if (tab_in_lds(TAB)) {                                                // one instance for the whole batch: its op table -> LDS
        stage_shared_table(lds, p.d.ops, p.d.jmax * p.d.mmax, (int)threadIdx.x);   // one instance: jmax rows are its J rows
        __syncthreads();
    }
    if (!alive || !selected) return;
    const bool ragged = p.d.jmin > 0 && p.d.jmin < p.d.jmax;
    if (MODE == kReset) {
        const int tid = tab_in_lds(TAB) ? 0 : __builtin_amdgcn_readfirstlane(p.d.table_of_env ? p.d.table_of_env[b] : b);
        ctx_from_instance(c, p, tid);
        wave_main<JPL, MODE, TAB>(p, c, h, ragged, a_in, lds, scratch);   // full width: a reset writes every row of the padded block
    } else if (MODE == kStep && a_in == JSS_ACTION_RESET) {
        // jss_step's "reset this env instead of stepping it" IS a reset: nothing of the old state is needed but the episode
        // counter, and the env may have been handed another (wider) instance since (table_of_env) -- the reset body, full width
        if (__builtin_amdgcn_readfirstlane(h.J) == 0) return;            // never reset: left alone, like by every step-type call
        const int tid = tab_in_lds(TAB) ? 0 : __builtin_amdgcn_readfirstlane(p.d.table_of_env ? p.d.table_of_env[b] : h.tid);
        ctx_from_instance(c, p, tid);
        wave_main<JPL, kReset, TAB>(p, c, h, ragged, a_in, lds, scratch);
    } else {
        // (J == 64 stays on the full-width body: the NOPE flag of its mask row lives at index 64, slot 1's first lane)
        if (NARROW && JPL == 2 && ragged && MODE != kSteps && __builtin_amdgcn_readfirstlane(h.J) < kWave)
            wave_main<1, MODE, TAB>(p, c, h, ragged, a_in, lds, scratch);
        else
            wave_main<JPL, MODE, TAB>(p, c, h, ragged, a_in, lds, scratch);
    }
}

template <int JPL, int MODE, int TAB>
__global__ __launch_bounds__(kBlock, wave_min_blocks(JPL, MODE, TAB)) void jss_kernel(Params p_arg) {
    HIP_DYNAMIC_SHARED(int32_t, lds)
    // by value where a launch is one step: measured faster than in place although it spills SGPRs; in place for the launches
    // that loop over steps with the state in registers (jss_common.hpp: trajectory mode +6.5 % / +10 % on config 4's share /
    // config 5, the 64-iteration rollout +6 % / +3 %, jss_steps +2.5 %: there the up-front loads would stay live for the whole loop)
    JSS_PARAMS_OF(p, p_arg, MODE == kTraj || MODE == kRollout || MODE == kSteps);
    wave_block<JPL, MODE, TAB>(p, (int)blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------
// Two envs per wavefront, one after the other (round 6).  A launch of the one-env-per-wavefront kernel over one round of
// resident waves is load burst -> compute -> store burst, strictly one after the other, because every wave is in the same
// phase at the same time (profiles/r05_misc/wave_timeline.txt: 4.4 k of a wave's 15.7 k cycles go by before its state has
// arrived, 8 192 waves asking for 15 MB at once).  Here a wavefront owns envs 2w and 2w + 1: it asks for BOTH envs' headers and
// state up front, steps the first while the second's state is on its way, claims the second's records before it issues the
// first env's stores (loads_landed), and the first env's stores drain under the second env's compute.  Same instructions per
// env, 9 more live VGPRs (the second env's raw records) and its 14 header words.  What it buys, measured
// (profiles/r06_misc/two_per_wave_ab.txt, wave_timeline_two_per_wave.txt): NOT the one-round launch it was built for -- at
// 8 192 envs half as many wavefronts (4 per SIMD) are latency-bound, a wavefront's two steps take 26.7 k cycles against 16.3 k
// for one, the launch is 15-18 % slower -- but launches of several rounds (65 536 envs in sub-batches: +6-7 %), where the
// second env's state arrives under the first env's step instead of at the head of a new wavefront's life.  The library uses it
// from JSS_TWO_PER_WAVE_MIN_BATCH envs per launch on (jss_kernels.hip).  One job per lane, the one-step modes (kRollout1, kStep).
// ---------------------------------------------------------------------------------------
template <int MODE, int TAB>
__device__ __forceinline__ void wave_one_of_two(const Params &p, int b, int lane, const HeaderWords &h, const RawEnv<1> &raw, int a_in,
                                                const int32_t *lds, float *scratch, const RawEnv<1> *next) {
    Ctx c;
    c.b = b;
    c.lane = lane;
    if (MODE == kStep && a_in == JSS_ACTION_RESET) {                     // (see wave_block: the reset body, nothing of `raw` is used)
        if (next) loads_landed(*next);                                   // (the reset body is all stores)
        if (__builtin_amdgcn_readfirstlane(h.J) == 0) return;
        const int tid = tab_in_lds(TAB) ? 0 : __builtin_amdgcn_readfirstlane(p.d.table_of_env ? p.d.table_of_env[b] : h.tid);
        ctx_from_instance(c, p, tid);
        wave_finish<1, kReset, TAB>(p, c, h, raw, a_in, lds, scratch);
    } else {
        if (next) wave_finish<1, MODE, TAB, true>(p, c, h, raw, a_in, lds, scratch, next);
        else wave_finish<1, MODE, TAB>(p, c, h, raw, a_in, lds, scratch);
    }
}

template <int MODE, int TAB>
__device__ __forceinline__ void wave_block2(const Params &p, int block, int32_t *lds) {
    static_assert(MODE == kRollout1 || MODE == kStep, "two envs per wavefront: the one-step modes");
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *scratch = reinterpret_cast<float *>(lds + p.table_lds_ints) + wave * p.obs_wave_floats;
    const int first = (block * kWavesPerBlock + wave) * 2;               // envs `first` and `first + 1`
    const bool alive0 = first < p.d.batch, alive1 = first + 1 < p.d.batch;
    const int b0 = alive0 ? first : p.d.batch - 1, b1 = alive1 ? first + 1 : p.d.batch - 1;   // (clamped: the loads stay in bounds)
    JSS_STAMP(p, b0, 0, lane);
    if (alive1) JSS_STAMP(p, b1, 0, lane);
    const HeaderWords h0 = load_header(p, b0), h1 = load_header(p, b1);
    int a0 = JSS_ACTION_SKIP, a1 = JSS_ACTION_SKIP;
    if (MODE == kStep) {
        a0 = __builtin_amdgcn_readfirstlane(p.actions[b0]);
        a1 = __builtin_amdgcn_readfirstlane(p.actions[b1]);
        if (p.flags & JSS_ROLLOUT_AUTORESET) {                           // jss_step_autoreset (wave_block)
            if (__builtin_amdgcn_readfirstlane((int)p.o.done[b0]) != 0) a0 = JSS_ACTION_RESET;
            if (__builtin_amdgcn_readfirstlane((int)p.o.done[b1]) != 0) a1 = JSS_ACTION_RESET;
        }
    }
    if (tab_in_lds(TAB)) {
        stage_shared_table(lds, p.d.ops, p.d.jmax * p.d.mmax, (int)threadIdx.x);
        __syncthreads();
    }
    if (!alive0) return;
    const bool ragged = p.d.jmin > 0 && p.d.jmin < p.d.jmax;
    // both envs' state loads, unconditionally and in straight-line code (an env that is reset instead of stepped reads rows
    // nobody looks at: one step in a few hundred) -- the wait in front of the first env's unpack then leaves exactly the
    // second env's loads outstanding
    const RawEnv<1> raw0 = wave_issue<1, TAB>(p, b0, lane, h0, ragged);
    const RawEnv<1> raw1 = wave_issue<1, TAB>(p, b1, lane, h1, ragged);
    wave_one_of_two<MODE, TAB>(p, b0, lane, h0, raw0, a0, lds, scratch, &raw1);
    if (alive1) wave_one_of_two<MODE, TAB>(p, b1, lane, h1, raw1, a1, lds, scratch, nullptr);
}

template <int MODE, int TAB>
__global__ __launch_bounds__(kBlock, JSS_WAVE_MIN_BLOCKS) void jss_kernel_two(Params p) {
    HIP_DYNAMIC_SHARED(int32_t, lds)
    wave_block2<MODE, TAB>(p, (int)blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------
// The resident step-session kernel (include/jss_hip.h, jss_session_*), one wavefront per env.  Same protocol as
// jss_packed_session_kernel (jss_packed_env.hpp): a wavefront owns `slots` envs (env = wave index + slot * waves in
// the grid); with one env the state lives in registers from open to close, with several each env is parked in LDS
// between its visits as the records it would be stored as (pack_env / unpack_env) plus its header words.
// ---------------------------------------------------------------------------------------
// LDS footprint of one parked env, in int4: the job records per lane, the machine clocks (one int per lane), the header
template <int JPL, int TAB>
constexpr int wave_park_rows() { return tab_compact(TAB) ? JPL : 2 * JPL; }
template <int JPL, int TAB>
constexpr int wave_park_int4() { return wave_park_rows<JPL, TAB>() * kWave + kWave / 4 + 1; }

template <int JPL, int TAB>
__device__ __forceinline__ void park_env(int4 *park, int slot, int lane, const RawEnv<JPL> &r, int clock, int episode, int step, int status) {
    int4 *q = park + (size_t)slot * wave_park_int4<JPL, TAB>();
    int k = 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        q[(k++) * kWave + lane] = r.lo[s];
        if (!tab_compact(TAB)) q[(k++) * kWave + lane] = r.hi[s];
    }
    int4 *t = q + wave_park_rows<JPL, TAB>() * kWave;
    reinterpret_cast<int *>(t)[lane] = r.tm;
    if (lane == 0) t[kWave / 4] = make_int4(clock, episode, step, status);
}
template <int JPL, int TAB>
__device__ __forceinline__ RawEnv<JPL> unpark_env(const int4 *park, int slot, int lane, int &clock, int &episode, int &step, int &status) {
    const int4 *q = park + (size_t)slot * wave_park_int4<JPL, TAB>();
    RawEnv<JPL> r;
    int k = 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        r.lo[s] = q[(k++) * kWave + lane];
        r.hi[s] = make_int4(0, 0, 0, 0);
        if (!tab_compact(TAB)) r.hi[s] = q[(k++) * kWave + lane];
    }
    const int4 *t = q + wave_park_rows<JPL, TAB>() * kWave;
    r.tm = reinterpret_cast<const int *>(t)[lane];
    const int4 w = t[kWave / 4];
    clock = __builtin_amdgcn_readfirstlane(w.x);
    episode = __builtin_amdgcn_readfirstlane(w.y);
    step = __builtin_amdgcn_readfirstlane(w.z);
    status = __builtin_amdgcn_readfirstlane(w.w);
    return r;
}

template <int JPL, int TAB>
__global__ __launch_bounds__(kBlock, JPL == 1 ? 6 : 4) void jss_session_kernel(Params p_arg) {
    HIP_DYNAMIC_SHARED(int32_t, lds)
    JSS_PARAMS_OF(p, p_arg, false);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *scratch = reinterpret_cast<float *>(lds + p.table_lds_ints) + wave * p.obs_wave_floats;
    const int slots = p.slots;
    int4 *park = reinterpret_cast<int4 *>(lds + p.park_off_ints) + (size_t)wave * slots * wave_park_int4<JPL, TAB>();
    const int n_waves = (int)gridDim.x * kWavesPerBlock;
    const int gw = (int)blockIdx.x * kWavesPerBlock + wave;
    if (tab_in_lds(TAB)) {
        stage_shared_table(lds, p.d.ops, p.d.jmax * p.d.mmax, (int)threadIdx.x);
        __syncthreads();
    }
    if (gw >= p.d.batch) return;
    if (gw == 0 && lane == 0) wt_store(p.status + 3, slots);
    const bool ragged = p.d.jmin > 0 && p.d.jmin < p.d.jmax;
    int my_slots = 0;
    for (int slot = 0; slot < slots; ++slot) my_slots += gw + slot * n_waves < p.d.batch ? 1 : 0;

    Ctx c;
    c.lane = lane;
    Env<JPL> e;
    Header hd;
    hd.episode = hd.step = 0;
    // the env's constants record is read-only while the session is open: it is re-read (scalar loads, cached) at every
    // visit of a parked env; the header words travel with the parked records
    auto enter = [&](int slot) {
        c.b = gw + slot * n_waves;
        const HeaderWords h = load_header(p, c.b);
        ctx_from_header(c, h);
        ctx_table<TAB>(c, p, lds);
        return h;
    };
    // ---- the state of every env of this wavefront: loaded once ----
    for (int slot = 0; slot < my_slots; ++slot) {
        const HeaderWords h = enter(slot);
        const RawEnv<JPL> raw = issue_loads<JPL, TAB>(c.b, lane, p, ragged ? c.J : p.d.jmax);
        const int clock = __builtin_amdgcn_readfirstlane(h.clock), status = __builtin_amdgcn_readfirstlane(h.status);
        hd.episode = __builtin_amdgcn_readfirstlane(h.episode);
        hd.step = __builtin_amdgcn_readfirstlane(h.step);
        if (slots > 1) park_env<JPL, TAB>(park, slot, lane, raw, clock, hd.episode, hd.step, status);
        else unpack_env<JPL, TAB>(e, c, raw, clock, status, reinterpret_cast<int32_t *>(scratch));
    }
    if (slots > 1) wave_lds_sync();

    // ---- step after step ----
    const size_t B = (size_t)p.d.batch;
    auto granule = [&](int step, int slot) -> const unsigned long long * {
        return p.mail + (size_t)(step % p.depth) * B + (gw + slot * n_waves);
    };
    int step = 0, pending = 0;
    bool closing = false, timed_out = false;
    unsigned long long x = fresh_load(granule(0, 0));
    while (!closing) {
        for (int slot = 0; slot < my_slots; ++slot) {
            const int nslot = slot + 1 < my_slots ? slot + 1 : 0;
            const int nstep = step + (nslot == 0 ? 1 : 0);
            const unsigned long long xn = fresh_load(granule(nstep, nslot));   // the next (step, env) pair's granule
            const unsigned want = (unsigned)step + 1u;
            long long t0 = 0;
            unsigned spins = 0;
            while ((unsigned)(__builtin_amdgcn_readfirstlane((int)(x >> 32))) != want) {
                if (pending) {
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
            const int a = __builtin_amdgcn_readfirstlane((int)(unsigned)x);
            if (timed_out || a == JSS_ACTION_CLOSE) {
                closing = true;
                break;
            }
            if (slots > 1) {
                enter(slot);
                int clock, status;
                const RawEnv<JPL> raw = unpark_env<JPL, TAB>(park, slot, lane, clock, hd.episode, hd.step, status);
                unpack_env<JPL, TAB>(e, c, raw, clock, status, reinterpret_cast<int32_t *>(scratch));
            }
            if (c.J != 0) {                           // (J == 0: the env was never reset -- the session leaves it alone)
                const StepResult res = step_compute<JPL, TAB, true>(e, hd, c, p, lds, a);
                const bool fresh = res.restart;
                if (pending) {                        // the previous step's stores have had this step's compute to drain: the
                    wt_drain();                       // progress word goes out BEFORE any output store of this step is issued
                    if (lane == 0) wt_store(p.progress + gw, pending);
                    pending = 0;
                }
                step_outputs<JPL, true>(e, c, p, res);
                store_mask<JPL, true>(e, c, p.o.action_mask + (size_t)c.b * (p.d.jmax + 1), p.d.jmax);
                store_obs<JPL, true>(e, c, p.o.real_obs + (size_t)c.b * p.d.jmax * 7, scratch, fresh ? p.d.jmax : c.J);
            }
            if (slots > 1) {
                park_env<JPL, TAB>(park, slot, lane, pack_env<JPL, TAB>(e, c), e.t, hd.episode, hd.step,
                                   (e.err & 0xFF) | (e.noop ? JSS_STATUS_NOOP : 0));
                wave_lds_sync();
            }
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
    // ---- the state goes back to memory (every row: nothing was kept to compare with) ----
    for (int slot = 0; slot < my_slots; ++slot) {
        if (slots > 1) {
            enter(slot);
            int clock, status;
            const RawEnv<JPL> raw = unpark_env<JPL, TAB>(park, slot, lane, clock, hd.episode, hd.step, status);
            unpack_env<JPL, TAB>(e, c, raw, clock, status, reinterpret_cast<int32_t *>(scratch));
        }
        if (c.J != 0) store_env<JPL, TAB>(e, c, p, hd, blank_raw<JPL, TAB>(), true);
    }
    if (lane == 0) {
        if (timed_out) atomicAdd(p.status + 0, 1);
        atomicAdd(p.status + 2, 1);
    }
}

}  // namespace jss
