#pragma once
// Wave-per-env kernel: one 64-lane wavefront simulates one env (J <= 128, M <= 64).
#include "jss_common.hpp"

namespace jss {

// Per-env constants, all wave-uniform.
struct Ctx {
    int b;
    int J, M;
    int max_time_op, max_time_jobs, sum_op;
    const int32_t *ops;  // LDS, row stride `stride`
    int stride;
    int lane;
};

template <int JPL>
struct Env {
    int t;                                                               // current_time_step
    int todo[JPL], cur[JPL], left[JPL], perf[JPL], idle[JPL], idle_last[JPL], f4[JPL];
    uint64_t legal[JPL], blocked[JPL];                                   // job sets, wave-uniform
    uint64_t valid[JPL];                                                 // lanes holding a real job
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

// ---------------------------------------------------------------------------------------
// reset(): jss_env.py:145-181
// ---------------------------------------------------------------------------------------
template <int JPL>
__device__ __forceinline__ void reset_env(Env<JPL> &e, const Ctx &c, const Params &p) {
    e.t = 0;                                                             // :154
    e.tm = 0;                                                            // :164
    e.noop = 0;                                                          // :161
    e.err = 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        int j = s * kWave + c.lane;
        bool v = j < c.J;
        e.valid[s] = __ballot(v);
        e.todo[s] = 0;                                                   // :166
        e.cur[s] = v ? c.ops[j * c.stride] : -1;                         // :174-176 needed machine = op 0
        e.left[s] = e.perf[s] = e.idle[s] = e.idle_last[s] = 0;          // :165-170
        e.f4[s] = 0;                                                     // :180 state zeros
        e.legal[s] = e.valid[s];                                         // :160
        e.blocked[s] = 0;                                                // :171-172
    }
    // solution = -1 (:163); coalesced rows of the padded [jmax][mmax] block
    int32_t *sol = p.s.solution + (size_t)c.b * p.d.jmax * p.d.mmax;
    int n = c.J * p.d.mmax;
    for (int i = c.lane; i < n; i += kWave) sol[i] = -1;
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
        const bool v = (e.valid[s] >> c.lane) & 1;
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
        int ncur = e.cur[s];
        if (fin[s]) ncur = (e.todo[s] < c.M) ? c.ops[j * c.stride + e.todo[s]] : -1;  // :562-566 / :581
        e.cur[s] = ncur;
        // feature 4 numerator: max(0, tm_old[need] - d) (:569-578) == tm_new[need]
        const int tm_need = __shfl(e.tm, (ncur >> 16) & 63);
        if (fin[s]) e.f4[s] = ncur >= 0 ? tm_need : JSS_F4_ONE;          // :586 "1.0" when the job is complete
        // re-legalisation :616-634: need[j] on a free machine, not legal, not blocked.
        // (a job that just completed has cur = -1 and is never legal, :589-591)
        const bool v = (e.valid[s] >> c.lane) & 1;
        const bool can = v && ncur >= 0 && ((free_m >> ((ncur >> 16) & 63)) & 1);
        e.legal[s] |= __ballot(can) & ~e.blocked[s];
    }
    return hole;
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
        const int j = s * kWave + c.lane;
        const bool lg = (e.legal[s] >> c.lane) & 1;
        nf[s] = false;
        if (lg && e.todo[s] < c.M - 1) {                                 // :219-239 non-final, next machine idle
            const int next_m = c.ops[j * c.stride + e.todo[s] + 1] >> 16;  // :227
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
template <int JPL>
__device__ __forceinline__ void check_no_op(Env<JPL> &e, const Ctx &c) {
    e.noop = 0;                                                          // :278
    const int nl = nb_legal(e);
    if (nl > 4 || nl == 0) return;                                       // :287 (nl == 0: no legal machine, U can never match)
    const uint64_t busy = __ballot(e.tm > 0);
    if (busy == 0) return;                                               // :285 len(next_time_step) > 0
    // PASS 1 (:305-321): sequential in ascending job index over the <= 4 legal jobs; max_horizon
    // sees the running prefix minimum of max_horizon_machine, so the order matters.
    int mm0 = -1, mm1 = -1, mm2 = -1;        // the <= 3 legal machines ...
    int mv0 = 0, mv1 = 0, mv2 = 0;           // ... and their max_horizon_machine
    int n_ml = 0;                            // nb_machine_legal
    int cf[4];                               // packed current op of the i-th legal job (ascending job index)
    {
        uint64_t b0 = e.legal[0];
        uint64_t b1 = JPL > 1 ? e.legal[JPL - 1] : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cf[i] = -1;
            if (i < nl) {
                if (b0) {
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
            if (m != mm0 && m != mm1 && m != mm2) {
                if (n_ml == 0) mm0 = m; else if (n_ml == 1) mm1 = m; else if (n_ml == 2) mm2 = m;
                ++n_ml;
            }
        }
    }
    if (n_ml > 3) return;                                                // :286
    const int nxt = e.t + wave_min(e.tm > 0 ? e.tm : kBig);              // :293 next_time_step[0]
    int mh = e.t;                                                        // :296
    mv0 = mv1 = mv2 = e.t + c.max_time_op;                               // :300-302
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nl) {
            const int m = cf[i] >> 16;
            const int end = e.t + (cf[i] & kDurMask);                    // :310
            if (end < nxt) return;                                       // :314-315
            int h;
            if (m == mm0) { mv0 = imin(mv0, end); h = mv0; }             // :318
            else if (m == mm1) { mv1 = imin(mv1, end); h = mv1; }
            else { mv2 = imin(mv2, end); h = mv2; }
            mh = imax(mh, h);                                            // :321
        }
    }
    // PASS 2 (:324-401): every illegal job walks its future ops; order-free, so one lane per job.
    int u = 0;  // bit i: legal machine i "would be better used by waiting"
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        const bool v = (e.valid[s] >> c.lane) & 1;
        const bool lg = (e.legal[s] >> c.lane) & 1;
        const bool bl = (e.blocked[s] >> c.lane) & 1;
        const bool caseA = v && !lg && e.left[s] > 0 && e.todo[s] + 1 < c.M;      // :327-330
        const bool caseB = v && !lg && !caseA && !bl && e.todo[s] < c.M;          // :366-369
        const int tm_need = __shfl(e.tm, (e.cur[s] >> 16) & 63);                   // :376
        int k = caseA ? e.todo[s] + 1 : e.todo[s];                                // :332 / :370
        int tn = caseA ? e.t + e.left[s] : e.t + tm_need;                         // :334-337 / :374-377
        if (caseA || caseB) {
            while (k < c.M - 1 && mh > tn) {                                       // :340-342 / :380-382
                const int op = c.ops[j * c.stride + k];
                const int m = op >> 16;
                if (m == mm0 && mv0 > tn) u |= 1;                                  // :346-351 (mm* are the legal machines)
                if (m == mm1 && mv1 > tn) u |= 2;
                if (m == mm2 && mv2 > tn) u |= 4;
                tn += op & kDurMask;                                               // :362
                ++k;
            }
        }
    }
    const int covered = (__ballot(u & 1) != 0) + (__ballot(u & 2) != 0) + (__ballot(u & 4) != 0);
    e.noop = (covered == n_ml) ? 1 : 0;                                  // :357-359 / :395-397
}

// ---------------------------------------------------------------------------------------
// step(): jss_env.py:403-481.  `a` is wave-uniform.  Returns the reward numerator
// (reward * max_time_op, an exact integer: scheduled duration minus idle-machine time).
// ---------------------------------------------------------------------------------------
template <int JPL>
__device__ __forceinline__ int step_env(Env<JPL> &e, const Ctx &c, const Params &p, int a) {
    if (a == JSS_ACTION_SKIP) return 0;
    if (a < 0 || a > c.J) {
        e.err |= JSS_ERR_BAD_ACTION;
        return 0;
    }
    int rn = 0;
    if (a == c.J) {                                                      // :419 NOPE
#pragma unroll
        for (int s = 0; s < JPL; ++s) {                                  // :422-428
            e.blocked[s] |= e.legal[s];
            e.legal[s] = 0;
        }
        for (;;) {                                                       // :429-430
            if (__ballot(e.tm > 0) == 0) {  // reference: IndexError (pop from empty list, :517)
                e.err |= JSS_ERR_NOPE_IDLE;
                break;
            }
            rn -= advance(e, c);
            if (any_legal(e)) break;
        }
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
        if (c.lane == 0) p.s.solution[((size_t)c.b * p.d.jmax + a) * p.d.mmax + k] = e.t;  // :454
#pragma unroll
        for (int s = 0; s < JPL; ++s) {
            const bool v = (e.valid[s] >> c.lane) & 1;
            const uint64_t same = __ballot(v && e.cur[s] >= 0 && (e.cur[s] >> 16) == m);
            e.legal[s] &= ~same;                                         // :455-463
            e.blocked[s] &= ~same;                                       // :464-467
        }
        while (!any_legal(e) && __ballot(e.tm > 0) != 0 && !(p.ablate & JSS_ABLATE_ADVANCE)) rn -= advance(e, c);  // :469-470
    }
    if (!(p.ablate & JSS_ABLATE_PRIORITIZE)) prioritize(e, c);           // :432 / :471
    if (!(p.ablate & JSS_ABLATE_CHECK_NO_OP)) check_no_op(e, c);         // :433 / :472
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

template <int JPL>
__device__ __forceinline__ int select_action(const Env<JPL> &e, const Ctx &c, int kind, uint64_t seed, uint32_t explore_q16,
                                             uint64_t env_id, uint32_t episode, uint32_t step) {
    const int nl = nb_legal(e);
    const int n = nl + (e.noop ? 1 : 0);
    if (n == 0) return -1;
    if (kind == JSS_POLICY_RANDOM) {
        // README.md:58-60: uniform over the set bits of the mask, NOPE included
        const uint32_t r = rng_u32(seed, env_id, episode, step);
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
    if (kind == JSS_POLICY_CR) {                                         // dispatching.py:365-408
        CrKey best;
        best.num = 0x3fffffff;
        best.den = 1;
        best.idx = kCrNone;
#pragma unroll
        for (int s = 0; s < JPL; ++s) {
            const int j = s * kWave + c.lane;
            const bool lg = (e.legal[s] >> c.lane) & 1;
            int total = 0, remaining = 0;
            if (lg)
                for (int k = 0; k < c.M; ++k) {
                    const int d = c.ops[j * c.stride + k] & kDurMask;
                    total += d;
                    if (k >= e.todo[s]) remaining += d;
                }
            CrKey key;
            key.num = lg ? 3 * total - 2 * e.t : 0x3fffffff;
            key.den = lg ? remaining : 1;
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
            else {                                                           // MWR / LWR :187-189 / :230-232
                v = 0;
                if (lg)
                    for (int k = e.todo[s]; k < c.M; ++k) v += c.ops[j * c.stride + k] & kDurMask;
            }
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
    if (e.noop && explore_q16 != 0) {                                    // dispatching.py:113: 10 % NOPE when NOPE is legal
        const uint32_t r = rng_u32(seed ^ kExploreSeedXor, env_id, episode, step);
        if ((r >> 16) < explore_q16) a = c.J;
    }
    return a;
}

// ---------------------------------------------------------------------------------------
// HBM <-> registers.  One 32-byte record per job (two dwordx4 per lane), one int4 header per env.
// ---------------------------------------------------------------------------------------
struct Header {
    int clock, episode, step, status;
};

template <int JPL>
struct RawEnv {  // loads issued before the op table is staged; unpacked after the barrier
    int4 h;
    int4 lo[JPL], hi[JPL];
    int tm;
};

template <int JPL>
__device__ __forceinline__ RawEnv<JPL> issue_loads(int b, int lane, const Params &p) {
    RawEnv<JPL> r;
    const int jm = p.d.jmax;
    r.h = reinterpret_cast<const int4 *>(p.s.env)[b];
    const int4 *js = reinterpret_cast<const int4 *>(p.s.job) + (size_t)b * jm * 2;
    r.tm = p.s.machine[(size_t)b * p.d.mmax + (lane < p.d.mmax ? lane : 0)];
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + lane;
        const int jc = j < jm ? j : 0;
        r.lo[s] = js[jc * 2];
        r.hi[s] = js[jc * 2 + 1];
    }
    return r;
}

template <int JPL>
__device__ __forceinline__ Header unpack_env(Env<JPL> &e, const Ctx &c, const RawEnv<JPL> &r) {
    Header hd;
    hd.clock = __builtin_amdgcn_readfirstlane(r.h.x);
    hd.episode = __builtin_amdgcn_readfirstlane(r.h.y);
    hd.step = __builtin_amdgcn_readfirstlane(r.h.z);
    hd.status = __builtin_amdgcn_readfirstlane(r.h.w);
    e.t = hd.clock;
    e.err = hd.status & 0xFF;
    e.noop = (hd.status & JSS_STATUS_NOOP) ? 1 : 0;
    e.tm = c.lane < c.M ? r.tm : 0;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        const bool v = j < c.J;
        const int4 lo = r.lo[s], hi = r.hi[s];
        e.valid[s] = __ballot(v);
        e.todo[s] = v ? lo.x : 0;
        e.cur[s] = v ? lo.y : -1;
        e.left[s] = v ? lo.z : 0;
        e.perf[s] = v ? lo.w : 0;
        e.idle[s] = v ? hi.x : 0;
        e.idle_last[s] = v ? hi.y : 0;
        e.f4[s] = v ? hi.z : 0;
        e.legal[s] = __ballot(v && (hi.w & JSS_FLAG_LEGAL));
        e.blocked[s] = __ballot(v && (hi.w & JSS_FLAG_BLOCKED));
    }
    return hd;
}

template <int JPL>
__device__ __forceinline__ void store_env(const Env<JPL> &e, const Ctx &c, const Params &p, const Header &hd) {
    const int jm = p.d.jmax;
    int4 *js = reinterpret_cast<int4 *>(p.s.job) + (size_t)c.b * jm * 2;
    uint8_t *mk = p.o.action_mask + (size_t)c.b * (jm + 1);
    if (c.lane == 0) {
        reinterpret_cast<int4 *>(p.s.env)[c.b] =
            make_int4(e.t, hd.episode, hd.step, (e.err & 0xFF) | (e.noop ? JSS_STATUS_NOOP : 0));
        mk[c.J] = (uint8_t)e.noop;
    }
    if (c.lane < c.M) p.s.machine[(size_t)c.b * p.d.mmax + c.lane] = e.tm;
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        if (j < c.J) {
            const int lg = (int)((e.legal[s] >> c.lane) & 1), bl = (int)((e.blocked[s] >> c.lane) & 1);
            js[j * 2] = make_int4(e.todo[s], e.cur[s], e.left[s], e.perf[s]);
            js[j * 2 + 1] = make_int4(e.idle[s], e.idle_last[s], e.f4[s], lg | (bl << 1));
            mk[j] = (uint8_t)lg;
        }
    }
}

// The (J,7) observation of jss_env.py:102-111, float32.  Every column is a function of the
// integer state (column 4 of its own stored numerator: the reference writes it only when an
// op finishes).  Transposed through LDS so the HBM write is jmax*7 contiguous floats.
template <int JPL>
__device__ __forceinline__ void store_obs(const Env<JPL> &e, const Ctx &c, const Params &p, float *scratch) {
    const float f_op = (float)c.max_time_op, f_jobs = (float)c.max_time_jobs, f_sum = (float)c.sum_op, f_m = (float)c.M;
    const float r_op = refined_rcp(f_op), r_jobs = refined_rcp(f_jobs), r_sum = refined_rcp(f_sum), r_m = refined_rcp(f_m);
#pragma unroll
    for (int s = 0; s < JPL; ++s) {
        const int j = s * kWave + c.lane;
        if (j < p.d.jmax) {
            const bool v = j < c.J;  // padding rows are written as zeros
            float *row = scratch + j * 7;
            row[0] = v ? (float)((e.legal[s] >> c.lane) & 1) : 0.f;                       // :130
            row[1] = div_by((float)e.left[s], f_op, r_op);                               // :448, :539
            row[2] = div_by((float)e.todo[s], f_m, r_m);                                 // :559
            row[3] = div_by((float)e.perf[s], f_jobs, r_jobs);                           // :545
            row[4] = e.f4[s] == JSS_F4_ONE ? 1.0f : div_by((float)e.f4[s], f_op, r_op);  // :569-586
            row[5] = div_by((float)e.idle_last[s], f_sum, r_sum);                        // :555, :600
            row[6] = div_by((float)e.idle[s], f_sum, r_sum);                             // :553, :601
        }
    }
    wave_lds_sync();
    float *dst = p.o.real_obs + (size_t)c.b * p.d.jmax * 7;
    const int n = p.d.jmax * 7;
    for (int i = c.lane; i < n; i += kWave) dst[i] = scratch[i];
    wave_lds_sync();
}

// ---------------------------------------------------------------------------------------
// the kernel: one mode per instantiation
// ---------------------------------------------------------------------------------------
template <int JPL, int MODE>
__global__ __launch_bounds__(kBlock, (MODE == kRollout && JPL == 2) ? 6 : 8) void jss_kernel(Params p) {
    HIP_DYNAMIC_SHARED(int32_t, lds)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_regions = p.shared_table ? 1 : kWavesPerBlock;
    int32_t *table = lds + (p.shared_table ? 0 : wave * p.region_ints);
    float *scratch = reinterpret_cast<float *>(lds + n_regions * p.region_ints) + wave * (p.d.jmax * 7);

    const int b_raw = blockIdx.x * kWavesPerBlock + wave;                 // one env per wave
    const bool alive = b_raw < p.d.batch;
    const int b = alive ? b_raw : p.d.batch - 1;
    // 1. state loads first: they depend on nothing but the env index
    const RawEnv<JPL> raw = issue_loads<JPL>(b, lane, p);
    int a_in = JSS_ACTION_SKIP;
    if (MODE == kStep) a_in = __builtin_amdgcn_readfirstlane(p.actions[b]);
    bool selected = true;
    if ((MODE == kReset || MODE == kAdvance) && p.which) selected = __builtin_amdgcn_readfirstlane((int)p.which[b]) != 0;
    // 2. instance constants + op table -> LDS
    const int tid = __builtin_amdgcn_readfirstlane(p.d.table_of_env ? p.d.table_of_env[b] : (p.d.n_tables == 1 ? 0 : b));
    Ctx c;
    c.b = b;
    c.lane = lane;
    c.J = __builtin_amdgcn_readfirstlane(p.d.jobs[tid]);
    c.M = __builtin_amdgcn_readfirstlane(p.d.machines[tid]);
    c.max_time_op = __builtin_amdgcn_readfirstlane(p.d.max_time_op[tid]);
    c.max_time_jobs = __builtin_amdgcn_readfirstlane(p.d.max_time_jobs[tid]);
    c.sum_op = __builtin_amdgcn_readfirstlane(p.d.sum_op[tid]);
    c.ops = table;
    c.stride = p.stride;
    if (p.shared_table) {
        const int n0 = p.d.jobs[0] * p.d.mmax;
        stage_table(lds, p.d.ops, p.d.ops16, 0, n0, (int)threadIdx.x, kBlock);
    } else {
        const int n = c.J * p.d.mmax;
        stage_table(table, p.d.ops, p.d.ops16, (size_t)tid * p.d.jmax * p.d.mmax, n, lane, kWave);
    }
    __syncthreads();
    if (!alive) return;

    Env<JPL> e;
    Header hd = unpack_env(e, c, raw);
    if (MODE == kReset) {
        if (!selected) return;
        hd.episode += 1;
        hd.step = 0;
        reset_env(e, c, p);
        if (lane == 0) {
            p.o.reward[b] = 0.f;
            p.o.done[b] = 0;
        }
    } else if (MODE == kStep) {
        const int rn = step_env(e, c, p, a_in);
        const bool called = a_in != JSS_ACTION_SKIP;
        const bool done = !any_legal(e);
        if (called) hd.step += 1;
        if (lane == 0) {
            p.o.reward[b] = (float)rn / (float)c.max_time_op;            // :483-493 (0 for skipped / ignored actions)
            p.o.done[b] = done ? 1 : 0;                                  // :639-653
            if (called && done) p.o.makespan[b] = e.t;                   // last_time_step :650
            if (p.s.counters && called) {
                add_counters(p.s.counters + (size_t)b * 4, 1, done ? 1 : 0, done ? e.t : 0, rn);
            }
        }
    } else if (MODE == kAdvance) {
        if (!selected) return;
        int hole = 0;
        if (__ballot(e.tm > 0) == 0) e.err |= JSS_ERR_NOPE_IDLE;        // reference: IndexError (:517)
        else hole = advance(e, c);
        if (lane == 0 && p.hole) p.hole[b] = hole;
    } else if (MODE == kPolicy) {
        const int a = select_action(e, c, p.kind, p.seed, p.explore_q16, (uint64_t)(p.d.env_ids ? p.d.env_ids[b] : p.d.env_id_base + b),
                                    (uint32_t)hd.episode, (uint32_t)hd.step);
        if (lane == 0) p.actions_out[b] = a;
        return;
    } else {  // kRollout / kRollout1: n_iter x (policy + step), state stays in registers
        const uint64_t env_id = (uint64_t)(p.d.env_ids ? p.d.env_ids[b] : p.d.env_id_base + b);
        int n_steps = 0, n_done = 0, sum_makespan = 0, sum_rn = 0;
        int last_rn = 0, last_makespan = -1;
        const int n_iter = MODE == kRollout1 ? 1 : p.n_iter;
        for (int it = 0; it < n_iter; ++it) {
            if (!any_legal(e)) {                                         // done (:639-653)
                if (!(p.flags & JSS_ROLLOUT_AUTORESET)) break;           // frozen
                reset_env(e, c, p);
                hd.episode += 1;
                hd.step = 0;
                continue;
            }
            const int a = select_action(e, c, p.kind, p.seed, p.explore_q16, env_id, (uint32_t)hd.episode,
                                        (uint32_t)hd.step);
            last_rn = step_env(e, c, p, a);
            hd.step += 1;
            n_steps += 1;
            sum_rn += last_rn;
            if (!any_legal(e)) {
                n_done += 1;
                sum_makespan += e.t;
                last_makespan = e.t;
            }
        }
        if (lane == 0) {
            if (n_steps) p.o.reward[b] = (float)last_rn / (float)c.max_time_op;
            p.o.done[b] = any_legal(e) ? 0 : 1;
            if (last_makespan >= 0) p.o.makespan[b] = last_makespan;
            if (p.s.counters) {
                add_counters(p.s.counters + (size_t)b * 4, n_steps, n_done, sum_makespan, sum_rn);
            }
        }
    }
    store_env(e, c, p, hd);
    if (!(p.ablate & JSS_ABLATE_OBS)) store_obs(e, c, p, scratch);
}

}  // namespace jss
