#pragma once
// Packed-8 kernel: EIGHT envs per wavefront for instances with J <= 16 and M <= 16.
//
// Why: the hardware holds 32 waves per CU.  With 4 envs per wave (jss_packed_env.hpp, G = 16) a
// 65 536-env launch needs two rounds of resident waves, and its time is rounds x contended wave latency
// (profiles/README.md, dispatch timeline).  Here an env owns 8 lanes and every lane carries two jobs
// (j = gl and gl + 8) and two machines (m = gl and gl + 8), so 65 536 envs are 8 192 waves: one round.
//
// Same structure as the G = 16 kernel -- group-uniform values replicated in VGPRs, ballots shifted down
// to the group, DPP reductions (3 steps: lane^1, lane^2, half-row mirror), ds_bpermute lookups -- with
// two slots per lane.  Semantics and citations: jss_wave_env.hpp / reference JSSEnv/envs/jss_env.py.
#include "jss_common.hpp"

namespace jss {

constexpr int kG8 = 8;                  // lanes per env
constexpr int kE8 = kWave / kG8;        // envs per wave

struct Q8Ctx {
    int lane, gl, gbase;
    int b;
    bool alive;
    bool jvalid[2], mvalid[2];
    int J, M, max_time_op, max_time_jobs, sum_op;
    const int32_t *ops;
    int stride;
};

struct Q8Env {
    int t;
    int todo[2], cur[2], left[2], perf[2], idle[2], idle_last[2], f4[2];
    int tm[2];
    bool legal[2], blocked[2];
    int noop, err;
};

// bits 0-7: slot 0 of the group's lanes, bits 8-15: slot 1  ==  bit j for job / machine j
__device__ __forceinline__ uint32_t q8_ballot(bool p0, bool p1, int gbase) {
    const uint64_t w0 = __ballot(p0), w1 = __ballot(p1);
    const uint32_t h0 = (gbase & 32) ? (uint32_t)(w0 >> 32) : (uint32_t)w0;
    const uint32_t h1 = (gbase & 32) ? (uint32_t)(w1 >> 32) : (uint32_t)w1;
    const int sh = gbase & 31;
    return ((h0 >> sh) & 0xFFu) | (((h1 >> sh) & 0xFFu) << 8);
}
__device__ __forceinline__ bool q8_any(bool p0, bool p1, int gbase) { return q8_ballot(p0, p1, gbase) != 0; }

__device__ __forceinline__ int q8_min(int v) {   // over the 8 lanes of the group
    v = imin(v, JSS_DPP(v, 0xB1));
    v = imin(v, JSS_DPP(v, 0x4E));
    v = imin(v, JSS_DPP(v, 0x141));               // row_half_mirror: the two quads of the 8-lane half row
    return v;
}
__device__ __forceinline__ int q8_max(int v) {
    v = imax(v, JSS_DPP(v, 0xB1));
    v = imax(v, JSS_DPP(v, 0x4E));
    v = imax(v, JSS_DPP(v, 0x141));
    return v;
}
__device__ __forceinline__ int q8_or(int v) {
    v |= JSS_DPP(v, 0xB1);
    v |= JSS_DPP(v, 0x4E);
    v |= JSS_DPP(v, 0x141);
    return v;
}

// job / machine `idx` (group-uniform index): the owning lane picks the slot, one cross-lane read
__device__ __forceinline__ int q8_read_u(const int (&v)[2], int idx, int gbase) {
    const int x = (idx & 8) ? v[1] : v[0];
    return __builtin_amdgcn_ds_bpermute((gbase + (idx & 7)) << 2, x);
}
// time_until_available_machine of machine m, m different per lane: both slots travel in one word
// (durations, hence machine times, are <= 65535)
__device__ __forceinline__ int q8_pack_tm(const Q8Env &e) { return (e.tm[0] & 0xFFFF) | (e.tm[1] << 16); }
__device__ __forceinline__ int q8_tm_of(int packed_tm, int m, int gbase) {
    const int w = __builtin_amdgcn_ds_bpermute((gbase + (m & 7)) << 2, packed_tm);
    return (m & 8) ? (int)((uint32_t)w >> 16) : (w & 0xFFFF);
}

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void q8_reset(Q8Env &e, const Q8Ctx &c, const Params &p, bool on) {
    if (on) {
        e.t = 0;
        e.noop = 0;
        e.err = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int j = s * 8 + c.gl;
            e.tm[s] = 0;
            e.todo[s] = 0;
            e.cur[s] = c.jvalid[s] ? c.ops[j * c.stride] : -1;           // :174-176
            e.left[s] = e.perf[s] = e.idle[s] = e.idle_last[s] = 0;
            e.f4[s] = 0;
            e.legal[s] = c.jvalid[s];
            e.blocked[s] = false;
        }
        if (c.alive) {                                                   // solution = -1 (:163)
            int32_t *sol = p.s.solution + (size_t)c.b * p.d.jmax * p.d.mmax;
            const int n = c.J * p.d.mmax;
            for (int i = c.gl; i < n; i += kG8) sol[i] = -1;
        }
    }
}

__device__ __forceinline__ int q8_next_event(const Q8Env &e) {
    const int a = e.tm[0] > 0 ? e.tm[0] : kBig, b = e.tm[1] > 0 ? e.tm[1] : kBig;
    return q8_min(imin(a, b));
}

__device__ __forceinline__ void q8_prefetch_next_op(const Q8Env &e, const Q8Ctx &c, int (&nxt)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        nxt[s] = (c.jvalid[s] && e.todo[s] + 1 < c.M) ? c.ops[(s * 8 + c.gl) * c.stride + e.todo[s] + 1] : -1;
}

// increase_time_step(): jss_env.py:495-637
__device__ __forceinline__ int q8_advance(Q8Env &e, const Q8Ctx &c, bool act, int d, const int (&next_op)[2]) {
    const int idle_machines = __popc(q8_ballot(c.mvalid[0] && e.tm[0] < d, c.mvalid[1] && e.tm[1] < d, c.gbase));
    const int hole = d * idle_machines;                                  // :606-608
    bool fin[2] = {false, false};
    if (act) {
        e.t += d;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int was = e.left[s];
            if (was > 0) {                                               // :529 running
                const int nl = imax(0, was - d);
                e.perf[s] += imin(d, was);
                e.left[s] = nl;
                if (nl == 0) {                                           // :550
                    e.idle[s] += d - was;
                    e.idle_last[s] = d - was;
                    e.todo[s] += 1;
                    fin[s] = true;
                    e.cur[s] = next_op[s];                               // :562-566 / :581
                }
            } else if (c.jvalid[s] && e.todo[s] < c.M) {                 // :594 waiting
                e.idle[s] += d;
                e.idle_last[s] += d;
            }
            e.tm[s] = imax(0, e.tm[s] - d);                              // :611
        }
    }
    const int ptm = q8_pack_tm(e);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int tm_need = q8_tm_of(ptm, e.cur[s] >> 16, c.gbase);
        if (fin[s]) e.f4[s] = e.cur[s] >= 0 ? tm_need : JSS_F4_ONE;      // :569-586
        if (act && c.jvalid[s] && e.cur[s] >= 0 && tm_need == 0 && !e.blocked[s]) e.legal[s] = true;  // :616-634
    }
    return hole;
}

// _prioritization_non_final(): jss_env.py:183-254
__device__ __forceinline__ void q8_prioritize(Q8Env &e, const Q8Ctx &c, bool on) {
    const bool f0 = on && e.legal[0] && e.todo[0] == c.M - 1, f1 = on && e.legal[1] && e.todo[1] == c.M - 1;
    uint32_t bits = q8_ballot(f0, f1, c.gbase);
    if (__ballot(bits != 0) == 0) return;
    bool nf[2];
    const int ptm = q8_pack_tm(e);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bool cand = on && e.legal[s] && e.todo[s] < c.M - 1;
        const int next_m = cand ? (c.ops[(s * 8 + c.gl) * c.stride + e.todo[s] + 1] >> 16) : 0;
        const int tm_next = q8_tm_of(ptm, next_m, c.gbase);             // collective: evaluated on every lane
        nf[s] = cand && tm_next == 0;                                    // :234
    }
    while (__ballot(bits != 0) != 0) {
        const int l = bits ? __ffs(bits) - 1 : 0;
        const int cf = q8_read_u(e.cur, l, c.gbase);
        const int mf = cf >> 16, df = cf & kDurMask;
        const bool h0 = bits != 0 && nf[0] && (e.cur[0] >> 16) == mf && (e.cur[0] & kDurMask) < df;
        const bool h1 = bits != 0 && nf[1] && (e.cur[1] >> 16) == mf && (e.cur[1] & kDurMask) < df;
        const bool hit = q8_any(h0, h1, c.gbase);
        if (bits != 0 && hit && c.gl == (l & 7)) {
            if (l & 8) e.legal[1] = false; else e.legal[0] = false;      // :253-254
        }
        bits &= bits - 1;
    }
}

// _check_no_op(): jss_env.py:256-401 (see p_check_no_op in jss_packed_env.hpp for the formulation)
__device__ __forceinline__ void q8_check_no_op(Q8Env &e, const Q8Ctx &c, bool on, int32_t *mvtab) {
    if (on) e.noop = 0;
    const uint32_t lm = q8_ballot(e.legal[0], e.legal[1], c.gbase);
    const int nl = __popc(lm);
    const int d_next = q8_next_event(e);
    const bool busy = d_next < kBig;
    bool gate = on && nl >= 1 && nl <= 4 && busy;
    if (__ballot(gate) == 0) return;
    const int nxt = e.t + d_next;
    const int cap = e.t + c.max_time_op;
    int my_m[2], my_end[2], h[2], mv[2];
    bool m_legal[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        my_m[s] = e.cur[s] >> 16;
        my_end[s] = e.t + (e.cur[s] & kDurMask);
        h[s] = imin(cap, my_end[s]);
        mv[s] = cap;
        m_legal[s] = false;
    }
    int cf[4];
    bool has[4];
    {
        uint32_t bits = lm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            has[r] = gate && bits != 0;
            const int l = bits ? __ffs(bits) - 1 : 0;
            cf[r] = q8_read_u(e.cur, l, c.gbase);
            bits &= bits - 1;
        }
    }
    const int ptm = q8_pack_tm(e);
    int tm_need[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) tm_need[s] = q8_tm_of(ptm, my_m[s], c.gbase);
    {
        uint32_t bits = lm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = bits ? __ffs(bits) - 1 : 0;
            const int m_r = cf[r] >> 16, end_r = e.t + (cf[r] & kDurMask);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int j = s * 8 + c.gl;
                if (has[r] && e.legal[s] && l < j && m_r == my_m[s]) h[s] = imin(h[s], end_r);   // :318
                if (has[r] && j == m_r) {                                                         // machine lane
                    mv[s] = imin(mv[s], end_r);
                    m_legal[s] = true;
                }
            }
            bits &= bits - 1;
        }
    }
    const uint32_t legal_machines = q8_ballot(m_legal[0], m_legal[1], c.gbase);
    gate = gate && __popc(legal_machines) <= 3;                          // :286
    const bool ends_early = q8_any(e.legal[0] && my_end[0] < nxt, e.legal[1] && my_end[1] < nxt, c.gbase);
    gate = gate && !ends_early;                                          // :314-315
    const int mh = q8_max(imax(e.legal[0] ? h[0] : e.t, e.legal[1] ? h[1] : e.t));  // :296, :321
    int32_t *tab = mvtab + c.gbase * 2;                                  // 16 ints per group
    tab[c.gl] = m_legal[0] ? mv[0] : -kBig;
    tab[8 + c.gl] = m_legal[1] ? mv[1] : -kBig;
    wave_lds_sync();
    int u = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bool caseA = c.jvalid[s] && !e.legal[s] && e.left[s] > 0 && e.todo[s] + 1 < c.M;       // :327-330
        const bool caseB = c.jvalid[s] && !e.legal[s] && !caseA && !e.blocked[s] && e.todo[s] < c.M;  // :366-369
        int k = caseA ? e.todo[s] + 1 : e.todo[s];
        int tn = caseA ? e.t + e.left[s] : e.t + tm_need[s];
        if (gate && (caseA || caseB)) {
            while (k < c.M - 1 && mh > tn) {                             // :340-342 / :380-382
                const int op = c.ops[(s * 8 + c.gl) * c.stride + k];
                const int m = op >> 16;
                if (tab[m] > tn) u |= 1 << m;                            // :346-351
                tn += op & kDurMask;
                ++k;
            }
        }
    }
    const int covered = q8_or(u);
    if (gate && (uint32_t)covered == legal_machines) e.noop = 1;         // :357-359 / :395-397
    wave_lds_sync();
}

// step(): jss_env.py:403-481
__device__ __forceinline__ int q8_step(Q8Env &e, const Q8Ctx &c, const Params &p, int a, int32_t *mvtab) {
    const bool is_nope = c.alive && a == c.J;
    const bool is_job = c.alive && a >= 0 && a < c.J;
    if (c.alive && (a < JSS_ACTION_SKIP || a > c.J)) e.err |= JSS_ERR_BAD_ACTION;
    const bool mine0 = c.gl == a, mine1 = c.gl + 8 == a;
    const bool a_legal = q8_any(mine0 && e.legal[0], mine1 && e.legal[1], c.gbase);
    if (is_job && !a_legal) e.err |= JSS_ERR_ILLEGAL_ACTION;
    const bool alloc = is_job && a_legal;
    const int ca = q8_read_u(e.cur, a, c.gbase);
    const int m = ca >> 16, d = ca & kDurMask;
    int rn = 0;
    if (alloc) {
        rn = d;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s * 8 + c.gl == m) e.tm[s] = d;                          // :446
            if (s * 8 + c.gl == a) {
                e.left[s] = d;                                           // :447
                p.s.solution[((size_t)c.b * p.d.jmax + a) * p.d.mmax + e.todo[s]] = e.t;  // :454
            }
            if (e.cur[s] >= 0 && (e.cur[s] >> 16) == m) {
                e.legal[s] = false;                                      // :455-463
                e.blocked[s] = false;                                    // :464-467
            }
        }
    }
    if (is_nope) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            e.blocked[s] = e.blocked[s] || e.legal[s];
            e.legal[s] = false;
        }
    }
    const bool stepping = alloc || is_nope;
    for (;;) {
        const bool none_legal = !q8_any(e.legal[0], e.legal[1], c.gbase);
        if (__ballot(stepping && none_legal) == 0) break;
        int next_op[2];
        q8_prefetch_next_op(e, c, next_op);
        const int dd = q8_next_event(e);
        const bool busy = dd < kBig;
        bool act = stepping && none_legal;
        if (act && !busy && is_nope) e.err |= JSS_ERR_NOPE_IDLE;
        act = act && busy;
        if (__ballot(act) == 0 || (p.ablate & JSS_ABLATE_ADVANCE)) break;
        const int hole = q8_advance(e, c, act, dd, next_op);
        if (act) rn -= hole;
    }
    if (!(p.ablate & JSS_ABLATE_PRIORITIZE)) q8_prioritize(e, c, stepping);
    if (!(p.ablate & JSS_ABLATE_CHECK_NO_OP)) q8_check_no_op(e, c, stepping, mvtab);
    return rn;
}

// action selectors
__device__ __forceinline__ int q8_select(const Q8Env &e, const Q8Ctx &c, int kind, uint64_t seed, uint32_t explore_q16,
                                         uint64_t env_id, uint32_t episode, uint32_t step) {
    const uint32_t lm = q8_ballot(e.legal[0], e.legal[1], c.gbase);
    const int nl = __popc(lm);
    const int n = nl + (e.noop ? 1 : 0);
    int a;
    if (kind == JSS_POLICY_RANDOM) {
        const uint32_t r = rng_u32(seed, env_id, episode, step);
        const int pick = (int)__umulhi(r, (uint32_t)n);
        const int below0 = __popc(lm & ((1u << c.gl) - 1u)), below1 = __popc(lm & ((1u << (c.gl + 8)) - 1u));
        const uint32_t hit = q8_ballot(e.legal[0] && below0 == pick, e.legal[1] && below1 == pick, c.gbase);
        a = hit ? __ffs(hit) - 1 : c.J;
    } else {
        if (kind == JSS_POLICY_CR) {
            CrKey best;
            best.num = 0x3fffffff;
            best.den = 1;
            best.idx = kCrNone;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                int total = 0, remaining = 0;
                if (e.legal[s])
                    for (int k = 0; k < c.M; ++k) {
                        const int dd = c.ops[(s * 8 + c.gl) * c.stride + k] & kDurMask;
                        total += dd;
                        if (k >= e.todo[s]) remaining += dd;
                    }
                CrKey key;
                key.num = e.legal[s] ? 3 * total - 2 * e.t : 0x3fffffff;
                key.den = e.legal[s] ? remaining : 1;
                key.idx = e.legal[s] ? s * 8 + c.gl : kCrNone;
                if (cr_better(key, best)) best = key;
            }
            best = cr_argmin<kG8>(best);
            a = best.idx < kCrNone ? best.idx : c.J;
        } else {
            const bool larger = (kind == JSS_POLICY_FIFO || kind == JSS_POLICY_MWR || kind == JSS_POLICY_MOR);
            int key[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                int v;
                if (kind == JSS_POLICY_FIFO) v = e.idle_last[s];
                else if (kind == JSS_POLICY_SPT) v = e.cur[s] & kDurMask;
                else if (kind == JSS_POLICY_MOR || kind == JSS_POLICY_LOR) v = c.M - e.todo[s];
                else {
                    v = 0;
                    if (e.legal[s])
                        for (int k = e.todo[s]; k < c.M; ++k) v += c.ops[(s * 8 + c.gl) * c.stride + k] & kDurMask;
                }
                key[s] = e.legal[s] ? (larger ? v : -v) : -kBig;
            }
            const int best = q8_max(imax(key[0], key[1]));
            const uint32_t hit = q8_ballot(e.legal[0] && key[0] == best, e.legal[1] && key[1] == best, c.gbase);
            a = hit ? __ffs(hit) - 1 : c.J;
        }
        if (e.noop && explore_q16 != 0) {
            const uint32_t r = rng_u32(seed ^ kExploreSeedXor, env_id, episode, step);
            if ((r >> 16) < explore_q16) a = c.J;
        }
    }
    return n == 0 ? -1 : a;
}

// HBM <-> registers
struct Q8Raw {
    int4 h, lo[2], hi[2];
    int tm[2];
};
struct Q8Header {
    int episode, step;
};

__device__ __forceinline__ Q8Raw q8_issue_loads(int b, int gl, const Params &p) {
    Q8Raw r;
    const int jm = p.d.jmax, mm = p.d.mmax;
    r.h = reinterpret_cast<const int4 *>(p.s.env)[b];
    const int4 *js = reinterpret_cast<const int4 *>(p.s.job) + (size_t)b * jm * 2;
    const int32_t *ms = p.s.machine + (size_t)b * mm;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = s * 8 + gl;
        const int jc = j < jm ? j : 0, mc = j < mm ? j : 0;
        r.lo[s] = js[jc * 2];
        r.hi[s] = js[jc * 2 + 1];
        r.tm[s] = ms[mc];
    }
    return r;
}

__device__ __forceinline__ Q8Header q8_unpack(Q8Env &e, const Q8Ctx &c, const Q8Raw &r) {
    e.t = r.h.x;
    e.err = r.h.w & 0xFF;
    e.noop = (r.h.w & JSS_STATUS_NOOP) ? 1 : 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bool v = c.jvalid[s];
        e.tm[s] = c.mvalid[s] ? r.tm[s] : 0;
        e.todo[s] = v ? r.lo[s].x : 0;
        e.cur[s] = v ? r.lo[s].y : -1;
        e.left[s] = v ? r.lo[s].z : 0;
        e.perf[s] = v ? r.lo[s].w : 0;
        e.idle[s] = v ? r.hi[s].x : 0;
        e.idle_last[s] = v ? r.hi[s].y : 0;
        e.f4[s] = v ? r.hi[s].z : 0;
        e.legal[s] = v && (r.hi[s].w & JSS_FLAG_LEGAL);
        e.blocked[s] = v && (r.hi[s].w & JSS_FLAG_BLOCKED);
    }
    Q8Header hd;
    hd.episode = r.h.y;
    hd.step = r.h.z;
    return hd;
}

__device__ __forceinline__ void q8_store(const Q8Env &e, const Q8Ctx &c, const Params &p, const Q8Header &hd) {
    if (!c.alive) return;
    const int jm = p.d.jmax;
    uint8_t *mk = p.o.action_mask + (size_t)c.b * (jm + 1);
    if (c.gl == 0) {
        reinterpret_cast<int4 *>(p.s.env)[c.b] =
            make_int4(e.t, hd.episode, hd.step, (e.err & 0xFF) | (e.noop ? JSS_STATUS_NOOP : 0));
        mk[c.J] = (uint8_t)e.noop;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = s * 8 + c.gl;
        if (c.mvalid[s]) p.s.machine[(size_t)c.b * p.d.mmax + j] = e.tm[s];
        if (c.jvalid[s]) {
            int4 *js = reinterpret_cast<int4 *>(p.s.job) + ((size_t)c.b * jm + j) * 2;
            js[0] = make_int4(e.todo[s], e.cur[s], e.left[s], e.perf[s]);
            js[1] = make_int4(e.idle[s], e.idle_last[s], e.f4[s],
                              (e.legal[s] ? JSS_FLAG_LEGAL : 0) | (e.blocked[s] ? JSS_FLAG_BLOCKED : 0));
            mk[j] = e.legal[s] ? 1 : 0;
        }
    }
}

__device__ __forceinline__ void q8_store_obs(const Q8Env &e, const Q8Ctx &c, const Params &p, float *scratch, int first_env,
                                             bool wave_whole) {
    const int row_floats = p.d.jmax * 7;
    const float f_op = (float)c.max_time_op, f_jobs = (float)c.max_time_jobs, f_sum = (float)c.sum_op, f_m = (float)c.M;
    const float r_op = refined_rcp(f_op), r_jobs = refined_rcp(f_jobs), r_sum = refined_rcp(f_sum), r_m = refined_rcp(f_m);
    float *mine = scratch + (c.gbase / kG8) * row_floats;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = s * 8 + c.gl;
        if (j < p.d.jmax) {
            float *row = mine + j * 7;
            row[0] = e.legal[s] ? 1.0f : 0.0f;
            row[1] = div_by((float)e.left[s], f_op, r_op);
            row[2] = div_by((float)e.todo[s], f_m, r_m);
            row[3] = div_by((float)e.perf[s], f_jobs, r_jobs);
            row[4] = e.f4[s] == JSS_F4_ONE ? 1.0f : div_by((float)e.f4[s], f_op, r_op);
            row[5] = div_by((float)e.idle_last[s], f_sum, r_sum);
            row[6] = div_by((float)e.idle[s], f_sum, r_sum);
        }
    }
    wave_lds_sync();
    const int n = kE8 * row_floats;
    if (wave_whole && (n & 3) == 0) {
        const float4 *src = reinterpret_cast<const float4 *>(scratch);
        float4 *dst = reinterpret_cast<float4 *>(p.o.real_obs + (size_t)first_env * row_floats);
        for (int i = c.lane; i < (n >> 2); i += kWave) dst[i] = src[i];
    } else if (c.alive) {
        float *dst = p.o.real_obs + (size_t)c.b * row_floats;
        for (int i = c.gl; i < row_floats; i += kG8) dst[i] = mine[i];
    }
    wave_lds_sync();
}

template <int MODE>
__device__ __forceinline__ void q8_body(Q8Env &e, Q8Header &hd, const Q8Ctx &c, const Params &p, int a_in, bool selected,
                                        int32_t *mvtab) {
    if (MODE == kReset) {
        const bool on = c.alive && selected;
        q8_reset(e, c, p, on);
        if (on) {
            hd.episode += 1;
            hd.step = 0;
            if (c.gl == 0) {
                p.o.reward[c.b] = 0.f;
                p.o.done[c.b] = 0;
            }
        }
    } else if (MODE == kStep) {
        const int rn = q8_step(e, c, p, a_in, mvtab);
        const bool called = a_in != JSS_ACTION_SKIP;
        const bool done = !q8_any(e.legal[0], e.legal[1], c.gbase);
        if (called) hd.step += 1;
        if (c.alive && c.gl == 0) {
            p.o.reward[c.b] = (float)rn / (float)c.max_time_op;
            p.o.done[c.b] = done ? 1 : 0;
            if (called && done) p.o.makespan[c.b] = e.t;
            if (p.s.counters && called)
                add_counters(p.s.counters + (size_t)c.b * 4, 1, done ? 1 : 0, done ? e.t : 0, rn);
        }
    } else if (MODE == kAdvance) {
        const bool on = c.alive && selected;
        int next_op[2];
        q8_prefetch_next_op(e, c, next_op);
        const int d = q8_next_event(e);
        const bool busy = d < kBig;
        if (on && !busy) e.err |= JSS_ERR_NOPE_IDLE;
        const int hole = q8_advance(e, c, on && busy, d, next_op);
        if (on && c.gl == 0 && p.hole) p.hole[c.b] = busy ? hole : 0;
    } else if (MODE == kPolicy) {
        const int a = q8_select(e, c, p.kind, p.seed, p.explore_q16,
                                (uint64_t)(p.d.env_ids ? p.d.env_ids[c.b] : p.d.env_id_base + c.b), (uint32_t)hd.episode,
                                (uint32_t)hd.step);
        if (c.alive && c.gl == 0) p.actions_out[c.b] = a;
    } else {  // kRollout / kRollout1
        const uint64_t env_id = (uint64_t)(p.d.env_ids ? p.d.env_ids[c.b] : p.d.env_id_base + c.b);
        int n_steps = 0, n_done = 0, last_rn = 0, last_makespan = -1, sum_makespan = 0, sum_rn = 0;
        const bool autoreset = (p.flags & JSS_ROLLOUT_AUTORESET) != 0;
        const int n_iter = MODE == kRollout1 ? 1 : p.n_iter;
        for (int it = 0; it < n_iter; ++it) {
            const bool done0 = !q8_any(e.legal[0], e.legal[1], c.gbase);
            const bool do_reset = c.alive && done0 && autoreset;
            const bool do_step = c.alive && !done0;
            if (MODE != kRollout1 && __ballot(do_reset || do_step) == 0) break;
            q8_reset(e, c, p, do_reset);
            if (do_reset) {
                hd.episode += 1;
                hd.step = 0;
            }
            int a = q8_select(e, c, p.kind, p.seed, p.explore_q16, env_id, (uint32_t)hd.episode, (uint32_t)hd.step);
            if (!do_step) a = JSS_ACTION_SKIP;
            const int rn = q8_step(e, c, p, a, mvtab);
            const bool done1 = !q8_any(e.legal[0], e.legal[1], c.gbase);
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
        }
        const bool done = !q8_any(e.legal[0], e.legal[1], c.gbase);
        if (c.alive && c.gl == 0) {
            if (n_steps) p.o.reward[c.b] = (float)last_rn / (float)c.max_time_op;
            p.o.done[c.b] = done ? 1 : 0;
            if (last_makespan >= 0) p.o.makespan[c.b] = last_makespan;
            if (p.s.counters) add_counters(p.s.counters + (size_t)c.b * 4, n_steps, n_done, sum_makespan, sum_rn);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock, MODE == kRollout ? 4 : 8) void jss_packed8_kernel(Params p) {
    HIP_DYNAMIC_SHARED(int32_t, lds)
    constexpr int EB = kE8 * kWavesPerBlock;          // 32 envs per workgroup
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int grp_in_block = threadIdx.x / kG8;
    float *scratch = reinterpret_cast<float *>(lds + p.obs_off_ints) + wave * p.obs_wave_floats;
    int32_t *mvtab = lds + p.mv_off_ints + wave * (2 * kWave);

    Q8Ctx c;
    c.lane = lane;
    c.gl = lane & (kG8 - 1);
    c.gbase = lane & ~(kG8 - 1);
    const int b_raw = blockIdx.x * EB + grp_in_block;
    const int first_env = blockIdx.x * EB + wave * kE8;
    const bool wave_whole = first_env + kE8 <= p.d.batch;
    c.alive = b_raw < p.d.batch;
    c.b = c.alive ? b_raw : p.d.batch - 1;
    const Q8Raw raw = q8_issue_loads(c.b, c.gl, p);
    int a_in = JSS_ACTION_SKIP;
    if (MODE == kStep) a_in = p.actions[c.b];
    bool selected = true;
    if ((MODE == kReset || MODE == kAdvance) && p.which) selected = p.which[c.b] != 0;
    const int tid = p.shared_table ? 0 : (p.d.table_of_env ? p.d.table_of_env[c.b] : c.b);
    c.J = p.d.jobs[tid];
    c.M = p.d.machines[tid];
    c.max_time_op = p.d.max_time_op[tid];
    c.max_time_jobs = p.d.max_time_jobs[tid];
    c.sum_op = p.d.sum_op[tid];
    c.stride = p.stride;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        c.jvalid[s] = s * 8 + c.gl < c.J;
        c.mvalid[s] = s * 8 + c.gl < c.M;
    }
    int32_t *table = lds + (p.shared_table ? 0 : grp_in_block * p.region_ints);
    c.ops = table;
    if (p.shared_table) {
        stage_table(lds, p.d.ops, p.d.ops16, 0, p.d.jobs[0] * p.d.mmax, (int)threadIdx.x, kBlock);
    } else {
        stage_table(table, p.d.ops, p.d.ops16, (size_t)tid * p.d.jmax * p.d.mmax, c.J * p.d.mmax, c.gl, kG8);
    }
    __syncthreads();

    Q8Env e;
    Q8Header hd = q8_unpack(e, c, raw);
    q8_body<MODE>(e, hd, c, p, a_in, selected, mvtab);
    if (MODE == kPolicy) return;
    q8_store(e, c, p, hd);
    if (!(p.ablate & JSS_ABLATE_OBS)) q8_store_obs(e, c, p, scratch, first_env, wave_whole);
}

}  // namespace jss
