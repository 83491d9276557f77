// jssenv_amd/csrc/jss_cpu.cpp -- libjss_cpu.so: the host-core twin of libjss_hip.so.
//
// Same C ABI (include/jss_hip.h: identical symbols, structs and memory layouts; every pointer is a host
// pointer, `stream` is ignored, calls are synchronous), written from the kernels' queue-free restatement of
// the simulator: scalar C++ working in place on the 32-byte job records, OpenMP over envs.  It exists for
// BASELINE config 1 ("runs without a GPU"), for `device="cpu"` users of the package, and as the multi-core
// CPU baseline bench.py times next to the GPU (cpu_baseline kind "twin").  It shares no code with oracle/
// (the literal restatement of the reference used as the checker) -- tests/ compares the two.
//
// Reference semantics (JSSEnv/envs/jss_env.py, cited per function) in the queue-free form: the reference's
// sorted event list is {t + tm[m] : tm[m] > 0}, its M x J illegal_actions matrix is blocked[j] && need[j] == m,
// and nb_legal_actions / machine_legal / nb_machine_legal are functions of the legal flags.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "jss_hip.h"

namespace {

constexpr int kBig = 0x3fffffff;
constexpr int kDurMask = 0xffff;
constexpr uint64_t kExploreSeedXor = 0x5851F42D4C957F2DULL;

struct Call {   // one ABI call
    JssDesc d;
    JssState s;
    JssOut o;
    const int32_t *actions = nullptr;
    int32_t *actions_out = nullptr;
    const uint8_t *which = nullptr;
    int32_t *hole = nullptr;
    JssTraj t = JssTraj();
    uint64_t seed = 0;
    uint32_t explore_q16 = 0;
    int kind = 0;
    int n_iter = 0;
    int flags = 0;
};

// One env: pointers into the batch tensors + its instance.
struct Env {
    int J, M, max_time_op, tid;
    const int32_t *norm;        // the six observation normalisers: max_time_jobs, sum_op, four float32 reciprocals
    const int32_t *ops, *rem;   // [jmax][mmax] tables of my instance (rem may be null)
    int stride, jmax, mmax;
    int32_t *hdr;               // [JSS_NH]
    int32_t *cst;               // [JSS_NC] the env's constants record
    int32_t *job;               // [jmax][JSS_NF]
    int32_t *tm;                // [mmax]
    int32_t *sol;               // [jmax][mmax]

    // word f (JSS_F_*) of job j's record.  `job` is the batch tensor itself for full records; for compact 16-byte
    // records (JSS_FC_*) it is a thread-local full-layout copy that load_compact() fills and store_compact() writes back
    int32_t *packed;            // compact / medium records of my env in the batch tensor (nullptr with full records)
    int packed_ints;            // JSS_NFC or JSS_NFM (0 with full records)
    int32_t &w(int j, int f) const { return job[j * JSS_NF + f]; }
    int op_at(int j, int k) const { return k < M ? ops[j * stride + k] : -1; }
    int cur(int j) const { return w(j, JSS_F_CUR); }                                     // current op, -1 = job finished
    int nxt(int j) const { return w(j, JSS_F_NEXT); }
    int todo(int j) const { return w(j, JSS_F_TODO) & JSS_TODO_MASK; }
    bool legal(int j) const { return w(j, JSS_F_TODO) & JSS_FLAG_LEGAL; }
    bool blocked(int j) const { return w(j, JSS_F_TODO) & JSS_FLAG_BLOCKED; }
    void set_legal(int j, bool v) const { w(j, JSS_F_TODO) = (w(j, JSS_F_TODO) & ~JSS_FLAG_LEGAL) | (v ? JSS_FLAG_LEGAL : 0); }
    void set_blocked(int j, bool v) const { w(j, JSS_F_TODO) = (w(j, JSS_F_TODO) & ~JSS_FLAG_BLOCKED) | (v ? JSS_FLAG_BLOCKED : 0); }
    int next2(int j) const {                                             // op table entry [j][todo + 2], -1 = none
        const unsigned v = (unsigned)w(j, JSS_F_TODO) >> JSS_NEXT2_SHIFT;
        return v ? (int)v : -1;
    }
    // the record's cached ops after todo(j) changed (or at reset): cur <- [todo], next <- [todo + 1], next2 <- [todo + 2]
    void refresh_ops(int j, bool valid) const {
        const int k = todo(j);
        w(j, JSS_F_CUR) = valid ? op_at(j, k) : -1;
        w(j, JSS_F_NEXT) = valid ? op_at(j, k + 1) : -1;
        set_next2(j, valid ? op_at(j, k + 2) : -1);
    }
    void set_next2(int j, int op) const {
        w(j, JSS_F_TODO) = (int32_t)(((unsigned)w(j, JSS_F_TODO) & ((1u << JSS_NEXT2_SHIFT) - 1u)) |
                                     (op >= 0 ? (unsigned)op << JSS_NEXT2_SHIFT : 0u));
    }
    int &t() const { return hdr[JSS_H_CLOCK]; }
    int noop() const { return (hdr[JSS_H_STATUS] & JSS_STATUS_NOOP) ? 1 : 0; }
    void set_noop(int v) const { hdr[JSS_H_STATUS] = (hdr[JSS_H_STATUS] & ~JSS_STATUS_NOOP) | (v ? JSS_STATUS_NOOP : 0); }
    void flag(int err) const { hdr[JSS_H_STATUS] |= err; }
};

// compact 16-byte records <-> the full-layout working copy (the cached ops are what the op table says)
void load_compact(const Env &e) {
    for (int j = 0; j < e.jmax; ++j) {
        const int32_t *r = e.packed + j * JSS_NFC;
        const unsigned w0 = (unsigned)r[JSS_FC_W0], w1 = (unsigned)r[JSS_FC_LEFT_F4];
        const int todo = (int)(w0 & JSS_FC_TODO_MASK);
        const bool v = j < e.J;
        e.w(j, JSS_F_TODO) = todo | ((w0 & JSS_FC_FLAG_LEGAL) ? JSS_FLAG_LEGAL : 0) | ((w0 & JSS_FC_FLAG_BLOCKED) ? JSS_FLAG_BLOCKED : 0);
        e.w(j, JSS_F_LEFT) = (int)(w1 & 0xffffu);
        e.w(j, JSS_F_PERF) = (int)(w0 >> JSS_FC_PERF_SHIFT);
        e.w(j, JSS_F_IDLE) = r[JSS_FC_IDLE];
        e.w(j, JSS_F_IDLE_LAST) = r[JSS_FC_IDLE_LAST];
        e.w(j, JSS_F_F4) = (w0 & JSS_FC_FLAG_F4_ONE) ? JSS_F4_ONE : (int)(w1 >> 16);
        e.refresh_ops(j, v);
    }
    // no machine clocks in memory: a machine is busy for exactly as long as the job on it (:446-449, :521-530)
    for (int m = 0; m < e.mmax; ++m) e.tm[m] = 0;
    for (int j = 0; j < e.J; ++j)
        if (e.w(j, JSS_F_LEFT) > 0 && e.cur(j) >= 0) e.tm[e.cur(j) >> 16] = e.w(j, JSS_F_LEFT);
}
// 24-byte medium records (JSS_FM_*: per-env instances, jobs / machines <= 32) <-> the full-layout working copy
void load_medium(const Env &e) {
    for (int j = 0; j < e.jmax; ++j) {
        const int32_t *r = e.packed + j * JSS_NFM;
        const unsigned w0 = (unsigned)r[JSS_FM_W0], w1 = (unsigned)r[JSS_FM_LEFT_F4], w2 = (unsigned)r[JSS_FM_PERF_NEXT], w3 = (unsigned)r[JSS_FM_NEXT_NEXT2];
        const unsigned cur = (w0 >> JSS_FM_CUR_SHIFT) & JSS_FM_OP_MASK, nxt = (w2 >> 21) | ((w3 & 0x3FFu) << 11), nxt2 = (w3 >> 10) & JSS_FM_OP_MASK;
        e.w(j, JSS_F_TODO) = (int)(w0 & JSS_FM_TODO_MASK) | ((w0 & JSS_FM_FLAG_LEGAL) ? JSS_FLAG_LEGAL : 0) | ((w0 & JSS_FM_FLAG_BLOCKED) ? JSS_FLAG_BLOCKED : 0);
        e.w(j, JSS_F_CUR) = cur ? (int)cur : -1;
        e.w(j, JSS_F_LEFT) = (int)(w1 & 0xffffu);
        e.w(j, JSS_F_PERF) = (int)(w2 & JSS_FM_OP_MASK);
        e.w(j, JSS_F_IDLE) = r[JSS_FM_IDLE];
        e.w(j, JSS_F_IDLE_LAST) = r[JSS_FM_IDLE_LAST];
        e.w(j, JSS_F_F4) = (w0 & JSS_FM_FLAG_F4_ONE) ? JSS_F4_ONE : (int)(w1 >> 16);
        e.w(j, JSS_F_NEXT) = nxt ? (int)nxt : -1;
        e.set_next2(j, nxt2 ? (int)nxt2 : -1);
    }
    for (int m = 0; m < e.mmax; ++m) e.tm[m] = 0;                         // no machine clocks in memory (load_compact)
    for (int j = 0; j < e.J; ++j)
        if (e.w(j, JSS_F_LEFT) > 0 && e.cur(j) >= 0) e.tm[e.cur(j) >> 16] = e.w(j, JSS_F_LEFT);
}
void store_medium(const Env &e) {
    for (int j = 0; j < e.jmax; ++j) {
        int32_t *r = e.packed + j * JSS_NFM;
        const int w0 = e.w(j, JSS_F_TODO), f4 = e.w(j, JSS_F_F4);
        const unsigned cur = e.cur(j) >= 0 ? (unsigned)e.cur(j) : 0u, nxt = e.nxt(j) >= 0 ? (unsigned)e.nxt(j) : 0u;
        const unsigned nxt2 = e.next2(j) >= 0 ? (unsigned)e.next2(j) : 0u;
        r[JSS_FM_W0] = (int32_t)((unsigned)(w0 & JSS_TODO_MASK) | ((w0 & JSS_FLAG_LEGAL) ? JSS_FM_FLAG_LEGAL : 0u) |
                                 ((w0 & JSS_FLAG_BLOCKED) ? JSS_FM_FLAG_BLOCKED : 0u) | (f4 == JSS_F4_ONE ? JSS_FM_FLAG_F4_ONE : 0u) |
                                 (cur << JSS_FM_CUR_SHIFT));
        r[JSS_FM_LEFT_F4] = (int32_t)((unsigned)e.w(j, JSS_F_LEFT) | ((unsigned)(f4 == JSS_F4_ONE ? 0 : f4) << 16));
        r[JSS_FM_PERF_NEXT] = (int32_t)((unsigned)e.w(j, JSS_F_PERF) | (nxt << 21));
        r[JSS_FM_NEXT_NEXT2] = (int32_t)((nxt >> 11) | (nxt2 << 10));
        r[JSS_FM_IDLE] = e.w(j, JSS_F_IDLE);
        r[JSS_FM_IDLE_LAST] = e.w(j, JSS_F_IDLE_LAST);
    }
}
void store_compact(const Env &e) {
    if (!e.packed) return;
    if (e.packed_ints == JSS_NFM) {
        store_medium(e);
        return;
    }
    for (int j = 0; j < e.jmax; ++j) {
        int32_t *r = e.packed + j * JSS_NFC;
        const int w0 = e.w(j, JSS_F_TODO), f4 = e.w(j, JSS_F_F4);
        r[JSS_FC_W0] = (int32_t)((unsigned)(w0 & JSS_TODO_MASK) | ((w0 & JSS_FLAG_LEGAL) ? JSS_FC_FLAG_LEGAL : 0u) |
                                 ((w0 & JSS_FLAG_BLOCKED) ? JSS_FC_FLAG_BLOCKED : 0u) | (f4 == JSS_F4_ONE ? JSS_FC_FLAG_F4_ONE : 0u) |
                                 ((unsigned)e.w(j, JSS_F_PERF) << JSS_FC_PERF_SHIFT));
        r[JSS_FC_LEFT_F4] = (int32_t)((unsigned)e.w(j, JSS_F_LEFT) | ((unsigned)(f4 == JSS_F4_ONE ? 0 : f4) << 16));
        r[JSS_FC_IDLE] = e.w(j, JSS_F_IDLE);
        r[JSS_FC_IDLE_LAST] = e.w(j, JSS_F_IDLE_LAST);
    }
}

// The env's view of the batch.  from_instance: a reset -- its shape and normalisers come from its instance record
// (env -> table_of_env -> record) and are copied into its header; every other call reads them back from the header.
Env env_of(const Call &c, int b, bool from_instance) {
    const JssDesc &d = c.d;
    const size_t region = (size_t)d.jmax * d.mmax;
    Env e;
    e.hdr = c.s.env + (size_t)b * JSS_NH;
    e.cst = c.s.env_const + (size_t)b * JSS_NC;
    if (from_instance) {
        e.tid = d.table_of_env ? d.table_of_env[b] : (d.n_tables == 1 ? 0 : b);
        const int32_t *inst = d.inst + (size_t)e.tid * JSS_NI;
        e.J = inst[JSS_I_JOBS];
        e.M = inst[JSS_I_MACHINES];
        e.max_time_op = inst[JSS_I_MAX_TIME_OP];
        e.norm = inst + JSS_I_MAX_TIME_JOBS;
    } else {
        e.J = e.cst[JSS_C_JOBS];
        e.M = e.cst[JSS_C_MACHINES];
        e.max_time_op = e.cst[JSS_C_MAX_TIME_OP];
        e.tid = e.cst[JSS_C_TABLE];
        e.norm = e.cst + JSS_C_MAX_TIME_JOBS;
    }
    e.ops = d.ops + e.tid * region;
    e.rem = d.rem ? d.rem + e.tid * region : nullptr;
    e.stride = d.mmax;
    e.jmax = d.jmax;
    e.mmax = d.mmax;
    if (d.record_ints == JSS_NFC || d.record_ints == JSS_NFM) {
        static thread_local int32_t unpacked[JSS_MAX_JOBS * JSS_NF], clocks[JSS_MAX_MACHINES];
        e.packed = c.s.job + (size_t)b * d.jmax * d.record_ints;
        e.packed_ints = d.record_ints;
        e.job = unpacked;
        e.tm = clocks;
        if (d.record_ints == JSS_NFC) load_compact(e);
        else load_medium(e);
    } else {
        e.packed = nullptr;
        e.packed_ints = 0;
        e.job = c.s.job + (size_t)b * d.jmax * JSS_NF;
        e.tm = c.s.machine + (size_t)b * d.mmax;
    }
    e.sol = c.s.solution + b * region;
    return e;
}

int n_legal(const Env &e) {
    int n = 0;
    for (int j = 0; j < e.J; ++j) n += e.legal(j);
    return n;
}

bool any_busy(const Env &e) {
    for (int m = 0; m < e.M; ++m)
        if (e.tm[m] > 0) return true;
    return false;
}

// ---- reset(): jss_env.py:145-181 ---------------------------------------------------------------------------
void reset_env(const Env &e) {
    e.t() = 0;                                                            // :154
    e.hdr[JSS_H_STATUS] = 0;                                              // NOPE illegal (:161), error bits cleared
    // the instance constants travel with the env from here on (include/jss_hip.h JSS_C_*)
    e.cst[JSS_C_JOBS] = e.J;
    e.cst[JSS_C_MACHINES] = e.M;
    e.cst[JSS_C_MAX_TIME_OP] = e.max_time_op;
    e.cst[JSS_C_TABLE] = e.tid;
    if (e.norm != e.cst + JSS_C_MAX_TIME_JOBS)
        for (int i = 0; i < 6; ++i) e.cst[JSS_C_MAX_TIME_JOBS + i] = e.norm[i];
    e.cst[10] = e.cst[11] = 0;
    for (int m = 0; m < e.mmax; ++m) e.tm[m] = 0;                         // :164
    for (int j = 0; j < e.jmax; ++j) {                                    // rows behind J: "no job" (todo 0, no op)
        const bool v = j < e.J;
        e.w(j, JSS_F_TODO) = v ? JSS_FLAG_LEGAL : 0;                      // todo 0 (:166), legal (:160), not blocked (:171)
        e.refresh_ops(j, v);                                              // :174-176 needed machine = op 0
        e.w(j, JSS_F_LEFT) = e.w(j, JSS_F_PERF) = e.w(j, JSS_F_IDLE) = e.w(j, JSS_F_IDLE_LAST) = 0;   // :165-170
        e.w(j, JSS_F_F4) = 0;                                             // :180
    }
    for (int i = 0; i < e.jmax * e.stride; ++i) e.sol[i] = -1;            // :163, the whole padded block
}

// ---- increase_time_step(): jss_env.py:495-637; caller guarantees a busy machine ------------------------------
int advance(const Env &e) {
    int d = kBig;                                                         // :517-522 next event = earliest release
    for (int m = 0; m < e.M; ++m)
        if (e.tm[m] > 0 && e.tm[m] < d) d = e.tm[m];
    e.t() += d;
    int hole = 0;
    for (int m = 0; m < e.M; ++m) {                                       // :604-613
        if (e.tm[m] < d) hole += d - e.tm[m];                             // :606-608 (only idle machines: tm == 0)
        e.tm[m] = e.tm[m] > d ? e.tm[m] - d : 0;                          // :611
    }
    for (int j = 0; j < e.J; ++j) {                                       // :525-601
        const int was = e.w(j, JSS_F_LEFT);
        if (was > 0) {                                                    // :529 running
            e.w(j, JSS_F_PERF) += d < was ? d : was;                      // :531, :544
            e.w(j, JSS_F_LEFT) = was > d ? was - d : 0;                   // :534
            if (was <= d) {                                               // :550 op finished
                e.w(j, JSS_F_IDLE) += d - was;                            // :552
                e.w(j, JSS_F_IDLE_LAST) = d - was;                        // :554
                const int k = e.todo(j) + 1;                              // :558
                e.w(j, JSS_F_TODO) = (e.w(j, JSS_F_TODO) & ~JSS_TODO_MASK) | k;
                e.refresh_ops(j, true);                                   // :562-566 the job moves on (-1: complete, :581)
                const int cur = e.cur(j);
                e.w(j, JSS_F_F4) = cur >= 0 ? e.tm[cur >> 16] : JSS_F4_ONE;   // :569-586 (machine clocks already advanced)
            }
        } else if (e.todo(j) < e.M) {                                     // :594 waiting
            e.w(j, JSS_F_IDLE) += d;                                      // :596
            e.w(j, JSS_F_IDLE_LAST) += d;                                 // :597
        }
    }
    for (int j = 0; j < e.J; ++j) {                                       // :616-634 re-legalisation
        const int cur = e.cur(j);
        if (cur >= 0 && e.tm[cur >> 16] == 0 && !e.blocked(j)) e.set_legal(j, true);
    }
    return hole;
}

// ---- _prioritization_non_final(): jss_env.py:183-254 --------------------------------------------------------
void prioritize(const Env &e) {
    bool any_final = false;
    for (int j = 0; j < e.J; ++j) any_final |= e.legal(j) && e.todo(j) == e.M - 1;   // :217
    if (!any_final) return;
    // shortest legal non-final job per machine whose NEXT machine is idle (:219-239)
    int min_nf[JSS_MAX_MACHINES];
    for (int m = 0; m < e.M; ++m) min_nf[m] = kBig;
    for (int j = 0; j < e.J; ++j) {
        if (!e.legal(j) || e.todo(j) >= e.M - 1) continue;
        if (e.tm[e.nxt(j) >> 16] != 0) continue;                // :234
        const int cur = e.cur(j);
        if ((cur & kDurMask) < min_nf[cur >> 16]) min_nf[cur >> 16] = cur & kDurMask;
    }
    for (int j = 0; j < e.J; ++j) {                                       // :244-254
        if (!e.legal(j) || e.todo(j) != e.M - 1) continue;
        const int cur = e.cur(j);
        if ((cur & kDurMask) > min_nf[cur >> 16]) e.set_legal(j, false);
    }
}

// ---- _check_no_op(): jss_env.py:256-401 ----------------------------------------------------------------------
void check_no_op(const Env &e) {
    e.set_noop(0);                                                        // :278
    const int t = e.t();
    int legal_jobs[4], nl = 0;
    for (int j = 0; j < e.J; ++j)
        if (e.legal(j)) {
            if (nl == 4) return;                                          // :287 more than 4 legal jobs
            legal_jobs[nl++] = j;
        }
    if (nl == 0) return;
    int d_next = kBig;                                                    // :285, :293
    for (int m = 0; m < e.M; ++m)
        if (e.tm[m] > 0 && e.tm[m] < d_next) d_next = e.tm[m];
    if (d_next == kBig) return;
    const int next_event = t + d_next;
    // pass 1 (:305-321), in ascending job order: per legal machine the running minimum of the ends
    int horizon[JSS_MAX_MACHINES];                                        // max_horizon_machine; kBig = machine not legal
    bool m_legal[JSS_MAX_MACHINES];
    int n_ml = 0;
    for (int m = 0; m < e.M; ++m) m_legal[m] = false;
    for (int i = 0; i < nl; ++i) {
        const int m = e.cur(legal_jobs[i]) >> 16;
        if (!m_legal[m]) {
            m_legal[m] = true;
            ++n_ml;
        }
    }
    if (n_ml > 3) return;                                                 // :286
    for (int m = 0; m < e.M; ++m) horizon[m] = t + e.max_time_op;         // :300-302
    int max_horizon = t;                                                  // :296
    for (int i = 0; i < nl; ++i) {
        const int cur = e.cur(legal_jobs[i]);
        const int end = t + (cur & kDurMask);                             // :310
        if (end < next_event) return;                                     // :314-315
        if (end < horizon[cur >> 16]) horizon[cur >> 16] = end;           // :318
        if (horizon[cur >> 16] > max_horizon) max_horizon = horizon[cur >> 16];   // :321
    }
    // pass 2 (:324-401): every illegal job looks ahead along its ops
    bool covered[JSS_MAX_MACHINES];
    for (int m = 0; m < e.M; ++m) covered[m] = false;
    for (int j = 0; j < e.J; ++j) {
        if (e.legal(j)) continue;
        const int todo = e.todo(j);
        const bool caseA = e.w(j, JSS_F_LEFT) > 0 && todo + 1 < e.M;      // :327-330 running, has a next op
        const bool caseB = !caseA && !e.blocked(j) && todo < e.M;         // :366-369 waiting for its machine
        if (!caseA && !caseB) continue;
        int k = caseA ? todo + 1 : todo;                                  // :332 / :370
        int tn = caseA ? t + e.w(j, JSS_F_LEFT) : t + e.tm[e.cur(j) >> 16];   // :334-337 / :374-377
        while (k < e.M - 1 && max_horizon > tn) {                         // :340-342 / :380-382
            const int op = k == todo ? e.cur(j)
                           : k == todo + 1 ? e.nxt(j)
                           : k == todo + 2 ? e.next2(j) : e.ops[j * e.stride + k];
            const int m = op >> 16;
            if (m_legal[m] && horizon[m] > tn) covered[m] = true;         // :346-351
            tn += op & kDurMask;                                          // :362
            ++k;
        }
    }
    for (int m = 0; m < e.M; ++m)
        if (m_legal[m] && !covered[m]) return;
    e.set_noop(1);                                                        // :357-359 / :395-397
}

// ---- step(): jss_env.py:403-481; returns the reward numerator ---------------------------------------------------
int step_env(const Env &e, int a) {
    if (a == JSS_ACTION_SKIP) return 0;
    if (a < 0 || a > e.J) {
        e.flag(JSS_ERR_BAD_ACTION);
        return 0;
    }
    int rn = 0;
    if (a == e.J) {                                                       // :419 NOPE
        for (int j = 0; j < e.J; ++j)                                     // :422-428
            if (e.legal(j)) {
                e.set_blocked(j, true);
                e.set_legal(j, false);
            }
        for (;;) {                                                        // :429-430
            if (!any_busy(e)) {                                           // reference: IndexError (:517)
                e.flag(JSS_ERR_NOPE_IDLE);
                break;
            }
            rn -= advance(e);
            if (n_legal(e)) break;
        }
    } else {                                                              // :441 allocate job a
        if (!e.legal(a)) {                                                // outside the mask: ignored + flagged
            e.flag(JSS_ERR_ILLEGAL_ACTION);
            return 0;
        }
        const int cur = e.cur(a);
        const int m = cur >> 16, d = cur & kDurMask;                      // :443-444
        rn = d;                                                           // :445
        e.tm[m] = d;                                                      // :446
        e.w(a, JSS_F_LEFT) = d;                                           // :447
        e.sol[a * e.stride + e.todo(a)] = e.t();                          // :454
        for (int j = 0; j < e.J; ++j) {
            const int cj = e.cur(j);
            if (cj >= 0 && (cj >> 16) == m) {
                e.set_legal(j, false);                                    // :455-463
                e.set_blocked(j, false);                                  // :464-467
            }
        }
        while (!n_legal(e) && any_busy(e)) rn -= advance(e);              // :469-470
    }
    prioritize(e);                                                        // :432 / :471
    check_no_op(e);                                                       // :433 / :472
    return rn;
}

// ---- action selectors ---------------------------------------------------------------------------------------
uint32_t fmix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
uint32_t rng_u32(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t step) {
    const uint32_t a = (uint32_t)seed + (uint32_t)env_id * 0x9E3779B9u + episode * 0x85EBCA6Bu + step * 0xC2B2AE35u;
    const uint32_t b = (uint32_t)(seed >> 32) ^ ((uint32_t)(env_id >> 32) * 0x27D4EB2Fu);
    return fmix32(fmix32(a) ^ b);
}

int select_action(const Env &e, const Call &c, uint64_t env_id) {
    const uint32_t episode = (uint32_t)e.hdr[JSS_H_EPISODE], step = (uint32_t)e.hdr[JSS_H_STEP];
    const int nl = n_legal(e);
    const int n = nl + e.noop();
    if (n == 0) return -1;
    const int kind = c.kind & 0xFF;                                       // bits 8-23: CriticalRatio's due-date factor p / q (0 = 3 / 2)
    const long long cr_p = ((c.kind >> 8) & 0xFF) ? ((c.kind >> 8) & 0xFF) : 3, cr_q = ((c.kind >> 8) & 0xFF) ? ((c.kind >> 16) & 0xFF) : 2;
    if (kind == JSS_POLICY_RANDOM) {                                      // README.md:58-60 uniform over the mask's set bits
        int pick = (int)(((uint64_t)rng_u32(c.seed, env_id, episode, step) * (uint32_t)n) >> 32);
        for (int j = 0; j < e.J; ++j)
            if (e.legal(j) && pick-- == 0) return j;
        return e.J;
    }
    if (nl == 0) return e.J;                                              // only NOPE is legal (dispatching.py:96-97)
    int best = -1;
    long long best_num = 0, best_den = 1;                                 // CR: exact fraction compare
    int best_v = 0;
    const bool cr_f64 = kind == JSS_POLICY_CR && ((c.kind >> 24) & 1);    // JSS_POLICY_CR_F64: the reference's doubles themselves
    double best_ratio = 0.0;
    for (int j = 0; j < e.J; ++j) {
        if (!e.legal(j)) continue;
        const int todo = e.todo(j);
        if (cr_f64) {                                                     // dispatching.py:351-363, :391-399
            volatile double due = (double)e.rem[j * e.stride] * c.d.cr_factor;       // (volatile: each operation rounded on its own)
            volatile double left = due - (double)e.t();
            const double ratio = left / (double)e.rem[j * e.stride + todo];
            if (best < 0 || ratio < best_ratio) {
                best = j;
                best_ratio = ratio;
            }
            continue;
        }
        if (kind == JSS_POLICY_CR) {                                      // dispatching.py:365-408, (p L - q t) / remaining
            const long long num = cr_p * e.rem[j * e.stride] - cr_q * e.t(), den = e.rem[j * e.stride + todo];
            if (best < 0 || num * best_den < best_num * den) {
                best = j;
                best_num = num;
                best_den = den;
            }
            continue;
        }
        int v;
        bool larger = false;
        switch (kind) {
        case JSS_POLICY_FIFO: v = e.w(j, JSS_F_IDLE_LAST); larger = true; break;      // :146
        case JSS_POLICY_SPT: v = e.cur(j) & kDurMask; break;                 // :105-106
        case JSS_POLICY_MWR: v = e.rem[j * e.stride + todo]; larger = true; break;    // :187-189
        case JSS_POLICY_LWR: v = e.rem[j * e.stride + todo]; break;                   // :230-232
        case JSS_POLICY_MOR: v = e.M - todo; larger = true; break;                    // :273
        default: v = e.M - todo; break;                                               // LOR :314
        }
        if (best < 0 || (larger ? v > best_v : v < best_v)) {             // strict: the lowest index wins ties
            best = j;
            best_v = v;
        }
    }
    if (e.noop() && c.explore_q16 != 0) {                                 // dispatching.py:113
        if ((rng_u32(c.seed ^ kExploreSeedXor, env_id, episode, step) >> 16) < c.explore_q16) best = e.J;
    }
    return best;
}

// ---- outputs ------------------------------------------------------------------------------------------------
float as_float(int32_t bits) {
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}
// the kernels' division: quotient estimate with the record's reciprocal, one residual correction (bit-identical)
float div_by(float a, float b, float rb) {
    const float q = a * rb;
    return std::fmaf(std::fmaf(-q, b, a), rb, q);
}

// _reward_scaler (jss_env.py:483-493), evaluated like the kernels do (reciprocal + one residual correction)
float reward_of(const Env &e, int rn) { return div_by((float)rn, (float)e.max_time_op, as_float(e.norm[2])); }

// observation rows < J and the mask row of one env (padding: obs rows behind J are zeroed when `pad`)
void write_obs_mask(const Env &e, int jm, float *obs, uint8_t *mk, bool pad) {
    const float f_op = (float)e.max_time_op, f_jobs = (float)e.norm[0], f_sum = (float)e.norm[1];
    const float f_m = (float)e.M;
    const float r_op = as_float(e.norm[2]), r_jobs = as_float(e.norm[3]);
    const float r_sum = as_float(e.norm[4]), r_m = as_float(e.norm[5]);
    if (obs) {
        for (int j = 0; j < e.J; ++j) {                                   // jss_env.py:102-111
            float *row = obs + j * 7;
            row[0] = e.legal(j) ? 1.f : 0.f;                              // :130
            row[1] = div_by((float)e.w(j, JSS_F_LEFT), f_op, r_op);       // :448, :539
            row[2] = div_by((float)e.todo(j), f_m, r_m);                  // :559
            row[3] = div_by((float)e.w(j, JSS_F_PERF), f_jobs, r_jobs);   // :545
            row[4] = e.w(j, JSS_F_F4) == JSS_F4_ONE ? 1.f : div_by((float)e.w(j, JSS_F_F4), f_op, r_op);   // :569-586
            row[5] = div_by((float)e.w(j, JSS_F_IDLE_LAST), f_sum, r_sum);    // :555, :600
            row[6] = div_by((float)e.w(j, JSS_F_IDLE), f_sum, r_sum);     // :553, :601
        }
        if (pad)
            for (int i = e.J * 7; i < jm * 7; ++i) obs[i] = 0.f;          // padding rows
    }
    if (mk) {
        for (int j = 0; j < e.J; ++j) mk[j] = e.legal(j) ? 1 : 0;
        mk[e.J] = (uint8_t)e.noop();
        for (int j = e.J + 1; j <= jm; ++j) mk[j] = 0;
    }
}

void write_outputs(const Env &e, const Call &c, int b) {
    const int jm = c.d.jmax;
    write_obs_mask(e, jm, c.o.real_obs + (size_t)b * jm * 7, c.o.action_mask + (size_t)b * (jm + 1), true);
}

void add_counters(const Call &c, int b, int steps, int episodes, long long makespans, long long reward_num) {
    if (!c.s.counters) return;
    int64_t *cn = c.s.counters + (size_t)b * 4;
    cn[0] += steps;
    cn[1] += episodes;
    cn[2] += makespans;
    cn[3] += reward_num;
}

uint64_t env_id_of(const Call &c, int b) { return (uint64_t)(c.d.env_ids ? c.d.env_ids[b] : c.d.env_id_base + b); }

enum Mode { kReset, kStep, kAdvance, kPolicy, kRollout, kTraj, kSteps };

void restart(const Env &e, const Call &c, int b) {                       // reset() + the bookkeeping around it
    const int episode = e.hdr[JSS_H_EPISODE];
    reset_env(e);
    e.hdr[JSS_H_EPISODE] = episode + 1;
    e.hdr[JSS_H_STEP] = 0;
    c.o.reward[b] = 0.f;
    c.o.done[b] = 0;
}

// One jss_step call of env b: a = job, J (NOPE), JSS_ACTION_SKIP or JSS_ACTION_RESET.  called = the env was stepped.
void step_call(Env &e, const Call &c, int b, int a, bool &called, int &rn) {
    called = false;
    rn = 0;
    if (a == JSS_ACTION_SKIP) return;                                     // untouched: reward / done / makespan stay
    if (a == JSS_ACTION_RESET) {                                          // reset() instead of a step; the env may have been
        e = env_of(c, b, true);                                           // given another instance since (table_of_env)
        restart(e, c, b);
        return;
    }
    rn = step_env(e, a);
    called = true;
    const bool done = n_legal(e) == 0;                                    // :639-653
    e.hdr[JSS_H_STEP] += 1;
    c.o.reward[b] = reward_of(e, rn);                                     // :483-493
    c.o.done[b] = done ? 1 : 0;
    if (done) c.o.makespan[b] = e.t();                                    // :650
    add_counters(c, b, 1, done ? 1 : 0, done ? e.t() : 0, rn);
}

void run_env(const Call &c, int mode, int b) {
    if (mode == kReset) {
        if (c.which && !c.which[b]) return;
        const Env e = env_of(c, b, true);
        restart(e, c, b);
        write_outputs(e, c, b);
        store_compact(e);
        return;
    }
    Env e = env_of(c, b, false);
    if (e.J == 0) return;                                                 // never reset: nothing to step
    switch (mode) {
    case kStep: {
        bool called;
        int rn;
        // jss_step_autoreset: an env that reported done on the previous call is reset instead of stepped
        const int a = ((c.flags & JSS_ROLLOUT_AUTORESET) && c.o.done[b]) ? JSS_ACTION_RESET : c.actions[b];
        step_call(e, c, b, a, called, rn);
        break;
    }
    case kSteps: {                                                        // n_iter x kStep, actions [n_iter][B], every step optionally recorded
        const int jm = c.d.jmax;
        for (int it = 0; it < c.n_iter; ++it) {
            const size_t slot = (size_t)it * (c.t.stride ? (size_t)c.t.stride : (size_t)c.d.batch) + b;
            bool called;
            int rn;
            step_call(e, c, b, c.actions[slot], called, rn);
            write_obs_mask(e, jm, c.t.real_obs ? c.t.real_obs + slot * jm * 7 : nullptr,
                           c.t.action_mask ? c.t.action_mask + slot * (jm + 1) : nullptr, false);
            if (c.t.reward) c.t.reward[slot] = called ? reward_of(e, rn) : 0.f;
            if (c.t.done) c.t.done[slot] = n_legal(e) == 0 ? 1 : 0;
        }
        break;
    }
    case kAdvance:
        if (c.which && !c.which[b]) return;
        {
            int hole = 0;
            if (!any_busy(e)) e.flag(JSS_ERR_NOPE_IDLE);                  // reference: IndexError (:517)
            else hole = advance(e);
            if (c.hole) c.hole[b] = hole;
        }
        break;
    case kPolicy:
        c.actions_out[b] = select_action(e, c, env_id_of(c, b));
        return;                                                           // no outputs rewritten
    default: {                                                            // n_iter x (policy + step), dispatching.py:55-75
        const uint64_t env_id = env_id_of(c, b);                          // kTraj: every iteration recorded (JssTraj)
        const int jm = c.d.jmax;
        const bool autoreset = (c.flags & JSS_ROLLOUT_AUTORESET) != 0;
        int n_steps = 0, n_done = 0, last_rn = 0, last_makespan = -1;
        long long sum_makespan = 0, sum_rn = 0;
        for (int it = 0; it < c.n_iter; ++it) {
            const size_t slot = (size_t)it * (c.t.stride ? (size_t)c.t.stride : (size_t)c.d.batch) + b;
            if (mode == kTraj)                                            // what the policy sees in this slot
                write_obs_mask(e, jm, c.t.real_obs ? c.t.real_obs + slot * jm * 7 : nullptr,
                               c.t.action_mask ? c.t.action_mask + slot * (jm + 1) : nullptr, false);
            if (n_legal(e) == 0) {                                        // done
                if (mode == kTraj) {
                    if (c.t.action) c.t.action[slot] = autoreset ? JSS_ACTION_RESET : JSS_ACTION_SKIP;
                    if (c.t.reward) c.t.reward[slot] = 0.f;
                    if (c.t.done) c.t.done[slot] = autoreset ? 0 : 1;
                }
                if (!autoreset) {
                    if (mode == kTraj) continue;                          // frozen: every remaining slot says so
                    break;
                }
                const int episode = e.hdr[JSS_H_EPISODE];
                reset_env(e);
                e.hdr[JSS_H_EPISODE] = episode + 1;
                e.hdr[JSS_H_STEP] = 0;
                continue;
            }
            const int a = select_action(e, c, env_id);
            last_rn = step_env(e, a);
            e.hdr[JSS_H_STEP] += 1;
            n_steps += 1;
            sum_rn += last_rn;
            const bool done = n_legal(e) == 0;
            if (done) {
                n_done += 1;
                sum_makespan += e.t();
                last_makespan = e.t();
            }
            if (mode == kTraj) {
                if (c.t.action) c.t.action[slot] = a;
                if (c.t.reward) c.t.reward[slot] = reward_of(e, last_rn);
                if (c.t.done) c.t.done[slot] = done ? 1 : 0;
            }
        }
        if (n_steps) c.o.reward[b] = reward_of(e, last_rn);
        c.o.done[b] = n_legal(e) == 0 ? 1 : 0;
        if (last_makespan >= 0) c.o.makespan[b] = last_makespan;
        add_counters(c, b, n_steps, n_done, sum_makespan, sum_rn);
        break;
    }
    }
    write_outputs(e, c, b);
    store_compact(e);
}

int run(const Call &c, int mode) {
    const int B = c.d.batch;
#ifdef _OPENMP
    if (c.d.threads > 0) {
#pragma omp parallel for schedule(static) num_threads(c.d.threads)
        for (int b = 0; b < B; ++b) run_env(c, mode, b);
    } else {
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; ++b) run_env(c, mode, b);
    }
#else
    for (int b = 0; b < B; ++b) run_env(c, mode, b);
#endif
    return 0;
}

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
    if (d->record_ints == JSS_NFC && d->n_tables != 1) return JSS_E_SHAPE;
    if (d->record_ints == JSS_NFM && (d->mmax > 32 || d->n_tables == 1)) return JSS_E_SHAPE;
    return 0;
}

// f64_ok: the calls whose policy is a launch of its own on the GPU (JSS_POLICY_CR_F64, include/jss_hip.h): same answers here
int check_kind(const JssDesc *d, int kind_arg, bool f64_ok = false) {
    const int kind = kind_arg & 0xFF, fp = (kind_arg >> 8) & 0xFF, fq = (kind_arg >> 16) & 0xFF;
    if (kind_arg < 0 || (kind_arg >> 25) || kind >= JSS_N_POLICIES) return JSS_E_KIND;
    if ((kind_arg >> 24) & 1) {
        if (!f64_ok || kind != JSS_POLICY_CR || fp || fq || !(d->cr_factor > 0.0) || !(d->cr_factor < 1e300)) return JSS_E_KIND;
    }
    if (fp || fq) {                                  // a due-date factor p / q: CriticalRatio only, q a power of two <= 64
        if (kind != JSS_POLICY_CR || fp < 1 || fq < 1 || fq > 64 || (fq & (fq - 1))) return JSS_E_KIND;
    }
    if ((kind == JSS_POLICY_MWR || kind == JSS_POLICY_LWR || kind == JSS_POLICY_CR) && !d->rem) return JSS_E_NULL;
    return 0;
}

}  // namespace

extern "C" {

int jss_abi_version(void) { return JSS_ABI_VERSION; }

const char *jss_backend(void) {
#ifdef _OPENMP
    return "cpu:openmp";
#else
    return "cpu:serial";
#endif
}

const char *jss_error_string(int code) {
    switch (code) {
    case 0: return "ok";
    case JSS_E_NULL: return "null pointer in JssDesc/JssState/JssOut or arguments";
    case JSS_E_SHAPE: return "bad shape (batch/jmax/mmax/n_tables/n_sub)";
    case JSS_E_KIND: return "unknown policy kind or kernel flavour";
    case JSS_E_LDS: return "batch shape needs more LDS per workgroup than the device provides";
    case JSS_E_RESIDENT: return "the batch does not fit the chip as one round of resident workgroups (step session)";
    case JSS_E_SESSION: return "step session: bad step range (mailbox ring overrun, or the session was never opened)";
    default: return "unknown error";
    }
}

int jss_reset(const JssDesc *desc, const JssState *state, const JssOut *out, const uint8_t *which, void *) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    Call c;
    c.d = *desc; c.s = *state; c.o = *out; c.which = which;
    return run(c, kReset);
}

int jss_step(const JssDesc *desc, const JssState *state, const int32_t *actions, const JssOut *out, void *) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (!actions) return JSS_E_NULL;
    Call c;
    c.d = *desc; c.s = *state; c.o = *out; c.actions = actions;
    return run(c, kStep);
}

int jss_step_autoreset(const JssDesc *desc, const JssState *state, const int32_t *actions, const JssOut *out, void *) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (!actions) return JSS_E_NULL;
    Call c;
    c.d = *desc; c.s = *state; c.o = *out; c.actions = actions; c.flags = JSS_ROLLOUT_AUTORESET;
    return run(c, kStep);
}

int jss_advance(const JssDesc *desc, const JssState *state, const uint8_t *which, int32_t *hole, const JssOut *out, void *) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    Call c;
    c.d = *desc; c.s = *state; c.o = *out; c.which = which; c.hole = hole;
    return run(c, kAdvance);
}

int jss_policy(const JssDesc *desc, const JssState *state, int kind, uint64_t seed, uint32_t explore_q16, int32_t *actions,
               void *) {
    int rc = check_args(desc, state, nullptr, false);
    if (rc) return rc;
    if (!actions) return JSS_E_NULL;
    if ((rc = check_kind(desc, kind, true))) return rc;
    Call c;
    c.d = *desc; c.s = *state; c.o = JssOut(); c.actions_out = actions; c.kind = kind; c.seed = seed; c.explore_q16 = explore_q16;
    return run(c, kPolicy);
}

int jss_rollout(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed, uint32_t explore_q16,
                int32_t n_iter, int32_t flags, void *) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if ((rc = check_kind(desc, kind))) return rc;
    if (n_iter < 0) return JSS_E_SHAPE;
    Call c;
    c.d = *desc; c.s = *state; c.o = *out; c.kind = kind; c.seed = seed; c.explore_q16 = explore_q16;
    c.n_iter = n_iter; c.flags = flags;
    return run(c, kRollout);
}

int jss_trajectory(const JssDesc *desc, const JssState *state, const JssOut *out, const JssTraj *traj, int kind,
                   uint64_t seed, uint32_t explore_q16, int32_t n_steps, int32_t flags, void *) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (!traj) return JSS_E_NULL;
    if ((rc = check_kind(desc, kind))) return rc;
    if (n_steps < 0) return JSS_E_SHAPE;
    Call c;
    c.d = *desc; c.s = *state; c.o = *out; c.t = *traj; c.kind = kind; c.seed = seed; c.explore_q16 = explore_q16;
    c.n_iter = n_steps; c.flags = flags;
    return run(c, kTraj);
}

int jss_steps(const JssDesc *desc, const JssState *state, const JssOut *out, const JssTraj *traj, const int32_t *actions,
              int32_t n_steps, void *) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (n_steps < 0) return JSS_E_SHAPE;
    if (n_steps == 0) return 0;                       // nothing to do (an empty action buffer has no address)
    if (!actions) return JSS_E_NULL;
    Call c;
    c.d = *desc; c.s = *state; c.o = *out; c.actions = actions; c.n_iter = n_steps;
    if (traj) c.t = *traj;
    c.t.action = nullptr;
    return run(c, kSteps);
}

// Step session on the host cores: nothing is resident between calls here (the "chip" is the cache hierarchy), so an open
// session is a record of what it steps; post executes its steps on the spot -- the same jss_step semantics, one step
// after the other -- and publishes the mailbox and progress words the way the device does; wait and close have nothing
// left to wait for.
struct HostSession {
    JssDesc d;
    JssState s;
    JssOut o;
    int next_step;
};
static std::mutex g_sessions_mutex;
static std::unordered_map<const void *, HostSession> g_sessions;

int jss_session_open(const JssDesc *desc, const JssState *state, const JssOut *out, const JssSession *session, void *) {
    int rc = check_args(desc, state, out, true);
    if (rc) return rc;
    if (!session || !session->mail || !session->progress || !session->status) return JSS_E_NULL;
    if (session->depth < 1 || session->timeout_ms < 0 || desc->batch < 1) return JSS_E_SHAPE;
    const int want = session->slots;
    if (want != 0 && want != 1 && want != 2 && want != 4 && want != 8) return JSS_E_SHAPE;
    std::lock_guard<std::mutex> lock(g_sessions_mutex);
    g_sessions[session->progress] = HostSession{*desc, *state, *out, 0};
    session->status[3] = want ? want : 1;
    return 0;
}

int jss_session_post(const JssDesc *desc, const JssSession *session, const int32_t *actions, int32_t first_step,
                     int32_t n_steps, int32_t waited, void *) {
    if (!desc || !session || !session->mail || !actions) return JSS_E_NULL;
    if (first_step < 0 || n_steps < 1 || waited < 0 || waited > first_step || first_step + n_steps - waited > session->depth)
        return JSS_E_SESSION;
    HostSession hs;
    {
        std::lock_guard<std::mutex> lock(g_sessions_mutex);
        const auto it = g_sessions.find(session->progress);
        if (it == g_sessions.end() || it->second.next_step != first_step) return JSS_E_SESSION;   // not open / not the next step
        it->second.next_step += n_steps;
        hs = it->second;
    }
    const size_t B = (size_t)desc->batch;
    for (int k = 0; k < n_steps; ++k) {
        const int step = first_step + k;
        for (size_t i = 0; i < B; ++i)
            session->mail[(size_t)(step % session->depth) * B + i] =
                ((uint64_t)(uint32_t)(step + 1) << 32) | (uint32_t)actions[(size_t)k * B + i];
        const int rc = jss_step(&hs.d, &hs.s, actions + (size_t)k * B, &hs.o, nullptr);
        if (rc) return rc;
        for (size_t i = 0; i < B; ++i) session->progress[i] = step + 1;
    }
    return 0;
}

int jss_session_wait(const JssDesc *desc, const JssSession *session, int32_t steps_done, void *) {
    if (!desc || !session || !session->progress || !session->status) return JSS_E_NULL;
    if (steps_done < 0) return JSS_E_SESSION;
    std::lock_guard<std::mutex> lock(g_sessions_mutex);
    const auto it = g_sessions.find(session->progress);
    if (it == g_sessions.end()) return JSS_E_SESSION;
    if (it->second.next_step < steps_done) session->status[1] += 1;       // would never arrive: report it like a timed-out wait
    return 0;
}

int jss_session_step(const JssDesc *desc, const JssSession *session, const int32_t *actions, int32_t step, void *) {
    return jss_session_post(desc, session, actions, step, 1, step, nullptr);
}

int jss_session_close(const JssDesc *desc, const JssSession *session, int32_t next_step, void *) {
    if (!desc || !session || !session->mail) return JSS_E_NULL;
    if (next_step < 0) return JSS_E_SESSION;
    std::lock_guard<std::mutex> lock(g_sessions_mutex);
    g_sessions.erase(session->progress);
    session->status[2] += 1;
    return 0;
}

int jss_sync_check(void *) { return 0; }                                  // every call of this library is synchronous

// envs are independent and the call is synchronous: n_steps one-step rollouts of every env ARE one n_steps-iteration
// rollout per env; sub-batches and streams have nothing to overlap here
int jss_rollout_steps(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                      uint32_t explore_q16, int32_t n_steps, int32_t flags, int32_t n_sub, void *const *streams) {
    if (n_steps < 0 || n_sub < 1 || n_sub > 16) return desc && state ? JSS_E_SHAPE : JSS_E_NULL;
    if (!streams) return JSS_E_NULL;
    if (n_steps == 0) {                               // no step: nothing is touched (the HIP library launches nothing); arguments checked
        const int rc = check_args(desc, state, out, true);
        return rc ? rc : check_kind(desc, kind);
    }
    return jss_rollout(desc, state, out, kind, seed, explore_q16, n_steps, flags, nullptr);
}

// the un-fused loop: envs are independent and every call is synchronous, so sub-batches and streams have nothing to overlap
int jss_policy_step_steps(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                          uint32_t explore_q16, int32_t *actions, int32_t n_steps, int32_t flags, int32_t n_sub,
                          void *const *streams) {
    if (n_steps < 0 || n_sub < 1 || n_sub > 16) return desc && state ? JSS_E_SHAPE : JSS_E_NULL;
    if (!streams || !actions) return JSS_E_NULL;
    for (int s = 0; s < n_steps; ++s) {
        int rc = jss_policy(desc, state, kind, seed, explore_q16, actions, nullptr);
        if (!rc) rc = (flags & JSS_ROLLOUT_AUTORESET) ? jss_step_autoreset(desc, state, actions, out, nullptr)
                                                     : jss_step(desc, state, actions, out, nullptr);
        if (rc) return rc;
    }
    return 0;
}

// several independent env sets: each is its own synchronous rollout here
int jss_rollout_steps_multi(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states,
                            const JssOut *const *outs, int kind, uint64_t seed, uint32_t explore_q16, int32_t n_steps,
                            int32_t flags, void *const *streams) {
    if (!descs || !states || !outs || !streams) return JSS_E_NULL;
    if (n_sets < 1 || n_sets > 16 || n_steps < 0) return JSS_E_SHAPE;
    for (int i = 0; i < n_sets; ++i) {
        int rc = check_args(descs[i], states[i], outs[i], true);
        if (!rc) rc = check_kind(descs[i], kind);
        if (!rc && n_steps > 0)                       // (n_steps == 0: nothing is touched, like the HIP library, which launches nothing)
            rc = jss_rollout(descs[i], states[i], outs[i], kind, seed, explore_q16, n_steps, flags & JSS_ROLLOUT_AUTORESET, nullptr);
        if (rc) return rc;
    }
    return 0;
}

// several env sets per call (include/jss_hip.h jss_multi_*): the host twin has no launches to fuse -- set after set
int jss_multi_reset(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const JssOut *const *outs,
                    const uint8_t *const *which, void *stream) {
    if (!descs || !states || !outs) return JSS_E_NULL;
    if (n_sets < 1 || n_sets > 16) return JSS_E_SHAPE;
    for (int i = 0; i < n_sets; ++i) {
        const int rc = jss_reset(descs[i], states[i], outs[i], which ? which[i] : nullptr, stream);
        if (rc) return rc;
    }
    return 0;
}

int jss_multi_step(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const int32_t *const *actions,
                   const JssOut *const *outs, int32_t flags, void *stream) {
    if (!descs || !states || !outs || !actions) return JSS_E_NULL;
    if (n_sets < 1 || n_sets > 16) return JSS_E_SHAPE;
    for (int i = 0; i < n_sets; ++i) {
        const int rc = (flags & JSS_ROLLOUT_AUTORESET) ? jss_step_autoreset(descs[i], states[i], actions[i], outs[i], stream)
                                                       : jss_step(descs[i], states[i], actions[i], outs[i], stream);
        if (rc) return rc;
    }
    return 0;
}

int jss_multi_policy(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, int kind, uint64_t seed,
                     uint32_t explore_q16, int32_t *const *actions, void *stream) {
    if (!descs || !states || !actions) return JSS_E_NULL;
    if (n_sets < 1 || n_sets > 16) return JSS_E_SHAPE;
    for (int i = 0; i < n_sets; ++i) {
        const int rc = jss_policy(descs[i], states[i], kind, seed, explore_q16, actions[i], stream);
        if (rc) return rc;
    }
    return 0;
}

int jss_multi_rollout(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const JssOut *const *outs,
                      int kind, uint64_t seed, uint32_t explore_q16, int32_t n_steps, int32_t flags, int32_t n_sub,
                      void *const *streams) {
    if (!descs || !states || !outs || !streams) return JSS_E_NULL;
    if (n_sets < 1 || n_sets > 16 || n_steps < 0 || n_sub < 1 || n_sub > 16) return JSS_E_SHAPE;
    void *stream = nullptr;
    for (int i = 0; i < n_sets; ++i) {       // n_steps x rollout(n_iter = 1) == rollout(n_iter = n_steps) on the state; `out` holds the last step either way
        int rc = check_args(descs[i], states[i], outs[i], true);
        if (!rc) rc = check_kind(descs[i], kind);
        if (!rc && n_steps > 0)                       // (n_steps == 0: nothing is touched, like the HIP library, which launches nothing)
            rc = jss_rollout(descs[i], states[i], outs[i], kind, seed, explore_q16, n_steps, flags & JSS_ROLLOUT_AUTORESET, stream);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
