// tests/san/twin_driver.cpp -- TEST INFRASTRUCTURE.  Drives libjss_cpu's entry points (jssenv_amd/csrc/jss_cpu.cpp is
// compiled INTO this executable with -fsanitize=address,undefined) through the C ABI of include/jss_hip.h on random
// instances: full-size shapes, edge shapes, forced NOPEs against the mask, out-of-range actions, partial resets,
// the trajectory recorder.  Any out-of-bounds access, use-after-free, signed overflow or misaligned access in the
// twin aborts the run; the driver itself checks the invariants that need no oracle (every op scheduled exactly
// once, machines never overlap, done <=> nothing legal).  Built and run by tests/test_sanitizers.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jss_hip.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

struct Batch {
    int B, J, M;
    std::vector<int32_t> ops, rem, inst, env, envc, job, machine, solution, makespan, actions, hole;
    std::vector<int64_t> counters;
    std::vector<float> obs, reward;
    std::vector<uint8_t> mask, done, which;
    JssDesc d;
    JssState s;
    JssOut o;
};

static float rcp(int v) { return 1.0f / (float)v; }
static int32_t bits(float f) {
    int32_t i;
    std::memcpy(&i, &f, 4);
    return i;
}

static void build(Batch &b, int B, int J, int M, int max_dur, bool per_env_tables) {
    b.B = B; b.J = J; b.M = M;
    const int T = per_env_tables ? B : 1;
    b.ops.assign((size_t)T * J * M, 0);
    b.rem.assign((size_t)T * J * M, 0);
    b.inst.assign((size_t)T * JSS_NI, 0);
    for (int t = 0; t < T; ++t) {
        int max_op = 0, max_job = 0, sum = 0;
        for (int j = 0; j < J; ++j) {
            std::vector<int> perm(M);
            for (int m = 0; m < M; ++m) perm[m] = m;
            for (int m = M - 1; m > 0; --m) std::swap(perm[m], perm[rnd() % (m + 1)]);
            int len = 0;
            for (int k = 0; k < M; ++k) {
                const int dur = 1 + (int)(rnd() % max_dur);
                b.ops[((size_t)t * J + j) * M + k] = (perm[k] << 16) | dur;
                len += dur;
                if (dur > max_op) max_op = dur;
            }
            int suffix = 0;
            for (int k = M - 1; k >= 0; --k) {
                suffix += b.ops[((size_t)t * J + j) * M + k] & 0xffff;
                b.rem[((size_t)t * J + j) * M + k] = suffix;
            }
            sum += len;
            if (len > max_job) max_job = len;
        }
        int32_t *r = &b.inst[(size_t)t * JSS_NI];
        r[JSS_I_JOBS] = J; r[JSS_I_MACHINES] = M; r[JSS_I_MAX_TIME_OP] = max_op; r[JSS_I_MAX_TIME_JOBS] = max_job;
        r[JSS_I_SUM_OP] = sum;
        r[JSS_I_RCP_MAX_TIME_OP] = bits(rcp(max_op)); r[JSS_I_RCP_MAX_TIME_JOBS] = bits(rcp(max_job));
        r[JSS_I_RCP_SUM_OP] = bits(rcp(sum)); r[JSS_I_RCP_MACHINES] = bits(rcp(M));
    }
    b.env.assign((size_t)B * JSS_NH, 0); b.envc.assign((size_t)B * JSS_NC, 0);
    b.job.assign((size_t)B * J * JSS_NF, 0); b.machine.assign((size_t)B * M, 0);
    b.solution.assign((size_t)B * J * M, 0); b.counters.assign((size_t)B * 4, 0);
    b.obs.assign((size_t)B * J * 7, 0.f); b.mask.assign((size_t)B * (J + 1), 0);
    b.reward.assign(B, 0.f); b.done.assign(B, 0); b.makespan.assign(B, 0);
    b.actions.assign(B, 0); b.hole.assign(B, 0); b.which.assign(B, 0);
    std::memset(&b.d, 0, sizeof b.d);
    b.d.batch = B; b.d.jmax = J; b.d.mmax = M; b.d.n_tables = T;
    b.d.ops = b.ops.data(); b.d.rem = b.rem.data(); b.d.inst = b.inst.data();
    b.d.env_id_base = 1000; b.d.threads = 2; b.d.jmin = J;
    b.s = JssState{b.env.data(), b.envc.data(), b.job.data(), b.machine.data(), b.solution.data(), b.counters.data()};
    b.o = JssOut{b.obs.data(), b.mask.data(), b.reward.data(), b.done.data(), b.makespan.data()};
}

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); std::exit(2); } } while (0)

static void check_finished_schedule(const Batch &b, int i) {
    const int J = b.J, M = b.M;
    const int t = b.d.n_tables == 1 ? 0 : i;
    std::vector<std::vector<std::pair<int, int>>> per_machine(M);
    int makespan = 0;
    for (int j = 0; j < J; ++j) {
        int prev_end = 0;
        for (int k = 0; k < M; ++k) {
            const int start = b.solution[((size_t)i * J + j) * M + k];
            const int op = b.ops[((size_t)t * J + j) * M + k];
            CHECK(start >= prev_end);                       // ops of a job in order, every one scheduled
            prev_end = start + (op & 0xffff);
            per_machine[op >> 16].push_back({start, prev_end});
            if (prev_end > makespan) makespan = prev_end;
        }
    }
    for (int m = 0; m < M; ++m)
        for (size_t a = 0; a < per_machine[m].size(); ++a)
            for (size_t c = a + 1; c < per_machine[m].size(); ++c)
                CHECK(per_machine[m][a].second <= per_machine[m][c].first || per_machine[m][c].second <= per_machine[m][a].first);
    CHECK(makespan == b.makespan[i] && makespan == b.env[(size_t)i * JSS_NH + JSS_H_CLOCK]);
}

static void run_shape(int B, int J, int M, int max_dur, bool per_env) {
    Batch b;
    build(b, B, J, M, max_dur, per_env);
    CHECK(jss_reset(&b.d, &b.s, &b.o, nullptr, nullptr) == 0);
    // one random episode per env, frozen at the end
    CHECK(jss_rollout(&b.d, &b.s, &b.o, JSS_POLICY_RANDOM, 7, 0, 4 * J * M + 64, 0, nullptr) == 0);
    for (int i = 0; i < B; ++i) {
        CHECK(b.done[i] == 1);
        check_finished_schedule(b, i);
        CHECK((b.env[(size_t)i * JSS_NH + JSS_H_STATUS] & 0xff) == 0);
    }
    // every rule, auto-restart, through rollout, rollout_steps and the trajectory recorder
    void *streams[2] = {nullptr, nullptr};
    for (int kind = 0; kind < JSS_N_POLICIES; ++kind) {
        CHECK(jss_reset(&b.d, &b.s, &b.o, nullptr, nullptr) == 0);
        CHECK(jss_rollout(&b.d, &b.s, &b.o, kind, 3, 6554, J * M + 9, JSS_ROLLOUT_AUTORESET, nullptr) == 0);
        CHECK(jss_rollout_steps(&b.d, &b.s, &b.o, kind, 3, 6554, 5, JSS_ROLLOUT_AUTORESET, 2, streams) == 0);
        const int K = 9;
        std::vector<float> tobs((size_t)K * B * J * 7), trew((size_t)K * B);
        std::vector<uint8_t> tmask((size_t)K * B * (J + 1)), tdone((size_t)K * B);
        std::vector<int32_t> tact((size_t)K * B);
        JssTraj tr{tobs.data(), tmask.data(), tact.data(), trew.data(), tdone.data()};
        CHECK(jss_trajectory(&b.d, &b.s, &b.o, &tr, kind, 3, 6554, K, JSS_ROLLOUT_AUTORESET, nullptr) == 0);
        for (size_t x = 0; x < tact.size(); ++x) CHECK(tact[x] >= JSS_ACTION_RESET && tact[x] <= J);
        JssTraj none{nullptr, nullptr, nullptr, nullptr, nullptr};
        CHECK(jss_trajectory(&b.d, &b.s, &b.o, &none, kind, 3, 0, 3, 0, nullptr) == 0);
    }
    // step() with hostile actions: forced NOPEs against the mask, out-of-range, illegal jobs, skips, resets
    CHECK(jss_reset(&b.d, &b.s, &b.o, nullptr, nullptr) == 0);
    for (int it = 0; it < 6 * J * M; ++it) {
        CHECK(jss_policy(&b.d, &b.s, JSS_POLICY_RANDOM, 11, 0, b.actions.data(), nullptr) == 0);
        for (int i = 0; i < B; ++i) {
            const uint32_t r = rnd() % 16;
            if (r == 0) b.actions[i] = J;                    // NOPE whether or not the mask allows it
            else if (r == 1) b.actions[i] = (int)(rnd() % (J + 4)) - 2;   // anything in [-2, J + 1]
            else if (r == 2) b.actions[i] = J + 1 + (int)(rnd() % 1000);
            else if (b.done[i]) b.actions[i] = JSS_ACTION_RESET;
        }
        CHECK(jss_step(&b.d, &b.s, b.actions.data(), &b.o, nullptr) == 0);
        if (it % 7 == 0) {
            for (int i = 0; i < B; ++i) b.which[i] = rnd() % 3 == 0;
            CHECK(jss_advance(&b.d, &b.s, b.which.data(), b.hole.data(), &b.o, nullptr) == 0);
        }
        if (it % 31 == 0) {
            for (int i = 0; i < B; ++i) b.which[i] = rnd() % 5 == 0;
            CHECK(jss_reset(&b.d, &b.s, &b.o, b.which.data(), nullptr) == 0);
        }
        for (int i = 0; i < B; ++i) {
            bool any = false;
            for (int j = 0; j < J; ++j) any = any || b.mask[(size_t)i * (J + 1) + j];
            for (int j = 0; j < J * 7; ++j) CHECK(b.obs[(size_t)i * J * 7 + j] >= 0.f && b.obs[(size_t)i * J * 7 + j] <= 1.f);
            (void)any;
        }
    }
    // the external-action forms (ABI v8): K steps per call with hostile actions, recorded; the vector-env step with the
    // auto-reset folded in; a step session (on the host cores: post executes its steps on the spot)
    {
        const int K = 13;
        std::vector<int32_t> acts((size_t)K * B);
        for (size_t x = 0; x < acts.size(); ++x) acts[x] = (int)(rnd() % (J + 5)) - 3;       // [-3, J + 1]: CLOSE is "bad action" here
        std::vector<float> tobs((size_t)K * B * J * 7), trew((size_t)K * B);
        std::vector<uint8_t> tmask((size_t)K * B * (J + 1)), tdone((size_t)K * B);
        JssTraj tr{tobs.data(), tmask.data(), nullptr, trew.data(), tdone.data()};
        CHECK(jss_steps(&b.d, &b.s, &b.o, &tr, acts.data(), K, nullptr) == 0);
        CHECK(jss_steps(&b.d, &b.s, &b.o, nullptr, acts.data(), K, nullptr) == 0);
        CHECK(jss_steps(&b.d, &b.s, &b.o, nullptr, nullptr, 0, nullptr) == 0);
        for (int it = 0; it < 40; ++it) {
            CHECK(jss_policy(&b.d, &b.s, JSS_POLICY_SPT, 5, 0, b.actions.data(), nullptr) == 0);
            CHECK(jss_step_autoreset(&b.d, &b.s, b.actions.data(), &b.o, nullptr) == 0);
        }
        std::vector<uint64_t> mail((size_t)4 * B, 0);
        std::vector<int32_t> progress(B, 0), status(4, 0);
        JssSession ss{mail.data(), progress.data(), status.data(), 4, 0, 0, 0};
        CHECK(jss_session_open(&b.d, &b.s, &b.o, &ss, nullptr) == 0);
        CHECK(jss_session_post(&b.d, &ss, acts.data(), 0, 4, 0, nullptr) == 0);
        CHECK(jss_session_post(&b.d, &ss, acts.data(), 4, 1, 0, nullptr) == JSS_E_SESSION);     // ring overrun
        CHECK(jss_session_wait(&b.d, &ss, 4, nullptr) == 0);
        CHECK(jss_session_step(&b.d, &ss, acts.data() + (size_t)4 * B, 4, nullptr) == 0);
        CHECK(jss_session_close(&b.d, &ss, 5, nullptr) == 0);
        CHECK(progress[0] == 5 && status[1] == 0);
        CHECK(jss_policy(&b.d, &b.s, JSS_POLICY_CR_FACTOR(5, 4), 0, 0, b.actions.data(), nullptr) == 0 || b.d.rem == nullptr);
    }
    CHECK(jss_sync_check(nullptr) == 0);
    std::printf("shape %dx%d x %d envs (%s tables): ok\n", J, M, B, per_env ? "per-env" : "shared");
}

int main() {
    CHECK(jss_abi_version() == JSS_ABI_VERSION);
    run_shape(24, 15, 15, 99, false);      // ta01-shaped, one shared table
    run_shape(6, 100, 20, 99, true);       // ta80-shaped, one table per env
    run_shape(9, 1, 2, 5, true);           // edge shapes
    run_shape(9, 2, 2, 3, false);
    run_shape(5, 3, 7, 9, true);
    run_shape(2, 128, 64, 999, true);      // the ABI's limits
    run_shape(16, 4, 3, 2, true);          // tiny durations: many simultaneous events
    // argument errors must be reported, not crash
    Batch b;
    build(b, 2, 3, 3, 5, false);
    JssState bad = b.s;
    bad.env_const = nullptr;
    CHECK(jss_reset(&b.d, &bad, &b.o, nullptr, nullptr) == JSS_E_NULL);
    CHECK(jss_policy(&b.d, &b.s, 99, 0, 0, b.actions.data(), nullptr) == JSS_E_KIND);
    CHECK(jss_rollout(&b.d, &b.s, &b.o, 0, 0, 0, -1, 0, nullptr) == JSS_E_SHAPE);
    CHECK(jss_policy(&b.d, &b.s, JSS_POLICY_CR_FACTOR(3, 5), 0, 0, b.actions.data(), nullptr) == JSS_E_KIND);      // q not a power of two
    CHECK(jss_policy(&b.d, &b.s, JSS_POLICY_SPT | (2 << 8) | (1 << 16), 0, 0, b.actions.data(), nullptr) == JSS_E_KIND);   // a factor on another rule
    std::printf("SANITIZED-TWIN-OK\n");
    return 0;
}
