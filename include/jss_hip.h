/*
 * include/jss_hip.h -- C ABI of libjss_hip.so, the MI355X (gfx950) batched
 * Job-Shop-Scheduling simulator, and of libjss_cpu.so, its host-core twin
 * (identical symbols and layouts; pointers are host pointers there and `stream`
 * is ignored).
 *
 * The reference (prosysscience/JSSEnv v1.1.0) is pure Python with no FFI; the
 * boundary this library replaces is the method surface of
 * JSSEnv/envs/jss_env.py::JssEnv, lifted over a batch axis:
 *
 *   jss_reset    <- JssEnv.reset()                        jss_env.py:145-181
 *   jss_step     <- JssEnv.step(action)                   jss_env.py:403-481
 *                   (+ _prioritization_non_final :183-254, _check_no_op :256-401,
 *                      increase_time_step :495-637, _get_current_state_representation
 *                      :121-134, _reward_scaler :483-493, _is_done :639-653)
 *   jss_advance  <- JssEnv.increase_time_step()           jss_env.py:495-637
 *                   (public; the reference's tests call it directly,
 *                    tests/test_solutions.py:66)
 *   jss_policy   <- the action selectors callers put in front of step():
 *                   README.md:58-60 (random masked), JSSEnv/dispatching.py
 *                   FIFO :133-156, SPT :92-116, MWR :173-199, LWR :216-242,
 *                   MOR :259-283, LOR :300-324, CR :365-408
 *   jss_rollout  <- DispatchingRule.run_episode's loop    dispatching.py:55-75
 *                   (policy + step fused, n iterations per launch, optional
 *                    auto-restart of finished episodes)
 *   jss_trajectory <- the same loop with EVERY step's transition written out, step-major:
 *                   what the policy saw (observation, mask), the action it took, the
 *                   reward and done flag it got -- the (s, a, r, d) stream a behaviour
 *                   policy (random, a dispatching rule) collects for a learner, K steps
 *                   per launch with the env state held in registers in between
 *   jss_steps    <- K consecutive JssEnv.step(action) calls per env with the K actions given up front ([K][B], e.g. a
 *                   recorded or planned action trace): ONE launch, the env state in registers in between, every
 *                   step's (obs, mask, reward, done) optionally written step-major
 *   jss_session_* <- the interactive loop `obs, r, done, _, _ = env.step(policy(obs))` (README.md:53-64) with the env
 *                   state RESIDENT on the chip between steps: a kernel that lives across steps takes each step's
 *                   actions from a device mailbox the caller's stream posts to, and writes that step's outputs
 *   jss_rollout_steps <- DispatchingRule.run_episode's loop (dispatching.py:55-75) issued as n_sub independent
 *                   sub-batches on n_sub streams, so that consecutive steps of different sub-batches
 *                   overlap on the device (env instances are independent)
 *   jss_step_autoreset <- JssEnv.step(action) / JssEnv.reset() (jss_env.py:403-481, :145-181) as a vector env calls them:
 *                   an env that reported done on the previous call is reset instead of stepped, in the same launch
 *   jss_policy_step_steps <- the un-fused loop `a = policy(obs); obs, r, done, _, _ = env.step(a)` (README.md:53-64), K
 *                   times, policy and step as launches of their own, pipelined over sub-batches
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer in JssDesc/JssState/JssOut is a
 *     DEVICE pointer owned by the caller (the Python host allocates them as torch
 *     tensors).  The library never allocates, frees or synchronises.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*).
 *   - return value: 0 = launched, <0 = argument error (JSS_E_*), >0 = hipError_t.
 *   - kernels: when every env of the batch has jobs, machines <= 32, 64/G envs share a
 *     wavefront (G = 16 or 32 lanes per env); otherwise one wavefront simulates one env
 *     (job j on lane j % 64, slot j / 64; machine m on lane m).  Limits: jobs <= 128,
 *     machines <= 64, durations in [1, 65535].
 *
 * Data layout (all row-major, batch outermost)
 *   op table      int32 [n_tables][jmax][mmax]   machine << 16 | duration, 0 = padding
 *   work table    int32 [n_tables][jmax][mmax]   rem[j][k] = sum of the durations of ops k..M-1 of job j
 *                                                (MWR / LWR / CR read it; rem[j][0] = job length)
 *   instance rec  int32 [n_tables][JSS_NI]       JSS_I_*: J, M, the observation's normalisers and their
 *                                                float32 reciprocals
 *   job state     int32 [B][jmax][JSS_NF]        one 32-byte record per job (JSS_F_* words below):
 *                                                a lane moves its job with two dwordx4 accesses
 *                 or    [B][jmax][JSS_NFC]       16-byte compact records (JSS_FC_*) when the batch shares one instance
 *                                                and JssDesc.record_ints says so
 *                 or    [B][jmax][JSS_NFM]       24-byte medium records (JSS_FM_*): per-env instances with machines <= 32
 *   env header    int32 [B][JSS_NH]              JSS_H_*: clock, episode, step, status
 *   env constants int32 [B][JSS_NC]              JSS_C_*: the env's instance constants (copied in by reset)
 *   machine state int32 [B][mmax]                time_until_available_machine (full records only)
 *   action_mask   uint8 [B][jmax + 1]            legal_actions (output; rebuilt from the flag bits
 *                                                every call); NOPE flag at index J(env), zeros after it
 *   solution      int32 [B][jmax][mmax]          start time of op k of job j, -1 = unscheduled
 *   real_obs      float [B][jmax][7]             the reference's (J,7) observation, zero rows after J(env)
 */
#ifndef JSS_HIP_H
#define JSS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JSS_ABI_VERSION 10

#define JSS_MAX_JOBS 128
#define JSS_MAX_MACHINES 64

/* words of the per-job record */
#define JSS_F_TODO 0      /* bits 0-7 todo_time_step_job, bit 8 legal_actions[j], bit 9 action_illegal_no_op[j],
                             bits 10-31 the op after the next one (op table entry [j][todo + 2] as machine << 16 |
                             duration: 22 bits), 0 = none                                                       */
#define JSS_F_CUR 1       /* current op, machine << 16 | duration; -1 = job finished.
                             needed_machine_jobs == cur >> 16 (arithmetic shift)   */
#define JSS_F_LEFT 2      /* time_until_finish_current_op_jobs                    */
#define JSS_F_PERF 3      /* total_perform_op_time_jobs                           */
#define JSS_F_IDLE 4      /* total_idle_time_jobs                                 */
#define JSS_F_IDLE_LAST 5 /* idle_time_jobs_last_op                               */
#define JSS_F_F4 6        /* numerator of observation feature 4 (written only when an
                             op finishes, jss_env.py:569-586); JSS_F4_ONE = "1.0"  */
#define JSS_F_NEXT 7      /* the op after the current one (op table entry [j][todo + 1]), -1 = none.  The record
                             carries the job's next THREE ops (cur, next, bits 10-31 of word 0) so that a step
                             touches the op table only when a job moves on, or when a look-ahead walk of
                             _check_no_op goes further than three ops */
#define JSS_NF 8
/* The compact record (JssDesc.record_ints == JSS_NFC; only for a batch that shares ONE instance, n_tables == 1): the
 * three cached ops are what the op table says at [j][todo .. todo + 2], and with one table for the whole batch every
 * workgroup has that table in LDS anyway -- so the record does not carry them; and within the library's limits
 * (durations <= 65535, machines <= 64) the remaining fields fit four words: ONE 16-byte access per job instead of two,
 * half the state traffic.  A compact batch keeps no machine clocks in memory either (JssState.machine is not
 * read or written and may be NULL): time_until_available_machine[m] is time_until_finish_current_op_jobs of the job
 * running on m -- both are set to the op's duration when it is scheduled (jss_env.py:446-449) and count down together
 * (:521-530) -- 0 for an idle machine, and is rebuilt from the records at every load.  Words: */
#define JSS_FC_W0 0        /* bits 0-6 todo_time_step_job (<= 64), bit 7 legal_actions[j], bit 8 action_illegal_no_op[j],
                              bit 9 observation feature 4 is "1.0" (JSS_F4_ONE), bits 10-31 total_perform_op_time_jobs
                              (<= 64 x 65535 < 2^22)                                                                   */
#define JSS_FC_LEFT_F4 1   /* bits 0-15 time_until_finish_current_op_jobs, bits 16-31 the feature-4 numerator (0 when bit
                              9 of word 0 is set)                                                                     */
#define JSS_FC_IDLE 2      /* total_idle_time_jobs                                  */
#define JSS_FC_IDLE_LAST 3 /* idle_time_jobs_last_op                                */
#define JSS_NFC 4
#define JSS_FC_TODO_MASK 127
#define JSS_FC_FLAG_LEGAL 128
#define JSS_FC_FLAG_BLOCKED 256
#define JSS_FC_FLAG_F4_ONE 512
#define JSS_FC_PERF_SHIFT 10
/* The medium record (JssDesc.record_ints == JSS_NFM): 24 bytes, for batches whose envs differ in instance (so the op
 * table is not in LDS and the record has to carry the job's next three ops) and whose instances have at most 32 machines
 * (mmax <= 32, any number of jobs, either kernel flavour; JSS_E_SHAPE otherwise).  With machines <= 32 an op is 21
 * bits (machine << 16 | duration, 0 = none), todo <= 32 is 6 bits, total_perform_op_time_jobs <= 32 x 65535 < 2^21: 189
 * of the 192 bits.  No machine clocks in memory either (JssState.machine may be NULL), as with compact records.  Measured
 * faster than full records on the 16-lane packed shapes (profiles/README.md) and on the one-job-per-lane shapes of the
 * one-wavefront-per-env flavour (profiles/r06_misc/medium_records_wave.txt): that is where the host uses it.  Words: */
#define JSS_FM_W0 0        /* bits 0-5 todo_time_step_job, bit 6 legal_actions[j], bit 7 action_illegal_no_op[j], bit 8 observation
                              feature 4 is "1.0", bits 9-29 the current op (0 = job finished)                              */
#define JSS_FM_LEFT_F4 1   /* bits 0-15 time_until_finish_current_op_jobs, bits 16-31 the feature-4 numerator          */
#define JSS_FM_PERF_NEXT 2 /* bits 0-20 total_perform_op_time_jobs, bits 21-31 the low 11 bits of the next op          */
#define JSS_FM_NEXT_NEXT2 3 /* bits 0-9 the high 10 bits of the next op, bits 10-30 the op after the next one           */
#define JSS_FM_IDLE 4      /* total_idle_time_jobs                                   */
#define JSS_FM_IDLE_LAST 5 /* idle_time_jobs_last_op                                 */
#define JSS_NFM 6
#define JSS_FM_TODO_MASK 63
#define JSS_FM_FLAG_LEGAL 64
#define JSS_FM_FLAG_BLOCKED 128
#define JSS_FM_FLAG_F4_ONE 256
#define JSS_FM_CUR_SHIFT 9
#define JSS_FM_OP_MASK 0x1FFFFF
#define JSS_F4_ONE (-1)
#define JSS_TODO_MASK 255
#define JSS_FLAG_LEGAL 256
#define JSS_FLAG_BLOCKED 512
#define JSS_NEXT2_SHIFT 10

/* words of the per-env header (16 bytes: read and rewritten by every call) */
#define JSS_H_CLOCK 0    /* current_time_step                                     */
#define JSS_H_EPISODE 1  /* episodes started (RNG key)                            */
#define JSS_H_STEP 2     /* env steps since reset (RNG key)                       */
#define JSS_H_STATUS 3   /* bits 0-7 JSS_ERR_*, bit 8 legal_actions[J] (NOPE)     */
#define JSS_NH 4
#define JSS_STATUS_NOOP 256

/* words of the per-env constants record (JssState.env_const, 48 bytes): a copy of the env's instance record and
 * its index, written by every reset of the env and read-only in between.  A step-type call of a batch whose envs
 * differ in instance gets everything it needs to know about its env from here, in the same memory round trip as the
 * state -- no env -> instance -> J/M chain of dependent loads in front of the job records, no second fetch of the
 * observation's normalisers behind them.  (A batch that shares ONE instance reads the instance record itself, a
 * wave-uniform scalar load, and never touches this tensor outside reset.) */
#define JSS_C_JOBS 0           /* J of the env's instance (0 = the env was never reset: step-type calls leave it alone) */
#define JSS_C_MACHINES 1       /* M                                               */
#define JSS_C_MAX_TIME_OP 2    /* jss_env.py:86                                   */
#define JSS_C_TABLE 3          /* index of the env's instance in ops / rem / inst */
#define JSS_C_MAX_TIME_JOBS 4  /* jss_env.py:89                                   */
#define JSS_C_SUM_OP 5         /* jss_env.py:88                                   */
#define JSS_C_RCP_MAX_TIME_OP 6    /* float32 bits, as JSS_I_RCP_*                */
#define JSS_C_RCP_MAX_TIME_JOBS 7
#define JSS_C_RCP_SUM_OP 8
#define JSS_C_RCP_MACHINES 9
#define JSS_NC 12              /* record stride in ints (words 10, 11 are 0)      */

/* words of the per-instance record */
#define JSS_I_JOBS 0
#define JSS_I_MACHINES 1
#define JSS_I_MAX_TIME_OP 2    /* jss_env.py:86 */
#define JSS_I_MAX_TIME_JOBS 3  /* jss_env.py:89 */
#define JSS_I_SUM_OP 4         /* jss_env.py:88 */
#define JSS_I_RCP_MAX_TIME_OP 5   /* float32 bits of 1 / max_time_op  (correctly rounded) */
#define JSS_I_RCP_MAX_TIME_JOBS 6 /* float32 bits of 1 / max_time_jobs */
#define JSS_I_RCP_SUM_OP 7        /* float32 bits of 1 / sum_op */
#define JSS_I_RCP_MACHINES 8      /* float32 bits of 1 / M */
#define JSS_NI 12              /* record stride in ints (48 bytes: three dwordx4) */

/* per-env error bits (low byte of the header's status word, sticky until reset) */
#define JSS_ERR_ILLEGAL_ACTION 1 /* job action outside the mask: ignored (reference: silent corruption) */
#define JSS_ERR_NOPE_IDLE 2      /* NOPE/advance with no busy machine (reference: IndexError, jss_env.py:517) */
#define JSS_ERR_BAD_ACTION 4     /* action < -2 or > J: ignored (reference: IndexError) */

#define JSS_ACTION_SKIP (-1) /* batched step: this env is not stepped; its state, reward, done and makespan
                                are left as they were (observation and mask are rewritten unchanged) */
#define JSS_ACTION_RESET (-2) /* batched step: this env is reset() instead of stepped (reward 0, done 0, episode + 1):
                                 gymnasium.vector "next-step" auto-reset in the same launch as the other envs' steps */
#define JSS_ACTION_CLOSE (-3) /* step-session mailbox only (posted by jss_session_close): the resident kernel stores the env
                                 state and exits.  Anywhere else it is an out-of-range action (JSS_ERR_BAD_ACTION) */

/* policies */
#define JSS_POLICY_RANDOM 0
#define JSS_POLICY_FIFO 1
#define JSS_POLICY_SPT 2
#define JSS_POLICY_MWR 3
#define JSS_POLICY_LWR 4
#define JSS_POLICY_MOR 5
#define JSS_POLICY_LOR 6
#define JSS_POLICY_CR 7 /* critical ratio (1.5 * job length - now) / remaining work, dispatching.py:365-408 */
#define JSS_N_POLICIES 8
/* `kind` arguments: bits 0-7 = the policy.  For JSS_POLICY_CR bits 8-15 / 16-23 may carry a due-date factor p / q other than
 * the reference's default 3 / 2 (CriticalRatio(due_date_factor=...), dispatching.py:337-360): 1 <= p <= 255, q a power of
 * two <= 64 -- the factors for which factor * job_length is exact in the reference's doubles, so that the device's exact
 * fraction comparison (p * job_length - q * now) / remaining sees the reference's order AND its ties.  0 = 3 / 2. */
#define JSS_POLICY_CR_FACTOR(p, q) (JSS_POLICY_CR | ((p) << 8) | ((q) << 16))
/* CriticalRatio with ANY due-date factor the reference accepts (a Python float, dispatching.py:337-360): the factor travels as
 * the double JssDesc.cr_factor and the selector evaluates the reference's own float64 expression -- fl(fl(fl(length * factor)
 * - now) / remaining), IEEE round-to-nearest, no contraction -- so order and ties are the reference's bit for bit.  Accepted by
 * the calls whose policy is a launch of its own (jss_policy, jss_multi_policy, jss_policy_step_steps); the fused rollouts
 * (jss_rollout, jss_rollout_steps, jss_trajectory, jss_multi_rollout) keep the integer-exact p / q form above and answer
 * JSS_E_KIND: their kernels carry no float64 code. */
#define JSS_POLICY_CR_F64 (JSS_POLICY_CR | (1 << 24))

/* jss_rollout flags */
#define JSS_ROLLOUT_AUTORESET 1 /* an env found done is reset instead of stepped (iteration not counted) */
#define JSS_ROLLOUT_FORK_JOIN 2 /* jss_rollout_steps only: the library orders streams[1..] behind streams[0] before its first
                                   launch and streams[0] behind all of them after its last (hipEventRecord /
                                   hipStreamWaitEvent on events it owns, one set per streams[0]), so that the call is
                                   stream-ordered on streams[0] like any other -- the caller does not fork / join.  Calls
                                   that share streams[0] must come from one host thread at a time. */

/* argument errors */
#define JSS_E_NULL (-1)
#define JSS_E_SHAPE (-2)
#define JSS_E_KIND (-3)
#define JSS_E_LDS (-4) /* the batch shape needs more LDS per workgroup than the device has */
#define JSS_E_RESIDENT (-5) /* jss_session_open: the batch does not fit the chip as ONE round of resident workgroups (with
                               room left for the caller's own kernels), even with 8 env sets per wavefront */
#define JSS_E_SESSION (-6)  /* jss_session_post / wait / close: bad step range (ring overrun, not the next step, closed) */

/* JssDesc.kernel: JSS_KERNEL_AUTO packs 64/G envs per wavefront when every env of the batch fits a 16- or
 * 32-lane group (jmax, mmax <= 32) and uses one wavefront per env otherwise; JSS_KERNEL_WAVE forces one
 * wavefront per env (A/B runs, tests).  JSS_KERNEL_ONE_ENV_PER_WAVE (a bit, OR-ed in): the one-wavefront-per-env launches
 * of the one-step calls (jss_step, jss_rollout(n_iter = 1), jss_rollout_steps, jss_multi_*) never let a wavefront serve two
 * envs in turn -- what they do by default when a launch covers JSS_TWO_PER_WAVE_MIN_BATCH envs or more (one job per lane,
 * per-env tables, full or medium records; results identical either way: A/B runs, tests); JSS_KERNEL_TWO_ENVS_PER_WAVE: they do so
 * whatever the size of the launch (tests on small batches).  Per call, not per process.  The CPU twin ignores the field. */
#define JSS_KERNEL_AUTO 0
#define JSS_KERNEL_WAVE 1
#define JSS_KERNEL_ONE_ENV_PER_WAVE 2
#define JSS_KERNEL_TWO_ENVS_PER_WAVE 4

typedef struct JssDesc {
    int32_t batch;               /* B: envs in this shard                                   */
    int32_t jmax, mmax;          /* padded job / machine extents of every per-env row       */
    int32_t n_tables;            /* distinct instances                                      */
    const int32_t *ops;          /* [n_tables][jmax][mmax]                                  */
    const int32_t *rem;          /* [n_tables][jmax][mmax] remaining-work table             */
    const int32_t *inst;         /* [n_tables][JSS_NI] instance records                     */
    const int32_t *table_of_env; /* [B] instance of env i; NULL: 0 if n_tables==1 else i    */
    const int64_t *env_ids;      /* [B] explicit global ids (shape-bucketed batches); NULL: env_id_base + i */
    int64_t env_id_base;         /* global id of env 0 (keys the RNG stream under sharding) */
    int32_t kernel;              /* JSS_KERNEL_*                                            */
    int32_t threads;             /* libjss_cpu.so: OpenMP threads for this call (0 = runtime default);
                                    libjss_hip.so ignores it                                */
    int32_t jmin;                /* hint: smallest J among the batch's instances (0 = unknown, read as jmax).  When
                                    jmin < jmax (a ragged, padded batch) the one-wavefront-per-env kernels read the
                                    instance record BEFORE the job records and never load the rows behind J(env)  */
    int32_t record_ints;         /* ints per job record: 0 or JSS_NF = full records; JSS_NFC = compact records
                                    (n_tables == 1 only, else JSS_E_SHAPE); JSS_NFM = medium records (n_tables > 1 and
                                    mmax <= 32, else JSS_E_SHAPE)                           */
    double cr_factor;            /* CriticalRatio's due_date_factor for kind = JSS_POLICY_CR_F64 (> 0); ignored otherwise */
    int32_t jclass, mclass;      /* jss_multi_* only (0, 0 = jmax, mmax): the largest J / M among THIS set's envs when the set is a
                                    range of a batch that is padded wider (a shape class of a ragged population inside ONE set of
                                    padded tensors): the grid picks the set's body from the class -- 4 or 2 envs per wavefront,
                                    one wavefront per env, two jobs per lane -- and takes the strides from jmax / mmax.  Such an
                                    env keeps its class for life (table_of_env may move it between instances of the class only),
                                    and the tensors must have been zero-filled: a class body touches the rows it owns, not
                                    the padding behind them.  The single-set calls ignore both fields */
} JssDesc;

typedef struct JssState {
    int32_t *env;      /* [B][JSS_NH]  JSS_H_*                                         */
    int32_t *env_const;/* [B][JSS_NC]  JSS_C_*: written by reset, read by the step-type calls */
    int32_t *job;      /* [B][jmax][JSS_NF] (or [B][jmax][JSS_NFC] / [JSS_NFM], JssDesc.record_ints) */
    int32_t *machine;  /* [B][mmax]; unused (may be NULL) with compact and medium records */
    int32_t *solution; /* [B][jmax][mmax]                                              */
    int64_t *counters; /* [B][4]: env steps, finished episodes, sum of makespans,
                          sum of reward numerators (reward * max_time_op); may be NULL */
} JssState;

typedef struct JssOut {
    float *real_obs;      /* [B][jmax][7]; rows J..jmax-1 are written as zeros            */
    uint8_t *action_mask; /* [B][jmax+1]; bytes J+1..jmax are written as zeros            */
    float *reward;        /* [B] reward of the last step (jss_env.py:483-493)             */
    uint8_t *done;        /* [B] nb_legal_actions == 0 (jss_env.py:639-653)               */
    int32_t *makespan;    /* [B] clock at the last done transition (last_time_step, :650) */
} JssOut;

/* Per-step record of jss_trajectory, step-major (slot k of env i at [k][i]).  Any pointer may be NULL: that stream
 * is not recorded. */
typedef struct JssTraj {
    float *real_obs;      /* [K][B][jmax][7]  the observation the policy saw in slot k.  Rows J(env)..jmax-1 are never
                             written: allocate the buffer zero-filled                                                */
    uint8_t *action_mask; /* [K][B][jmax+1]   the mask the policy saw (whole row written, zeros behind the NOPE flag) */
    int32_t *action;      /* [K][B]  the action taken: job, J = NOPE; JSS_ACTION_RESET = the env was found done and was
                             reset instead of stepped (JSS_ROLLOUT_AUTORESET); JSS_ACTION_SKIP = found done, left frozen */
    float *reward;        /* [K][B]  reward of that step (0 in RESET / SKIP slots)   */
    uint8_t *done;        /* [K][B]  done after that step                            */
    int64_t stride;       /* envs between slot k and slot k + 1 of one env: 0 = desc->batch.  A call on a RANGE of a larger
                             batch (a shape class of a batch dealt out by class, JssDesc.jclass: every per-env pointer moved to
                             the range's first env, these five and jss_steps' actions included) passes the WHOLE batch's size,
                             so that the range's records land in its columns of the whole batch's [K][B] buffers (ABI v10) */
} JssTraj;

int jss_abi_version(void);
const char *jss_error_string(int code);
/* "hip:gfx950" for libjss_hip.so, "cpu:openmp" / "cpu:serial" for libjss_cpu.so */
const char *jss_backend(void);

/* reset every env (which == NULL) or the envs with which[i] != 0 */
int jss_reset(const JssDesc *desc, const JssState *state, const JssOut *out, const uint8_t *which, void *stream);

/* one step() per env; actions[i] in [0, J], JSS_ACTION_SKIP or JSS_ACTION_RESET */
int jss_step(const JssDesc *desc, const JssState *state, const int32_t *actions, const JssOut *out, void *stream);

/* JssEnv.step() (jss_env.py:403-481) with gymnasium.vector "next-step" auto-reset -- JssEnv.reset() (:145-181) of the envs
 * that finished -- folded in: an env whose out->done is set (it reported done on the
 * previous call) is reset instead of stepped -- its action is ignored, reward 0, done 0, episode + 1 -- exactly as if the
 * caller had put JSS_ACTION_RESET into actions[i].  One launch for the whole `obs, r, done = envs.step(a)` of a vector env. */
int jss_step_autoreset(const JssDesc *desc, const JssState *state, const int32_t *actions, const JssOut *out, void *stream);

/* one increase_time_step() per env with which[i] != 0 (NULL = all); hole[i] = returned idle time (may be NULL) */
int jss_advance(const JssDesc *desc, const JssState *state, const uint8_t *which, int32_t *hole, const JssOut *out,
                void *stream);

/* actions[i] = policy(state i); explore_q16 = probability (x 65536) of answering NOPE when NOPE is
 * legal (the reference's rules use 0.1, dispatching.py:113; 0 = deterministic) */
int jss_policy(const JssDesc *desc, const JssState *state, int kind, uint64_t seed, uint32_t explore_q16,
               int32_t *actions, void *stream);

/* n_iter x (policy + step) per env inside one launch, state held in registers */
int jss_rollout(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                uint32_t explore_q16, int32_t n_iter, int32_t flags, void *stream);

/* jss_rollout(n_iter = n_steps) that also records every iteration in `traj` (slot k = iteration k): state, `out`
 * and the counters end up exactly as after jss_rollout with the same arguments; an iteration that finds its env done
 * records JSS_ACTION_RESET (auto-reset: gymnasium.vector "next-step" semantics, the slot after it holds the fresh
 * episode's first observation) or JSS_ACTION_SKIP, reward 0.  State is read and written once per call. */
int jss_trajectory(const JssDesc *desc, const JssState *state, const JssOut *out, const JssTraj *traj, int kind,
                   uint64_t seed, uint32_t explore_q16, int32_t n_steps, int32_t flags, void *stream);

/* n_steps x jss_step per launch: actions[k * B + i] = the action of env i in step k (the codes of jss_step: job, J = NOPE,
 * JSS_ACTION_SKIP, JSS_ACTION_RESET).  State, `out` and the counters end exactly as after n_steps jss_step calls with
 * actions + k * B; the state is read and written ONCE.  With `traj` (may be NULL; any of its streams may be NULL) slot k
 * holds real_obs / action_mask AFTER step k and the reward / done of step k; for an env that step k did not step
 * (JSS_ACTION_SKIP, JSS_ACTION_RESET) the slot records reward 0 and done = "no legal action in the state it is in" -- not
 * the carried-over values jss_step would leave in `out` (an env skipped because it is done records done 1, a restarted one
 * done 0).  traj->action is not written. */
int jss_steps(const JssDesc *desc, const JssState *state, const JssOut *out, const JssTraj *traj, const int32_t *actions,
              int32_t n_steps, void *stream);

/* ---- step session: the env state stays on the chip between steps -----------------------------------------------
 * jss_session_open launches ONE resident kernel on `stream` -- a stream of its own AND of a priority of its own
 * (hipStreamCreateWithPriority): HIP maps the streams of one priority onto a small pool of hardware queues, and a kernel
 * queued behind the resident one in the same queue would never start; everything else the caller does goes to streams
 * of another priority.  Its wavefronts load their envs' state once and then, step after step, wait for the step's actions
 * in the mailbox, execute jss_step's semantics on registers (env sets parked in LDS when a wavefront owns several), and
 * write that step's real_obs / action_mask / reward / done / makespan / solution entry to `out` write-through, followed
 * by a per-wavefront progress word.  Nothing of the state is read from or written to memory between open and close: a
 * step moves the action in (8 bytes per env) and the outputs out.
 *   mailbox    mail[(step % depth) * B + i] = (uint64_t)(step + 1) << 32 | (uint32_t)action of env i in `step`: the tag
 *              makes a granule self-validating (one 8-byte store, no separate flag, no fence)
 *   post       jss_session_post (a small kernel on the CALLER's stream, e.g. behind its policy network) writes the granules
 *              of steps [first_step, first_step + n_steps) from an int32 [n_steps][B] action buffer
 *   wait       jss_session_wait (a one-workgroup kernel on the caller's stream) returns once every wavefront has
 *              published `steps_done` steps.  `out` is ONE set of buffers that every step overwrites: kernels enqueued
 *              behind the wait see step `steps_done`'s outputs, whole, only if nothing beyond it has been posted (the
 *              resident kernel does not hold step s + 1's stores back for a reader of step s); with steps posted ahead
 *              the outputs are defined again once the caller has waited for everything it posted
 *   close      jss_session_close posts JSS_ACTION_CLOSE for step `next_step`: the resident kernel writes the state back
 *              (job records, header, machine clocks) and adds its counters; after the session's stream has drained the
 *              batch is an ordinary batch again.  State tensors and counters are NOT current while a session is open.
 * The ring holds `depth` steps: the caller must not post step s before it has waited for step s - depth (the host
 * functions check `first_step + n_steps - waited <= depth` from the numbers they are given; JSS_E_SESSION otherwise).
 * While a session is open the caller synchronises its own streams, never the device (hipDeviceSynchronize would wait for
 * the resident kernel).  Every device-side wait is bounded: a wavefront that sees no mail for timeout_ms stores its state, bumps status[0] and
 * exits (the session is then dead: close it); a wait that times out bumps status[1].
 * Residency: the grid must fit the chip in ONE round with room to spare for the caller's kernels (two workgroup slots per
 * CU and 64 KB of its LDS stay free); jss_session_open picks the smallest number of env sets per wavefront (1, 2, 4, 8)
 * that does, JSS_E_RESIDENT if none.  One session per device at a time: a second resident grid would not find the room
 * the first one was promised.
 * The env -> instance map (JssDesc.table_of_env) is fixed while a session is open: a JSS_ACTION_RESET restarts the env
 * on the instance it has. */
typedef struct JssSession {
    uint64_t *mail;      /* [depth][B] action granules; zero-filled by the caller before open                          */
    int32_t *progress;   /* [B] (one word per wavefront is used): steps published; zero-filled by the caller          */
    int32_t *status;     /* [4]: wavefronts that timed out, waits that timed out, wavefronts that exited, env sets per
                            wavefront chosen by open; zero-filled by the caller                                       */
    int32_t depth;       /* steps the mailbox ring holds (>= 1)                                                       */
    int32_t timeout_ms;  /* bound of every device-side wait (0 = 2000)                                                */
    int32_t slots;       /* env sets per wavefront: 0 = the smallest of 1, 2, 4, 8 that fits; otherwise exactly this many
                            (JSS_E_RESIDENT if that does not fit)                                                     */
    int32_t reserved;
} JssSession;

int jss_session_open(const JssDesc *desc, const JssState *state, const JssOut *out, const JssSession *session, void *stream);
/* `waited` = the number of steps the caller has already waited for (flow control of the ring, see above) */
int jss_session_post(const JssDesc *desc, const JssSession *session, const int32_t *actions, int32_t first_step,
                     int32_t n_steps, int32_t waited, void *stream);
int jss_session_wait(const JssDesc *desc, const JssSession *session, int32_t steps_done, void *stream);
/* post of ONE step + wait for it in a single launch on the caller's stream (the kernel's first workgroup stays until the step
 * is finished): `obs, r, done = step(actions)` as one call.  step = the step's number (== steps posted so far == waited). */
int jss_session_step(const JssDesc *desc, const JssSession *session, const int32_t *actions, int32_t step, void *stream);
int jss_session_close(const JssDesc *desc, const JssSession *session, int32_t next_step, void *stream);

/* Debug aid: waits for everything queued on `stream` and returns the first asynchronous error (0 = none).  The
 * launching calls above only report what the launch itself reports; a fault inside a kernel surfaces here.  The
 * only call of the library that blocks. */
int jss_sync_check(void *stream);

/* n_steps x jss_rollout(n_iter = 1) over the whole batch, issued as n_sub contiguous sub-batches (boundaries at
 * multiples of 64 envs): step s of sub-batch i is launched on streams[i] and depends only on step s - 1 of the same
 * sub-batch, so the tail of one sub-batch's launch overlaps the head of another's.  Same results as n_steps calls
 * of jss_rollout.  The caller orders streams[] against its own stream (fork before, join after) unless it passes
 * JSS_ROLLOUT_FORK_JOIN.  1 <= n_sub <= 16. */
int jss_rollout_steps(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                      uint32_t explore_q16, int32_t n_steps, int32_t flags, int32_t n_sub, void *const *streams);

/* The UN-fused loop `a = policy(obs); obs, r, done = envs.step(a)` (README.md:53-64 with a vector env), n_steps times, with
 * the policy a launch of its own (jss_policy as the stand-in for the caller's policy network) and the actions going through
 * memory: per step and sub-batch jss_policy -> `actions` -> jss_step (jss_step_autoreset with JSS_ROLLOUT_AUTORESET), the
 * sub-batches on n_sub streams so that one sub-batch's policy overlaps another's step.  Same results as n_steps x
 * (jss_policy, jss_step) over the whole batch.  `actions` = int32 [B] scratch of the caller.  Streams and
 * JSS_ROLLOUT_FORK_JOIN as jss_rollout_steps. */
int jss_policy_step_steps(const JssDesc *desc, const JssState *state, const JssOut *out, int kind, uint64_t seed,
                          uint32_t explore_q16, int32_t *actions, int32_t n_steps, int32_t flags, int32_t n_sub,
                          void *const *streams);

/* The same for SEVERAL independent env sets at once (the shape classes of a ragged population, each a compact batch of
 * its own: jssenv_amd.BucketedJssEnv): n_steps x jss_rollout(n_iter = 1) per set, set i on streams[i], the launches
 * issued step-major (step s of every set before step s + 1 of any), so that sets whose streams share a hardware queue
 * lose their overlap and nothing more.  One host call per window instead of one per set and chunk.  flags as
 * jss_rollout_steps (JSS_ROLLOUT_FORK_JOIN: streams[1..] are ordered against streams[0] by the library).
 * 1 <= n_sets <= 16. */
int jss_rollout_steps_multi(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states,
                            const JssOut *const *outs, int kind, uint64_t seed, uint32_t explore_q16, int32_t n_steps,
                            int32_t flags, void *const *streams);

/* ---- several independent env sets in ONE launch ----------------------------------------------------------------
 * The shape classes of a ragged population (jssenv_amd.BucketedJssEnv: one compact batch per class -- 16-lane groups,
 * 32-lane groups, one wavefront per env, two jobs per lane) stepped without padding and without one launch (and one
 * stream) per class: ONE grid per call covers every set, a workgroup finding its set by its index.  Each call does to
 * every set i exactly what the single-set call does with descs[i] / states[i] / outs[i] -- jss_multi_reset = jss_reset
 * (JssEnv.reset, jss_env.py:145-181), jss_multi_step = jss_step / jss_step_autoreset (JssEnv.step, :403-481;
 * flags = JSS_ROLLOUT_AUTORESET or 0), jss_multi_policy = jss_policy, jss_multi_rollout = n_steps x jss_rollout(n_iter = 1)
 * (n_steps launches per part, see below) -- results identical.  The fused grid covers sets with per-env instance tables
 * (n_tables > 1; full records, or medium records on the 16- / 32-lane shapes), 2 to 6 of them; any other combination is
 * issued as one plain launch per set (and part, below) in the same stream order (same results).  `which` (jss_multi_reset) may be NULL, and so may
 * its entries: every env of that set.  1 <= n_sets <= 16. */
int jss_multi_reset(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const JssOut *const *outs,
                    const uint8_t *const *which, void *stream);
int jss_multi_step(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const int32_t *const *actions,
                   const JssOut *const *outs, int32_t flags, void *stream);
int jss_multi_policy(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, int kind, uint64_t seed,
                     uint32_t explore_q16, int32_t *const *actions, void *stream);
/* n_sub, streams, JSS_ROLLOUT_FORK_JOIN as jss_rollout_steps: with n_sub > 1 every set is cut into n_sub contiguous parts
 * (at multiples of 64 envs; at most 4 are used) and part i of ALL sets is one grid per step on streams[i], so that one part's
 * drain overlaps another's fill; n_sub = 1 is one grid per step on streams[0].  Sets without a body in the fused grid are cut
 * into the same parts, part i of every set as plain launches on streams[i].  n_steps == 0 launches nothing and touches nothing
 * (jss_rollout_steps, jss_rollout_steps_multi and jss_policy_step_steps likewise; both libraries). */
int jss_multi_rollout(int32_t n_sets, const JssDesc *const *descs, const JssState *const *states, const JssOut *const *outs,
                      int kind, uint64_t seed, uint32_t explore_q16, int32_t n_steps, int32_t flags, int32_t n_sub,
                      void *const *streams);

#ifdef JSS_PROFILING
/* Instrumented builds only (tools/build_instrumented.py compiles with -DJSS_PROFILING; the shipped library does
 * not export this).  JSS_PROF_ABLATE: bit mask of phases the kernels skip -- results become WRONG -- used to
 * attribute kernel time; JSS_PROF_LDS_PAD: extra dynamic LDS bytes per workgroup, to cap occupancy. */
#define JSS_PROF_ABLATE 1
#define JSS_PROF_LDS_PAD 2
#define JSS_ABLATE_CHECK_NO_OP 1
#define JSS_ABLATE_PRIORITIZE 2
#define JSS_ABLATE_OBS 4
#define JSS_ABLATE_SELECT 8
#define JSS_ABLATE_ADVANCE 16
int jss_profiling_set(int option, int value);
/* device buffer [B][16] uint64 (NULL = off): lane 0 of every wavefront of the one-wavefront-per-env kernels records the shader
 * clock at its phase boundaries (slot 0 entry, 1 header words there, 2 state unpacked, 3 action selected, 7 / 8 / 9 inside step():
 * after the allocation + event jump, _prioritization_non_final, _check_no_op, 4 step done, 5 state / mask stores issued, 6 end) */
int jss_profiling_stamps(void *device_buffer);
#endif

#ifdef __cplusplus
}
#endif
#endif
