/*
 * oracle/jss_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar C restatement of the reference simulator prosysscience/JSSEnv v1.1.0
 * (JSSEnv/envs/jss_env.py) and of the action selectors on the hot path
 * (JSSEnv/dispatching.py FIFO/SPT, README.md:53-64 random masked loop).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  tools/make_golden.py drives the live reference in
 * the build container and commits traces under tests/golden/; tests/
 * test_oracle_golden.py replays them through this library and demands exact
 * equality on every integer and on the float64 observation and reward.
 */
#ifndef JSS_ORACLE_H
#define JSS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-env error bits; the same values as include/jss_hip.h JSS_ERR_* */
#define ORC_ERR_ILLEGAL_ACTION 1 /* job action outside the mask: ignored            */
#define ORC_ERR_NOPE_IDLE      2 /* NOPE / advance with no busy machine (ref: IndexError at jss_env.py:517) */
#define ORC_ERR_BAD_ACTION     4 /* action < 0 or > J: ignored                      */

typedef struct OrcEnv OrcEnv;

OrcEnv *orc_create(int jobs, int machines, const int32_t *machine_jm, const int32_t *duration_jm);
void orc_destroy(OrcEnv *e);

void orc_reset(OrcEnv *e);                          /* jss_env.py:145-181 */
/* jss_env.py:403-481.  strict=0: the reference's behaviour for every action it
 * survives (any NOPE, any job action).  strict=1: the total semantics the
 * device implements (job actions outside the mask are ignored + flagged). */
int  orc_step(OrcEnv *e, int action, int strict, double *reward, int *done);
int  orc_increase_time_step(OrcEnv *e, int *hole);  /* jss_env.py:495-637; <0 when the queue is empty */

/* scalar getters */
int  orc_jobs(const OrcEnv *e);
int  orc_machines(const OrcEnv *e);
int  orc_current_time_step(const OrcEnv *e);
int  orc_nb_legal_actions(const OrcEnv *e);
int  orc_nb_machine_legal(const OrcEnv *e);
int  orc_next_time_step_len(const OrcEnv *e);
int  orc_err(const OrcEnv *e);
int  orc_max_time_op(const OrcEnv *e);
int  orc_max_time_jobs(const OrcEnv *e);
int  orc_sum_op(const OrcEnv *e);
long orc_last_reward_numerator(const OrcEnv *e);

/* array views (owned by the env, valid until orc_destroy) */
const int32_t *orc_todo_time_step_job(const OrcEnv *e);            /* [J]   */
const int32_t *orc_needed_machine_jobs(const OrcEnv *e);           /* [J]   */
const int32_t *orc_time_until_finish_current_op_jobs(const OrcEnv *e);
const int32_t *orc_total_perform_op_time_jobs(const OrcEnv *e);
const int32_t *orc_total_idle_time_jobs(const OrcEnv *e);
const int32_t *orc_idle_time_jobs_last_op(const OrcEnv *e);
const int32_t *orc_time_until_available_machine(const OrcEnv *e);  /* [M]   */
const int32_t *orc_solution(const OrcEnv *e);                      /* [J*M] */
const int32_t *orc_next_time_step(const OrcEnv *e);                /* [len] */
const uint8_t *orc_legal_actions(const OrcEnv *e);                 /* [J+1] */
const uint8_t *orc_action_illegal_no_op(const OrcEnv *e);          /* [J]   */
const uint8_t *orc_machine_legal(const OrcEnv *e);                 /* [M]   */
const uint8_t *orc_illegal_actions(const OrcEnv *e);               /* [M*J] */
const double  *orc_state(const OrcEnv *e);                         /* [J*7] */

/* action selectors --------------------------------------------------- */
#define ORC_POLICY_RANDOM 0 /* uniform over set bits of the mask, NOPE included (README.md:58-60) */
#define ORC_POLICY_FIFO   1 /* dispatching.py:133-156, exploration disabled */
#define ORC_POLICY_SPT    2 /* dispatching.py:92-116,  exploration disabled */
#define ORC_POLICY_MWR    3 /* dispatching.py:173-199 */
#define ORC_POLICY_LWR    4 /* dispatching.py:216-242 */
#define ORC_POLICY_MOR    5 /* dispatching.py:259-283 */
#define ORC_POLICY_LOR    6 /* dispatching.py:300-324 */
#define ORC_POLICY_CR     7 /* dispatching.py:365-408, due_date_factor 1.5 */

uint32_t orc_rng_u32(uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t step);
int orc_policy(const OrcEnv *e, int kind, uint64_t seed, uint64_t env_id, uint32_t episode, uint32_t step);
/* same with the rules' NOPE exploration (dispatching.py:113): when NOPE is legal, answer NOPE with
 * probability explore_q16 / 65536, drawn from the counter RNG keyed with seed ^ ORC_EXPLORE_SEED_XOR */
#define ORC_EXPLORE_SEED_XOR 0x5851F42D4C957F2DULL
int orc_policy_explore(const OrcEnv *e, int kind, uint64_t seed, uint32_t explore_q16, uint64_t env_id, uint32_t episode,
                       uint32_t step);

/* Run `steps` policy+step iterations with auto-restart (an env found done is
 * reset instead of stepped; that iteration is not counted).  Accumulates
 * counters[0]=env steps, [1]=finished episodes, [2]=sum of makespans,
 * and reward_sum.  Returns the env steps executed.  Used as the CPU baseline
 * and to cross-check the device rollout (same counter RNG). */
long orc_rollout(OrcEnv *e, int kind, uint64_t seed, uint64_t env_id, uint32_t *episode, uint32_t *step_in_episode,
                 long iterations, long counters[3], double *reward_sum);

/* The same loop for a whole batch of independent envs from a fresh reset, OpenMP over envs; returns every env's
 * final integer state, counters and float64 observation (any output may be NULL).  Lets the GPU tests compare EVERY
 * env of a full-size batch with this restatement, not a sample.  0 = ok. */
int orc_rollout_batch(int n_envs, int n_tables, int jmax, int mmax, const int32_t *jobs_of_table,
                      const int32_t *machines_of_table, const int32_t *machine_tjm, const int32_t *duration_tjm,
                      const int32_t *table_of_env, int kind, uint64_t seed, uint64_t env_id_base, const int64_t *env_ids,
                      uint32_t explore_q16, long iterations, int autoreset, int threads,
                      int32_t *clock, int32_t *episode, int32_t *step_in_episode, int32_t *job_fields, int32_t *tm,
                      int32_t *solution, uint8_t *mask, uint8_t *blocked, int64_t *counters, double *obs, int32_t *err);

#ifdef __cplusplus
}
#endif
#endif
