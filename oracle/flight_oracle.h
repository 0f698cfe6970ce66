/*
 * oracle/flight_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64, one env at a time) of the reference's
 * flight_easy / flight environment step+reward path.  It is the checker the HIP
 * path is compared against; nothing under cooperative-search_amd/ may include,
 * link, import or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it.
 *
 * Parity status: PINNED against golden vectors captured from the imported
 * reference (the .npz files under tests/golden/, generator tests/golden/gen_golden.py).
 * The reference has no tests of its own.
 *
 * Third-party arithmetic restated here: NumPy's legacy global RandomState
 * (MT19937 + 53-bit `rand` + polar `randn`), numpy unpinned by the reference
 * (2.2.6 in the build container); algorithm facts in SURVEY.md Appendix B.
 */
#ifndef FLIGHT_ORACLE_H
#define FLIGHT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_AGENTS 8
#define ORC_MAX_TARGETS 16
#define ORC_MAP_MAX 64

typedef struct orc_config {
    int32_t variant;      /* 0 = flight_easy (env/flight_env_easy.py), 1 = flight (env/flight_env.py) */
    int32_t n_agents;
    int32_t n_targets;
    int32_t map_size;
    int32_t view_range;
    int32_t time_limit;
    int32_t agent_mode;   /* 0..3 */
    int32_t target_mode;  /* 0 = file + jitter, 1 = uniform */
    double velocity;      /* args.agent_velocity */
    double safe_dist;
    double detect_prob;
    double force_dist;
    double force_factor;  /* POTENTIAL_FORCE_FACTOR */
    /* circle_dict rows (main.py:19-32), units of map_size/10 */
    double cx[ORC_MAX_TARGETS], cy[ORC_MAX_TARGETS], dx[ORC_MAX_TARGETS], dy[ORC_MAX_TARGETS];
    int32_t deter[ORC_MAX_TARGETS]; /* 1 = 't' (fixed), 0 = 'f' (jittered) */
} orc_config;

typedef struct orc_env orc_env;

orc_env *orc_create(const orc_config *cfg);
void orc_destroy(orc_env *e);
void orc_seed(orc_env *e, uint32_t seed);           /* np.random.seed(seed) for this env's private stream */
void orc_reset(orc_env *e, int init);               /* env.reset(init) */
int orc_step(orc_env *e, const int32_t *actions, int32_t *reward, int32_t *terminated, int32_t *win);
void orc_get_obs(const orc_env *e, double *out);    /* [n][4] easy, [n][map*map+4] flight */
void orc_get_state(const orc_env *e, double *out);  /* [4n+3m] */
/* raw state */
void orc_get_agents(const orc_env *e, double *pos_xy, double *yaw, int32_t *out_flag);
void orc_set_agents(orc_env *e, const double *pos_xy, const double *yaw);
void orc_set_trig_mode(int mode); /* 0 (default): libm sin/cos like the reference; 1: correctly rounded table (HIP-equivalent) */
void orc_set_exact_pow(int on); /* 1 (default): `**2` = libm pow(x,2.0) like the reference; 0: x*x (timed baseline) */
void orc_get_targets(const orc_env *e, double *pos_xy, int32_t *found);
void orc_set_targets(orc_env *e, const double *pos_xy, const int32_t *found);
void orc_get_counters(const orc_env *e, int32_t *out6); /* target_find, win, time_step, total_reward, curr_reward, newly_mask */
void orc_get_prob_map(const orc_env *e, double *out);   /* [map][map], first index = x cell */
void orc_set_prob_map(orc_env *e, const double *in);
uint64_t orc_words_consumed(const orc_env *e);          /* 32-bit MT outputs consumed since orc_seed */
void orc_set_draw_log(orc_env *e, double *buf, int64_t cap); /* log every env-level rand() */
int64_t orc_draw_log_count(const orc_env *e);
void orc_clear_draw_log(orc_env *e);

/* raw RNG access (for RNG known-answer tests) */
uint32_t orc_rng_u32(orc_env *e);
double orc_rng_rand(orc_env *e);
double orc_rng_randn(orc_env *e);

/* batch driver used by bench.py's cpu_baseline leg: B independent envs, OpenMP over envs */
typedef struct orc_batch orc_batch;
orc_batch *orc_batch_create(const orc_config *cfg, int64_t batch, const uint32_t *seeds);
void orc_batch_destroy(orc_batch *b);
void orc_batch_reset(orc_batch *b, int init, const uint8_t *mask, int threads);
/* one step of every env (+ obs/state emission in fp32 when the pointers are non-null);
 * auto_reset != 0 resets envs that were terminated on entry (same rule as the HIP path) */
void orc_batch_step(orc_batch *b, const int32_t *actions, float *reward, uint8_t *terminated, uint8_t *win,
                    float *obs, float *state, int auto_reset, int freeze_done, int threads);
/* T steps of every env in ONE OpenMP region, env-major; outputs [T][B][...] as T orc_batch_step calls would write */
void orc_batch_rollout(orc_batch *b, const int32_t *actions, int T, float *reward, uint8_t *terminated, uint8_t *win,
                       float *obs, float *state, int auto_reset, int freeze_done, int threads);
void orc_batch_rollout_rep(orc_batch *b, const int32_t *actions, int T, int repeat, float *reward, uint8_t *terminated,
                           uint8_t *win, float *obs, float *state, int auto_reset, int freeze_done, int threads);
orc_env *orc_batch_env(orc_batch *b, int64_t i);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
