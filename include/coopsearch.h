/*
 * include/coopsearch.h -- C ABI of libcoopsearch_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of WZN1ng/Cooperative-Search: the flight_easy / flight environment
 * reset + step + reward + obs/state emission, batched over B independent environments that live in HBM.
 * The reference has no FFI: its boundary is a duck-typed Python object.  Every
 * entry point below names the reference method it replaces (paths relative to the reference repo); the
 * Python mirror of that object protocol lives in cooperative-search_amd/env.py.  Two bindings sit on these symbols:
 * torch.ops.coopsearch.* (csrc/torch_ops.cpp, the default: tensor checks in C++, torch's current HIP stream) and
 * ctypes (_lib.py, the torch-free route; INTEGRATION.md shows both stubs).
 *
 * Conventions
 *   - plain C types only; every *_dev pointer is DEVICE memory owned by the caller (e.g. a torch tensor's
 *     data_ptr()); `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *   - nothing allocates, synchronises or copies on the step path; all work is enqueued on `stream`;
 *   - return value: 0 = OK, negative = CS_E_*; cs_last_error() gives a thread-local message;
 *   - n_targets <= 16 and n_agents <= 8: every kernel layout (16 lanes per env with lane t owning target t; 8 lanes per env
 *     with lane t owning agent t and targets t, t + 8; one env per lane) is built on these bounds.  Which layout a call runs
 *     on is chosen by batch size: the dispatch table is in DESIGN.md section 4; all of them give the same results.
 *
 * Numerics contract: agent positions and yaw are fp64 and follow the reference's
 * float operations one for one (same order, no FMA contraction, correctly rounded sin/cos of the
 * accumulated yaw); the uniform draws are NumPy's MT19937 stream per env, consumed in the reference's
 * order; rewards, flags and counts are integers.  Emitted obs/state/reward tensors are fp32.
 */
#ifndef COOPSEARCH_H
#define COOPSEARCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_ABI_VERSION 7   /* 2: cs_layout.ahead_off (pre-twisted MT words), cs_mt_canonical; 3: cs_layout.job_off;
                              4: cs_source_hash, CS_KERNEL_OCT; 5: CS_KERNEL_ODE; 6: cs_epsilon (exploration schedule),
                              cs_epsilon_step, CS_KERNEL_LANEV; 7: CS_CHECK_ACTIONS, cs_has_legacy_kernels */
#define CS_MAX_AGENTS 8
#define CS_MAX_TARGETS 16
#define CS_MAX_MAP 64
#define CS_MT_PAD 32      /* words 0..31 of an env's MT19937 row are mirrored behind word 623 */
#define CS_MT_STRIDE 672  /* uint32 words per env row: 624 state + 32 mirror + 16 unused (21 x 128 bytes) */
#define CS_TAPE_STRIDE 16 /* uint32 words per env of the lane kernel's hit tape (cs_layout.tape_off) */
#define CS_JOB_BYTES 256  /* bytes per env and record of the flight map-update job records (cs_layout.job_off) */

enum { CS_OK = 0, CS_E_CONFIG = -1, CS_E_ARG = -2, CS_E_LAUNCH = -3 };

/* action selection of cs_policy_forward / cs_rollout_policy */
enum {
    CS_SELECT_SOFTMAX = 1, /* agent/agent.py:77-97 (_choose_action_from_softmax, alg == 'reinforce'):
                              prob = (1 - epsilon) * softmax(q) + epsilon / n_actions; default (0) is :68-75, argmax with
                              epsilon-greedy exploration */
    CS_SELECT_SAMPLE = 2   /* with CS_SELECT_SOFTMAX: draw from Categorical(prob); without it argmax(prob) -- the
                              reference samples unless (epsilon == 0 and evaluate) */
};

/* cs_step / cs_rollout flags */
enum {
    CS_FREEZE_DONE = 1,  /* an env that was terminated on entry is left untouched: reward 0, terminated 1.
                            (The reference has no terminal guard, flight_env_easy.py:303-314; leave this
                            flag clear to reproduce that, as the B = 1 adapter does.) */
    CS_AUTO_RESET = 2,   /* an env that was terminated on entry is reset(init=False) first, then stepped */
    CS_ACTIONS_I64 = 4,  /* actions_dev holds int64 (torch.long) instead of int32 */
    CS_KERNEL_GROUP = 8, /* flight_easy: force the 16-lanes-per-env kernels (the default of cs_step up to the lane kernels' range) */
    CS_KERNEL_LANE = 16, /* flight_easy: force the first-generation lane-per-env kernel (the default for teams of 6 to 8 at very
                            large batches; smaller teams get CS_KERNEL_LANEV's kernel); same results */
    CS_KERNEL_SOLO = 32, /* (rounds 1-2: the 16-lanes-per-env ROLLOUT kernels, one wavefront or a kinematics / detection pair per four
                            envs.  Removed in round 6 -- no dispatch row had selected them since round 3: cs_rollout answers this flag
                            and the next with CS_E_CONFIG; the values stay reserved) */
    CS_KERNEL_DUO = 64,
    CS_KERNEL_OCT = 128, /* cs_rollout, flight_easy: force the 8-lanes-per-env kernel (lane t owns agent t and targets t, t + 8;
                            nothing replicated but the header: four and more wavefronts per SIMD -- the default between
                            the pair kernel's range and the lane kernel's); same results */
    CS_KERNEL_OD = 256,  /* cs_rollout, flight_easy: force the octet PAIR kernel (the 8-lane layout with a kinematics wavefront
                            running steps ahead of a detection wavefront, per 8 envs); same results */
    CS_KERNEL_ODE = 512, /* ... with a third wavefront per 8 envs that writes the outputs (default up to 8192 envs when obs
                            and state are both requested); same results */
    CS_KERNEL_LANEV = 1024, /* flight_easy, teams of up to 5: force the second-generation lane-per-env kernel (targets in
                            registers, half-wavefront staging of the get_state rows: three to four wavefronts per SIMD instead of
                            two; CS_KERNEL_LANE forces the first generation); same results */
    CS_CHECK_ACTIONS = 2048 /* debug aid: validate every action of the call on the device BEFORE anything is stepped.  A value
                            outside 0..2 makes the call return CS_E_ARG with the reference's IndexError wording ("list index out
                            of range": dyaw[act], flight_env_easy.py:259-262) and leaves the env state untouched.  Without the
                            flag the kernels treat any value other than 1 / 2 as 0 (no bounds check in the hot loops).  Python's
                            negative indices (dyaw[-1]) are NOT accepted by the batched path; the B = 1 adapter maps them like the
                            reference.  Synchronises the stream: a call made while the stream is being captured (or while the
                            capture state cannot be queried) is NOT checked, silently.  A library built with
                            -DCS_CHECK_ACTIONS_ALWAYS checks every call; the torch op layer sets the flag for batches of up to
                            64 envs (COOPSEARCH_CHECK_ACTIONS=0 / 1 turns that off / on for every batch). */
};

/* Exploration schedule of RolloutWorker.generate_episode -- common/rollout.py:35-41 (episode / epoch scale: one anneal before
 * the episode: the caller's), :75-76 (STEP scale, the QMIX and DOP default, common/arguments.py:78-82, 137-141: after every
 * executed env step `epsilon = epsilon - anneal_epsilon if epsilon > min_epsilon else epsilon`), :133-135 (the value persists
 * across episodes).  The reference has ONE worker and one scalar; a batch of B envs is B independent workers, each with its own
 * epsilon annealed by its own executed steps (a frozen, finished env does not anneal -- its episode loop has ended).
 *   eps_dev == NULL: every env uses `epsilon` for the whole call (no schedule).
 *   eps_dev != NULL: double [B], env b's epsilon: read when the call starts, annealed on the device after every step env b
 *                    executes (if per_step), written back when the call ends.
 * All arithmetic is the reference's: IEEE doubles, one subtraction per executed step. */
typedef struct cs_epsilon {
    double epsilon;       /* value of every env when eps_dev is NULL */
    double anneal;        /* args.anneal_epsilon */
    double min_epsilon;   /* args.min_epsilon */
    int32_t per_step;     /* != 0: args.epsilon_anneal_scale == 'step' */
    int32_t reserved;
    double *eps_dev;      /* NULL or double [B], in / out */
    double *trace_dev;    /* NULL or double [T][B], out: the epsilon env b's action selection of step t used */
} cs_epsilon;

/* Environment constants: common/arguments.py:27-34 (map_size, target_num, target_mode, agent_mode, n_agents,
 * view_range) and :233-284 (get_flight_args / get_flight_easy_args), plus the parsed target file
 * (main.py:19-32 load_targets; units of map_size/10). */
typedef struct cs_config {
    int32_t variant;      /* 0 = flight_easy (env/flight_env_easy.py), 1 = flight (env/flight_env.py) */
    int32_t n_agents;     /* 1..CS_MAX_AGENTS */
    int32_t n_targets;    /* 1..CS_MAX_TARGETS (reference: 15) */
    int32_t map_size;     /* reference: 50 */
    int32_t view_range;   /* reference: 7 */
    int32_t time_limit;   /* reference: 200 */
    int32_t agent_mode;   /* 0..3, flight_env_easy.py:139-180 */
    int32_t target_mode;  /* 0 = target file + gaussian jitter, 1 = uniform; flight_env_easy.py:95-136 */
    double velocity;      /* args.agent_velocity = 1 */
    double safe_dist;     /* 1 */
    double detect_prob;   /* 0.9 */
    double force_dist;    /* 3 */
    double force_factor;  /* POTENTIAL_FORCE_FACTOR = 0.8, flight_env_easy.py:65 */
    double cx[CS_MAX_TARGETS], cy[CS_MAX_TARGETS], dx[CS_MAX_TARGETS], dy[CS_MAX_TARGETS];
    int32_t deter[CS_MAX_TARGETS]; /* 1 = 't' fixed, 0 = 'f' jittered */
    int64_t batch;        /* B: environments resident on this device */
} cs_config;

/* Byte offsets of the per-env arrays inside the caller-allocated state blob (all 256-byte aligned). */
typedef struct cs_layout {
    size_t total_bytes;
    size_t tgt_off;    /* double [B][16][2]   target (x, y)                                             */
    size_t agent_off;  /* double [B][8][4]    agent (x, y, yaw, spare)                                   */
    size_t hdr_off;    /* int32  [B][16]      CS_H_* words below                                         */
    size_t mt_off;     /* uint32 [B][CS_MT_STRIDE] MT19937 state, circular (incremental) form + cursor in hdr;
                          words 624..655 mirror words 0..31                                          */
    size_t ahead_off;  /* int32  [B]          number of words at the cursor that are ALREADY twisted (their outputs
                          are temper(word)): the lane-per-env kernel regenerates the state 192 words at a time,
                          coalesced, ahead of consumption; 0 = the plain circular form.  0 <= ahead <= 624   */
    size_t tape_off;   /* uint32 [B][CS_TAPE_STRIDE] hit tape of the twisted words: bit r of words 0..9 = "the draw made of
                          stream words 2r, 2r+1 counted from the stream position `base` satisfies rand() <= detect_prob";
                          words 10, 11 = base (value of CS_H_WORDS_LO/HI when written), 12, 13 = the integer threshold
                          it was built for.  Derived data: written by cs_mt_advance / the lane kernel, ignored (and
                          rebuilt) whenever it does not match the cursor                                        */
    size_t prob_off;   /* float  [B][map*map] probability map, first index = x cell (flight only)        */
    size_t job_off;    /* uint8  [2][B][CS_JOB_BYTES] flight only: what the map sweep needs of the step that ran before
                          it (pending-pass flags, newly found masks, target cells, agent positions), double buffered so
                          that cs_rollout can sweep step t's map while step t + 1 runs.  Derived data: rewritten by
                          every cs_reset / cs_step / cs_rollout before it is read                              */
} cs_layout;

/* words of the per-env header */
enum {
    CS_H_FOUND = 0,     /* bit j = target j found                       (Target.find, flight_env_easy.py:12) */
    CS_H_NEWLY = 1,     /* targets found by the last detection pass     (tgt_found, flight_env.py:238)       */
    CS_H_TARGET_FIND = 2, /* env.target_find                             (flight_env_easy.py:42)              */
    CS_H_FLAGS = 3,     /* bit0 win_flag, bit1 map update pending, bit2 reset-time map update pending,
                           bits 8..15 out_flag[i]                                                            */
    CS_H_TIME_STEP = 4, /* env.time_step                                                                      */
    CS_H_TOTAL_REWARD = 5, /* env.total_reward (integer)                                                      */
    CS_H_MT_POS = 6,    /* cursor into the circular MT19937 state, 0..623, always even                        */
    CS_H_EPISODES = 7,  /* resets performed                                                                  */
    CS_H_WORDS_LO = 8,  /* 32-bit MT outputs consumed since cs_seed (u64, lo/hi)                             */
    CS_H_WORDS_HI = 9,
    CS_H_CURR_REWARD = 10, /* env.curr_reward of the last detection pass                                     */
    CS_H_NEWLY_RESET = 11, /* flight: targets found by the reset-time pass of a fused auto-reset              */
    CS_H_WORDS = 16
};

int cs_abi_version(void);
/* Hash of the sources this library was compiled from (cooperative-search_amd/build.py:source_hash; "" for a build that
 * did not pass it): how the loader tells a library built from other sources, instead of comparing file mtimes. */
const char *cs_source_hash(void);
/* 0 (since round 6: the 16-lanes-per-env ROLLOUT kernels of rounds 1-2 are gone; CS_KERNEL_SOLO / CS_KERNEL_DUO make cs_rollout
 * return CS_E_CONFIG, a CS_KERNEL_GROUP rollout is T launches of the 16-lane step kernel).  Kept for ABI 7 callers. */
int cs_has_legacy_kernels(void);
const char *cs_last_error(void);

/* Fills `out` with the state-blob layout for cfg (cfg->batch envs).  Host only. */
int cs_state_layout(const cs_config *cfg, cs_layout *out);

/* Zeroes the blob and sets every probability map to 0.5 (FlightSearchEnv.__init__, flight_env.py:53). */
int cs_init(const cs_config *cfg, void *state_dev, void *stream);

/* np.random.seed(seeds[b]) for env b's private stream (the reference shares one global stream between a
 * single env and everything else, and never seeds it: SURVEY.md Appendix C Q1). */
int cs_seed(const cs_config *cfg, void *state_dev, const uint32_t *seeds_dev, void *stream);

/* env.reset(init) -- flight_env_easy.py:79-182, flight_env.py:83-191 -- for every env with mask[b] != 0
 * (mask_dev NULL = all).  Ends with the reference's reset-time detection pass (and map update for flight).
 * obs_dev / state_dev_out (either may be NULL) receive get_obs() / get_state() of ALL envs afterwards:
 *   obs   float [B][n][4]            flight_easy   (get_obs, flight_env_easy.py:218-221)
 *         float [B][n][map*map + 4]  flight        (get_obs, flight_env.py:223-230: map first, 4 features last)
 *   state float [B][4n + 3m]                       (get_state, flight_env_easy.py:190-216) */
int cs_reset(const cs_config *cfg, void *state_dev, const uint8_t *mask_dev, int init,
             float *obs_dev, float *state_out_dev, void *stream);

/* env.step(act_list) -- flight_env_easy.py:303-314 (_agent_step :255-291, _potential_energy_force :293-301,
 * _update_obs :223-253), flight_env.py:357-368 (+ _update_prob_map :275-303) -- for every env.
 *   actions_dev    int32 or int64 [B][n], values 0/1/2 (validated only with CS_CHECK_ACTIONS)
 *   reward_dev     float [B]   (integer valued)     terminated_dev / win_dev  uint8 [B]
 *   obs_dev / state_out_dev as in cs_reset (may be NULL) */
int cs_step(const cs_config *cfg, void *state_dev, const void *actions_dev, int flags,
            float *reward_dev, uint8_t *terminated_dev, uint8_t *win_dev,
            float *obs_dev, float *state_out_dev, void *stream);

/* T consecutive env.step calls from ONE call: actions [T][B][n]; reward [T][B]; terminated/win [T][B];
 * obs [T][B][n][obs width]; state_out [T][B][4n+3m] (obs/state_out may be NULL).  Same results as T cs_step calls
 * with the same flags.  flight_easy: one launch with the env resident in registers; flight: one launch per step in
 * which the map sweep of step t (update + the map part of get_obs) runs beside the kinematics / detection of step
 * t + 1 -- the two are independent once the sweep reads the step's job record (cs_layout.job_off).
 * flight_easy: obs_dev must be 16-byte aligned (CS_E_ARG otherwise): an (env, agent) observation is one 16-byte store;
 * state_out_dev may be anywhere (a table that is not 16-byte aligned, or whose rows per step are not, takes scalar stores). */
int cs_rollout(const cs_config *cfg, void *state_dev, const void *actions_dev, int T, int flags,
               float *reward_dev, uint8_t *terminated_dev, uint8_t *win_dev,
               float *obs_dev, float *state_out_dev, void *stream);

/* MT19937 pre-pass: for every env with fewer than `min_ahead` twisted words ahead of its cursor, twist the whole row
 * ahead (ahead -> 624) in one coalesced sweep.  Does not change any stream: only WHEN its words are regenerated.
 * cs_rollout's lane-per-env path runs it before every 64-step chunk for teams of 6 and more (smaller teams refresh
 * their rows inside the kernels).  Callers that drive cs_step should call it every ~32 steps with min_ahead ~400
 * (BatchedFlightEnv.step does): a single step reads its draws from the env's hit tape while that is valid, which keeps
 * the MT19937 window -- a load that depends on the header's cursor -- off the critical path of the launch (7.3 instead
 * of 7.8 us per step at 4096 envs); a row that runs out simply falls back to twisting on demand. */
int cs_mt_advance(const cs_config *cfg, void *state_dev, int min_ahead, void *stream);

/* Writes every env's MT19937 row in CANONICAL form -- all 624 words twisted ahead of the cursor -- to
 * rows_out_dev (uint32 [B][CS_MT_STRIDE]) without changing the state.  Two states that describe the same position of the same
 * stream have equal canonical rows (and equal CS_H_MT_POS), however much of the row each kernel had pre-twisted. */
int cs_mt_canonical(const cs_config *cfg, void *state_dev, uint32_t *rows_out_dev, void *stream);

/* get_obs() + get_state() of every env without stepping. */
int cs_emit(const cs_config *cfg, void *state_dev, float *obs_dev, float *state_out_dev, void *stream);

/* Per-device partial sums of the evaluation metrics of runner.py:86-96 / rollout.py:190-198:
 * out4_dev[0] += sum total_reward, [1] += sum win_flag, [2] += sum target_find, [3] += number of envs.
 * (double[4]; the caller zeroes it, then all-gathers / all-reduces the partials over RCCL.) */
int cs_metrics(const cs_config *cfg, void *state_dev, double *out4_dev, void *stream);

/* ---- caller-side row f3: fused forward of the shared recurrent agent network ------------
 * Replaces the per-agent batch-1 loop of agent/agent.py:33-75 (choose_action) over network/base_net.py:5-46
 * ([conv ->] fc1 -> ReLU -> GRUCell(64) -> fc2[Linear, ReLU, Linear]): ONE launch for all rows = B * n_agents, fp32 on
 * the matrix cores, epsilon-greedy choice on the device; for flight, one more launch for the conv front end. */

/* floats in a packed weight blob */
size_t cs_policy_packed_floats(void);

/* HOST: torch-layout weights (fc1.weight [64][in_dim], rnn.weight_ih / weight_hh [192][64], fc2.0.weight [64][64],
 * fc2.2.weight [n_actions][64] and their biases) -> packed_host[cs_policy_packed_floats()], to be copied to the device.
 * in_dim = [16 conv features +] 4 + n_actions + n_agents <= 32 (agent.py:41-52, base_net.py:31-39).
 * The network runs on the 16-bit matrix pipe with every fp32 operand split into two halves (csrc/policy_dev.h): a WEIGHT whose
 * magnitude exceeds 65504 (or is not finite) is refused with CS_E_ARG; an ACTIVATION (a ReLU output of fc1 / fc2) beyond 65504
 * SATURATES there (min(relu(.), 65504): one v_med3_f32 per conversion, values inside the range unchanged bit for bit) -- with
 * |obs| <= 1 and |h| < 1 no activation gets that far while the rows of fc1 / fc2.0 have an L1 norm below ~6e4, as any trained
 * network's do; the q-values stay finite either way (tests/test_gpu_policy.py). */
int cs_policy_pack(const float *fc1_w, const float *fc1_b, const float *w_ih, const float *b_ih, const float *w_hh,
                   const float *b_hh, const float *fc2a_w, const float *fc2a_b, const float *fc2b_w, const float *fc2b_b,
                   int in_dim, int n_actions, float *packed_host);

/* One forward over rows = B*n (row r: env r / n_agents, agent r % n_agents).  obs row r = 4 floats at
 * obs_dev + r*obs_stride + obs_offset (floats); last_dev[r] = previous action or < 0 for none
 * (last_dev NULL = raw mode: the row at obs_dev + r*obs_stride + obs_offset already holds all 4 + n_actions +
 * n_agents non-conv inputs);
 * feat_dev (NULL for flight_easy): float [rows / rows_per_feat][16] conv features (cs_policy_conv_features) that go in
 * front of the obs columns; rows r*rows_per_feat .. +rows_per_feat-1 share feature row r (rows_per_feat = n_agents
 * when the features were computed once per env);
 * hidden_dev float [rows][64] updated in place; q_dev float [rows][n_actions] or NULL; actions_dev int64 [rows]:
 * select = 0: argmax_a q (first maximum), or with probability epsilon a uniform action; CS_SELECT_SOFTMAX: the softmax
 * rule of agent.py:77-97.  Random choices come from a counter-based generator keyed by (seed, step, row0 + row): row0 =
 * global index of this call's row 0 (env_offset * n_agents for a sharded batch), so the noise does not depend on the
 * sharding.
 * avail_actions (agent/agent.py:70, :87 mask q / prob with the env's get_avail_agent_actions) is not an input: both envs
 * of this path return all ones for every agent (flight_env_easy.py:184-188, flight_env.py:193-197), so the mask is the
 * identity; a caller with a real mask applies it to q_dev and selects on its side. */
int cs_policy_forward(const float *packed_dev, const float *obs_dev, int obs_stride, int obs_offset,
                      const int64_t *last_dev, const float *feat_dev, int rows_per_feat, float *hidden_dev, float *q_dev,
                      int64_t *actions_dev, int rows, int n_agents, int n_actions, float epsilon, const double *eps_env_dev,
                      uint64_t seed, uint32_t step, uint64_t row0, int select, void *stream);
/* (eps_env_dev: NULL, or double [rows / n_agents]: row r explores with epsilon eps_env_dev[r / n_agents] instead of `epsilon` --
 * the per-env schedule of cs_epsilon for callers that drive the loop step by step; cs_epsilon_step anneals it.) */

/* One step of the exploration schedule for a caller that drives the loop itself (cs_policy_forward -> cs_epsilon_step ->
 * cs_step): for every env that the NEXT cs_step(flags) will execute (not: terminated on entry and frozen),
 * eps_dev[b] = eps_dev[b] - anneal if eps_dev[b] > min_epsilon else eps_dev[b] (common/rollout.py:75-76); trace_row_dev (NULL or
 * double [B]) receives the values BEFORE the anneal, i.e. what the action selection of this step used. */
int cs_epsilon_step(const cs_config *cfg, void *state_dev, int flags, double *eps_dev, double anneal, double min_epsilon,
                    double *trace_row_dev, void *stream);

/* flight: the conv front end of base_net.py:9-18,31-36 with the reference's hyper-parameters (common/arguments.py:256-265:
 * Conv2d(1,4,k=4,s=2) -> ReLU -> Conv2d(4,1,k=3,s=1,p=1) -> ReLU -> Linear(576,16)) on n_maps 50x50 probability maps;
 * map m = 2500 floats at maps_dev + m*map_stride.  The weight pointers are the torch tensors as they are (device).
 * All agents of an env observe the same map (flight_env.py:223-230): run it once per env on the env's first obs row
 * (map_stride = n_agents * 2504) and pass rows_per_feat = n_agents to cs_policy_forward. */
int cs_policy_conv_features(const float *conv1_w_dev, const float *conv1_b_dev, const float *conv2_w_dev,
                            const float *conv2_b_dev, const float *lin_w_dev, const float *lin_b_dev,
                            const float *maps_dev, int64_t map_stride, int n_maps, float *feat_dev, void *stream);
const char *cs_policy_last_error(void);

/* Fused closed loop (flight_easy, n_agents <= 5): T x (cs_policy_forward -> cs_step) in ONE launch, i.e. the body of
 * RolloutWorker.generate_episode's loop (common/rollout.py:43-76) for all B envs with the hidden state, the chosen
 * actions and the envs resident on chip between steps.  Same results, bit for bit, as T pairs of
 * cs_policy_forward(..., step = step0 + s) and cs_step(flags) calls.
 *   packed_dev   cs_policy_pack output            hidden_dev  float [B*n][64], in/out
 *   last_dev     int64 [B][n] action before the first step (< 0 = none)
 *   actions_dev  int64 [T][B][n] chosen actions (out); the other outputs as in cs_rollout.
 * Same results, bit for bit, as T triples cs_policy_forward(step0 + s, eps_env_dev) -> cs_epsilon_step -> cs_step. */
int cs_rollout_policy(const cs_config *cfg, void *state_dev, const float *packed_dev, float *hidden_dev,
                      const int64_t *last_dev, int T, int flags, const cs_epsilon *eps, uint64_t seed, uint32_t step0,
                      uint64_t row0, int select, int64_t *actions_dev, float *reward_dev, uint8_t *terminated_dev,
                      uint8_t *win_dev, float *obs_dev, float *state_out_dev, void *stream);
/* (eps: the exploration schedule, see cs_epsilon; with eps->eps_dev the per-step anneal of rollout.py:75-76 runs INSIDE the
 * launch, env by env, and the carried values come back in eps_dev.) */

/* Closed loop for flight (the loop body of common/rollout.py:43-76 with the conv network of network/base_net.py:9-36):
 * T x (cs_policy_conv_features -> cs_policy_forward -> cs_step) enqueued by ONE call.  Bit for bit the results of the T
 * triples of calls; what changes is the data movement: the conv front end reads every env's probability map where it
 * lives in the state blob (one 10 KB read per env instead of a pass over the n observation copies), and with
 * obs_dev == NULL the n copies of the map that get_obs emits (flight_env.py:223-230) are never written -- the network
 * is their only consumer inside the loop.  The map update still runs every step.
 *   conv1_w_dev .. lin_b_dev   the six tensors of cs_policy_conv_features
 *   scratch_dev                float [B][16 + 4 n_agents] (conv features; the agents' own 4 observation floats)
 *   last_dev                   int64 [B][n] action before the first step (< 0 = none)
 *   actions_dev                int64 [T][B][n] chosen actions (out); obs_dev / state_out_dev NULL or as in cs_rollout */
int cs_rollout_policy_flight(const cs_config *cfg, void *state_dev, const float *packed_dev, const float *conv1_w_dev,
                             const float *conv1_b_dev, const float *conv2_w_dev, const float *conv2_b_dev,
                             const float *lin_w_dev, const float *lin_b_dev, float *hidden_dev, const int64_t *last_dev,
                             float *scratch_dev, int T, int flags, const cs_epsilon *eps, uint64_t seed, uint32_t step0, uint64_t row0,
                             int select, int64_t *actions_dev, float *reward_dev, uint8_t *terminated_dev, uint8_t *win_dev,
                             float *obs_dev, float *state_out_dev, void *stream);

/* ---- caller-side rows f1 / f2: episode batch assembly ------------------------------------------------------------
 * common/rollout.py:66-76,105-132 (the eleven per-episode arrays and their padding: steps after termination are zero
 * rows with padded = 1, terminated = 1) and common/replay_buffer.py:41-61 (store_episode) in one pass over the
 * collector's step-major tables.  All destinations are float32 [slots][T][...] arrays (a fresh [B][T][...] batch, or
 * the ring of a replay buffer); env b's episode goes to slot slot_dev[b] (NULL: slot b).
 *   o_tab [T+1][B][n][obs_w]   s_tab [T+1][B][state_w]   u_tab int64 [T][B][n]   r_tab [T][B]   term_tab u8 [T][B] */
typedef struct cs_episode_out {
    float *o, *u, *s, *r, *o_next, *s_next, *avail_u, *avail_u_next, *u_onehot, *padded, *terminated;
} cs_episode_out;

int cs_store_episodes(int B, int T, int n_agents, int n_actions, int obs_w, int state_w, const float *o_tab_dev,
                      const float *s_tab_dev, const int64_t *u_tab_dev, const float *r_tab_dev,
                      const uint8_t *term_tab_dev, const int64_t *slot_dev, const cs_episode_out *out, void *stream);
const char *cs_episodes_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
