/* include/cagpu.h -- C ABI of libcagpu.so, the MI355X (gfx950) hot path of the batched
 * collision-avoidance simulator.
 *
 * This is the drop-in boundary.  The reference has no device boundary at
 * all: its only FFI is the Cython `rvo2.PyRVOSimulator` binding.  Each entry point below names
 * the reference interface it replaces (paths relative to
 * /root/reference/gym_collision_avoidance/envs/).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types; every data pointer is a DEVICE pointer
 *     into memory owned by the caller (torch tensors in the Python host), 16-byte aligned, valid
 *     until the stream reaches the call.  The library never allocates or frees device memory.
 *   - CaParams / CaState / CaOut / CaAutoReset are HOST structs, copied at call time.
 *   - asynchronous and stream-ordered on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream).  Re-entrant across streams and devices; the only host global is the
 *     thread-local last-error string.  Device globals: the fault word (cagpu_device_faults) and
 *     the per-CU progress table of the n-step kernel's progress-fair priorities -- words tagged
 *     with a per-process launch counter, read for issue priorities only: concurrent launches can
 *     perturb each other's priorities through it, never their results.
 *   - returns 0 on success, a negative CA_E* code otherwise; never throws across the boundary.
 *   - layout: agent-major SoA, index e*num_agents + a (agent fastest), one array per field, so a
 *     wavefront's 64 lanes load 64 consecutive elements.  State is float64 because the
 *     reference's state is (its discrete events -- at-goal, collision, time-out, sort buckets --
 *     are decided on float64 values); observations / rewards leave as float32, the dtype the
 *     reference declares for them (config.py:93-170).
 */
#ifndef CAGPU_H_
#define CAGPU_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAGPU_VERSION 11

/* error codes */
enum { CA_OK = 0, CA_EINVAL = -1, CA_EUNSUPPORTED = -2, CA_ELAUNCH = -3, CA_ENODEVICE = -4 };

/* per-agent flag word.  Bits 0-5 are the reference's Agent booleans (agent.py:108-112,138;
 * env.py:421-424,534-535); bits 6-7 come from the agent's Policy object (Policy.py:11-14,
 * config.py:152-157); bits 8-15 select the built-in policy / dynamics plugin. */
enum {
  CA_AT_GOAL = 1u << 0,
  CA_WAS_AT_GOAL = 1u << 1,
  CA_IN_COLLISION = 1u << 2,
  CA_WAS_IN_COLLISION = 1u << 3,
  CA_OUT_OF_TIME = 1u << 4,
  CA_DONE = 1u << 5,
  CA_IS_LEARNING = 1u << 6,
  CA_STILL_LEARNING = 1u << 7,
  CA_POLICY_SHIFT = 8,  /* 4 bits */
  CA_DYNAMICS_SHIFT = 12, /* 4 bits */
  /* Ragged batches: this agent SLOT holds no agent in the env's current episode.  The reference draws the agent count per
   * episode (test_cases.py:224-227: randint(2, MAX_NUM_AGENTS_IN_ENVIRONMENT + 1)) and every loop of its step runs over
   * len(self.agents) (collision_avoidance_env.py:345-367); here num_agents is the batch-wide MAXIMUM and an env with fewer
   * agents leaves its last slots absent.  Set by cagpu_reset / an auto-reset for a case row whose radius is <= 0 (the
   * padding rows of a ragged table); an absent slot is no neighbour, no collision partner, is not sensed, does not count
   * for game over or the episode statistics, and its outputs are zeros (observation row, reward) / done = 1 -- the zero
   * padding of wrappers.py:143-173.  Absent slots carry CA_DONE | CA_AT_GOAL | CA_WAS_AT_GOAL as well. */
  CA_ABSENT = 1u << 16,
  /* CaState.next_action holds this agent's action for the NEXT step (software-pipelined policy, see next_action).
   * Whoever writes the state arrays of an env directly must clear this bit for its agents (cagpu_reset does). */
  CA_PLAN_VALID = 1u << 17
};
/* policy plugin ids (test_cases.py:68-85 `policy_dict`) */
enum {
  CA_POL_RVO = 0,           /* policies/RVOPolicy.py + rvo2 (ORCA)           */
  CA_POL_NONCOOP = 1,       /* policies/NonCooperativePolicy.py              */
  CA_POL_STATIC = 2,        /* policies/StaticPolicy.py                      */
  CA_POL_EXTERNAL = 3,      /* policies/ExternalPolicy.py: raw [speed, dheading] from ext_actions */
  CA_POL_LEARNING = 4,      /* policies/LearningPolicy.py: scaled ext_actions */
  CA_POL_LEARNING_GA3C = 5, /* policies/LearningPolicyGA3C.py: discrete index in ext_actions[.,0] */
  CA_POL_GA3C_CADRL = 6     /* policies/GA3CCADRLPolicy.py: discrete index written into ext_actions[.,0] by cagpu_ga3c */
};
/* dynamics plugin ids (test_cases.py:93-96 `dynamics_dict`) */
enum {
  CA_DYN_UNICYCLE = 0,      /* dynamics/UnicycleDynamics.py:14-47            */
  CA_DYN_MAX_TURN_RATE = 1, /* dynamics/UnicycleDynamicsMaxTurnRate.py:17-43 */
  CA_DYN_EXTERNAL = 2       /* dynamics/ExternalDynamics.py                  */
};
/* OtherAgentsStatesSensor.agent_sorting_method (sensors/OtherAgentsStatesSensor.py:34-52) */
enum { CA_SORT_CLOSEST_FIRST = 0, CA_SORT_CLOSEST_LAST = 1, CA_SORT_TIME_TO_IMPACT = 2 };
/* game_over rule (collision_avoidance_env.py:537-551) */
enum { CA_OVER_ALL_DONE = 0 /* EVALUATE_MODE */, CA_OVER_AGENT0 = 1 /* TRAIN_SINGLE_AGENT */, CA_OVER_LEARNING_DONE = 2 };

/* The Config constants read on the hot path (config.py:28-86,174) + batch geometry. */
typedef struct CaParams {
  int32_t num_envs, num_agents;
  int32_t max_obs;           /* K = MAX_NUM_OTHER_AGENTS_OBSERVED; obs row = 6 + 7*K floats */
  int32_t sort_mode, game_over_mode;
  int32_t rvo_max_neighbors; /* MAX_NUM_AGENTS_IN_ENVIRONMENT (RVOPolicy.py:15) */
  int32_t obs_clip;          /* OtherAgentsStatesSensor.max_num_other_agents_observed (<= max_obs): only the
                                obs_clip closest others are emitted, the remaining rows stay zero
                                (OtherAgentsStatesSensor.py:39,112) */
  int32_t ragged;            /* != 0: envs may hold fewer than num_agents agents (CA_ABSENT slots: a case row with radius <= 0);
                                0 = every slot holds an agent, whatever its radius (the kernels skip the absent-slot tests) */
  double dt, near_goal_threshold, max_time_ratio, getting_close_range, sensing_horizon;
  double reward_at_goal, reward_collision, reward_time_step, reward_wiggly, wiggly_threshold;
  double reward_min, reward_max; /* np.clip bounds (collision_avoidance_env.py:589-599) */
  double rvo_time_horizon, rvo_collab_coeff;
  double max_heading_change; /* env-wide pi/3 (collision_avoidance_env.py:87), LearningPolicy.py:30 */
  double reward_collision_wall; /* REWARD_COLLISION_WITH_WALL (config.py:33; collision_avoidance_env.py:425-429) */
  double rvo_dt;             /* RVOPolicy.dt = Config.DT (RVOPolicy.py:13): rvo2's timeStep and the 1/dt of the speed
                                read-back (:26, :106) -- NOT the dt of this step() call, which only the dynamics see */
} CaParams;

/* Device pointers to the simulator state; [E*N] unless noted. */
typedef struct CaState {
  double *pos_x, *pos_y;       /* Agent.pos_global_frame                       */
  double *vel_x, *vel_y;       /* Agent.vel_global_frame                       */
  double *heading;             /* Agent.heading_global_frame                   */
  double *goal_x, *goal_y;     /* Agent.goal_global_frame                      */
  double *radius, *pref_speed;
  double *time_remaining;      /* Agent.time_remaining_to_reach_goal           */
  double *t;                   /* Agent.t                                      */
  double *slt;                 /* Agent.straight_line_time_to_reach_goal       */
  double *ep_reward;           /* running sum of this episode's rewards (env_utils.py:51) */
  float *last_action;          /* [E*N,2] Agent.past_actions[0] = [speed, delta heading] */
  uint32_t *flags;
  int32_t *step_num;           /* Agent.step_num                               */
  int32_t *episode_step;       /* [E] CollisionAvoidanceEnv.episode_step_number */
  int32_t *reset_count;        /* [E] auto-resets taken so far                 */
  double *env_stats;           /* [E,8] episodes, collision eps, all-at-goal eps, stuck eps, sum steps,
                                  sum total_reward, sum time_to_goal, sum extra_time_to_goal
                                  (experiments/src/env_utils.py:56-87, reduced to counters) */
  float *next_action;          /* [E*N,4] or NULL.  Software-pipelined policy query: the built-in RVO policy of step t+1
                                  reads the post-move state of step t only (collision_avoidance_env.py:305-323 runs it
                                  BEFORE anyone moves), exactly what the sensing / reward half of step t reads -- so with
                                  this array a step computes both side by side on disjoint waves and stores, per agent,
                                  {speed, delta heading (the float32 `all_actions` pair, env.py:305-307), ORCA velocity x, y}
                                  for the next step, flagged CA_PLAN_VALID; the next cagpu_step / cagpu_rollout then
                                  starts at the move.  Same arithmetic on the same inputs: results are bit-identical to
                                  next_action == NULL.  Agents without a valid plan (after a reset, after the host wrote
                                  the state, external / learning policies) are queried at the start of the step as before. */
  double *turning_dir;         /* [E*N] or NULL (not maintained).  Agent.turning_dir: the CADRL value network's turning
                                  memory, updated by UnicycleDynamics.step only (UnicycleDynamics.py:41-47), zeroed by
                                  Agent.reset (agent.py:133) */
  /* Per-step inputs of RVOPolicy's two stochastic branches (policies/RVOPolicy.py:77-90, :118-119), drawn by the CALLER
   * on the device before the launch (the host mirror does it with torch's device generator: core.BatchedSim
   * .set_rvo_stochastic) -- both NULL in the deterministic case, i.e. always in the reference's shipped configurations:
   *   rvo_collab        float  [E*N] or NULL: the collaboration coefficient of each agent AS THE EGO of its query
   *                     (setAgentCollabCoeff, :86-90) -- Config.RVO_COLLAB_COEFF, or 0 while an anti-collaborative agent
   *                     (RVO_COLLAB_COEFF < 0) is in its non-cooperative phase; NULL: CaParams.rvo_collab_coeff for all.
   *   rvo_heading_noise double [E*N] or NULL: added to the delta heading of an RVO agent's action after the pi/6 clip
   *                     (`delta_heading + np.random.normal(0, 0.5)`, :118-119); 0 for agents without heading_noise.
   * With either one set the policy is queried at the start of the step (the pipelined plan -- next_action -- is not
   * used: the draws belong to the step that consumes them). */
  const float *rvo_collab;
  const double *rvo_heading_noise;
  /* Externally integrated motion, applied AT THE MOVE of this step: device float64 [E*N, 5] = px, py, vx, vy, heading, or NULL.
   * An agent with CA_DYN_EXTERNAL whose row holds no NaN takes that state where the built-in models integrate theirs
   * (Agent.take_action, agent.py:214-220, calls `self.dynamics_model.step(action, dt)` -- here the caller ran its own
   * Dynamics subclass on the host with the action of this step), i.e. AFTER every policy of the step has been queried on
   * the pre-step state, and before the at-goal test, the clocks, collisions and sensing of the same step.  Rows of NaN /
   * agents with another dynamics id are ignored.  With it the policy is queried at the start of the step (like rvo_*). */
  const double *ext_state;
} CaState;

/* Device pointers to what a step hands back (collision_avoidance_env.py:225-234). */
typedef struct CaOut {
  float *obs;        /* [E,N,6+7K]: is_learning, num_other_agents, dist_to_goal, heading_ego_frame,
                        pref_speed, radius, other_agents_states[K][7] -- the array layout of
                        wrappers.py:143-173 (MultiagentDictToMultiagentArrayWrapper) */
  float *rewards;    /* [E,N]                                           */
  uint8_t *done;     /* [E,N] which_agents_done                         */
  uint8_t *game_over;/* [E]                                             */
  float *actions;    /* [E,N,2] the float32 `all_actions` array (env.py:305-307); may be NULL */
  float *orca_vel;   /* [E,N,2] or NULL: for every agent whose RVOPolicy was queried in this step, the velocity rvo2 chose
                        (PyRVOSimulator.doStep + getAgentVelocity, RVOPolicy.py:93) -- what cagpu_orca returns for the same
                        float inputs, bit for bit; 0 for the agents that were not queried.  Parity hook for the ORCA phases of
                        the step kernel itself. */
  void *workspace;   /* device scratch for envs with MORE THAN 64 AGENTS, or NULL.  Up to 64 agents an env is one workgroup tile
                        and every per-(agent, other) quantity lives in LDS; beyond that (the reference's make_testcase_huge /
                        get_testcase_huge, test_cases.py:914-1018: 100 agents) the step runs a one-thread-per-agent kernel
                        (num_agents <= 1024: workgroups of 256 / 512 / 1024 threads) whose per-pair columns live here.
                        cagpu_workspace_bytes(p) says how much. */
  uint64_t workspace_bytes;
} CaOut;

/* Fixture-table auto-reset (the batched form of vec_env.py:120-128 + test_cases.py:593-624):
 * when env e's episode ends its statistics are added to env_stats[e], its k-th reset loads case
 * (env_id_offset + e + k*case_stride) % n_cases of `table` and the observation handed back is the
 * reset observation (rewards / done / game_over stay those of the terminal step). */
typedef struct CaAutoReset {
  const double *table; /* device, [n_cases, N, 6] = px, py, gx, gy, pref_speed, radius */
  int32_t n_cases;
  int64_t env_id_offset; /* global id of this shard's env 0 (multi-GPU sharding) */
  int64_t case_stride;   /* normally the global number of envs */
  const float *reset_obs; /* device float [n_cases, N, 6+7*max_obs] or NULL: the reset observation of every case,
                             i.e. o->obs of cagpu_reset(num_envs = n_cases, cases = table) with the same CaParams.
                             With it an auto-reset copies the row; without it the tile runs a second sensing pass. */
  const float *reset_plan; /* device float [n_cases, N, 4] or NULL: CaState.next_action of every case's reset state (cagpu_plan
                              on the state cagpu_reset(num_envs = n_cases, cases = table) leaves), so that an auto-reset
                              env starts its new episode with a valid plan; only read when CaState.next_action is set. */
  uint64_t heading_seed;  /* 0: the initial heading of a reset agent points at its goal (EVALUATE_MODE, test_cases.py:555-557).
                             Otherwise training mode (test_cases.py:558-559: np.random.uniform(-pi, pi)): heading =
                             -pi + 2 pi u, u the Philox4x32-10 uniform of (heading_seed; global env id, reset count, agent)
                             -- a pure function of those, whatever the batch size or sharding; reset_obs is then not
                             used (the observation depends on the heading: second sensing pass). */
} CaAutoReset;

/* Static occupancy grid shared by every env (Map.py:6-24; the env builds Map(16 m, 16 m, 0.1 m), env.py:378-392).
 * Bit-packed, row-major: cell (row, col) is bit (col & 31) of word static_bits[row * ((cols + 31) / 32) + col / 32];
 * row = floor(origin_r - y / cell), col = floor(origin_c + x / cell) (Map.py:26-32). */
typedef struct CaMap {
  const uint32_t *static_bits; /* device; NULL = no static obstacles (agents are still rasterised for the scan) */
  int32_t rows, cols;
  double cell, origin_r, origin_c;
} CaMap;

/* LaserScanSensor state + observation (sensors/LaserScanSensor.py:24-44): num_beams beams over
 * [min_angle, max_angle] around the heading, num_ranges samples every range_res metres. */
typedef struct CaScan {
  uint8_t *hist; /* device [E,N,num_to_store,num_beams]: range index per beam, 255 = nothing hit (state) */
  float *out;    /* device [E,N,num_to_store,num_beams]: the 'laserscan' observation in metres */
  int32_t num_beams, num_to_store, num_ranges, reserved0;
  double min_angle, max_angle, range_res, max_range;
} CaScan;

/* GA3C-CADRL network weights (policies/GA3C_CADRL/checkpoints/<run>/network_*.data-00000-of-00001): device float
 * pointers in the checkpoint's own layout, kernels row-major [in, out].  LSTM gate order i, j, f, o. */
typedef struct CaNet {
  const float *lstm_kernel, *lstm_bias;     /* rnn/lstm_cell/{kernel [71,256], bias [256]}: input = [x_t (7), h (64)] */
  const float *layer1_kernel, *layer1_bias; /* layer1/{kernel [68,256], bias}: input = [host (4), h_final (64)]       */
  const float *layer2_kernel, *layer2_bias; /* layer2/{kernel [256,256], bias}                                        */
  const float *fc1_kernel, *fc1_bias;       /* fullyconnected1/{kernel [256,256], bias}                               */
  const float *logits_kernel, *logits_bias; /* logits_p/{kernel [256,11], bias [11]}                                  */
  const float *input_mean, *input_std;      /* graph constants `Const`, `Const_1` [138] (= config.py:93-149)          */
  /* Scratch for cagpu_ga3c, device int32 [num_envs * num_agents + 6] (the list, its count at [num_envs * num_agents], two
   * 64-bit counters behind it that are tagged with the call's epoch: the scratch needs NO initialisation and nothing an earlier
   * or aborted call left in it matters; the order of the packed rows across workgroups is unspecified), or NULL.  With it the agents that need an action
   * this step (GA3C-CADRL policy, not done: collision_avoidance_env.py:310-312 queries no others) are packed first and
   * only their rows are evaluated -- in steady state about half of the agents of an evaluation batch are done and wait
   * for their env's game over.  NULL: every 64-agent tile that holds at least one such agent is evaluated whole. */
  int32_t *rows_scratch;
  /* Which agents this checkpoint drives (the reference gives every agent its own policy object and network session,
   * GA3CCADRLPolicy.py:23-47, so agents of one scene may run different checkpoints): device int32 [num_envs * num_agents]
   * or NULL.  With it, cagpu_ga3c evaluates only the agents with agent_net[i] == net_index; the caller makes one call per
   * distinct checkpoint (each writes its own agents' entries of ext_actions).  NULL: every live GA3C-CADRL agent. */
  const int32_t *agent_net;
  int32_t net_index, reserved0;
  /* The four big weight matrices as fp16 planes in matrix-core fragment order: device buffer of cagpu_ga3c_packed_bytes()
   * bytes (16-byte aligned), filled ONCE per checkpoint by cagpu_ga3c_pack() from the float32 arrays above (which cagpu_ga3c
   * still reads for the x_t / host inputs, the biases and the logits layer).  Required: cagpu_ga3c fails with CA_EINVAL
   * without it.  The network computes on float32 operands carried as two fp16 planes (22 significant bits); see cagpu_ga3c. */
  const void *packed;
} CaNet;

int cagpu_version(void);
const char *cagpu_last_error(void);
/* Introspection for tests / bench.py: the kernel instantiation and launch geometry the last cagpu_step / rollout /
 * reset / observe call of THIS thread selected, e.g. "ca_kernel<256, false, 10, false, true, 4> grid=1024 ...".
 * The selection depends only on the call's arguments and the device's CU count (never on the environment). */
const char *cagpu_last_kernel(void);

/* Replaces: Agent.reset (agent.py:59-138) for every agent of the envs with mask[e] != 0 (mask NULL =
 * all), in the EVALUATE_MODE form of test_cases.py:545-590 (heading toward the goal unless
 * `headings` [E,N] is given), followed by the reset observation (collision_avoidance_env.py:276-282).
 * cases: device [E,N,6] = px, py, gx, gy, pref_speed, radius; a row with radius <= 0 leaves its slot absent
 * (CA_ABSENT: ragged batches, absent slots last).  The policy / dynamics / learning bits of `flags` must already be
 * set; reset_count[e] is zeroed, CA_PLAN_VALID cleared. */
int cagpu_reset(const CaParams *p, const CaState *s, const CaOut *o, const double *cases, const double *headings,
                const uint8_t *mask, void *stream);

/* Replaces: CollisionAvoidanceEnv.step (collision_avoidance_env.py:156-234) for every env:
 * policy queries on the pre-step state (RVOPolicy / rvo2.doStep, NonCooperative, Static, external),
 * Agent.take_action + UnicycleDynamics.step + update_ego_frame, _check_for_collisions,
 * _compute_rewards, OtherAgentsStatesSensor.sense + observation assembly, _check_which_agents_done.
 * ext_actions: device float64 [E,N,2], read only for agents with an external policy; may be NULL
 * (the reference's `env.step(None)`, env_utils.py:50).  ar == NULL: no auto-reset. */
int cagpu_step(const CaParams *p, const CaState *s, const CaOut *o, const double *ext_actions, const CaAutoReset *ar,
               void *stream);

/* cagpu_step with a static map: an agent whose disc covers an occupied static cell collides with the wall
 * (collision_avoidance_env.py:494-506, :425-429).  map == NULL or map->static_bits == NULL: same as cagpu_step. */
int cagpu_step_map(const CaParams *p, const CaState *s, const CaOut *o, const double *ext_actions, const CaAutoReset *ar,
                   const CaMap *map, void *stream);

/* Replaces: Map.add_agents_to_map (Map.py:46-64) + LaserScanSensor.sense (sensors/LaserScanSensor.py:49-101) for every
 * agent of every env, on the CURRENT state (call after cagpu_reset / cagpu_step): the env's agents are rasterised as
 * discs into a copy of the static grid held in LDS, every beam is marched through it (the agent's own disc is
 * transparent), the result is the range of the last sample before the SECOND hit (the reference's
 * `cumsum == 1` indexing, LaserScanSensor.py:77-81).  An agent with step_num == 0 takes its first measurement (all
 * history rows filled, :84-85), otherwise the history is rolled (:86-88). */
int cagpu_laserscan(const CaParams *p, const CaState *s, const CaMap *map, const CaScan *scan, void *stream);

/* Replaces: GA3CCADRLPolicy.find_next_action (policies/GA3CCADRLPolicy.py:49-84) + NetworkVPCore.predict_p
 * (GA3C_CADRL/network.py:24-41, the TF1 graph of the checkpoint) for every agent whose policy is CA_POL_GA3C_CADRL and
 * that is not done: the policy vector X[138] = obs[1:] (zero-padded / cropped), (X - mean) / std, a 64-unit LSTM over
 * the first num_other_agents of the 19 other-agent slots, three 256-wide ReLU layers, logits_p; the argmax (index into
 * network.Actions, network.py:7-16) goes to ext_actions[e,n,0] (and 0 to [e,n,1]), where cagpu_step turns it into
 * [pref_speed * a0, a1] exactly as for CA_POL_LEARNING_GA3C.  obs: device float [E,N,6+7*max_obs], the observation of
 * the CURRENT state (what the reference hands to the policy, collision_avoidance_env.py:319-323) -- or NULL: FUSED SENSING,
 * the kernel computes the ego-centric observation of every agent it evaluates from the state arrays itself
 * (OtherAgentsStatesSensor.sense + the observation assembly, with p->obs_clip / sort_mode / sensing_horizon), bit-identical to
 * the stored row; needs num_agents <= 32 and closest_first / closest_last sorting.  logits (nullable):
 * device float [E,N,11], written for the same agents.  Arithmetic: float32 like the TF graph, on the F16 matrix cores
 * (v_mfma_f32_16x16x32_f16, f32 accumulate): both operands of every contraction are carried as two fp16 planes
 * (x ~ hi + lo, hi = fp16(x), lo = fp16(x - hi): 22 of 24 significant bits, fp16 denormals kept) and three of the four
 * plane products are accumulated -- a product is off by less than
 * 2^-21 of itself (an f32 multiply: 2^-24); layer1's four host inputs and the logits layer run on the exact f32 MFMA
 * (v_mfma_f32_16x16x4_f32; the LSTM does not: that instruction holds the SIMD's VALU).  Logits agree with a float32 evaluation of the graph to ~1e-6 (tests: rtol 1e-4, atol 2e-4).
 * Needs net->packed (cagpu_ga3c_pack). */
int cagpu_ga3c(const CaParams *p, const CaState *s, const float *obs, const CaNet *net, double *ext_actions,
               float *logits, void *stream);

/* The size of CaNet.packed, and the one-time split of a checkpoint's weights into it: reads net->lstm_kernel,
 * layer1_kernel, layer2_kernel, fc1_kernel (device float32, the checkpoint's [in, out] layout) and writes `bytes` =
 * cagpu_ga3c_packed_bytes() bytes at `packed` (device, 16-byte aligned); the LSTM kernel's columns are stored multiplied (in
 * float32) by -log2 e (gates i, f, o) / -2 log2 e (gate j): the kernel evaluates the gates as 1 / (1 + 2^z).  Replaces nothing in the reference (TF keeps its
 * variables in one layout); it is this library's equivalent of GA3CCADRLPolicy.initialize_network's checkpoint restore
 * (GA3CCADRLPolicy.py:23-47).  Call again after changing a weight array. */
uint64_t cagpu_ga3c_packed_bytes(void);
int cagpu_ga3c_pack(const CaNet *net, void *packed, uint64_t bytes, void *stream);

/* Replaces: generate_rand_test_case_multi (envs/policies/CADRL/scripts/multi/gen_rand_testcases.py:111-444) behind
 * test_cases.get_testcase_random (envs/test_cases.py:212-253), for num_cases scenarios at once: 15 % two-agent swap +
 * circle, 15 % circle, 70 % rejection-sampled starts / goals in a square of half side `side` (drawn per case from
 * [side_lo, side_hi] when side_hi > side_lo) that grows 1 % per attempt.  cases: device float64 [num_cases, num_agents, 6]
 * = px, py, gx, gy, pref_speed, radius -- the layout cagpu_reset and CaAutoReset.table take.  Randomness is
 * counter-based (Philox4x32-10 keyed by `seed`, counter = (draw, case index)): the same (seed, case index) gives the
 * same scenario whatever num_cases is.  status: device int32 [num_cases] or NULL (0 = every agent was accepted by the
 * reference's rules; 1 = an agent hit the attempt cap and was placed anyway). */
int cagpu_generate_cases(int64_t num_cases, int32_t num_agents, double side_lo, double side_hi, double speed_lo,
                         double speed_hi, double radius_lo, double radius_hi, uint64_t seed, double *cases, int32_t *status,
                         void *stream);

/* The same generator for RAGGED tables -- test_cases.get_testcase_random with num_agents=None and a side_length list
 * (envs/test_cases.py:224-241, the reference's default TEST_CASE_ARGS, config.py:118-131): the agent count of a case is
 * drawn first, uniform over n_min .. n_max (np.random.randint(2, MAX_NUM_AGENTS_IN_ENVIRONMENT + 1)), then the side
 * length from every entry of side_ranges (HOST float64 [n_ranges, 4] = count lo, count hi (exclusive), side lo, side hi;
 * n_ranges <= 8) that holds the count; every count in [n_min, n_max] must be held by an entry (the reference asserts it).
 * cases: device float64 [num_cases, max_agents, 6]; rows past the drawn count are zero (radius 0 = an empty slot of a
 * ragged batch, CaParams.ragged).  counts: device int32 [num_cases] or NULL. */
int cagpu_generate_cases_ragged(int64_t num_cases, int32_t max_agents, int32_t n_min, int32_t n_max,
                                const double *side_ranges, int32_t n_ranges, double speed_lo, double speed_hi,
                                double radius_lo, double radius_hi, uint64_t seed, double *cases, int32_t *counts,
                                int32_t *status, void *stream);

/* n_steps consecutive cagpu_step calls fused into ONE launch (every step still writes its
 * outputs; the buffers hold the last step's).  Envs never interact, so no grid-wide sync is
 * needed.  This is the batched form of env_utils.py:45-52 `while not terminated: env.step(None)`.
 * ext_actions (if any) are held constant over the n_steps.  The per-step inputs CaState.rvo_collab / rvo_heading_noise /
 * ext_state belong to the ONE step that consumes them: with n_steps > 1 any of them set is CA_EINVAL. */
int cagpu_rollout(const CaParams *p, const CaState *s, const CaOut *o, const double *ext_actions,
                  const CaAutoReset *ar, int32_t n_steps, void *stream);

/* cagpu_rollout whose every step KEEPS its outputs: the look-ahead ring behind `env.step(None)`.  With every policy internal
 * the reference's `env.step(None)` needs no input from the host (experiments/src/env_utils.py:45-52: `run_episode` passes
 * None until the episode is over), so the next n_steps steps can be computed in one launch and handed out one by one -- but
 * a gym caller wants the outputs of EVERY step, not only the last one's.  Here the output pointers of `o` name slot 0 of a
 * ring of n_steps slots and step t of the call (t = 0 .. n_steps - 1) writes slot t:
 *   o->obs [n_steps, E, N, 6+7K], o->rewards [n_steps, E, N], o->done [n_steps, E, N], o->game_over [n_steps, E],
 *   o->actions / o->orca_vel (if given) [n_steps, E, N, 2].
 * State, statistics and auto-resets are those of cagpu_rollout(n_steps) -- i.e. of n_steps cagpu_step calls, bit for bit
 * (tests/test_gpu_ring.py).  Like cagpu_rollout it takes no per-step inputs (CaState.rvo_collab / rvo_heading_noise /
 * ext_state must be NULL: CA_EINVAL; ext_actions, if any, are held constant).
 * snapshot_delta (bytes; 0 = none): the REWIND POINT.  A caller that runs ahead must be able to go back (an action arrives
 * for step t < n_steps: restore the state the call started from, cagpu_rollout(t), go on one step at a time).  With all state
 * arrays of `s` in ONE allocation and a second allocation of the same layout snapshot_delta bytes away, the kernel itself
 * stores every state element it loads at its start -- all of `s` that a step reads or writes, env_stats included -- at
 * (its address + snapshot_delta): when the call has run, the second allocation holds the state BEFORE the call.  Only the
 * pipelined n-step kernel does this (cagpu_ring_snapshots() says whether a call with these arguments would); otherwise
 * CA_EUNSUPPORTED and the caller copies the state itself ahead of the call. */
int cagpu_rollout_ring(const CaParams *p, const CaState *s, const CaOut *o, const double *ext_actions,
                       const CaAutoReset *ar, int32_t n_steps, int64_t snapshot_delta, void *stream);
/* 1: cagpu_rollout_ring with these arguments runs the kernel that takes the snapshot itself (snapshot_delta != 0 accepted);
 * 0: it does not; < 0: the arguments are invalid (CA_E*).  Host-only, launches nothing. */
int cagpu_ring_snapshots(const CaParams *p, const CaState *s, const CaOut *o, const CaAutoReset *ar, int32_t n_steps);

/* The policy query of the NEXT step ahead of time (collision_avoidance_env.py:305-323 for the built-in RVO policy):
 * fills s->next_action from the CURRENT state and sets CA_PLAN_VALID, without stepping.  cagpu_step / cagpu_rollout keep
 * the plan up to date by themselves; this entry point exists for states that did not come out of a step (the reset state
 * of a fixture table -> CaAutoReset.reset_plan).  Requires what the pipelined step kernel requires: s->next_action,
 * num_agents in {2, 3, 4, 5, 6, 8, 10}, closest_first sorting (CA_EUNSUPPORTED otherwise: the step kernels then query the
 * policy at the start of the step).  A plan is computed under the CaParams of THIS call (rvo_time_horizon,
 * rvo_collab_coeff, rvo_dt, sensing_horizon, rvo_max_neighbors): whoever changes one of them afterwards must clear
 * CA_PLAN_VALID in the flag words and recompute CaAutoReset.reset_plan / reset_obs -- the library cannot see that a
 * parameter differs from the one a stored plan was made with. */
int cagpu_plan(const CaParams *p, const CaState *s, void *stream);

/* Replaces: rvo2.PyRVOSimulator.doStep() + getAgentVelocity for every agent (call sites
 * RVOPolicy.py:25-28,70-74,86-93): one ORCA velocity per agent from C-float inputs.
 * pos/vel/pref: device float [E,N,2]; radius/max_speed: device float [E,N]; new_vel: device float [E,N,2]. */
int cagpu_orca(int32_t num_envs, int32_t num_agents, const float *pos, const float *vel, const float *pref,
               const float *radius, const float *max_speed, float collab_coeff, float time_horizon, float time_step,
               int32_t max_neighbors, float neighbor_dist, float *new_vel, void *stream);

/* Replaces: OtherAgentsStatesSensor.sense + the observation assembly (OtherAgentsStatesSensor.py:58-144,
 * agent.py:323-327) for the CURRENT state, without stepping: rewrites o->obs only. */
int cagpu_observe(const CaParams *p, const CaState *s, const CaOut *o, void *stream);

/* Bytes of CaOut.workspace the step / reset / observe / rollout calls want for these parameters: 0 up to 64 agents per env,
 * otherwise one share (60 B x threads x num_agents; threads = 256 / 512 / 1024 for up to 256 / 512 / 1024 agents) per
 * workgroup of the large-env kernel, for min(num_envs, 2 x CUs) workgroups (1 x CUs above 256 agents); a smaller workspace
 * works too: fewer workgroups walk the envs.  Host-only call. */
uint64_t cagpu_workspace_bytes(const CaParams *p);

/* Device-side fault word of the CURRENT device (synchronises it): bit 0 = a bounded hand-over poll inside the pipelined
 * step kernel ran out (csrc/cagpu_pipe.inc wait_for), i.e. some launch since the last clear may have produced wrong state.
 * bit 1 = an operand of the GA3C-CADRL network kernel left the range of its two-plane fp16 split (|x| >= 65504: a normalised
 * input or an activation; csrc/cagpu_ga3c.inc), i.e. some cagpu_ga3c call since the last clear chose its actions from
 * saturated values.
 * *faults receives the word; clear != 0 resets it.  0 in normal operation; check it wherever the host synchronises anyway. */
int cagpu_device_faults(uint32_t *faults, int32_t clear);

/* The same word WITHOUT a synchronisation (v11): queues a 4-byte copy into *host_dst -- PINNED host memory of the caller --
 * on `stream`, behind the work already submitted there; the caller reads *host_dst once an event recorded behind this call
 * has completed.  Does not clear.  For the product path of a caller that never synchronises (the look-ahead ring of
 * core.BatchedSim: one probe per refill). */
int cagpu_device_faults_async(uint32_t *host_dst, void *stream);

/* Parity hook (tests only, synchronous, HOST pointers, default stream): evaluates on the device, element by element, the
 * operations through which a step's results can differ from a CPU run of the same algorithm -- no reference analogue:
 *   op 0: out0 = atan2(a, b)            (ROCm's libm; RVOPolicy.py:100, Dynamics.py:36, test_cases.py:554)
 *   op 1: out0, out1 = sin a, cos a     (the step kernels' short-range kernel for a heading in [-pi, pi]; UnicycleDynamics.py:30-35)
 *   op 2 / 3: out0 = the lean float divide a / b / square root of a used inside the ORCA phases, out1 = the correctly rounded one
 *   op 4 / 5: the same for the float64 divide / square root (distances, preferred velocity, ego frame)
 *   op 6: out0, out1 = heading_ego_frame, dist_to_goal of an agent at the origin with heading 0 and goal (a, b)
 * tests/test_gpu_bench_geometry.py runs the CPU oracle on ops 0 / 1 to show that they are the ONLY difference (free-running
 * episodes then agree bit for bit); tests/test_gpu_parity.py pins the operand range in which ops 2 - 5 agree. */
int cagpu_debug_libm(int32_t op, int32_t n, const double *a, const double *b, double *out0, double *out1);

/* Measurement hook (profiles/ only, no reference analogue): dst[i] = src[i] for n float64 elements with the access shape of
 * the step kernels' state loads -- ONE 8-byte element per lane and instruction (global_load_dwordx2 / global_store_dwordx2,
 * 64 consecutive elements per wavefront), device pointers, asynchronous on `stream`.  Its traffic is known exactly (8 n
 * bytes read, 8 n written), which is what the rocprofv3 FETCH_SIZE / WRITE_SIZE counters of the step kernels are calibrated
 * against (profiles/r05_traffic.json: the counters under-report this pattern on gfx950). */
int cagpu_debug_copy8(int64_t n, const double *src, double *dst, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CAGPU_H_ */
