/* oracle/ca_oracle.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * C API of the CPU restatement of the reference's per-step hot path
 * (gym_collision_avoidance/envs/collision_avoidance_env.py:156-234 `step`, :236-282 `reset`).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline -- never as the product path.
 *
 * Everything is float64 like the reference's Python, except (a) the ORCA arithmetic, which is
 * C float like the rvo2 library (orca_ref.h; PARITY UNPINNED, see that header), and (b) the
 * action pair, which the reference rounds to float32 (`all_actions` is a float32 array,
 * collision_avoidance_env.py:305-307).
 */
#ifndef ORACLE_CA_ORACLE_H_
#define ORACLE_CA_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flag word (one per agent) */
enum {
  ORC_AT_GOAL = 1u << 0,          /* agent.is_at_goal               (agent.py:150-153) */
  ORC_WAS_AT_GOAL = 1u << 1,      /* agent.was_at_goal_already      (agent.py:203-204) */
  ORC_IN_COLLISION = 1u << 2,     /* agent.in_collision             (env.py:421-424)   */
  ORC_WAS_IN_COLLISION = 1u << 3, /* agent.was_in_collision_already (agent.py:205-206) */
  ORC_OUT_OF_TIME = 1u << 4,      /* agent.ran_out_of_time          (agent.py:238-239) */
  ORC_DONE = 1u << 5,             /* agent.is_done                  (env.py:534-535)   */
  ORC_IS_LEARNING = 1u << 6,      /* policy.str == "learning"       (config.py:152-157) */
  ORC_STILL_LEARNING = 1u << 7,   /* policy.is_still_learning       (Policy.py:13)     */
  ORC_ABSENT = 1u << 16           /* ragged batches (OrcParams.ragged): this SLOT holds no agent in the env's episode -- the
                                     reference's agent list is shorter than MAX_NUM_AGENTS_IN_ENVIRONMENT (test_cases.py:224-227) */
};
/* policy ids (test_cases.py:68-85 registry) */
enum { ORC_POL_RVO = 0, ORC_POL_NONCOOP = 1, ORC_POL_STATIC = 2, ORC_POL_EXTERNAL = 3, ORC_POL_LEARNING = 4, ORC_POL_LEARNING_GA3C = 5,
       ORC_POL_GA3C_CADRL = 6 /* index computed by oracle/ga3c_ref.py, handed in through ext like LEARNING_GA3C */ };
/* dynamics ids (test_cases.py:93-96 + UnicycleDynamicsMaxTurnRate.py) */
enum { ORC_DYN_UNICYCLE = 0, ORC_DYN_MAX_TURN_RATE = 1, ORC_DYN_EXTERNAL = 2 };
/* agent_sorting_method (OtherAgentsStatesSensor.py:34-52) */
enum { ORC_SORT_CLOSEST_FIRST = 0, ORC_SORT_CLOSEST_LAST = 1, ORC_SORT_TIME_TO_IMPACT = 2 };
/* game_over rule (env.py:537-551) */
enum { ORC_OVER_ALL_DONE = 0 /* EVALUATE_MODE */, ORC_OVER_AGENT0 = 1 /* TRAIN_SINGLE_AGENT */, ORC_OVER_LEARNING_DONE = 2 };

typedef struct {
  int32_t num_envs, num_agents, max_obs /* K: rows of the obs array */, sort_mode, game_over_mode, rvo_max_neighbors;
  int32_t obs_clip /* sensor.max_num_other_agents_observed <= K */;
  int32_t ragged /* != 0: a reset row with radius <= 0 leaves its slot empty (ORC_ABSENT) */;
  double dt, near_goal_threshold, max_time_ratio, getting_close_range, sensing_horizon;
  double reward_at_goal, reward_collision, reward_time_step, reward_wiggly, wiggly_threshold;
  double reward_min, reward_max; /* np.clip bounds, env.py:589-599 */
  double rvo_time_horizon, rvo_collab_coeff;
  double max_heading_change; /* env-wide pi/3, env.py:87, used by LearningPolicy.py:30 */
  double reward_collision_wall; /* config.py:33, env.py:425-429 */
  double rvo_dt;             /* RVOPolicy.py:13: Config.DT (timeStep of rvo2 and the 1/dt of :106) */
} OrcParams;

/* SoA state, index e*num_agents + a */
typedef struct {
  double *pos_x, *pos_y, *vel_x, *vel_y, *heading, *goal_x, *goal_y, *radius, *pref_speed;
  double *time_remaining, *t, *slt /* straight_line_time_to_reach_goal */, *ep_reward;
  double *turning_dir; /* Agent.turning_dir (agent.py:133; UnicycleDynamics.py:41-47) */
  float *last_action;  /* [.,2] past_actions[0] (agent.py:212-213) */
  uint32_t *flags;
  int32_t *policy, *dynamics, *step_num;
  /* per env */
  int32_t *episode_step, *reset_count;
  double *env_stats; /* [E,8]: episodes, collision_eps, all_at_goal_eps, stuck_eps, sum steps,
                        sum total_reward, sum time_to_goal, sum extra_time_to_goal (env_utils.py:56-87) */
  /* per-step inputs of RVOPolicy's stochastic branches, both nullable (RVOPolicy.py:77-90, :118-119; include/cagpu.h) */
  const float *rvo_collab;          /* [E*N] the collaboration coefficient of each agent as the ego of its query */
  const double *rvo_heading_noise;  /* [E*N] added to an RVO agent's delta heading after the pi/6 clip */
  const double *ext_state;          /* [E*N,5] px, py, vx, vy, heading a DYN_EXTERNAL agent takes at the move (NaN row: none) */
} OrcState;

typedef struct {
  double *obs;       /* [E,N,6+7K]: is_learning,num_other_agents,dist_to_goal,heading_ego_frame,pref_speed,radius,K x 7 */
  double *rewards;   /* [E,N] */
  uint8_t *done;     /* [E,N] which_agents_done */
  uint8_t *game_over;/* [E] */
  float *actions;    /* [E,N,2] the float32 all_actions array (debug / parity of the policy stage) */
  float *orca_vel;   /* [E,N,2] or NULL: the velocity rvo2 chose for every agent whose RVOPolicy was queried (0 otherwise) */
} OrcOut;

/* Static map + LaserScanSensor (Map.py:6-64, sensors/LaserScanSensor.py:24-101; env.py:494-506 wall collisions).
 * static_map: row-major uint8 [rows*cols] (1 = occupied) or NULL (all free). */
typedef struct {
  const uint8_t* static_map;
  int32_t rows, cols;
  double cell, origin_r, origin_c;
} OrcMap;
typedef struct {
  uint8_t* hist;  /* [E,N,num_to_store,num_beams] range index per beam (255 = no hit) */
  double* out;    /* [E,N,num_to_store,num_beams] meters */
  int32_t num_beams, num_to_store, num_ranges;
  double min_angle, max_angle, range_res, max_range;
} OrcScan;
/* laserscan observation of the CURRENT state (call after reset / step); an agent with step_num == 0 takes its
 * first measurement (all history rows filled), otherwise the history is rolled. */
int ca_oracle_laserscan(const OrcParams* p, const OrcState* s, const OrcMap* m, const OrcScan* sc);
/* env.step with a static map: wall collisions enter the reward / in_collision logic. */
int ca_oracle_step_map(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* ext_actions, const OrcMap* m);

int ca_oracle_version(void);

/* (Re)initialise the envs with mask[e]!=0 (mask NULL = all) from cases[e][a][6] =
 * px,py,gx,gy,pref_speed,radius (test_cases.py:545-557, EVALUATE_MODE heading = toward goal;
 * agent.py:59-138) and write their reset observation (env.py:276-282). headings may be NULL. */
int ca_oracle_reset(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* cases, const double* headings,
                    const uint8_t* mask);

/* One env.step for every env (no auto-reset). ext_actions [E,N,2] float64 may be NULL. */
int ca_oracle_step(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* ext_actions);

/* n_steps of step + DummyVecEnv-style auto-reset (vec_env.py:120-128) from a fixture table
 * table[n_cases][N][6]: env e's k-th reset loads case (env_id_offset + e + k*case_stride) % n_cases;
 * episode statistics are accumulated into env_stats at each game_over. */
int ca_oracle_rollout(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* table, int32_t n_cases,
                      int64_t env_id_offset, int64_t case_stride, int32_t n_steps);

/* ca_oracle_rollout with external actions (held constant over the n_steps; NULL = none) and a static map (NULL = none):
 * the auto-reset loop for the GA3C-CADRL and map workloads (tests at the bench geometries). */
int ca_oracle_rollout_ex(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* ext_actions, const double* table,
                         int32_t n_cases, int64_t env_id_offset, int64_t case_stride, int32_t n_steps, const OrcMap* map);

/* Route the step's three libm-dependent operations (atan2; cos / sin of the new heading) through the caller's functions
 * (NULL = glibc's): lets a test run the oracle on another libm's bits.  Process-wide. */
void ca_oracle_set_libm(double (*atan2_fn)(double, double), void (*sincos_fn)(double, double*, double*));

/* Stand-alone pieces (same arithmetic as inside step), for per-stage parity tests. */
/* ORCA: one new velocity per agent from float inputs (rvo2 doStep for every agent of every env). */
int ca_oracle_orca(int32_t num_envs, int32_t num_agents, const float* pos, const float* vel, const float* pref,
                   const float* radius, const float* max_speed, float collab, float time_horizon, float time_step,
                   int32_t max_neighbors, float neighbor_dist, float* new_vel);
/* numpy-style round(x, 2) key used by the sensor sort (OtherAgentsStatesSensor.py:107). */
double ca_oracle_round2(double x);

#ifdef __cplusplus
}
#endif
#endif
