// oracle/ca_oracle.cpp -- TEST INFRASTRUCTURE (CPU oracle), not product code.  See ca_oracle.h.
//
// Plain single-threaded C++ restatement of one `CollisionAvoidanceEnv.step` / `.reset`, written
// from the reference's Python (file:line cited at each stage; paths relative to
// /root/reference/gym_collision_avoidance/envs/).  float64 throughout, in the reference's
// operation order; build with -ffp-contract=off.  Pinned against the reference itself: the golden
// vectors under tests/golden/ are produced by oracle/gen_golden.py, which runs the UNMODIFIED
// reference Python (with the rvo2 module from oracle/rvo2_module/), and tests/test_oracle_golden.py
// replays them through this file.  The ORCA stage (orca_ref.h) is self-pinned -- "parity
// unpinned" -- because the reference's rvo2 submodule is empty.
#include "ca_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "orca_ref.h"

namespace {

const double kPi = 3.141592653589793;  // np.pi
const double kTwoPi = 2.0 * kPi;

// ---- libm hooks (tests only; ca_oracle_set_libm).  The three libm-dependent operations of the step -- atan2 (policy
// heading RVOPolicy.py:100, ego frame Dynamics.py:36, reset heading test_cases.py:554) and cos / sin of the new heading
// (UnicycleDynamics.py:30-35) -- go through these pointers, so that a test can run the oracle on ANOTHER libm's results
// (the GPU's: tests/test_gpu_bench_geometry.py::test_swap_cases_agree_on_the_gpus_libm_bits).  Default: glibc, like numpy.
typedef double (*Atan2Fn)(double, double);
typedef void (*SinCosFn)(double, double*, double*);
Atan2Fn g_atan2 = nullptr;
SinCosFn g_sincos = nullptr;
inline double m_atan2(double y, double x) { return g_atan2 ? g_atan2(y, x) : std::atan2(y, x); }
inline void m_sincos(double a, double& sn, double& cs) {
  if (g_sincos) g_sincos(a, &sn, &cs);
  else { cs = std::cos(a); sn = std::sin(a); }
}

// util.py:141-146
inline double wrap(double a) {
  while (a >= kPi) a -= kTwoPi;
  while (a < -kPi) a += kTwoPi;
  return a;
}

// numpy float64 `%` (Python-sign remainder), RVOPolicy.py:102
inline double pymod(double a, double b) {
  double m = std::fmod(a, b);
  if (m != 0.0) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = std::copysign(0.0, b);
  }
  return m;
}

inline double sgn(double x) { return (x > 0.0) - (x < 0.0); }  // np.sign

// round(np.float64, 2): numpy's scalar __round__ = rint(x*100)/100 (NOT CPython's correctly
// rounded decimal algorithm).  dist_2_other is an np.float64 whenever radius comes from a fixture
// array (test_cases.py:549-550), which is the benchmark path.  OtherAgentsStatesSensor.py:107
inline double round2(double x) { return std::nearbyint(x * 100.0) / 100.0; }

struct Ego {
  double dist_to_goal, prll_x, prll_y, orth_x, orth_y, heading_ego;
};

// agent.py:329-349 get_ref + dynamics/Dynamics.py:24-41 update_ego_frame (heading part)
inline Ego ego_frame(double px, double py, double gx, double gy, double heading) {
  Ego e;
  const double dx = gx - px, dy = gy - py;
  e.dist_to_goal = std::sqrt(dx * dx + dy * dy);
  if (e.dist_to_goal > 1e-8) {
    e.prll_x = dx / e.dist_to_goal;
    e.prll_y = dy / e.dist_to_goal;
  } else {
    e.prll_x = dx;
    e.prll_y = dy;
  }
  e.orth_x = -e.prll_y;
  e.orth_y = e.prll_x;
  e.heading_ego = wrap(heading - m_atan2(e.prll_y, e.prll_x));
  return e;
}

struct Cand {
  int j;
  double key, p_orth, tti;
};

// Ragged batches: the reference's env holds len(self.agents) <= MAX_NUM_AGENTS_IN_ENVIRONMENT agents and every loop of its
// step runs over that list (env.py:345-367, test_cases.py:224-227); here an env is num_agents SLOTS and the slots without an
// agent carry ORC_ABSENT.  `present` lists the slots that hold one, in slot (= list index) order.
inline bool absent(const OrcParams& p, const OrcState& s, int i) { return p.ragged && (s.flags[i] & ORC_ABSENT); }
inline std::vector<int> present(const OrcParams& p, const OrcState& s, int e) {
  std::vector<int> v;
  for (int a = 0; a < p.num_agents; ++a)
    if (!absent(p, s, e * p.num_agents + a)) v.push_back(a);
  return v;
}

// util.py:101-127 tangent_vecs_from_external_pt + util.py:23-83 compute_time_to_impact (float64, numpy order)
inline double time_to_impact(double hx, double hy, double ox, double oy, double hvx, double hvy, double ovx, double ovy,
                             double r) {
  const double v0 = hvx - ovx, v1 = hvy - ovy;  // v_rel
  const double xp = hx, yp = hy, a = ox, b = oy;
  const double sq = (xp - a) * (xp - a) + (yp - b) * (yp - b) - r * r;
  if (sq < 0) return 0.0;  // already inside the collision zone
  const double st = std::sqrt(sq);
  const double xnum1 = r * r * (xp - a), xnum2 = r * (yp - b) * st;
  const double ynum1 = r * r * (yp - b), ynum2 = r * (xp - a) * st;
  const double den = (xp - a) * (xp - a) + (yp - b) * (yp - b);
  const double c1x = ((xnum1 + xnum2) / den + a) - xp, c1y = ((ynum1 - ynum2) / den + b) - yp;
  const double c2x = ((xnum1 - xnum2) / den + a) - xp, c2y = ((ynum1 + ynum2) / den + b) - yp;
  const double x11 = c1x * v1 - c1y * v0, x12 = c1x * c2y - c1y * c2x;  // np.cross
  const double x21 = c2x * v1 - c2y * v0, x22 = c2x * c1y - c2y * c1x;
  const double inf = std::numeric_limits<double>::infinity();
  if (!(x11 * x12 >= 0 && x21 * x22 >= 0)) return inf;
  if (std::fabs(v0) < 1e-5 && std::fabs(v1) < 1e-5) return inf;
  const double px = hx, py = hy;
  double x1, x2, y1, y2;
  if (std::fabs(v0) < 1e-5) {  // vertical v_rel
    x1 = x2 = px;
    const double A = 1, B = -2 * b, Cc = b * b + (px - a) * (px - a) - r * r;
    y1 = (-B + std::sqrt(B * B - 4 * A * Cc)) / (2 * A);
    y2 = (-B - std::sqrt(B * B - 4 * A * Cc)) / (2 * A);
  } else {
    const double m = v1 / v0;
    const double A = 1 + m * m;
    const double B = -2 * a + 2 * m * (py - b - m * px);
    const double Cc = a * a - r * r + (m * px - (py - b)) * (m * px - (py - b));
    x1 = (-B + std::sqrt(B * B - 4 * A * Cc)) / (2 * A);
    x2 = (-B - std::sqrt(B * B - 4 * A * Cc)) / (2 * A);
    y1 = m * (x1 - px) + py;
    y2 = m * (x2 - px) + py;
  }
  const double d1 = std::sqrt((x1 - px) * (x1 - px) + (y1 - py) * (y1 - py));
  const double d2 = std::sqrt((x2 - px) * (x2 - px) + (y2 - py) * (y2 - py));
  const double d = d2 < d1 ? d2 : d1;  // min(d1, d2)
  return d / std::sqrt(v0 * v0 + v1 * v1);
}

// OtherAgentsStatesSensor.py:58-144 for host `a` of env `e` -> one obs row, written at `row`
void sense(const OrcParams& p, const OrcState& s, int e, int a, double* row) {
  const int N = p.num_agents, K = p.max_obs;
  const int base = e * N, h = base + a;
  if (absent(p, s, h)) {  // no agent in this slot: the zero row of wrappers.py:143-173
    std::memset(row, 0, sizeof(double) * (6 + 7 * K));
    return;
  }
  const Ego eg = ego_frame(s.pos_x[h], s.pos_y[h], s.goal_x[h], s.goal_y[h], s.heading[h]);
  std::vector<Cand> c;
  c.reserve(N);
  for (int j = 0; j < N; ++j) {
    if (j == a) continue;  // :79 (ids are the list indices in every builder of test_cases.py)
    const int o = base + j;
    if (absent(p, s, o)) continue;
    const double rx = s.pos_x[o] - s.pos_x[h], ry = s.pos_y[o] - s.pos_y[h];
    const double p_orth = rx * eg.orth_x + ry * eg.orth_y;
    const double dc = std::sqrt(rx * rx + ry * ry);  // util.py:148-153
    const double d2o = dc - s.radius[h] - s.radius[o];
    if (dc > p.sensing_horizon) continue;  // :90
    double tti = 0.0;
    if (p.sort_mode == ORC_SORT_TIME_TO_IMPACT)  // :96-104
      tti = time_to_impact(s.pos_x[h], s.pos_y[h], s.pos_x[o], s.pos_y[o], s.vel_x[h], s.vel_y[h], s.vel_x[o],
                           s.vel_y[o], s.radius[h] + s.radius[o]);
    Cand k = {j, round2(d2o), p_orth, tti};
    c.push_back(k);
  }
  // :34-52
  if (p.sort_mode == ORC_SORT_TIME_TO_IMPACT) {  // key (-tti, -dist, p_orth) for the clip and for the final order
    std::stable_sort(c.begin(), c.end(), [](const Cand& x, const Cand& y) {
      if (-x.tti != -y.tti) return -x.tti < -y.tti;
      if (-x.key != -y.key) return -x.key < -y.key;
      return x.p_orth < y.p_orth;
    });
  } else {
    std::stable_sort(c.begin(), c.end(), [](const Cand& x, const Cand& y) {
      if (x.key != y.key) return x.key < y.key;
      return x.p_orth < y.p_orth;
    });
  }
  const int clip = p.obs_clip < K ? p.obs_clip : K;  // :39 clip to the sensor's own limit; rows stay K (:112)
  if (static_cast<int>(c.size()) > clip) c.resize(clip);
  if (p.sort_mode == ORC_SORT_CLOSEST_LAST) {
    std::stable_sort(c.begin(), c.end(), [](const Cand& x, const Cand& y) {
      if (-x.key != -y.key) return -x.key < -y.key;
      return x.p_orth < y.p_orth;
    });
  }
  const uint32_t f = s.flags[h];
  row[0] = (f & ORC_IS_LEARNING) ? 1.0 : 0.0;
  row[1] = static_cast<double>(c.size());  // num_other_agents_observed :141
  row[2] = eg.dist_to_goal;
  row[3] = eg.heading_ego;
  row[4] = s.pref_speed[h];
  row[5] = s.radius[h];
  double* oa = row + 6;
  std::memset(oa, 0, sizeof(double) * 7 * K);  // :112
  for (size_t n = 0; n < c.size(); ++n) {
    const int o = base + c[n].j;
    const double rx = s.pos_x[o] - s.pos_x[h], ry = s.pos_y[o] - s.pos_y[h];
    double* r = oa + 7 * n;
    r[0] = rx * eg.prll_x + ry * eg.prll_y;
    r[1] = rx * eg.orth_x + ry * eg.orth_y;
    r[2] = s.vel_x[o] * eg.prll_x + s.vel_y[o] * eg.prll_y;
    r[3] = s.vel_x[o] * eg.orth_x + s.vel_y[o] * eg.orth_y;
    r[4] = s.radius[o];
    r[5] = s.radius[h] + s.radius[o];
    r[6] = std::sqrt(rx * rx + ry * ry) - s.radius[h] - s.radius[o];
  }
}

void observe_env(const OrcParams& p, const OrcState& s, const OrcOut& o, int e) {
  const int N = p.num_agents, W = 6 + 7 * p.max_obs;
  for (int a = 0; a < N; ++a) sense(p, s, e, a, o.obs + (static_cast<size_t>(e) * N + a) * W);
}

// test_cases.py:545-590 (EVALUATE_MODE) + agent.py:59-138
void reset_env(const OrcParams& p, const OrcState& s, int e, const double* cs /* [N][6] */, const double* hd) {
  const int N = p.num_agents;
  for (int a = 0; a < N; ++a) {
    const int i = e * N + a;
    const double* c = cs + 6 * a;
    s.pos_x[i] = c[0];
    s.pos_y[i] = c[1];
    s.goal_x[i] = c[2];
    s.goal_y[i] = c[3];
    s.pref_speed[i] = c[4];
    s.radius[i] = c[5];
    s.vel_x[i] = s.vel_y[i] = 0.0;
    s.heading[i] = hd ? hd[a] : m_atan2(c[3] - c[1], c[2] - c[0]);  // test_cases.py:554-556
    const double dx = c[0] - c[2], dy = c[1] - c[3];
    s.slt[i] = (std::sqrt(dx * dx + dy * dy) - p.near_goal_threshold) / c[4];  // agent.py:99
    double tr = p.max_time_ratio * s.slt[i];
    if (p.dt > tr) tr = p.dt;  // agent.py:104
    s.time_remaining[i] = tr;
    s.t[i] = 0.0;
    s.ep_reward[i] = 0.0;
    s.turning_dir[i] = 0.0;  // agent.py:133
    s.last_action[2 * i] = s.last_action[2 * i + 1] = 0.f;
    s.step_num[i] = 0;
    s.flags[i] &= (ORC_IS_LEARNING | ORC_STILL_LEARNING);
    if (p.ragged && !(c[5] > 0.0)) {  // padding row of a ragged table: an empty slot
      s.pos_x[i] = s.pos_y[i] = s.goal_x[i] = s.goal_y[i] = s.radius[i] = s.heading[i] = s.slt[i] = s.time_remaining[i] = 0.0;
      s.pref_speed[i] = 1.0;
      s.flags[i] |= ORC_ABSENT | ORC_DONE | ORC_AT_GOAL | ORC_WAS_AT_GOAL;
    }
  }
  s.episode_step[e] = 0;
}

// Map.py:26-32 world_coordinates_to_map_indices
inline void to_cell(const OrcMap& m, double x, double y, long& gr, long& gc, bool& in_map) {
  gr = static_cast<long>(std::floor(m.origin_r - y / m.cell));
  gc = static_cast<long>(std::floor(m.origin_c + x / m.cell));
  in_map = gr >= 0 && gc >= 0 && gr < m.rows && gc < m.cols;
}

// env.py:494-506: does the agent's disc (Map.py:54-58 get_agent_map_indices) cover an occupied static cell?
bool hits_wall(const OrcMap& m, double x, double y, double radius) {
  if (!m.static_map) return false;
  long gr, gc;
  bool in_map;
  to_cell(m, x, y, gr, gc, in_map);
  if (!in_map) return false;
  const double rr = (radius / m.cell) * (radius / m.cell);
  for (long r = 0; r < m.rows; ++r)
    for (long c = 0; c < m.cols; ++c) {
      const double dc = static_cast<double>(c - gc), dr = static_cast<double>(r - gr);
      if (dc * dc + dr * dr < rr && m.static_map[r * m.cols + c]) return true;
    }
  return false;
}

void step_env(const OrcParams& p, const OrcState& s, const OrcOut& o, const double* ext, int e, const OrcMap* map = nullptr) {
  const int N = p.num_agents;
  const int b = e * N;
  s.episode_step[e] += 1;  // env.py:183

  // ---- 1. actions from the PRE-step state (env.py:305-323) ----
  std::vector<float> act(2 * N, 0.f), ovel(2 * N, 0.f);
  std::vector<orca_ref::Body> body;
  bool have_bodies = false;
  const std::vector<int> P = present(p, s, e);  // the env's agent list
  const int NP = static_cast<int>(P.size());
  for (int ia = 0; ia < NP; ++ia) {
    const int a = P[ia];
    const int i = b + a;
    if (s.flags[i] & ORC_DONE) continue;  // env.py:311
    double spd = 0.0, dh = 0.0;
    switch (s.policy[i]) {
      case ORC_POL_RVO: {  // policies/RVOPolicy.py:50-122
        if (!have_bodies) {  // :57-74 (identical for every ego agent of this env: same pre-step state)
          body.resize(NP);
          for (int j = 0; j < NP; ++j) {
            const int q = b + P[j];
            const double vx = s.goal_x[q] - s.pos_x[q], vy = s.goal_y[q] - s.pos_y[q];
            const double sc = s.pref_speed[q] / std::sqrt(vx * vx + vy * vy);  // :67
            body[j].pos = orca_ref::mk(static_cast<float>(s.pos_x[q]), static_cast<float>(s.pos_y[q]));
            body[j].vel = orca_ref::mk(static_cast<float>(s.vel_x[q]), static_cast<float>(s.vel_y[q]));
            body[j].pref = orca_ref::mk(static_cast<float>(sc * vx), static_cast<float>(sc * vy));
            body[j].radius = static_cast<float>((1 + 5e-2) * s.radius[q]);  // :71
            body[j].max_speed = static_cast<float>(s.pref_speed[q]);        // :70
            body[j].collab = s.rvo_collab ? s.rvo_collab[q] : static_cast<float>(p.rvo_collab_coeff);  // :86-90 (only the ego's is used)
          }
          have_bodies = true;
        }
        const float ts = static_cast<float>(p.rvo_dt);  // RVOPolicy.py:13,26
        const orca_ref::Vec v = orca_ref::new_velocity(body.data(), NP, ia, static_cast<float>(p.sensing_horizon),
                                                       static_cast<size_t>(p.rvo_max_neighbors),
                                                       static_cast<float>(p.rvo_time_horizon), ts);
        ovel[2 * a] = v.x;
        ovel[2 * a + 1] = v.y;
        const orca_ref::Vec np_ = orca_ref::advance(body[ia].pos, v, ts);  // :93,:96
        const double dpx = static_cast<double>(np_.x) - s.pos_x[i];       // :97 float32 pos - float64 pos
        const double dpy = static_cast<double>(np_.y) - s.pos_y[i];
        const double ang = m_atan2(dpy, dpx) - 0.0;                    // :100-101
        const double nh = pymod(ang, kTwoPi);                             // :102
        dh = wrap(nh - s.heading[i]);                                     // :103
        spd = (1.0 / p.rvo_dt) * std::sqrt(dpx * dpx + dpy * dpy);        // :106
        if (std::fabs(dh) > kPi / 6) {                                    // :109-111
          dh = sgn(dh) * (kPi / 6);
          spd = 0.0;
        }
        if (s.rvo_heading_noise) dh = dh + s.rvo_heading_noise[i];        // :118-119 (the caller's draw)
      } break;
      case ORC_POL_NONCOOP: {  // policies/NonCooperativePolicy.py:21
        const Ego eg = ego_frame(s.pos_x[i], s.pos_y[i], s.goal_x[i], s.goal_y[i], s.heading[i]);
        spd = s.pref_speed[i];
        dh = -eg.heading_ego;
      } break;
      case ORC_POL_STATIC:  // policies/StaticPolicy.py:21-23
        s.goal_x[i] = s.pos_x[i];
        s.goal_y[i] = s.pos_y[i];
        break;
      case ORC_POL_EXTERNAL:  // policies/ExternalPolicy.py:14-16
        if (ext) { spd = ext[2 * i]; dh = ext[2 * i + 1]; }
        break;
      case ORC_POL_LEARNING:  // policies/LearningPolicy.py:29-33
        if (ext) { dh = p.max_heading_change * (2. * ext[2 * i + 1] - 1.); spd = s.pref_speed[i] * ext[2 * i]; }
        break;
      case ORC_POL_GA3C_CADRL:  // policies/GA3CCADRLPolicy.py:81-84 (the network itself: oracle/ga3c_ref.py)
      case ORC_POL_LEARNING_GA3C: {  // policies/LearningPolicyGA3C.py:24-26 + GA3C_CADRL/network.py:7-16
        static const double tab[11][2] = {{1, -kPi / 6}, {1, -kPi / 12}, {1, 0}, {1, kPi / 12}, {1, kPi / 6}, {0.5, -kPi / 6},
                                          {0.5, 0},      {0.5, kPi / 6}, {0, -kPi / 6}, {0, 0}, {0, kPi / 6}};
        if (ext) {
          int k = static_cast<int>(ext[2 * i]);
          if (k < 0) k = 0;
          if (k > 10) k = 10;
          spd = s.pref_speed[i] * tab[k][0];
          dh = tab[k][1];
        }
      } break;
      default: break;
    }
    act[2 * a] = static_cast<float>(spd);  // env.py:305-307: float32 storage
    act[2 * a + 1] = static_cast<float>(dh);
  }
  if (o.actions) std::memcpy(o.actions + 2 * b, act.data(), sizeof(float) * 2 * N);
  if (o.orca_vel) std::memcpy(o.orca_vel + 2 * b, ovel.data(), sizeof(float) * 2 * N);

  // ---- 2. every agent moves (agent.py:192-241) ----
  for (int a : P) {
    const int i = b + a;
    uint32_t f = s.flags[i];
    if (f & (ORC_AT_GOAL | ORC_OUT_OF_TIME | ORC_IN_COLLISION)) {  // :202-209
      if (f & ORC_AT_GOAL) f |= ORC_WAS_AT_GOAL;
      if (f & ORC_IN_COLLISION) f |= ORC_WAS_IN_COLLISION;
      s.flags[i] = f;
      s.vel_x[i] = s.vel_y[i] = 0.0;
      continue;
    }
    s.last_action[2 * i] = act[2 * a];  // :212-213
    s.last_action[2 * i + 1] = act[2 * a + 1];
    const double a0 = act[2 * a], a1 = act[2 * a + 1];
    if (s.dynamics[i] != ORC_DYN_EXTERNAL) {
      double nh;
      if (s.dynamics[i] == ORC_DYN_MAX_TURN_RATE) {  // dynamics/UnicycleDynamicsMaxTurnRate.py:31-33
        double tr = a1 / p.dt;
        tr = std::min(std::max(tr, -3.0), 3.0);
        nh = wrap(tr * p.dt + s.heading[i]);
      } else {
        nh = wrap(a1 + s.heading[i]);  // dynamics/UnicycleDynamics.py:28
      }
      double c, sn;
      m_sincos(nh, sn, c);
      s.pos_x[i] += a0 * c * p.dt;  // :30-32
      s.pos_y[i] += a0 * sn * p.dt;
      s.vel_x[i] = a0 * c;  // :34-35
      s.vel_y[i] = a0 * sn;
      s.heading[i] = nh;  // :39
      if (s.dynamics[i] == ORC_DYN_UNICYCLE) {  // :41-47 (UnicycleDynamicsMaxTurnRate.py has no such block)
        double& td = s.turning_dir[i];
        if (std::fabs(td) < 1e-5) td = 0.11 * ((nh > 0.0) - (nh < 0.0));
        else if (td * nh < 0.0) td = std::max(-kPi, std::min(kPi, -td + nh));
        else td = ((td > 0.0) - (td < 0.0)) * std::max(0.0, std::fabs(td) - 0.1);
      }
    }
    else if (s.ext_state) {  // a host-side Dynamics subclass moved this agent (agent.py:214-220: dynamics_model.step)
      const double* q = s.ext_state + 5 * i;
      if (!(std::isnan(q[0]) || std::isnan(q[1]) || std::isnan(q[2]) || std::isnan(q[3]) || std::isnan(q[4]))) {
        s.pos_x[i] = q[0]; s.pos_y[i] = q[1]; s.vel_x[i] = q[2]; s.vel_y[i] = q[3]; s.heading[i] = q[4];
      }
    }
    const double gx = s.pos_x[i] - s.goal_x[i], gy = s.pos_y[i] - s.goal_y[i];
    if (gx * gx + gy * gy <= p.near_goal_threshold * p.near_goal_threshold) f |= ORC_AT_GOAL;  // :150-153
    else f &= ~ORC_AT_GOAL;
    s.time_remaining[i] -= p.dt;  // :235-239
    s.t[i] += p.dt;
    s.step_num[i] += 1;
    if (s.time_remaining[i] <= 0.0) f |= ORC_OUT_OF_TIME;
    s.flags[i] = f;
  }

  // ---- 3. all unordered pairs, done agents included (env.py:458-512) ----
  std::vector<uint8_t> coll(N, 0);
  std::vector<double> nearest(N, std::numeric_limits<double>::infinity());
  for (int ii = 0; ii < NP; ++ii)
    for (int jj = ii + 1; jj < NP; ++jj) {
      const int i = P[ii], j = P[jj];
      const double dx = s.pos_x[b + i] - s.pos_x[b + j], dy = s.pos_y[b + i] - s.pos_y[b + j];
      const double d = std::sqrt(dx * dx + dy * dy);  // util.py:17-21
      const double cr = s.radius[b + i] + s.radius[b + j];
      if (d - cr < nearest[i]) nearest[i] = d - cr;
      if (d - cr < nearest[j]) nearest[j] = d - cr;
      if (d <= cr) coll[i] = coll[j] = 1;
    }

  // ---- 4. rewards (env.py:394-456) ----
  for (int a = 0; a < N; ++a) o.rewards[b + a] = 0.0;  // (empty slots)
  for (int a : P) {
    const int i = b + a;
    uint32_t f = s.flags[i];
    double r = p.reward_time_step;
    if (f & ORC_AT_GOAL) {
      if (!(f & ORC_WAS_AT_GOAL)) r = p.reward_at_goal;
    } else if (!(f & ORC_WAS_IN_COLLISION)) {
      if (coll[a]) {
        r = p.reward_collision;
        f |= ORC_IN_COLLISION;
      } else if (map && hits_wall(*map, s.pos_x[i], s.pos_y[i], s.radius[i])) {  // env.py:425-429
        r = p.reward_collision_wall;  // config.py:33
        f |= ORC_IN_COLLISION;
      } else {
        if (nearest[a] <= p.getting_close_range) r = -0.1 - nearest[a] / 2.0;
        if (std::fabs(static_cast<double>(s.last_action[2 * i + 1])) > p.wiggly_threshold) r += p.reward_wiggly;
      }
    }
    r = std::min(std::max(r, p.reward_min), p.reward_max);
    o.rewards[i] = r;
    s.ep_reward[i] += r;  // experiments/src/env_utils.py:51
    s.flags[i] = f;
  }

  // ---- 5. observations (env.py:555-575) ----
  observe_env(p, s, o, e);

  // ---- 6. done / game over (env.py:514-553) ----
  bool all_done = true, all_learning_done = true;
  for (int a = 0; a < N; ++a) o.done[b + a] = 1;  // (empty slots)
  for (int a : P) {
    const int i = b + a;
    uint32_t f = s.flags[i];
    const bool d = f & (ORC_AT_GOAL | ORC_OUT_OF_TIME | ORC_IN_COLLISION);
    if (d) f |= ORC_DONE; else f &= ~ORC_DONE;
    s.flags[i] = f;
    o.done[i] = d;
    all_done = all_done && d;
    if (f & ORC_STILL_LEARNING) all_learning_done = all_learning_done && d;
  }
  bool over = all_done;
  if (p.game_over_mode == ORC_OVER_AGENT0) over = o.done[b];
  else if (p.game_over_mode == ORC_OVER_LEARNING_DONE) over = all_learning_done;
  o.game_over[e] = over;
}

// experiments/src/env_utils.py:56-87 reduced to counters
void episode_stats(const OrcParams& p, const OrcState& s, int e) {
  const int N = p.num_agents, b = e * N;
  bool any_coll = false, all_goal = true;
  double tot_r = 0.0, ttg = 0.0, extra = 0.0;
  for (int a : present(p, s, e)) {
    const uint32_t f = s.flags[b + a];
    any_coll = any_coll || (f & ORC_IN_COLLISION);
    all_goal = all_goal && (f & ORC_AT_GOAL);
    tot_r += s.ep_reward[b + a];
    ttg += s.t[b + a];
    extra += s.t[b + a] - s.slt[b + a];
  }
  double* st = s.env_stats + 8 * static_cast<size_t>(e);
  st[0] += 1.0;
  if (any_coll) st[1] += 1.0;
  else if (all_goal) st[2] += 1.0;
  else st[3] += 1.0;
  st[4] += s.episode_step[e];
  st[5] += tot_r;
  st[6] += ttg;
  st[7] += extra;
}

}  // namespace

extern "C" {

int ca_oracle_version(void) { return 4; }

double ca_oracle_round2(double x) { return round2(x); }

int ca_oracle_reset(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* cases, const double* headings,
                    const uint8_t* mask) {
  const int N = p->num_agents;
  for (int e = 0; e < p->num_envs; ++e) {
    if (mask && !mask[e]) continue;
    reset_env(*p, *s, e, cases + static_cast<size_t>(e) * N * 6, headings ? headings + static_cast<size_t>(e) * N : nullptr);
    s->reset_count[e] = 0;
    observe_env(*p, *s, *o, e);
  }
  return 0;
}

int ca_oracle_step(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* ext_actions) {
  for (int e = 0; e < p->num_envs; ++e) step_env(*p, *s, *o, ext_actions, e);
  return 0;
}

int ca_oracle_rollout_ex(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* ext_actions, const double* table,
                         int32_t n_cases, int64_t env_id_offset, int64_t case_stride, int32_t n_steps, const OrcMap* map) {
  const int N = p->num_agents;
  for (int t = 0; t < n_steps; ++t)
    for (int e = 0; e < p->num_envs; ++e) {
      step_env(*p, *s, *o, ext_actions, e, (map && map->static_map) ? map : nullptr);
      if (o->game_over[e]) {  // vec_env.py:120-128: stats, then reset and hand back the reset observation
        episode_stats(*p, *s, e);
        s->reset_count[e] += 1;
        const int64_t c = (env_id_offset + e + static_cast<int64_t>(s->reset_count[e]) * case_stride) % n_cases;
        reset_env(*p, *s, e, table + static_cast<size_t>(c) * N * 6, nullptr);
        observe_env(*p, *s, *o, e);
      }
    }
  return 0;
}

int ca_oracle_rollout(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* table, int32_t n_cases,
                      int64_t env_id_offset, int64_t case_stride, int32_t n_steps) {
  return ca_oracle_rollout_ex(p, s, o, nullptr, table, n_cases, env_id_offset, case_stride, n_steps, nullptr);
}

void ca_oracle_set_tie_order(int reverse) { orca_ref::g_tie_reverse = reverse ? 1 : 0; }  // (tests only: see orca_ref.h)

void ca_oracle_set_libm(double (*atan2_fn)(double, double), void (*sincos_fn)(double, double*, double*)) {
  g_atan2 = atan2_fn;
  g_sincos = sincos_fn;
}

int ca_oracle_step_map(const OrcParams* p, const OrcState* s, const OrcOut* o, const double* ext_actions, const OrcMap* m) {
  for (int e = 0; e < p->num_envs; ++e) step_env(*p, *s, *o, ext_actions, e, m);
  return 0;
}

// Map.add_agents_to_map (Map.py:46-52) + LaserScanSensor.sense (LaserScanSensor.py:49-101)
int ca_oracle_laserscan(const OrcParams* p, const OrcState* s, const OrcMap* m, const OrcScan* sc) {
  const int N = p->num_agents, B = sc->num_beams, R = sc->num_ranges, H = sc->num_to_store;
  std::vector<uint8_t> grid(static_cast<size_t>(m->rows) * m->cols);
  std::vector<double> angles(B), ranges(R);
  {  // np.linspace(min, max, B) and np.arange(0, max_range, res)
    const double step = (sc->max_angle - sc->min_angle) / (B - 1);
    for (int b = 0; b < B; ++b) angles[b] = b * step + sc->min_angle;
    angles[B - 1] = sc->max_angle;
    for (int r = 0; r < R; ++r) ranges[r] = 0.0 + r * sc->range_res;
  }
  for (int e = 0; e < p->num_envs; ++e) {
    const int b0 = e * N;
    if (m->static_map) std::copy(m->static_map, m->static_map + grid.size(), grid.begin());
    else std::fill(grid.begin(), grid.end(), 0);
    for (int a = 0; a < N; ++a) {  // rasterise every agent as a disc around its (floored) cell
      long gr, gc;
      bool in_map;
      to_cell(*m, s->pos_x[b0 + a], s->pos_y[b0 + a], gr, gc, in_map);
      if (!in_map) continue;
      const double rr = (s->radius[b0 + a] / m->cell) * (s->radius[b0 + a] / m->cell);
      for (long r = 0; r < m->rows; ++r)
        for (long c = 0; c < m->cols; ++c) {
          const double dc = static_cast<double>(c - gc), dr = static_cast<double>(r - gr);
          if (dc * dc + dr * dr < rr) grid[r * m->cols + c] = 1;
        }
    }
    for (int a = 0; a < N; ++a) {
      const int i = b0 + a;
      long er, ec;
      bool ego_in;
      to_cell(*m, s->pos_x[i], s->pos_y[i], er, ec, ego_in);
      const double err = (s->radius[i] / m->cell) * (s->radius[i] / m->cell);
      uint8_t* hist = sc->hist + static_cast<size_t>(i) * H * B;
      const bool first = s->step_num[i] == 0;  // Sensor.num_measurements_made == 0
      if (!first)
        for (int h = H - 1; h > 0; --h) std::memcpy(hist + h * B, hist + (h - 1) * B, B);  // np.roll(..., 1, axis=0)
      for (int b = 0; b < B; ++b) {
        const double ang = angles[b] + s->heading[i];
        const double cs = std::cos(ang), sn = std::sin(ang);
        int hits = 0, idx = 255;
        for (int r = 0; r < R && hits < 2; ++r) {
          const double x = s->pos_x[i] + ranges[r] * cs, y = s->pos_y[i] + ranges[r] * sn;
          long gr, gc;
          bool in_map;
          to_cell(*m, x, y, gr, gc, in_map);
          bool hit = false;
          if (in_map) {
            const double dc = static_cast<double>(gc - ec), dr = static_cast<double>(gr - er);
            const bool ego = ego_in && (dc * dc + dr * dr < err);
            hit = grid[gr * m->cols + gc] && !ego;
          }
          hits += hit;
          if (hits == 1) idx = r;  // np.where(cumsum == 1) + duplicate-index assignment keeps the LAST such sample
        }
        hist[b] = static_cast<uint8_t>(idx);
      }
      if (first)
        for (int h = 1; h < H; ++h) std::memcpy(hist + h * B, hist, B);
      double* out = sc->out + static_cast<size_t>(i) * H * B;
      for (int q = 0; q < H * B; ++q) out[q] = hist[q] == 255 ? sc->max_range : ranges[hist[q]];
    }
  }
  return 0;
}

int ca_oracle_orca(int32_t num_envs, int32_t num_agents, const float* pos, const float* vel, const float* pref,
                   const float* radius, const float* max_speed, float collab, float time_horizon, float time_step,
                   int32_t max_neighbors, float neighbor_dist, float* new_vel) {
  std::vector<orca_ref::Body> body(num_agents);
  for (int e = 0; e < num_envs; ++e) {
    const size_t b = static_cast<size_t>(e) * num_agents;
    for (int j = 0; j < num_agents; ++j) {
      body[j].pos = orca_ref::mk(pos[2 * (b + j)], pos[2 * (b + j) + 1]);
      body[j].vel = orca_ref::mk(vel[2 * (b + j)], vel[2 * (b + j) + 1]);
      body[j].pref = orca_ref::mk(pref[2 * (b + j)], pref[2 * (b + j) + 1]);
      body[j].radius = radius[b + j];
      body[j].max_speed = max_speed[b + j];
      body[j].collab = collab;
    }
    for (int a = 0; a < num_agents; ++a) {
      const orca_ref::Vec v = orca_ref::new_velocity(body.data(), num_agents, a, neighbor_dist,
                                                     static_cast<size_t>(max_neighbors), time_horizon, time_step);
      new_vel[2 * (b + a)] = v.x;
      new_vel[2 * (b + a) + 1] = v.y;
    }
  }
  return 0;
}

}  // extern "C"
