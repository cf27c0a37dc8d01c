// gym_collision_avoidance_amd/csrc/cagpu.hip -- the MI355X (gfx950 / CDNA4) hot path behind include/cagpu.h.
//
// One fused kernel advances every env by one (or n) simulation step(s):
//   policy (ORCA / non-coop / static / external)  -> RVOPolicy.py:50-122 + rvo2, NonCooperativePolicy.py:21, ...
//   move (unicycle integration, ego frame)        -> agent.py:192-241, UnicycleDynamics.py:14-47, Dynamics.py:24-41
//   all-pairs collision + nearest distance        -> collision_avoidance_env.py:458-512
//   rewards                                       -> collision_avoidance_env.py:394-456
//   ego-centric sorted other-agent observation    -> OtherAgentsStatesSensor.py:58-144
//   done / game over / fixture auto-reset + stats -> collision_avoidance_env.py:514-553, vec_env.py:120-128
//
// Mapping (wave64, no MFMA: this is branchy element-wise geometry): one LANE per agent, one
// WORKGROUP per tile of WHOLE envs (tile = BLOCK / num_agents envs), so every neighbour an agent
// needs is owned by a lane of the same workgroup and travels through LDS, never through HBM:
//   * HBM loads/stores are agent-major SoA -> lane i touches element base+i: fully coalesced;
//   * the O(N^2) pairwise passes (ORCA half-planes, collisions, sensor) read the tile's positions /
//     velocities / radii from LDS (same-env lanes hit the same address -> broadcast);
//   * per-lane work arrays (ORCA lines, sort keys) live in LDS as [slot][lane] columns -> bank-conflict free;
//   * the observation rows are staged in LDS and leave as one contiguous, coalesced block.
// Envs never interact, so n-step rollouts need no grid-wide synchronisation.
//
// Numerics: float64 state in the reference's operation order, the action pair rounded to float32
// (env.py:305-307), ORCA in float with the RVO2 operation order.  Built with -ffp-contract=off: a
// fused multiply-add in `dx*dx + dy*dy <= r*r` would change discrete events.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "cagpu.h"

namespace {

constexpr double kPi = 3.141592653589793;
constexpr double kTwoPi = 2.0 * kPi;
constexpr float kRvoEps = 0.00001f;

enum { MODE_STEP = 0, MODE_RESET = 1, MODE_OBSERVE = 2 };

struct KArgs {
  CaParams p;
  CaState s;
  CaOut o;
  const double* ext;
  // auto reset
  const double* table;
  int32_t n_cases;
  int64_t env_id_offset, case_stride;
  // explicit reset
  const double* reset_cases;
  const double* reset_headings;
  const uint8_t* reset_mask;
  int32_t n_steps, mode, stage_obs;
};

// ---------------------------------------------------------------- small math helpers
struct F2 {
  float x, y;
};
__device__ __forceinline__ F2 f2(float x, float y) { F2 v; v.x = x; v.y = y; return v; }
__device__ __forceinline__ F2 operator+(F2 a, F2 b) { return f2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ F2 operator-(F2 a, F2 b) { return f2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ F2 operator*(float s, F2 a) { return f2(s * a.x, s * a.y); }
__device__ __forceinline__ float dotf(F2 a, F2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float detf(F2 a, F2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float sqf(float a) { return a * a; }
// IEEE correctly-rounded float divide / sqrt: plain `/` and sqrtf under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt (passed explicitly by build_native.py).  NOT __fsqrt_rn:
// in this ROCm that intrinsic maps to the native (not correctly rounded) square root.
__device__ __forceinline__ float divf(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrtf_rn(float a) { return sqrtf(a); }
// RVO2's Vector2 / float: multiply by the reciprocal
__device__ __forceinline__ F2 over(F2 a, float s) { const float inv = divf(1.0f, s); return f2(a.x * inv, a.y * inv); }
__device__ __forceinline__ F2 unitf(F2 a) { return over(a, sqrtf_rn(dotf(a, a))); }

__device__ __forceinline__ double wrap_pi(double a) {  // util.py:141-146 ([-pi, pi))
  for (int it = 0; it < 4096 && a >= kPi; ++it) a -= kTwoPi;
  for (int it = 0; it < 4096 && a < -kPi; ++it) a += kTwoPi;
  return a;
}

// ---------------------------------------------------------------- ORCA (RVO2 algorithm, C float)
// Lines live in LDS as float4 {point.x, point.y, dir.x, dir.y}, slot-major: line k of this lane
// is L[k * STRIDE].  The permitted half-plane is to the left of dir through point.
template <int STRIDE>
__device__ bool lp1(const float4* L, int k, float radius, F2 opt, bool dir_opt, F2& res) {
  const float4 lk = L[k * STRIDE];
  const F2 pk = f2(lk.x, lk.y), dk = f2(lk.z, lk.w);
  const float dp = dotf(pk, dk);
  const float disc = sqf(dp) + sqf(radius) - dotf(pk, pk);
  if (disc < 0.0f) return false;
  const float sd = sqrtf_rn(disc);
  float t_lo = -dp - sd;
  float t_hi = -dp + sd;
  for (int i = 0; i < k; ++i) {
    const float4 li = L[i * STRIDE];
    const F2 pi = f2(li.x, li.y), di = f2(li.z, li.w);
    const float den = detf(dk, di);
    const float num = detf(di, pk - pi);
    if (fabsf(den) <= kRvoEps) {
      if (num < 0.0f) return false;
      continue;
    }
    const float t = divf(num, den);
    if (den >= 0.0f) t_hi = (t < t_hi) ? t : t_hi;
    else t_lo = (t_lo < t) ? t : t_lo;
    if (t_lo > t_hi) return false;
  }
  if (dir_opt) {
    if (dotf(opt, dk) > 0.0f) res = pk + t_hi * dk;
    else res = pk + t_lo * dk;
  } else {
    const float t = dotf(dk, opt - pk);
    if (t < t_lo) res = pk + t_lo * dk;
    else if (t > t_hi) res = pk + t_hi * dk;
    else res = pk + t * dk;
  }
  return true;
}

template <int STRIDE>
__device__ int lp2(const float4* L, int n, float radius, F2 opt, bool dir_opt, F2& res) {
  if (dir_opt) res = radius * opt;
  else if (dotf(opt, opt) > sqf(radius)) res = radius * unitf(opt);
  else res = opt;
  for (int i = 0; i < n; ++i) {
    const float4 li = L[i * STRIDE];
    if (detf(f2(li.z, li.w), f2(li.x, li.y) - res) > 0.0f) {
      const F2 keep = res;
      if (!lp1<STRIDE>(L, i, radius, opt, dir_opt, res)) {
        res = keep;
        return i;
      }
    }
  }
  return n;
}

template <int STRIDE>
__device__ void lp3(const float4* L, float4* P, int n, int begin, float radius, F2& res) {
  float depth = 0.0f;
  for (int i = begin; i < n; ++i) {
    const float4 li = L[i * STRIDE];
    const F2 pi = f2(li.x, li.y), di = f2(li.z, li.w);
    if (detf(di, pi - res) > depth) {
      int m = 0;
      for (int j = 0; j < i; ++j) {
        const float4 lj = L[j * STRIDE];
        const F2 pj = f2(lj.x, lj.y), dj = f2(lj.z, lj.w);
        const float D = detf(di, dj);
        F2 pt;
        if (fabsf(D) <= kRvoEps) {
          if (dotf(di, dj) > 0.0f) continue;
          pt = 0.5f * (pi + pj);
        } else {
          pt = pi + divf(detf(dj, pi - pj), D) * di;
        }
        const F2 dr = unitf(dj - di);
        P[m * STRIDE] = make_float4(pt.x, pt.y, dr.x, dr.y);
        ++m;
      }
      const F2 keep = res;
      if (lp2<STRIDE>(P, m, radius, f2(-di.y, di.x), true, res) < m) res = keep;
      depth = detf(di, pi - res);
    }
  }
}

// Half-plane induced on `me` by `ot` (Agent::computeNewVelocity, agent part).
__device__ __forceinline__ float4 half_plane(F2 mpos, F2 mvel, float mrad, F2 opos, F2 ovel, float orad, float collab,
                                             float inv_h, float time_step) {
  const F2 rp = opos - mpos;
  const F2 rv = mvel - ovel;
  const float d2 = dotf(rp, rp);
  const float R = mrad + orad;
  const float R2 = sqf(R);
  F2 dir, u;
  if (d2 > R2) {
    const F2 w = rv - inv_h * rp;
    const float w2 = dotf(w, w);
    const float dp1 = dotf(w, rp);
    if (dp1 < 0.0f && sqf(dp1) > R2 * w2) {
      const float wl = sqrtf_rn(w2);
      const F2 uw = over(w, wl);
      dir = f2(uw.y, -uw.x);
      u = (R * inv_h - wl) * uw;
    } else {
      const float leg = sqrtf_rn(d2 - R2);
      if (detf(rp, w) > 0.0f) {
        dir = over(f2(rp.x * leg - rp.y * R, rp.x * R + rp.y * leg), d2);
      } else {
        const F2 t = over(f2(rp.x * leg + rp.y * R, -rp.x * R + rp.y * leg), d2);
        dir = f2(-t.x, -t.y);
      }
      const float dp2 = dotf(rv, dir);
      u = dp2 * dir - rv;
    }
  } else {
    const float inv_dt = divf(1.0f, time_step);
    const F2 w = rv - inv_dt * rp;
    const float wl = sqrtf_rn(dotf(w, w));
    const F2 uw = over(w, wl);
    dir = f2(uw.y, -uw.x);
    u = (R * inv_dt - wl) * uw;
  }
  const F2 pt = mvel + collab * u;
  return make_float4(pt.x, pt.y, dir.x, dir.y);
}

// New ORCA velocity of the agent on this lane.  fpx/fpy/fvx/fvy/frad: the tile's float bodies in LDS,
// `ebase` = LDS index of agent 0 of my env, `a` my agent index, N agents per env.  dcol: [N][STRIDE]
// float scratch column, L / P: [N-1][STRIDE] float4 line columns (all already offset to my lane).
template <int STRIDE>
__device__ F2 orca_velocity(const float* fpx, const float* fpy, const float* fvx, const float* fvy, const float* frad,
                            int ebase, int a, int N, F2 pref, float max_speed, float collab, float time_horizon,
                            float time_step, float neighbor_dist, int max_nb, float* dcol, float4* L, float4* P) {
  const F2 mpos = f2(fpx[ebase + a], fpy[ebase + a]);
  const F2 mvel = f2(fvx[ebase + a], fvy[ebase + a]);
  const float mrad = frad[ebase + a];
  const float range_sq = sqf(neighbor_dist);
  // neighbour list = others with distSq < rangeSq, ascending by distSq, ties by index, first max_nb
  int cnt = 0;
  for (int j = 0; j < N; ++j) {
    float d2 = INFINITY;
    if (j != a) {
      const F2 d = mpos - f2(fpx[ebase + j], fpy[ebase + j]);
      d2 = dotf(d, d);
      if (d2 < range_sq) ++cnt; else d2 = INFINITY;
    }
    dcol[j * STRIDE] = d2;
  }
  const int n = cnt < max_nb ? cnt : max_nb;
  const float inv_h = divf(1.0f, time_horizon);
  for (int j = 0; j < N; ++j) {
    const float dj = dcol[j * STRIDE];
    if (j == a || !(dj < INFINITY)) continue;
    int rank = 0;
    for (int q = 0; q < N; ++q) {
      const float dq = dcol[q * STRIDE];
      rank += (dq < dj) || (dq == dj && q < j);
    }
    if (rank >= n) continue;
    L[rank * STRIDE] = half_plane(mpos, mvel, mrad, f2(fpx[ebase + j], fpy[ebase + j]), f2(fvx[ebase + j], fvy[ebase + j]),
                                  frad[ebase + j], collab, inv_h, time_step);
  }
  F2 v;
  const int fail = lp2<STRIDE>(L, n, max_speed, pref, false, v);
  if (fail < n) lp3<STRIDE>(L, P, n, fail, max_speed, v);
  return v;
}

// ---------------------------------------------------------------- ego frame (agent.py:329-349, Dynamics.py:24-41)
struct Ego {
  double dist, prx, pry, orx, ory, heading_ego;
};
__device__ __forceinline__ Ego ego_frame(double px, double py, double gx, double gy, double heading) {
  Ego e;
  const double dx = gx - px, dy = gy - py;
  e.dist = sqrt(dx * dx + dy * dy);
  if (e.dist > 1e-8) {
    e.prx = dx / e.dist;
    e.pry = dy / e.dist;
  } else {
    e.prx = dx;
    e.pry = dy;
  }
  e.orx = -e.pry;
  e.ory = e.prx;
  e.heading_ego = wrap_pi(heading - atan2(e.pry, e.prx));
  return e;
}

// ---------------------------------------------------------------- LDS carve-up
// fixed part (per lane):  5 doubles (post-move pos/vel/radius) + 5 floats (ORCA bodies) + 3 doubles (stats) + 1 u32
// union part  (per lane): max( ORCA: N floats + 2*(N-1) float4 ,  sensor: 3*N doubles + N ints [+ W floats staging] )
__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~static_cast<size_t>(15); }
__host__ __device__ inline size_t lds_fixed_bytes(int block) { return align16(static_cast<size_t>(block) * (8 * 8 + 5 * 4 + 4)); }
__host__ __device__ inline size_t lds_orca_bytes(int block, int N) {
  return align16(static_cast<size_t>(block) * N * 4) + static_cast<size_t>(block) * 2 * (N > 1 ? N - 1 : 1) * 16;
}
__host__ __device__ inline size_t lds_sense_bytes(int block, int N, int W, int stage) {
  return align16(static_cast<size_t>(block) * N * (3 * 8 + 4)) + (stage ? align16(static_cast<size_t>(block) * W * 4) : 0);
}

struct Lane {  // per-lane registers of one agent
  double px, py, vx, vy, heading, gx, gy, rad, ps, tr, t, slt, epr;
  float act0, act1;
  uint32_t flags;
  int32_t step_num;
};

// test_cases.py:545-557 + agent.py:59-138
__device__ __forceinline__ void reset_lane(Lane& r, const double* c, const double* heading, const CaParams& p) {
  r.px = c[0]; r.py = c[1]; r.gx = c[2]; r.gy = c[3]; r.ps = c[4]; r.rad = c[5];
  r.vx = r.vy = 0.0;
  r.heading = heading ? *heading : atan2(c[3] - c[1], c[2] - c[0]);
  const double dx = c[0] - c[2], dy = c[1] - c[3];
  r.slt = (sqrt(dx * dx + dy * dy) - p.near_goal_threshold) / c[4];
  double tr = p.max_time_ratio * r.slt;
  if (p.dt > tr) tr = p.dt;
  r.tr = tr;
  r.t = 0.0;
  r.epr = 0.0;
  r.act0 = r.act1 = 0.f;
  r.step_num = 0;
  r.flags &= ~0x3Fu;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void ca_kernel(const KArgs k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const CaParams& p = k.p;
  const int N = p.num_agents, K = p.max_obs, W = 6 + 7 * K;
  const int tile_envs = BLOCK / N;
  const int tile_n = tile_envs * N;
  const int lane = threadIdx.x;
  const int le = lane / N, a = lane - le * N;
  const int ebase = le * N;
  const long env0 = static_cast<long>(blockIdx.x) * tile_envs;
  const long e = env0 + le;
  const bool active = (lane < tile_n) && (e < p.num_envs);
  const long i = e * N + a;
  const long tile_base = env0 * N;  // first global agent index of this tile
  long tile_cnt = static_cast<long>(p.num_envs) * N - tile_base;
  if (tile_cnt > tile_n) tile_cnt = tile_n;

  // ---- LDS
  double* sh_px = reinterpret_cast<double*>(smem);
  double* sh_py = sh_px + BLOCK;
  double* sh_vx = sh_py + BLOCK;
  double* sh_vy = sh_vx + BLOCK;
  double* sh_rad = sh_vy + BLOCK;
  double* sh_r0 = sh_rad + BLOCK;  // stats scratch
  double* sh_r1 = sh_r0 + BLOCK;
  double* sh_r2 = sh_r1 + BLOCK;
  float* sh_fpx = reinterpret_cast<float*>(sh_r2 + BLOCK);
  float* sh_fpy = sh_fpx + BLOCK;
  float* sh_fvx = sh_fpy + BLOCK;
  float* sh_fvy = sh_fvx + BLOCK;
  float* sh_frad = sh_fvy + BLOCK;
  uint32_t* sh_flag = reinterpret_cast<uint32_t*>(sh_frad + BLOCK);
  unsigned char* un = smem + lds_fixed_bytes(BLOCK);
  // ORCA view of the union
  float* dcol = reinterpret_cast<float*>(un) + lane;
  float4* Lcol = reinterpret_cast<float4*>(un + align16(static_cast<size_t>(BLOCK) * N * 4)) + lane;
  float4* Pcol = Lcol + static_cast<size_t>(N > 1 ? N - 1 : 1) * BLOCK;
  // sensor view of the union
  double* kcol = reinterpret_cast<double*>(un) + lane;             // [N][BLOCK] sort key (rint(100*d))
  double* ocol = kcol + static_cast<size_t>(N) * BLOCK;            // [N][BLOCK] p_orth
  double* d2col = ocol + static_cast<size_t>(N) * BLOCK;           // [N][BLOCK] dist_2_other
  int* rcol = reinterpret_cast<int*>(d2col + static_cast<size_t>(N) * BLOCK - lane) + lane;  // [N][BLOCK] rank
  float* sh_obs = reinterpret_cast<float*>(un + align16(static_cast<size_t>(BLOCK) * N * (3 * 8 + 4)));

  // ---- load my agent
  Lane r;
  if (active) {
    r.px = k.s.pos_x[i]; r.py = k.s.pos_y[i]; r.vx = k.s.vel_x[i]; r.vy = k.s.vel_y[i];
    r.heading = k.s.heading[i]; r.gx = k.s.goal_x[i]; r.gy = k.s.goal_y[i];
    r.rad = k.s.radius[i]; r.ps = k.s.pref_speed[i]; r.tr = k.s.time_remaining[i]; r.t = k.s.t[i];
    r.slt = k.s.slt[i]; r.epr = k.s.ep_reward[i];
    const float2 la = reinterpret_cast<const float2*>(k.s.last_action)[i];
    r.act0 = la.x; r.act1 = la.y;
    r.flags = k.s.flags[i];
    r.step_num = k.s.step_num[i];
  } else {
    r.px = r.py = r.vx = r.vy = r.heading = r.gx = r.gy = 0.0;
    r.rad = r.ps = 1.0; r.tr = r.t = r.slt = r.epr = 0.0;
    r.act0 = r.act1 = 0.f; r.flags = CA_DONE | CA_AT_GOAL; r.step_num = 0;
  }
  int ep_step = 0, reset_cnt = 0;
  if (active) { ep_step = k.s.episode_step[e]; reset_cnt = k.s.reset_count[e]; }

  bool do_sense = true;   // does my env need its observation (re)written
  float reward = 0.f;
  uint32_t done_out = 0, over_out = 0;

  if (k.mode == MODE_RESET) {
    do_sense = active && (!k.reset_mask || k.reset_mask[e]);
    if (do_sense) {
      reset_lane(r, k.reset_cases + i * 6, k.reset_headings ? k.reset_headings + i : nullptr, p);
      ep_step = 0;
      reset_cnt = 0;
    }
  }

  const int n_steps = (k.mode == MODE_STEP) ? k.n_steps : 1;
  for (int step = 0; step < n_steps; ++step) {
    if (k.mode == MODE_STEP) {
      ep_step += 1;  // env.py:183
      // ================= 1. policy: actions from the PRE-step state (env.py:305-323)
      sh_fpx[lane] = static_cast<float>(r.px);
      sh_fpy[lane] = static_cast<float>(r.py);
      sh_fvx[lane] = static_cast<float>(r.vx);
      sh_fvy[lane] = static_cast<float>(r.vy);
      sh_frad[lane] = static_cast<float>((1 + 5e-2) * r.rad);  // RVOPolicy.py:71
      __syncthreads();
      double spd = 0.0, dh = 0.0;
      const uint32_t pol = (r.flags >> CA_POLICY_SHIFT) & 0xF;
      if (active && !(r.flags & CA_DONE)) {  // env.py:311
        if (pol == CA_POL_RVO) {
          const double vx = r.gx - r.px, vy = r.gy - r.py;
          const double sc = r.ps / sqrt(vx * vx + vy * vy);  // RVOPolicy.py:66-67
          const F2 pref = f2(static_cast<float>(sc * vx), static_cast<float>(sc * vy));
          const float ts = static_cast<float>(p.dt);
          const F2 v = orca_velocity<BLOCK>(sh_fpx, sh_fpy, sh_fvx, sh_fvy, sh_frad, ebase, a, N, pref,
                                            static_cast<float>(r.ps), static_cast<float>(p.rvo_collab_coeff),
                                            static_cast<float>(p.rvo_time_horizon), ts,
                                            static_cast<float>(p.sensing_horizon), p.rvo_max_neighbors, dcol, Lcol, Pcol);
          // Agent::update: float position += v * timeStep; RVOPolicy.py:96-111
          const float npx = sh_fpx[lane] + v.x * ts, npy = sh_fpy[lane] + v.y * ts;
          const double dpx = static_cast<double>(npx) - r.px, dpy = static_cast<double>(npy) - r.py;
          const double ang = atan2(dpy, dpx);
          const double nh = (ang < 0.0) ? ang + kTwoPi : ((ang == 0.0) ? 0.0 : ang);  // `% (2*pi)`, :102
          dh = wrap_pi(nh - r.heading);
          spd = (1.0 / p.dt) * sqrt(dpx * dpx + dpy * dpy);
          if (fabs(dh) > kPi / 6) {
            dh = ((dh > 0.0) - (dh < 0.0)) * (kPi / 6);
            spd = 0.0;
          }
        } else if (pol == CA_POL_NONCOOP) {  // NonCooperativePolicy.py:21
          const Ego eg = ego_frame(r.px, r.py, r.gx, r.gy, r.heading);
          spd = r.ps;
          dh = -eg.heading_ego;
        } else if (pol == CA_POL_STATIC) {  // StaticPolicy.py:21-23
          r.gx = r.px;
          r.gy = r.py;
        } else if (k.ext) {
          const double e0 = k.ext[2 * i], e1 = k.ext[2 * i + 1];
          if (pol == CA_POL_EXTERNAL) {  // ExternalPolicy.py:14-16
            spd = e0;
            dh = e1;
          } else if (pol == CA_POL_LEARNING) {  // LearningPolicy.py:29-33
            dh = p.max_heading_change * (2. * e1 - 1.);
            spd = r.ps * e0;
          } else if (pol == CA_POL_LEARNING_GA3C) {  // LearningPolicyGA3C.py:24-26, network.py:7-16
            int q = static_cast<int>(e0);
            q = q < 0 ? 0 : (q > 10 ? 10 : q);
            const double s0 = (q < 5) ? 1.0 : ((q < 8) ? 0.5 : 0.0);
            const double h5[5] = {-kPi / 6, -kPi / 12, 0.0, kPi / 12, kPi / 6};
            const double h3[3] = {-kPi / 6, 0.0, kPi / 6};
            spd = r.ps * s0;
            dh = (q < 5) ? h5[q] : h3[(q - 5) % 3];
          }
        }
      }
      const float a0f = static_cast<float>(spd), a1f = static_cast<float>(dh);  // float32 `all_actions`
      if (active && k.o.actions) reinterpret_cast<float2*>(k.o.actions)[i] = make_float2(a0f, a1f);

      // ================= 2. move (agent.py:192-241)
      if (active) {
        if (r.flags & (CA_AT_GOAL | CA_OUT_OF_TIME | CA_IN_COLLISION)) {
          if (r.flags & CA_AT_GOAL) r.flags |= CA_WAS_AT_GOAL;
          if (r.flags & CA_IN_COLLISION) r.flags |= CA_WAS_IN_COLLISION;
          r.vx = r.vy = 0.0;
        } else {
          r.act0 = a0f;
          r.act1 = a1f;
          const double a0 = a0f, a1 = a1f;
          const uint32_t dyn = (r.flags >> CA_DYNAMICS_SHIFT) & 0xF;
          if (dyn != CA_DYN_EXTERNAL) {
            double nh;
            if (dyn == CA_DYN_MAX_TURN_RATE) {  // UnicycleDynamicsMaxTurnRate.py:31-33
              double trn = a1 / p.dt;
              trn = fmin(fmax(trn, -3.0), 3.0);
              nh = wrap_pi(trn * p.dt + r.heading);
            } else {
              nh = wrap_pi(a1 + r.heading);  // UnicycleDynamics.py:28
            }
            double sn, cs;
            sincos(nh, &sn, &cs);
            r.px += a0 * cs * p.dt;
            r.py += a0 * sn * p.dt;
            r.vx = a0 * cs;
            r.vy = a0 * sn;
            r.heading = nh;
          }
          const double qx = r.px - r.gx, qy = r.py - r.gy;
          if (qx * qx + qy * qy <= p.near_goal_threshold * p.near_goal_threshold) r.flags |= CA_AT_GOAL;
          else r.flags &= ~static_cast<uint32_t>(CA_AT_GOAL);
          r.tr -= p.dt;
          r.t += p.dt;
          r.step_num += 1;
          if (r.tr <= 0.0) r.flags |= CA_OUT_OF_TIME;
        }
      }
      __syncthreads();  // everyone is done with the ORCA view of the union
    }

    // ================= 3-6 run once, and a second time for envs that auto-reset this step
    for (int pass = 0; pass < 2; ++pass) {
      // publish the post-move tile
      sh_px[lane] = r.px; sh_py[lane] = r.py; sh_vx[lane] = r.vx; sh_vy[lane] = r.vy; sh_rad[lane] = r.rad;
      __syncthreads();

      // ---- pairwise pass: collisions + nearest (env.py:458-512) and sensor candidates (sensor :76-107)
      Ego eg;
      int cnt = 0;
      bool coll = false;
      double nearest = INFINITY;
      if (active && do_sense) {
        eg = ego_frame(r.px, r.py, r.gx, r.gy, r.heading);
        for (int j = 0; j < N; ++j) {
          double key = INFINITY, po = 0.0, d2o = 0.0;
          if (j != a) {
            const double ox = sh_px[ebase + j], oy = sh_py[ebase + j], orad = sh_rad[ebase + j];
            const double rx = ox - r.px, ry = oy - r.py;
            const double d = sqrt(rx * rx + ry * ry);
            const double cr = r.rad + orad;
            const double gap = d - cr;
            nearest = (gap < nearest) ? gap : nearest;
            coll = coll || (d <= cr);
            if (!(d > p.sensing_horizon)) {
              d2o = d - r.rad - orad;
              key = rint(d2o * 100.0);  // numpy scalar round(x, 2) bucket; ordering of k == ordering of k/100
              po = rx * eg.orx + ry * eg.ory;
              ++cnt;
            }
          }
          kcol[j * BLOCK] = key;
          ocol[j * BLOCK] = po;
          d2col[j * BLOCK] = d2o;
        }
      }

      // ---- rewards + collision flag (env.py:394-456); only on the stepping pass
      if (k.mode == MODE_STEP && pass == 0 && active) {
        double rw = p.reward_time_step;
        if (r.flags & CA_AT_GOAL) {
          if (!(r.flags & CA_WAS_AT_GOAL)) rw = p.reward_at_goal;
        } else if (!(r.flags & CA_WAS_IN_COLLISION)) {
          if (coll) {
            rw = p.reward_collision;
            r.flags |= CA_IN_COLLISION;
          } else {
            if (nearest <= p.getting_close_range) rw = -0.1 - nearest / 2.0;
            if (fabs(static_cast<double>(r.act1)) > p.wiggly_threshold) rw += p.reward_wiggly;
          }
        }
        rw = fmin(fmax(rw, p.reward_min), p.reward_max);
        r.epr += rw;
        reward = static_cast<float>(rw);
      }

      // ---- sensor: rank the candidates, emit rows (sensor :20-55,:109-143)
      if (active && do_sense) {
        const int keep = cnt < K ? cnt : K;
        float* row = k.stage_obs ? (sh_obs + static_cast<size_t>(lane) * W) : (k.o.obs + i * W);
        row[0] = (r.flags & CA_IS_LEARNING) ? 1.f : 0.f;
        row[1] = static_cast<float>(keep);
        row[2] = static_cast<float>(eg.dist);
        row[3] = static_cast<float>(eg.heading_ego);
        row[4] = static_cast<float>(r.ps);
        row[5] = static_cast<float>(r.rad);
        for (int q = 6 + 7 * keep; q < W; ++q) row[q] = 0.f;
        for (int j = 0; j < N; ++j) {
          const double kj = kcol[j * BLOCK];
          int rank = N;
          if (kj < INFINITY) {
            const double oj = ocol[j * BLOCK];
            rank = 0;
            for (int q = 0; q < N; ++q) {
              const double kq = kcol[q * BLOCK], oq = ocol[q * BLOCK];
              rank += (kq < kj) || (kq == kj && (oq < oj || (oq == oj && q < j)));
            }
          }
          rcol[j * BLOCK] = rank;
        }
        for (int j = 0; j < N; ++j) {
          int rank = rcol[j * BLOCK];
          if (rank >= keep) continue;
          if (p.sort_mode == CA_SORT_CLOSEST_LAST) {  // re-sort the kept ones by (-key, p_orth), stable
            const double kj = kcol[j * BLOCK], oj = ocol[j * BLOCK];
            int r2 = 0;
            for (int q = 0; q < N; ++q) {
              const int rq = rcol[q * BLOCK];
              if (rq >= keep) continue;
              const double kq = kcol[q * BLOCK], oq = ocol[q * BLOCK];
              r2 += (kq > kj) || (kq == kj && (oq < oj || (oq == oj && rq < rank)));
            }
            rank = r2;
          }
          const double ox = sh_px[ebase + j], oy = sh_py[ebase + j], orad = sh_rad[ebase + j];
          const double ovx = sh_vx[ebase + j], ovy = sh_vy[ebase + j];
          const double rx = ox - r.px, ry = oy - r.py;
          float* o7 = row + 6 + 7 * rank;
          o7[0] = static_cast<float>(rx * eg.prx + ry * eg.pry);
          o7[1] = static_cast<float>(rx * eg.orx + ry * eg.ory);
          o7[2] = static_cast<float>(ovx * eg.prx + ovy * eg.pry);
          o7[3] = static_cast<float>(ovx * eg.orx + ovy * eg.ory);
          o7[4] = static_cast<float>(orad);
          o7[5] = static_cast<float>(r.rad + orad);
          o7[6] = static_cast<float>(d2col[j * BLOCK]);
        }
      }

      // ---- done / game over (env.py:514-553); only on the stepping pass
      bool need_second = false;
      if (k.mode == MODE_STEP && pass == 0) {
        const bool d = (r.flags & (CA_AT_GOAL | CA_OUT_OF_TIME | CA_IN_COLLISION)) != 0;
        if (d) r.flags |= CA_DONE; else r.flags &= ~static_cast<uint32_t>(CA_DONE);
        done_out = d;
        sh_flag[lane] = r.flags;
        sh_r0[lane] = r.epr;
        sh_r1[lane] = r.t;
        sh_r2[lane] = r.t - r.slt;
        __syncthreads();
        bool all_done = true, all_learning_done = true, any_coll = false, all_goal = true;
        for (int j = 0; j < N; ++j) {
          const uint32_t f = sh_flag[ebase + j];
          const bool dj = (f & CA_DONE) != 0;
          all_done = all_done && dj;
          if (f & CA_STILL_LEARNING) all_learning_done = all_learning_done && dj;
          any_coll = any_coll || (f & CA_IN_COLLISION);
          all_goal = all_goal && (f & CA_AT_GOAL);
        }
        bool over = all_done;
        if (p.game_over_mode == CA_OVER_AGENT0) over = (sh_flag[ebase] & CA_DONE) != 0;
        else if (p.game_over_mode == CA_OVER_LEARNING_DONE) over = all_learning_done;
        over_out = over;
        // per-step outputs that are final regardless of auto-reset
        if (active) {
          k.o.rewards[i] = reward;
          k.o.done[i] = static_cast<uint8_t>(done_out);
          if (a == 0) k.o.game_over[e] = static_cast<uint8_t>(over_out);
        }
        // auto-reset (vec_env.py:120-128) + episode statistics (env_utils.py:56-87)
        do_sense = false;
        if (active && over && k.table) {
          if (a == 0) {
            double tot_r = 0.0, ttg = 0.0, extra = 0.0;
            for (int j = 0; j < N; ++j) {
              tot_r += sh_r0[ebase + j];
              ttg += sh_r1[ebase + j];
              extra += sh_r2[ebase + j];
            }
            double* st = k.s.env_stats + 8 * e;
            st[0] += 1.0;
            if (any_coll) st[1] += 1.0;
            else if (all_goal) st[2] += 1.0;
            else st[3] += 1.0;
            st[4] += ep_step;
            st[5] += tot_r;
            st[6] += ttg;
            st[7] += extra;
          }
          reset_cnt += 1;
          const long c = (k.env_id_offset + e + static_cast<long>(reset_cnt) * k.case_stride) % k.n_cases;
          reset_lane(r, k.table + (c * N + a) * 6, nullptr, p);
          ep_step = 0;
          do_sense = true;
          need_second = true;
        }
      }
      // ---- write the observation block of this tile (coalesced from the LDS staging area)
      const int again = __syncthreads_or(need_second ? 1 : 0);
      if (!again) {
        if (k.stage_obs) {
          // NOTE: in MODE_RESET with a mask, rows of unmasked envs were not staged: copy per env.
          const long total = tile_cnt * W;
          float* dst = k.o.obs + tile_base * W;
          if (k.mode == MODE_RESET && k.reset_mask) {
            for (long q = lane; q < total; q += BLOCK) {
              const long ag = q / W;  // agent within tile
              const long ee = env0 + ag / N;
              if (k.reset_mask[ee]) dst[q] = sh_obs[q];
            }
          } else if (((tile_base * W) & 3) == 0 && (total & 3) == 0) {
            const float4* src4 = reinterpret_cast<const float4*>(sh_obs);
            float4* dst4 = reinterpret_cast<float4*>(dst);
            for (long q = lane; q < (total >> 2); q += BLOCK) dst4[q] = src4[q];
          } else {
            for (long q = lane; q < total; q += BLOCK) dst[q] = sh_obs[q];
          }
        }
        break;
      }
      // some env in this tile was reset: stage everything written so far stays; redo sensing for reset envs
      // (rows of non-reset envs are already in the staging area / in HBM)
    }
    __syncthreads();  // staging area free again before the next step's ORCA view
    do_sense = true;
  }

  // ---- store my agent
  if (active && k.mode != MODE_OBSERVE && (k.mode == MODE_STEP || !k.reset_mask || k.reset_mask[e])) {
    k.s.pos_x[i] = r.px; k.s.pos_y[i] = r.py; k.s.vel_x[i] = r.vx; k.s.vel_y[i] = r.vy;
    k.s.heading[i] = r.heading; k.s.time_remaining[i] = r.tr; k.s.t[i] = r.t;
    k.s.ep_reward[i] = r.epr;
    k.s.goal_x[i] = r.gx; k.s.goal_y[i] = r.gy;
    k.s.radius[i] = r.rad; k.s.pref_speed[i] = r.ps; k.s.slt[i] = r.slt;
    reinterpret_cast<float2*>(k.s.last_action)[i] = make_float2(r.act0, r.act1);
    k.s.flags[i] = r.flags;
    k.s.step_num[i] = r.step_num;
    if (a == 0) { k.s.episode_step[e] = ep_step; k.s.reset_count[e] = reset_cnt; }
  }
}

// ---------------------------------------------------------------- stand-alone ORCA (rvo2 doStep replacement)
struct OrcaArgs {
  int32_t num_envs, num_agents, max_nb;
  const float *pos, *vel, *pref, *radius, *max_speed;
  float collab, time_horizon, time_step, neighbor_dist;
  float* new_vel;
};

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void orca_kernel(const OrcaArgs k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = k.num_agents;
  const int tile_envs = BLOCK / N, tile_n = tile_envs * N;
  const int lane = threadIdx.x;
  const int le = lane / N, a = lane - le * N, ebase = le * N;
  const long e = static_cast<long>(blockIdx.x) * tile_envs + le;
  const bool active = lane < tile_n && e < k.num_envs;
  const long i = e * N + a;
  float* sh_fpx = reinterpret_cast<float*>(smem);
  float* sh_fpy = sh_fpx + BLOCK;
  float* sh_fvx = sh_fpy + BLOCK;
  float* sh_fvy = sh_fvx + BLOCK;
  float* sh_frad = sh_fvy + BLOCK;
  unsigned char* un = smem + align16(static_cast<size_t>(BLOCK) * 5 * 4);
  float* dcol = reinterpret_cast<float*>(un) + lane;
  float4* Lcol = reinterpret_cast<float4*>(un + align16(static_cast<size_t>(BLOCK) * N * 4)) + lane;
  float4* Pcol = Lcol + static_cast<size_t>(N > 1 ? N - 1 : 1) * BLOCK;
  F2 pref = f2(0.f, 0.f);
  float ms = 0.f;
  if (active) {
    const float2 ps = reinterpret_cast<const float2*>(k.pos)[i], vl = reinterpret_cast<const float2*>(k.vel)[i];
    const float2 pf = reinterpret_cast<const float2*>(k.pref)[i];
    sh_fpx[lane] = ps.x; sh_fpy[lane] = ps.y; sh_fvx[lane] = vl.x; sh_fvy[lane] = vl.y;
    sh_frad[lane] = k.radius[i];
    pref = f2(pf.x, pf.y);
    ms = k.max_speed[i];
  }
  __syncthreads();
  if (active) {
    const F2 v = orca_velocity<BLOCK>(sh_fpx, sh_fpy, sh_fvx, sh_fvy, sh_frad, ebase, a, N, pref, ms, k.collab,
                                      k.time_horizon, k.time_step, k.neighbor_dist, k.max_nb, dcol, Lcol, Pcol);
    reinterpret_cast<float2*>(k.new_vel)[i] = make_float2(v.x, v.y);
  }
}

// ---------------------------------------------------------------- host side
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* detail = "") {
  std::snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}

int check_params(const CaParams* p, const CaState* s, const CaOut* o) {
  if (!p || !s || !o) return fail(CA_EINVAL, "cagpu: NULL params/state/out%s");
  if (p->num_envs < 1 || p->num_agents < 1) return fail(CA_EINVAL, "cagpu: num_envs and num_agents must be >= 1%s");
  if (p->num_agents > 64) return fail(CA_EUNSUPPORTED, "cagpu: num_agents > 64 not supported yet (ORCA line tile must fit the 160 KiB LDS)%s");
  if (p->max_obs < 0) return fail(CA_EINVAL, "cagpu: max_obs < 0%s");
  if (p->sort_mode != CA_SORT_CLOSEST_FIRST && p->sort_mode != CA_SORT_CLOSEST_LAST)
    return fail(CA_EUNSUPPORTED, "cagpu: only closest_first / closest_last sorting is implemented%s");
  if (p->game_over_mode < 0 || p->game_over_mode > 2) return fail(CA_EINVAL, "cagpu: bad game_over_mode%s");
  if (!(p->dt > 0.0)) return fail(CA_EINVAL, "cagpu: dt must be > 0%s");
  if (!o->obs || !o->rewards || !o->done || !o->game_over) return fail(CA_EINVAL, "cagpu: NULL output pointer%s");
  const void* ptrs[] = {s->pos_x, s->pos_y, s->vel_x, s->vel_y, s->heading, s->goal_x, s->goal_y, s->radius,
                        s->pref_speed, s->time_remaining, s->t, s->slt, s->ep_reward, s->last_action, s->flags,
                        s->step_num, s->episode_step, s->reset_count, s->env_stats};
  for (const void* q : ptrs)
    if (!q) return fail(CA_EINVAL, "cagpu: NULL state pointer%s");
  return CA_OK;
}

int pick_block(int N) { return N <= 64 ? 64 : (N <= 128 ? 128 : 256); }

template <int BLOCK>
int launch_main(const KArgs& k, hipStream_t st) {
  const int N = k.p.num_agents, W = 6 + 7 * k.p.max_obs;
  KArgs kk = k;
  size_t un_orca = lds_orca_bytes(BLOCK, N);
  size_t un_sense = lds_sense_bytes(BLOCK, N, W, 1);
  kk.stage_obs = 1;
  size_t total = lds_fixed_bytes(BLOCK) + (un_orca > un_sense ? un_orca : un_sense);
  if (total > 64 * 1024) {  // keep >= 2 workgroups per CU: give up the staging area first
    un_sense = lds_sense_bytes(BLOCK, N, W, 0);
    kk.stage_obs = 0;
    total = lds_fixed_bytes(BLOCK) + (un_orca > un_sense ? un_orca : un_sense);
  }
  if (total > 160 * 1024) return fail(CA_EUNSUPPORTED, "cagpu: num_agents too large for the 160 KiB LDS tile%s");
  if (total > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ca_kernel<BLOCK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(total));
    if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int tile_envs = BLOCK / N;
  const unsigned grid = static_cast<unsigned>((k.p.num_envs + tile_envs - 1) / tile_envs);
  hipLaunchKernelGGL(ca_kernel<BLOCK>, dim3(grid), dim3(BLOCK), total, st, kk);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
}

int launch_any(const KArgs& k, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (pick_block(k.p.num_agents)) {
    case 64: return launch_main<64>(k, st);
    case 128: return launch_main<128>(k, st);
    default: return launch_main<256>(k, st);
  }
}

template <int BLOCK>
int launch_orca(const OrcaArgs& k, hipStream_t st) {
  const int N = k.num_agents;
  const size_t total = align16(static_cast<size_t>(BLOCK) * 5 * 4) + lds_orca_bytes(BLOCK, N);
  if (total > 160 * 1024) return fail(CA_EUNSUPPORTED, "cagpu: num_agents too large for the 160 KiB LDS tile%s");
  if (total > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&orca_kernel<BLOCK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(total));
    if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int tile_envs = BLOCK / N;
  const unsigned grid = static_cast<unsigned>((k.num_envs + tile_envs - 1) / tile_envs);
  hipLaunchKernelGGL(orca_kernel<BLOCK>, dim3(grid), dim3(BLOCK), total, st, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
}

}  // namespace

extern "C" {

int cagpu_version(void) { return CAGPU_VERSION; }

const char* cagpu_last_error(void) { return g_err; }

int cagpu_reset(const CaParams* p, const CaState* s, const CaOut* o, const double* cases, const double* headings,
                const uint8_t* mask, void* stream) {
  int rc = check_params(p, s, o);
  if (rc) return rc;
  if (!cases) return fail(CA_EINVAL, "cagpu_reset: NULL cases%s");
  KArgs k;
  std::memset(&k, 0, sizeof(k));
  k.p = *p; k.s = *s; k.o = *o;
  k.reset_cases = cases; k.reset_headings = headings; k.reset_mask = mask;
  k.n_steps = 1; k.mode = MODE_RESET;
  return launch_any(k, stream);
}

static int step_impl(const CaParams* p, const CaState* s, const CaOut* o, const double* ext, const CaAutoReset* ar,
                     int32_t n_steps, void* stream) {
  int rc = check_params(p, s, o);
  if (rc) return rc;
  if (n_steps < 1) return fail(CA_EINVAL, "cagpu: n_steps must be >= 1%s");
  KArgs k;
  std::memset(&k, 0, sizeof(k));
  k.p = *p; k.s = *s; k.o = *o; k.ext = ext;
  if (ar) {
    if (!ar->table || ar->n_cases < 1) return fail(CA_EINVAL, "cagpu: bad CaAutoReset%s");
    k.table = ar->table; k.n_cases = ar->n_cases; k.env_id_offset = ar->env_id_offset; k.case_stride = ar->case_stride;
  }
  k.n_steps = n_steps; k.mode = MODE_STEP;
  return launch_any(k, stream);
}

int cagpu_step(const CaParams* p, const CaState* s, const CaOut* o, const double* ext_actions, const CaAutoReset* ar,
               void* stream) {
  return step_impl(p, s, o, ext_actions, ar, 1, stream);
}

int cagpu_rollout(const CaParams* p, const CaState* s, const CaOut* o, const double* ext_actions, const CaAutoReset* ar,
                  int32_t n_steps, void* stream) {
  return step_impl(p, s, o, ext_actions, ar, n_steps, stream);
}

int cagpu_observe(const CaParams* p, const CaState* s, const CaOut* o, void* stream) {
  int rc = check_params(p, s, o);
  if (rc) return rc;
  KArgs k;
  std::memset(&k, 0, sizeof(k));
  k.p = *p; k.s = *s; k.o = *o;
  k.n_steps = 1; k.mode = MODE_OBSERVE;
  return launch_any(k, stream);
}

int cagpu_orca(int32_t num_envs, int32_t num_agents, const float* pos, const float* vel, const float* pref,
               const float* radius, const float* max_speed, float collab_coeff, float time_horizon, float time_step,
               int32_t max_neighbors, float neighbor_dist, float* new_vel, void* stream) {
  if (num_envs < 1 || num_agents < 1) return fail(CA_EINVAL, "cagpu_orca: bad sizes%s");
  if (num_agents > 64) return fail(CA_EUNSUPPORTED, "cagpu_orca: num_agents > 64 not supported yet%s");
  if (!pos || !vel || !pref || !radius || !max_speed || !new_vel) return fail(CA_EINVAL, "cagpu_orca: NULL pointer%s");
  OrcaArgs k;
  k.num_envs = num_envs; k.num_agents = num_agents; k.max_nb = max_neighbors;
  k.pos = pos; k.vel = vel; k.pref = pref; k.radius = radius; k.max_speed = max_speed;
  k.collab = collab_coeff; k.time_horizon = time_horizon; k.time_step = time_step; k.neighbor_dist = neighbor_dist;
  k.new_vel = new_vel;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (pick_block(num_agents)) {
    case 64: return launch_orca<64>(k, st);
    case 128: return launch_orca<128>(k, st);
    default: return launch_orca<256>(k, st);
  }
}

}  // extern "C"
