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
// Mapping (wave64, no MFMA: this is branchy element-wise geometry): one WORKGROUP per tile of WHOLE
// envs (floor(64 / num_agents) envs = up to 64 agents; 4 envs at the benchmark's 10 agents), so every neighbour an
// agent needs is owned by the same workgroup and travels through LDS, never through HBM:
//   * HBM loads/stores are agent-major SoA -> lane i touches element base+i: fully coalesced;
//   * serial per-agent chains (the scan of the linear programme, float64 trig, rewards, flags) run one LANE per agent
//     on wave 0, the tile's "agent wave";
//   * the O(N^2) pairwise work (ORCA half-planes, the 1-D programmes of the linear programme, collision gaps, sensor
//     keys / ranks / rows) runs one THREAD per (agent, other) -- or per (agent, line), or per unordered pair -- across
//     all waves, reading the tile's positions / velocities / radii from LDS; where a wave holds whole agents the lanes
//     of an agent exchange through ds_bpermute / DPP instead;
//   * per-(agent, slot) work arrays live in LDS as [slot][agent] columns -> bank-conflict free;
//   * the observation rows go straight to HBM (or, for small launches, are staged in LDS and leave as one block).
// A workgroup's step is a chain of dependent phases; the launch is latency-bound, not bandwidth-bound (DESIGN.md
// section 4, profiles/r02_kernel_geometry.md): wave priorities, few barriers and short chains are what matter.
// Envs never interact, so n-step rollouts need no grid-wide synchronisation.
//
// Numerics: float64 state in the reference's operation order, the action pair rounded to float32
// (env.py:305-307), ORCA in float with the RVO2 operation order.  Built with -ffp-contract=off: a
// fused multiply-add in `dx*dx + dy*dy <= r*r` would change discrete events.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <type_traits>

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
  const float* reset_obs;  // [n_cases, N, W] reset observation of every case (nullptr: re-sense after an auto-reset)
  const float* reset_plan; // [n_cases, N, 4] next_action of every case's reset state (pipelined kernel only)
  unsigned long long heading_seed;  // != 0: random initial headings at an auto-reset (training mode)
  // explicit reset
  const double* reset_cases;
  const double* reset_headings;
  const uint8_t* reset_mask;
  CaMap map;  // static obstacles for wall collisions (static_bits == NULL: none)
  int32_t n_steps, mode, stage_obs;
  int32_t tile_envs;  // envs per workgroup (<= ROW / num_agents)
  int32_t col_stride; // columns of the per-(agent, slot) LDS tiles: ROW, or N for single-env tiles (N > 32)
  double inv_rvo_dt;  // 1 / p.rvo_dt (RVOPolicy.py:106), divided once on the host
  // cagpu_rollout_ring: step t of the launch writes its outputs to slot t of the caller's ring -- element strides between
  // the slots of the observation block (E N W), of the per-agent outputs (E N; x 2 for the action pairs) and of game_over (E);
  // all 0 everywhere else (every step writes the same buffers)
  int64_t ring_obs, ring_agent, ring_env;
  // cagpu_rollout_ring: != 0: the pipelined n-step kernel also stores the state it starts from at (address + snap_delta)
  int64_t snap_delta;
  int32_t ablate;  // timing experiments only (-DCAGPU_ABLATE + env CAGPU_ABLATE); 0 in product builds
  int32_t launch_seq;  // a per-process launch counter: tags the words of the n-step kernel's per-CU progress table (cagpu_pipe.inc)
  float inv_h_f, inv_dt_f;  // pipelined kernels: 1.0f / float(p.rvo_time_horizon), 1.0f / float(p.rvo_dt) -- IEEE float quotients, set by launch_pipe2
  int32_t yield_t;     // > 0: progress-fair priorities among the workgroups of a CU (PIPE_PRIO in cagpu_pipe.inc); set by launch_pipe2
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
// The same two operations for the step kernel's ORCA phases: the compiler's correctly rounded sequences WITHOUT their range
// handling (v_div_scale / v_div_fixup; the 2^32 rescaling and the class test of the square root) -- the same arithmetic
// whenever no operand, intermediate or result is denormal, huge or NaN: 8 instead of 11 and 9 instead of 14 instructions,
// all of them on dependent chains (linearProgram1 / 3 are chains of divisions).  scratch/divsqrt_check.hip: identical
// bits on 2^28 random operands with exponents in [-60, 60] (division, reciprocal) / x in [2^-80, 2^80] (square root).
// ORCA's operands are velocities, positions and their differences -- 0 (handled: 0 / b = 0, sqrt(0) = 0) or 1e-16 .. 1e8
// in magnitude.  The stand-alone cagpu_orca kernel keeps the compiler's `/` and sqrtf for arbitrary inputs, and both
// are held to the same oracle bit for bit (tests/test_gpu_parity.py).
#if defined(CAGPU_EXP) && (CAGPU_EXP & 4)
__device__ __forceinline__ float divq(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrtq(float a) { return sqrtf(a); }
#else
__device__ __forceinline__ float divq(float a, float b) {
  float y = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y, 1.0f);
  y = __builtin_fmaf(e, y, y);
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ float sqrtq(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
  const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  float o = (r_dn <= 0.0f) ? s_dn : s;
  o = (r_up > 0.0f) ? s_up : o;
  return (x == 0.0f) ? x : o;
}
#endif
// ... and for float64 (distances, preferred velocity, ego frame: UnicycleDynamics / Dynamics / the sensor): the compiler's
// sequences are rsq + 9 multiply-adds (square root) and rcp + 2 Newton steps + one residual correction (divide); the range
// handling around them (v_ldexp / v_cmp_class / v_div_scale / v_div_fixup) costs 8 and 3 more instructions per call.
// scratch/divsqrt_check.hip: identical bits on 2^28 random operands with exponents in [-300, 300] / [-600, 600].
#if defined(CAGPU_EXP) && (CAGPU_EXP & 4)
__device__ __forceinline__ double divd(double a, double b) { return a / b; }
__device__ __forceinline__ double sqrtd(double a) { return sqrt(a); }
#else
__device__ __forceinline__ double divd(double a, double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  const double q = a * y;
  return __builtin_fma(__builtin_fma(-b, q, a), y, q);
}
__device__ __forceinline__ double sqrtd(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return (x == 0.0) ? x : g;
}
#endif
// RVO2's Vector2 / float: multiply by the reciprocal
__device__ __forceinline__ F2 over(F2 a, float s) { const float inv = divf(1.0f, s); return f2(a.x * inv, a.y * inv); }
__device__ __forceinline__ F2 unitf(F2 a) { return over(a, sqrtf_rn(dotf(a, a))); }
__device__ __forceinline__ F2 unitq(F2 a) { const float inv = divq(1.0f, sqrtq(dotf(a, a))); return f2(a.x * inv, a.y * inv); }

__device__ __forceinline__ double wrap_pi(double a) {  // util.py:141-146 ([-pi, pi))
  if (a >= kPi) a -= kTwoPi;   // first iteration of the reference's while loops, branch-free
  if (a < -kPi) a += kTwoPi;
  if (!(a < kPi) || a < -kPi) {  // more than one turn out of range: the general loops (bounded)
    for (int it = 0; it < 4096 && a >= kPi; ++it) a -= kTwoPi;
    for (int it = 0; it < 4096 && a < -kPi; ++it) a += kTwoPi;
  }
  return a;
}

// sin and cos of a heading in [-pi, pi] (UnicycleDynamics.py:29-32): Cody-Waite reduction by pi/2 (two-part constant,
// |k| <= 2) and the fdlibm kernel polynomials -- ~45 instructions where the general-range libm sincos needs ~150.
// Accuracy (checked against long-double libm on 2e7 arguments): <= 1 ulp, except within 1e-11 of a multiple of pi/2
// where the ABSOLUTE error stays below 1e-26 (the result itself is ~1e-12 there); 97.6 % of the results are
// bit-identical to glibc's.  The products these feed are positions of order 1..10 m.
__device__ __forceinline__ void sincos_heading(double x, double& s, double& c) {
  const double kf = rint(x * 6.36619772367581382433e-01);
  const int k = static_cast<int>(kf);
  const double r = __builtin_fma(-kf, 1.57079632673412561417e+00, x);  // first 33 bits of pi/2: exact product
  const double w = kf * 6.07710050650619224932e-11;                     // pi/2 - the 33 bits
  const double y = r - w;
  const double yt = (r - y) - w;
  const double z = y * y;
  const double v = z * y;
  const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.58969099521155010221e-10,
                    -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                    8.33333333332248946124e-03);
  const double sn = y - ((z * (0.5 * yt - v * rs) - yt) - v * -1.66666666666666324348e-01);
  const double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                    -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                    2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double ww = 1.0 - hz;
  const double cs = ww + (((1.0 - ww) - hz) + (z * rc - y * yt));
  const bool swap = (k & 1) != 0;
  const double s0 = swap ? cs : sn, c0 = swap ? sn : cs;
  s = (k & 2) ? -s0 : s0;
  c = (((k + 1) & 2) != 0) ? -c0 : c0;
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

// The same half-plane without divergent branches (step kernel): the three cases of computeNewVelocity share ONE square
// root and ONE reciprocal on selected inputs -- cut-off disc: sqrt(w.w), 1 / |w| (also the overlap case, which is the
// cut-off disc of a one-step horizon); legs: sqrt(distSq - R^2), 1 / distSq -- so a wave whose lanes fall into different
// cases runs the long dependent chain once instead of once per case.  Every selected expression is the one above:
// bit-identical results (tests: the stand-alone cagpu_orca kernel keeps the branchy form, both match the oracle).
__device__ __forceinline__ float4 half_plane_sel(F2 mpos, F2 mvel, float mrad, F2 opos, F2 ovel, float orad, float collab,
                                                 float inv_h, float inv_dt) {
  const F2 rp = opos - mpos;
  const F2 rv = mvel - ovel;
  const float d2 = dotf(rp, rp);
  const float R = mrad + orad;
  const float R2 = sqf(R);
  const bool far = d2 > R2;
  const float kk = far ? inv_h : inv_dt;
  const F2 w = rv - kk * rp;
  const float w2 = dotf(w, w);
  const float dp1 = dotf(w, rp);
  const bool disc = !far || (dp1 < 0.0f && sqf(dp1) > R2 * w2);
  const float sq = sqrtq(disc ? w2 : (d2 - R2));  // |w|  or  the leg length
  const float inv = divq(1.0f, disc ? sq : d2);
  // cut-off disc (and overlap)
  const F2 uw = f2(w.x * inv, w.y * inv);
  const F2 dir_a = f2(uw.y, -uw.x);
  const F2 u_a = (R * kk - sq) * uw;
  // legs
  const bool left = detf(rp, w) > 0.0f;
  const float bx = left ? (rp.x * sq - rp.y * R) : (rp.x * sq + rp.y * R);
  const float by = left ? (rp.x * R + rp.y * sq) : (-rp.x * R + rp.y * sq);
  const F2 t = f2(bx * inv, by * inv);
  const F2 dir_b = left ? t : f2(-t.x, -t.y);
  const float dp2 = dotf(rv, dir_b);
  const F2 u_b = dp2 * dir_b - rv;
  const F2 dir = disc ? dir_a : dir_b;
  const F2 u = disc ? u_a : u_b;
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
  e.dist = sqrtd(dx * dx + dy * dy);
  if (e.dist > 1e-8) {
    e.prx = divd(dx, e.dist);
    e.pry = divd(dy, e.dist);
  } else {
    e.prx = dx;
    e.pry = dy;
  }
  e.orx = -e.pry;
  e.ory = e.prx;
  e.heading_ego = wrap_pi(heading - atan2(e.pry, e.prx));
  return e;
}

// time to impact of `host` with `other` (util.py:23-83 compute_time_to_impact + :101-127 tangent_vecs_from_external_pt),
// float64 in the numpy operation order; only used by the time_to_impact sorting mode of the sensor
__device__ double time_to_impact(double hx, double hy, double ox, double oy, double hvx, double hvy, double ovx,
                                 double ovy, double r) {
  const double v0 = hvx - ovx, v1 = hvy - ovy;
  const double xp = hx, yp = hy, a = ox, b = oy;
  const double sq = (xp - a) * (xp - a) + (yp - b) * (yp - b) - r * r;
  if (sq < 0) return 0.0;
  const double st = sqrt(sq);
  const double xnum1 = r * r * (xp - a), xnum2 = r * (yp - b) * st;
  const double ynum1 = r * r * (yp - b), ynum2 = r * (xp - a) * st;
  const double den = (xp - a) * (xp - a) + (yp - b) * (yp - b);
  const double c1x = ((xnum1 + xnum2) / den + a) - xp, c1y = ((ynum1 - ynum2) / den + b) - yp;
  const double c2x = ((xnum1 - xnum2) / den + a) - xp, c2y = ((ynum1 + ynum2) / den + b) - yp;
  const double x11 = c1x * v1 - c1y * v0, x12 = c1x * c2y - c1y * c2x;
  const double x21 = c2x * v1 - c2y * v0, x22 = c2x * c1y - c2y * c1x;
  if (!(x11 * x12 >= 0 && x21 * x22 >= 0)) return INFINITY;
  if (fabs(v0) < 1e-5 && fabs(v1) < 1e-5) return INFINITY;
  const double px = hx, py = hy;
  double x1, x2, y1, y2;
  if (fabs(v0) < 1e-5) {
    x1 = x2 = px;
    const double A = 1, B = -2 * b, Cc = b * b + (px - a) * (px - a) - r * r;
    y1 = (-B + sqrt(B * B - 4 * A * Cc)) / (2 * A);
    y2 = (-B - sqrt(B * B - 4 * A * Cc)) / (2 * A);
  } else {
    const double m = v1 / v0;
    const double A = 1 + m * m;
    const double B = -2 * a + 2 * m * (py - b - m * px);
    const double Cc = a * a - r * r + (m * px - (py - b)) * (m * px - (py - b));
    x1 = (-B + sqrt(B * B - 4 * A * Cc)) / (2 * A);
    x2 = (-B - sqrt(B * B - 4 * A * Cc)) / (2 * A);
    y1 = m * (x1 - px) + py;
    y2 = m * (x2 - px) + py;
  }
  const double d1 = sqrt((x1 - px) * (x1 - px) + (y1 - py) * (y1 - py));
  const double d2 = sqrt((x2 - px) * (x2 - px) + (y2 - py) * (y2 - py));
  const double d = d2 < d1 ? d2 : d1;
  return d / sqrt(v0 * v0 + v1 * v1);
}

// The device's fault word (cagpu_device_faults): bit 0 = a bounded hand-over poll of the pipelined step kernel ran out
// (cagpu_pipe.inc), bit 1 = a GA3C-CADRL operand left the fp16 range of the network kernel's two-plane split (cagpu_ga3c.inc).
__device__ unsigned int g_fault = 0u;

#include "cagpu_grouplp.inc"

// x / D for small non-negative x: with a compile-time D two full-rate instructions (24-bit multiply by ceil(2^20 / D),
// shift; exact while x * D < 2^20), otherwise the float reciprocal form (convert, add, multiply, convert).  The remainder
// x - q * D likewise takes the 24-bit multiply (v_mul_lo_u32 is a quarter-rate instruction).
template <int D>
__device__ __forceinline__ int div_small(const int x, const float inv_d) {
  if (D > 0) return static_cast<int>(__umul24(static_cast<unsigned>(x), static_cast<unsigned>(((1 << 20) + (D > 0 ? D : 1) - 1) / (D > 0 ? D : 1))) >> 20);
  return static_cast<int>((static_cast<float>(x) + 0.5f) * inv_d);
}

// for (q = 0; q < n; ++q) body(q) in blocks of BLK: with a run-time n (the generic kernel) the loads of a block are in
// flight together instead of one LDS round trip per iteration (N = 50: the rank loop of the sensor took 57 k of the
// step's 155 k cycles as a plain loop); with a compile-time n everything unrolls as before.
template <int BLK, typename F>
__device__ __forceinline__ void for_n(const int n, F&& body) {
  int q = 0;
  for (; q + BLK <= n; q += BLK) {
#pragma unroll
    for (int u = 0; u < BLK; ++u) body(q + u);
  }
#pragma unroll
  for (int u = 0; u < BLK - 1; ++u)
    if (q + u < n) body(q + u);
}

// ---------------------------------------------------------------- main kernel
// Work decomposition of one tile (ROW = 64 agent slots = floor(64/N) whole envs) on a workgroup of NT threads:
//   agent phases  (A*): one LANE per agent, on wave 0 only -- the serial per-agent chains (incremental LP,
//                       float64 atan2/sincos, flags).  The other waves wait at the barrier and cost no issue slots.
//   pair phases   (P*): one THREAD per ordered (agent, other) pair, spread over all NT threads -- ORCA
//                       neighbour distances + half-planes, collision gaps, sensor keys / ranks / row emission.
// Everything the phases exchange lives in LDS; per-(agent, slot) arrays are [slot][ROW] columns so that a phase that
// walks slots for a fixed agent (wave 0) and a phase that walks agents for a fixed slot both stay conflict-free.
constexpr int ROW = 64;
// Column stride of the per-(agent, slot) LDS tiles [slot][CS]: ODD, so that the lanes of one agent, which hold different
// slots -- the half-plane store Lmat[rank * CS + agent] of the ORCA phase: a stride of 64 float4 put all nine of them on
// the same four banks (SQ_LDS_BANK_CONFLICT 767 k cycles per launch at 4096 x 10, profiles/r02_rocprof_summary.md) --
// spread over the banks, while lanes that walk agents for a fixed slot stay consecutive.
#ifndef CAGPU_CSPAD
#define CAGPU_CSPAD 1
#endif
constexpr int CS_ROW = ROW + CAGPU_CSPAD;
constexpr int KEY_NONE = 2147483647;  // sort key of a pair that is not sensed (self, beyond the sensing horizon)
#ifdef CAGPU_ABLATE  // experiment build (scratch/): run-time switches in k.ablate + in-kernel phase timers
#define AB(bit) (k.ablate & (bit))
#define DUP(bit) for (int rep_ = 0; rep_ < ((k.ablate & (bit)) ? 2 : 1); ++rep_)  // run an idempotent phase twice: its cost in situ
__device__ unsigned long long g_prof[16];
__device__ unsigned long long g_wgprof[1024 * 16];
#define TICK(slot) do { if (tid == 0) { const unsigned long long now_ = clock64(); sh_prof[slot] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define AB(bit) false
#define DUP(bit)
#if defined(CAGPU_WGTIME)
// product-like build with one wall-clock stamp (100 MHz) per phase boundary on thread 0: g_wgphase[wg][slot] += elapsed
__device__ unsigned int g_wgphase[4096 * 16];
#define TICK(slot) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); wg_ph[slot] += static_cast<unsigned>(now_ - wg_prev); wg_prev = now_; } } while (0)
#else
#define TICK(slot) do {} while (0)
#endif
#endif
#ifdef CAGPU_WGTIME  // experiment build: per-workgroup wall-clock stamps of the LAST launch (100 MHz counter), no timers inside
__device__ unsigned long long g_wgtime[4096 * 8];
#endif
// Compile-time experiment switches (scratch/ builds: -DCAGPU_EXP=<bit mask>, libcagpu_exp<mask>_fast.so); 0 in the product
#ifndef CAGPU_EXP
#define CAGPU_EXP 0
#endif
#define EXP(bit) (((CAGPU_EXP) & (bit)) != 0)
#define LP1_UNROLL _Pragma("unroll")
// Workgroup barrier between phases that exchange data through LDS only: waits for this wave's LDS traffic, not for its
// outstanding global stores (__syncthreads() = workgroup fence + barrier also drains vmcnt, i.e. every observation
// store has to be acknowledged by the L2 before the phase after it may start).
#if (CAGPU_EXP & 2)
#define WG_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define WG_SYNC() __syncthreads()
#endif
// Wave priorities per phase (s_setprio 0..3).  Four workgroups share a CU, so every SIMD holds one wave of each of them,
// usually in different phases.  The agent wave's serial chains (one lane per agent: they are what the workgroup's other
// three waves are waiting for) and the linearProgram3 stragglers issue first, then pair phases in the order of the
// step (a workgroup that is behind overtakes one that is ahead): 21.6 -> 20.3 us per step, rollout 14.1 -> 12.6
// (profiles/r02_kernel_geometry.md).  PRIO(...) lists the level of a phase under schemes 1 .. 8 (-DCAGPU_PRIO=<n>; 0 =
// hardware default): 1 agent wave 3 / ORCA pairs 2 / sensor pairs 1; 2 only linearProgram3 raised; 3 "earlier phase =
// higher"; 4 agent wave 3, everything else 1; **5 = 1 with the emission phase and the post-linearProgram3 wait at 0 (the
// product)**; 6 - 8 variations of 5 (sensor phases one level up; ORCA pairs at 3; emission at 1), all 0.05 - 0.3 us behind 5.
#ifndef CAGPU_PRIO
#define CAGPU_PRIO 5
#endif
#define PRIO(a, b, c, d, e, ...) do { constexpr int pr_[] = {0, a, b, c, d, e, __VA_ARGS__}; if (CAGPU_PRIO) __builtin_amdgcn_s_setprio(pr_[CAGPU_PRIO]); } while (0)
// Pair items w = 0 .. n_items-1 are dealt to the NT threads in rounds; odd rounds run BACKWARDS over the threads, so the
// last, partial round lands on the highest waves and wave 0 -- the agent wave -- is free for its one-lane-per-agent work
// while the other waves finish the pair phase (400 items on 256 threads: wave 0 has one round, waves 2 and 3 two).
#define FOR_ITEMS_UPTO(w, count)                                                       \
  for (int base_ = 0, odd_ = 0; base_ < (count); base_ += NT, odd_ ^= 1)              \
    if (const int w = base_ + (odd_ ? (NT - 1 - tid) : tid); w < (count))
#define FOR_PAIR_ITEMS(w) FOR_ITEMS_UPTO(w, n_items)

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~static_cast<size_t>(15); }
#include "cagpu_scan.inc"
#include "cagpu_ga3c.inc"
#include "cagpu_gen.inc"

// fixed: 7 f64 + 10 f32 + 4 u32 per agent slot (the 3 f64 of the episode scratch alias six ORCA float arrays) + the
// linearProgram3 queue (length + up to ROW entries, then the number of ORCA queries)
__host__ __device__ inline size_t lds_fixed_bytes(int row = ROW) {
  return static_cast<size_t>(row) * (7 * 8 + 10 * 4 + 4 * 4) + align16(static_cast<size_t>(row + 3) * 4);
}
// union, ORCA view: half-planes [N-1][CS] float4, the solution of every line's 1-D programme [N-1][CS] float2 + its
// feasibility byte; the projected lines of linearProgram3 live in the registers of the solving group
__host__ __device__ inline size_t lds_orca_lines(int N, int cs) { return static_cast<size_t>(cs) * (N > 1 ? N - 1 : 1); }
__host__ __device__ inline size_t lds_orca_bytes(int N, int cs = ROW) {
  return lds_orca_lines(N, cs) * (16 + 8) + align16(lds_orca_lines(N, cs));
}
// union, sensor view: p_orth / gap / time-to-impact [N][CS] f64, key i32, dist_2_other f32, rank u8,
// obs staging [ROW*W] f32
__host__ __device__ inline size_t lds_sense_bytes(int N, int W, int stage, int tti, int cs = ROW, int row = ROW) {
  return align16(static_cast<size_t>(cs) * N * ((tti ? 3 : 2) * 8 + 9)) + (stage ? align16(static_cast<size_t>(row) * W * 4) : 0);
}

struct Lane {  // per-lane registers of one agent (wave 0)
  double px, py, vx, vy, heading, gx, gy, rad, ps, tr, t, slt, epr;
  float act0, act1;
  uint32_t flags;
  int32_t step_num;
  double td;  // Agent.turning_dir (only where CaState.turning_dir is given)
};

// Agent.turning_dir after UnicycleDynamics.step turned the agent to heading `nh` (UnicycleDynamics.py:41-47)
__device__ __forceinline__ double turning_dir_next(const double td, const double nh) {
  if (fabs(td) < 1e-5) return 0.11 * ((nh > 0.0) - (nh < 0.0));
  if (td * nh < 0.0) return fmax(-kPi, fmin(kPi, -td + nh));
  return ((td > 0.0) - (td < 0.0)) * fmax(0.0, fabs(td) - 0.1);
}

// test_cases.py:545-557 + agent.py:59-138
// (the heading travels by value: a pointer to a local kept a 16-byte scratch object alive in every instantiation)
template <typename Params>  // (CaParams, or the pipelined kernel's address-space-qualified view of it)
__device__ __forceinline__ void reset_lane(Lane& r, const double* c, const bool has_heading, const double heading,
                                           const Params& p) {
  r.px = c[0]; r.py = c[1]; r.gx = c[2]; r.gy = c[3]; r.ps = c[4]; r.rad = c[5];
  r.vx = r.vy = 0.0;
  r.heading = has_heading ? heading : atan2(c[3] - c[1], c[2] - c[0]);
  const double dx = c[0] - c[2], dy = c[1] - c[3];
  r.slt = (sqrt(dx * dx + dy * dy) - p.near_goal_threshold) / c[4];
  double tr = p.max_time_ratio * r.slt;
  if (p.dt > tr) tr = p.dt;
  r.tr = tr;
  r.t = 0.0;
  r.epr = 0.0;
  r.td = 0.0;
  r.act0 = r.act1 = 0.f;
  r.step_num = 0;
  r.flags &= ~(0x3Fu | CA_ABSENT | CA_PLAN_VALID);
  if (p.ragged && !(c[5] > 0.0)) {  // a padding row of a ragged table: the slot holds no agent in this episode
    r.px = r.py = r.gx = r.gy = r.rad = r.heading = r.slt = r.tr = 0.0;
    r.ps = 1.0;
    r.flags |= CA_ABSENT | CA_DONE | CA_AT_GOAL | CA_WAS_AT_GOAL;
  }
}

template <int NT, bool STAGE, int NC, bool MULTI, bool RO, int TE = 0>
// n-step: <= 128 VGPRs (four 4-wave workgroups per CU) -- except N = 20, whose single-step form already needs 172 (two
// workgroups per CU either way): held to 128 its n-step form spilled ~200 VGPRs to scratch around the step loop
__global__ __launch_bounds__(NT, MULTI ? (NC == 20 ? 2 : 4) : 1) void ca_kernel(const KArgs k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const CaParams& p = k.p;
  const int N = NC ? NC : p.num_agents;  // NC > 0: compile-time agent count (loops unroll, divisions fold)
  const int K = p.max_obs, W = 6 + 7 * K;
  const float inv_n = 1.0f / static_cast<float>(N);
  const int tile_envs = TE ? TE : (NC ? ROW / (NC ? NC : 1) : k.tile_envs);  // compile-time in the specialised kernels
  const int tile_n = tile_envs * N;
  const int n_items = tile_n * N;
  const int tid = threadIdx.x;
  // Wave 0 is the workgroup's AGENT WAVE (one lane per agent).  It needs no rotation across workgroups: the dispatcher
  // starts each workgroup of a CU on a different SIMD (scratch/hwid.hip: with 4 workgroups of 4 waves per CU every SIMD
  // holds exactly one wave 0), and rotating the role measured no difference (profiles/r02_kernel_geometry.md).
  const bool wave0 = tid < ROW;
  // identity of the agent on this lane (meaningful on wave 0 only)
  const int lane = tid & (ROW - 1);
  const int le = lane / N, a = lane - le * N;
  const long env0 = static_cast<long>(blockIdx.x) * tile_envs;
  const long e = env0 + le;
  const bool active = wave0 && (lane < tile_n) && (e < p.num_envs);
  const int ebase = active ? le * N : 0;
  const long i = e * N + a;
  const long tile_base = env0 * N;
  float* obs_tile = k.o.obs + tile_base * (6 + 7 * p.max_obs);  // the tile's observation rows (uniform: scalar arithmetic)
  long ring_a = 0, ring_e = 0;  // n-step kernel: this step's slot of the caller's output ring (cagpu_rollout_ring), in elements
  long tile_cnt = static_cast<long>(p.num_envs) * N - tile_base;
  if (tile_cnt > tile_n) tile_cnt = tile_n;

  // ---- LDS
  double* sh_px = reinterpret_cast<double*>(smem);
  double* sh_py = sh_px + ROW;
  double* sh_vx = sh_py + ROW;
  double* sh_vy = sh_vx + ROW;
  double* sh_rad = sh_vy + ROW;
  double* sh_prx = sh_rad + ROW;  // ego frame ref_prll (ref_orth = (-pry, prx))
  double* sh_pry = sh_prx + ROW;
  float* sh_fpx = reinterpret_cast<float*>(sh_pry + ROW);  // ORCA's float bodies (dead once the policy phase is over)
  float* sh_fpy = sh_fpx + ROW;
  float* sh_fvx = sh_fpy + ROW;
  float* sh_fvy = sh_fvx + ROW;
  float* sh_frad = sh_fvy + ROW;
  float* sh_fms = sh_frad + ROW;                      // speed limit (float pref_speed)
  double* sh_r0 = reinterpret_cast<double*>(sh_fpx);  // episode-stat scratch of A3 -> A4: aliases the six arrays above
  double* sh_r1 = sh_r0 + ROW;
  double* sh_r2 = sh_r1 + ROW;
  uint32_t* sh_flag = reinterpret_cast<uint32_t*>(sh_fms + ROW);
  int* sh_q = reinterpret_cast<int*>(sh_flag + ROW);  // 1: this agent queries ORCA this step
  int* sh_nb = sh_q + ROW;                            // its neighbour count n
  int* sh_sense = sh_nb + ROW;                        // 1: (re)write this agent's observation in this pass
  float* sh_vrx = reinterpret_cast<float*>(sh_sense + ROW);  // ORCA velocity of each agent
  float* sh_vry = sh_vrx + ROW;
  float* sh_fprx = sh_vry + ROW;                      // its preferred velocity (float)
  float* sh_fpry = sh_fprx + ROW;
  int* sh_q3 = reinterpret_cast<int*>(sh_fpry + ROW);  // linearProgram3 queue: [0] = length, [1 ..] = entries
  unsigned char* un = smem + lds_fixed_bytes(ROW);
  // Column stride of the per-(agent, slot) tiles: ROW = 64 (a shift) in general; for single-env tiles (N > 32) the N
  // columns actually used, which is what lets two 50-agent workgroups share a CU's LDS.
  const int CS = NC ? CS_ROW : k.col_stride;
  // ORCA view of the union
  float4* Lmat = reinterpret_cast<float4*>(un);                                          // [N-1][CS] half-planes
  float2* Rmat = reinterpret_cast<float2*>(Lmat + lds_orca_lines(N, CS));                // [N-1][CS] 1-D optimum on line i
  uint8_t* okmat = reinterpret_cast<uint8_t*>(Rmat + lds_orca_lines(N, CS));             // [N-1][CS] line i feasible
  // sensor view of the union
  // p_orth and the collision gap stay float64 (they decide order / collisions); the sort bucket rint(100 d) is an
  // integer (int32, KEY_NONE = not sensed) and dist_2_other is only emitted as float32: 25 B per pair instead of 33.
  double* omat = reinterpret_cast<double*>(un);          // [N][CS] p_orth
  double* gmat = omat + static_cast<size_t>(N) * CS;     // [N][CS] centre distance - combined radius
  const int has_tti = (p.sort_mode == CA_SORT_TIME_TO_IMPACT) ? 1 : 0;
  double* tmat = gmat + static_cast<size_t>(N) * CS;     // [N][CS] time to impact (time_to_impact sorting only)
  int* kmat = reinterpret_cast<int*>(tmat + static_cast<size_t>(has_tti ? N : 0) * CS);  // [N][CS] sort key
  float* d2mat = reinterpret_cast<float*>(kmat + static_cast<size_t>(N) * CS);           // [N][CS] dist_2_other
  // (one spare byte per pair behind d2mat: the first-pass ranks closest_last's second pass read until round 4)
  float* sh_obs = reinterpret_cast<float*>(un + align16(static_cast<size_t>(CS) * N * ((has_tti ? 3 : 2) * 8 + 9)));

  // ---- load my agent
  Lane r;
  r.px = r.py = r.vx = r.vy = r.heading = r.gx = r.gy = 0.0;
  r.rad = r.ps = 1.0; r.tr = r.t = r.slt = r.epr = 0.0; r.td = 0.0;
  r.act0 = r.act1 = 0.f; r.flags = CA_DONE | CA_AT_GOAL; r.step_num = 0;
  int ep_step = 0, reset_cnt = 0;
  if (active) {
    if (k.s.turning_dir) r.td = k.s.turning_dir[i];
    r.px = k.s.pos_x[i]; r.py = k.s.pos_y[i]; r.vx = k.s.vel_x[i]; r.vy = k.s.vel_y[i];
    r.heading = k.s.heading[i]; r.gx = k.s.goal_x[i]; r.gy = k.s.goal_y[i];
    r.rad = k.s.radius[i]; r.ps = k.s.pref_speed[i]; r.tr = k.s.time_remaining[i]; r.t = k.s.t[i];
    r.slt = k.s.slt[i]; r.epr = k.s.ep_reward[i];
    const float2 la = reinterpret_cast<const float2*>(k.s.last_action)[i];
    r.act0 = la.x; r.act1 = la.y;
    r.flags = k.s.flags[i];
    r.step_num = k.s.step_num[i];
    ep_step = k.s.episode_step[e];
    reset_cnt = k.s.reset_count[e];
  }
  bool statics_dirty = false;  // goal / radius / pref_speed / slt changed (reset, StaticPolicy)
  bool do_sense = active;
  float reward = 0.f;

  if (k.mode == MODE_RESET) {
    do_sense = active && (!k.reset_mask || k.reset_mask[e]);
    if (do_sense) {
      reset_lane(r, k.reset_cases + i * 6, k.reset_headings != nullptr, k.reset_headings ? k.reset_headings[i] : 0.0, p);
      ep_step = 0;
      reset_cnt = 0;
      statics_dirty = true;
    }
  }

#ifdef CAGPU_ABLATE
  __shared__ unsigned long long sh_prof[16];
  if (tid < 16) sh_prof[tid] = 0;
  __syncthreads();
  unsigned long long tprev_ = clock64();
#endif
#ifdef CAGPU_WGTIME
  unsigned long long wg_t0 = 0, wg_info = 0, wg_lp3 = 0, wg_prev = 0;
  unsigned wg_ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long wg_c0 = 0;
  if (tid == 0) { wg_t0 = wg_prev = wall_clock64(); wg_c0 = clock64(); }
#endif
  // The phases are straight-line code in the single-step kernel (lambdas inlined at their call sites): with the
  // n-step loop and the two-pass sensing loop around them the register allocator keeps ~100 more VGPRs alive around
  // the back edges (profiles/r01_kernel_geometry.md).  MULTI = true keeps the step loop for cagpu_rollout.
  auto one_step = [&]() {
    // In the n-step kernel, hide the loop invariance of everything derived from the thread id (one laundered copy
    // per iteration): otherwise LICM hoists dozens of per-thread addresses / predicates out of the step loop and
    // the allocator has to keep them alive across it.
    int tid_l = threadIdx.x;
    if (MULTI) asm volatile("" : "+v"(tid_l));
    const int tid = tid_l;
    const bool wave0 = tid < ROW;
    const int lane = tid & (ROW - 1);
    const int le = lane / N, a = lane - le * N;
    const long e = env0 + le;
    const bool active = wave0 && (lane < tile_n) && (e < p.num_envs);
    const int ebase = active ? le * N : 0;
    const long i = e * N + a;
    TICK(0);
    if (k.mode == MODE_STEP) {
      // ================= A1: who queries ORCA, float bodies (RVOPolicy.py:57-74)
      const uint32_t pol = (r.flags >> CA_POLICY_SHIFT) & 0xF;
      const bool query = active && !(r.flags & CA_DONE);  // env.py:311
      const bool rvo = query && pol == CA_POL_RVO;
      if (wave0) PRIO(3, 0, 3, 3, 3, 3, 3, 3); else PRIO(2, 0, 3, 1, 2, 2, 3, 2);
      if (wave0) {
        ep_step += 1;  // env.py:183
        // (an absent slot of a ragged batch sits at infinity: distSq = inf = "not a neighbour" for everybody)
        sh_fpx[lane] = (p.ragged && (r.flags & CA_ABSENT)) ? INFINITY : static_cast<float>(r.px);
        sh_fpy[lane] = static_cast<float>(r.py);
        sh_fvx[lane] = static_cast<float>(r.vx);
        sh_fvy[lane] = static_cast<float>(r.vy);
        sh_frad[lane] = static_cast<float>((1 + 5e-2) * r.rad);  // RVOPolicy.py:71
        // compact list of the tile's ORCA queries: in steady state ~40 % of the agents are done and wait for their env's
        // game over (EVALUATE_MODE), and pair items dealt over ALL agents would idle in 4 of 10 lanes of every wave
        const unsigned long long qm = __ballot(rvo);
        if (rvo) sh_q[__popcll(qm & ((1ull << lane) - 1ull))] = lane;
        if (lane == 0) {
          sh_q3[0] = 0;
          sh_q3[ROW + 1] = __popcll(qm);
        }
      }
      WG_SYNC();
      TICK(1);
      F2 v_orca = f2(0.f, 0.f);  // this lane's ORCA velocity (agent wave)
      PRIO(2, 0, 3, 1, 2, 2, 3, 2);
      if (!AB(1)) {
        // ================= P2: every (querying agent, other) pair: neighbour rank (ascending distSq, ties by index:
        // Agent::insertAgentNeighbor) + ORCA half-plane
        const float range_sq = sqf(static_cast<float>(p.sensing_horizon));
        const bool unlimited = !(range_sq < INFINITY);
        const float inv_h = divf(1.0f, static_cast<float>(p.rvo_time_horizon));
        const float inv_dt = divf(1.0f, static_cast<float>(p.rvo_dt));  // RVOPolicy.py:13,26
        const float collab = static_cast<float>(p.rvo_collab_coeff);
        const int n_live = sh_q3[ROW + 1];
        // A wave holds WHOLE agents -- floor(64 / (N - 1)) of them, lane = (agent, one of its N - 1 others) -- so a lane
        // gets the other distances of its agent from its neighbour lanes (ds_bpermute: no recomputation, no LDS round
        // trip) and counts its rank.  At N = 10: 7 agents per wave, 28 per round of the workgroup.
        constexpr int NW = NT / 64;
        const int wv = tid >> 6, wl = tid & 63;
        const int GN = NC ? (NC > 1 ? NC - 1 : 1) : (N > 1 ? N - 1 : 1);  // lanes per agent
        const float inv_gn = 1.0f / static_cast<float>(GN);
        const int apw = 64 / GN;     // agents per wave
        const int apr = NW * apw;    // agents per round of the workgroup
        const int g = div_small<(NC > 1 ? NC - 1 : NC)>(wl, inv_gn), jo = wl - __mul24(g, GN);
        const int gbase = (wl - jo) << 2;  // ds_bpermute byte address of the first lane of my agent's group
#pragma unroll
        for (int c0 = 0; c0 < (NC ? tile_n : n_live); c0 += apr) {
          if (c0 >= n_live) continue;  // workgroup-uniform
          const int c = c0 + wv * apw + g;
          const bool valid = (g < apw) && (c < n_live);
          const int ag = sh_q[valid ? c : 0];
          const int eb = __mul24(div_small<NC>(ag, inv_n), N), aa = ag - eb;
          const int j = jo + ((jo >= aa) ? 1 : 0);  // my other agent (index order is kept: ties by index)
          const F2 mpos = f2(sh_fpx[ag], sh_fpy[ag]);
          float dj = INFINITY;
          if (valid && j < N) {
            const F2 d = mpos - f2(sh_fpx[eb + j], sh_fpy[eb + j]);
            dj = dotf(d, d);
            if (!unlimited && !(dj < range_sq)) dj = INFINITY;
          }
          // rank = number of others that are closer (ties by index: they need two floats to be EQUAL -- symmetric starts --
          // and are settled in a second walk by the waves that hold one)
          int rank = 0, cnt = (N > 1) ? N - 1 : 0, tie = 0;
          for_n<8>(GN, [&](const int q) {
            const float dq = __int_as_float(__builtin_amdgcn_ds_bpermute(gbase + 4 * q, __float_as_int(dj)));
            rank += static_cast<int>(dq < dj);
            tie += static_cast<int>(dq == dj);
          });
          if (__any(tie > 1 && dj < INFINITY)) {  // (a distance always equals itself)
            for_n<8>(GN, [&](const int q) {
              const float dq = __int_as_float(__builtin_amdgcn_ds_bpermute(gbase + 4 * q, __float_as_int(dj)));
              rank += static_cast<int>(dq == dj) & static_cast<int>(q < jo);
            });
          }
          if (!unlimited || p.ragged) {  // a finite neighborDist / absent slots: count who is a neighbour
            cnt = 0;
            for_n<8>(GN, [&](const int q) {
              cnt += static_cast<int>(__int_as_float(__builtin_amdgcn_ds_bpermute(gbase + 4 * q, __float_as_int(dj))) < INFINITY);
            });
          }
          // (neighborDist = inf -- Config.SENSING_HORIZON, RVOPolicy.py:27 -- makes every other agent a neighbour)
          const int n = cnt < p.rvo_max_neighbors ? cnt : p.rvo_max_neighbors;
          if (valid) {
            if (jo == 0) sh_nb[ag] = n;
            if (dj < INFINITY && rank < n) {
              // (the ego's own collaboration coefficient where the caller drew one per agent: RVOPolicy.py:77-90)
              const float cf = k.s.rvo_collab ? k.s.rvo_collab[env0 * N + ag] : collab;
              Lmat[rank * CS + ag] = half_plane_sel(mpos, f2(sh_fvx[ag], sh_fvy[ag]), sh_frad[ag],
                                                     f2(sh_fpx[eb + j], sh_fpy[eb + j]), f2(sh_fvx[eb + j], sh_fvy[eb + j]),
                                                     sh_frad[eb + j], cf, inv_h, inv_dt);
            }
          }
        }
        if (wave0 && rvo) {  // the preferred velocity (float64 sqrt + divide), while the other waves finish the pairs
          const double vx = r.gx - r.px, vy = r.gy - r.py;
          const double sc = divd(r.ps, sqrtd(vx * vx + vy * vy));  // RVOPolicy.py:66-67
          sh_fprx[lane] = static_cast<float>(sc * vx);
          sh_fpry[lane] = static_cast<float>(sc * vy);
          sh_fms[lane] = static_cast<float>(r.ps);
        }
        WG_SYNC();
        TICK(2);
        if (N > G16 && !EXP(32)) {
          // ================= more than 16 agents per env: one WAVE per querying agent solves the whole programme
          // cooperatively (lane j = line j): linearProgram2 as "first violated line" ballots with the 1-D programme of that
          // line solved by the lanes holding the lines before it, then linearProgram3 if it ends infeasible
          // (cagpu_grouplp.inc; bit-identical to the sequential algorithm).  Solving EVERY line's 1-D programme in advance,
          // as below, is N^3 / 2 intersections per env -- at N = 50 that and its scan took 37 k of the step's 138 k cycles,
          // for 1.3 .. 3 programmes a query really needs.
          const int wq = tid >> 6, jl = tid & 63;
          for (int c = wq; c < n_live; c += NT / 64) {  // wave-uniform
            const int ag = sh_q[c];
            const int nf = sh_nb[ag];
            const float4 ln = Lmat[((jl < nf) ? jl : 0) * CS + ag];
            F2 v;
            const int f2_ = lp2_group<64>(jl < nf, f2(ln.x, ln.y), f2(ln.z, ln.w), sh_fms[ag], f2(sh_fprx[ag], sh_fpry[ag]), false,
                                          v, jl, jl);
            if (f2_ != NOFAIL) lp3_group<64>(nf, f2_, f2(ln.x, ln.y), f2(ln.z, ln.w), sh_fms[ag], v, jl, jl);
            if (jl == 0) { sh_vrx[ag] = v.x; sh_vry[ag] = v.y; }
          }
          WG_SYNC();
          TICK(12);
          if (wave0 && rvo) v_orca = f2(sh_vrx[lane], sh_vry[lane]);
          if (wave0) PRIO(3, 0, 3, 3, 3, 3, 3, 3); else PRIO(1, 0, 2, 0, 0, 0, 0, 1);
        } else {
        // ================= P2b: linearProgram1 of EVERY line i against the lines before it, one thread per (agent, i).
        // The 1-D optimum on line i depends on the lines [0, i), the speed disc and the preferred velocity only -- not
        // on the running result of linearProgram2 -- so all of them are computed side by side (min / max are exact and
        // a 1-D programme is infeasible iff its final interval is empty), and linearProgram2 itself shrinks to the scan
        // below: per line one violation test and one select.  Bit-identical to the incremental form.
        // Items are line-major (w = i * n_live + c): the lanes of a wave hold (nearly) the same line index i, so the
        // loop over the lines m < i runs to a wave-uniform bound instead of N - 2 for everybody.
        const int n_lp1 = n_live * (N - 1);
        const float inv_live = 1.0f / static_cast<float>(n_live > 0 ? n_live : 1);
                for (int base_ = 0; base_ < n_lp1; base_ += NT) {
          const int w = base_ + tid;
          const bool item = w < n_lp1;
          int i_hi, i, ag;
          if (!EXP(32)) {
            // highest lines first: a second round (more than NT / (N - 1) queries in the tile) holds the lines with the
            // FEWEST predecessors, i.e. the cheap ones
            const int w_lo = base_ + (tid & ~63);  // the wave's first item holds its highest line
            i_hi = __builtin_amdgcn_readfirstlane((N - 2) - static_cast<int>((static_cast<float>(w_lo) + 0.5f) * inv_live));
            const int wq = item ? static_cast<int>((static_cast<float>(w) + 0.5f) * inv_live) : 0;
            i = item ? (N - 2) - wq : 0;
            ag = sh_q[item ? w - wq * n_live : 0];
          } else {
            const int w_hi = (base_ + (tid | 63) < n_lp1) ? base_ + (tid | 63) : n_lp1 - 1;  // the wave's last item
            i_hi = __builtin_amdgcn_readfirstlane(static_cast<int>((static_cast<float>(w_hi) + 0.5f) * inv_live));
            i = item ? static_cast<int>((static_cast<float>(w) + 0.5f) * inv_live) : 0;
            ag = sh_q[item ? w - i * n_live : 0];
          }
          const bool live = item && i < sh_nb[ag];
          const float4 li = Lmat[i * CS + ag];
          const F2 Pi = f2(li.x, li.y), Di = f2(li.z, li.w);
          const float radius = sh_fms[ag];
          const F2 opt = f2(sh_fprx[ag], sh_fpry[ag]);
          const float dp = dotf(Pi, Di);
          const float disc = sqf(dp) + sqf(radius) - dotf(Pi, Pi);
          bool ok = !(disc < 0.0f);
          const float sd = sqrtq(ok ? disc : 0.0f);
          float t_lo = -dp - sd;
          float t_hi = -dp + sd;
          auto against = [&](const int m) {  // intersect line i with line m (< i)
            const bool use = m < i;
            const float4 lm = Lmat[(use ? m : 0) * CS + ag];
            const F2 Pm = f2(lm.x, lm.y), Dm = f2(lm.z, lm.w);
            const float den = detf(Di, Dm);
            const float num = detf(Dm, Pi - Pm);
            const bool par = fabsf(den) <= kRvoEps;
            ok = ok && !(use && par && num < 0.0f);
            const float tt = divq(num, par ? 1.0f : den);
            const float c_hi = (use && !par && den >= 0.0f) ? tt : INFINITY;
            const float c_lo = (use && !par && den < 0.0f) ? tt : -INFINITY;
            t_hi = (c_hi < t_hi) ? c_hi : t_hi;
            t_lo = (t_lo < c_lo) ? c_lo : t_lo;
          };
          if (base_ + (tid & ~63) < n_lp1) {  // wave-uniform: this wave has items in this round
            // (run-time trip count on purpose: with all N - 2 iterations unrolled and every load hoisted the waves that
            // hold low lines lose more than the others gain: 18.9 -> 19.3 us)
LP1_UNROLL
            for (int m = 0; m < i_hi; ++m) against(m);
          }
          ok = ok && !(t_lo > t_hi);
          const float t = dotf(Di, opt - Pi);
          const float tc = (t < t_lo) ? t_lo : ((t > t_hi) ? t_hi : t);
          const F2 res = Pi + tc * Di;
          if (live) {
            Rmat[i * CS + ag] = make_float2(res.x, res.y);
            okmat[i * CS + ag] = ok ? 1 : 0;
          }
        }
        WG_SYNC();
        TICK(12);
        // ================= linearProgram2 = a scan over the lines, one lane per agent (agent wave); infeasible
        // programmes (4.6 % of the queries at N = 10) are queued for the cooperative linearProgram3 pass
        int failf = NOFAIL;
        if (wave0) PRIO(3, 0, 3, 3, 3, 3, 3, 3);
        if (wave0 && rvo) {
          const int n = sh_nb[lane];
          const float radius = sh_fms[lane];
          const F2 opt = f2(sh_fprx[lane], sh_fpry[lane]);
          F2 res = opt;
          if (dotf(opt, opt) > sqf(radius)) res = radius * unitq(opt);
          for_n<8>(NC ? NC - 1 : n, [&](const int i) {
            // The loads do not depend on the running result, so all of them can be in flight before the select chain
            // starts -- provided the tests stay branch-free: the conditions are combined as integers (written with &&
            // the compiler guards every load with its own branch + wait, one LDS round trip per line).
            const float4 li = Lmat[i * CS + lane];
            const float2 ri = Rmat[i * CS + lane];
            const int oki = okmat[i * CS + lane];
            const float dt = detf(f2(li.z, li.w), f2(li.x, li.y) - res);
            const int hit = static_cast<int>(i < n) & static_cast<int>(failf == NOFAIL) & static_cast<int>(dt > 0.0f);
            const int take = hit & (oki != 0 ? 1 : 0);
            res.x = take ? ri.x : res.x;
            res.y = take ? ri.y : res.y;
            failf = (hit & (take ^ 1)) ? i : failf;
          });
          v_orca = res;
          if (failf != NOFAIL) {
            sh_vrx[lane] = res.x;
            sh_vry[lane] = res.y;
            sh_q3[1 + atomicAdd(&sh_q3[0], 1)] = lane | (failf << 8) | (n << 16);
          }
        }
        WG_SYNC();
        TICK(15);
        // ================= linearProgram3 for the queued agents.  N <= 10 (at most 9 lines): one WAVE per agent, all pair
        // intersections of the embedded linearProgram2 in one step (lp3_wave8); otherwise one 16-lane group per agent
        // while N <= 16 (lane j = half-plane j; ballot + DPP row reductions), the whole wave beyond (cagpu_grouplp.inc)
        const int n3 = sh_q3[0];
#ifdef CAGPU_WGTIME
        if (tid == 0) {
          wg_info = static_cast<unsigned long long>(n3) | (static_cast<unsigned long long>(sh_q3[ROW + 1]) << 8);
          wg_lp3 = wall_clock64();
        }
#endif
        if (n3 > 0 && !AB(2) && !EXP(8)) {  // workgroup-uniform (EXP(8): experiment, results invalid)
          PRIO(3, 3, 3, 2, 3, 3, 3, 3);
          auto solve3 = [&](auto gs_tag) {
            constexpr int GS = decltype(gs_tag)::value;
            constexpr int GROUPS = NT / GS;
            // queue entry k goes to wave k % (number of waves) first: infeasible agents are solved side by side
            const int jl = tid & (GS - 1), g = tid / GS;
            constexpr int GPW = 64 / GS;  // groups per wave
            for (int q3 = (g % GPW) * (NT / 64) + (g / GPW); q3 < n3; q3 += GROUPS) {
              const int ent = sh_q3[1 + q3], agf = ent & 0xFF, ff = (ent >> 8) & 0xFF;
              const int nf = sh_nb[agf];
              const float4 ln = Lmat[((jl < nf) ? jl : 0) * CS + agf];
              F2 v = f2(sh_vrx[agf], sh_vry[agf]);
              lp3_group<GS>(nf, ff, f2(ln.x, ln.y), f2(ln.z, ln.w), sh_fms[agf], v, jl, tid & 63);
              if (jl == 0) { sh_vrx[agf] = v.x; sh_vry[agf] = v.y; }
            }
          };
          if (N <= 10) {
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
            for (int q3 = wv; q3 < n3; q3 += NT / 64) {
              const int ent = sh_q3[1 + q3], agf = ent & 0xFF, ff = (ent >> 8) & 0xFF;  // (bits 16 ..: the line count)
              F2 v = f2(sh_vrx[agf], sh_vry[agf]);
#ifdef CAGPU_WGTIME
              int it3 = 0;
              TICK(13);  // (slot 13: barrier + queue / entry loads; slot 14: lp3_wave8 itself, wave 0's entries only)
              lp3_wave8(Lmat + agf, CS, (ent >> 16) & 0xFF, ff, sh_fms[agf], v, tid & 63, &it3);
              TICK(14);
              if (tid == 0) wg_info += static_cast<unsigned long long>(it3) << 24;
#else
              lp3_wave8(Lmat + agf, CS, (ent >> 16) & 0xFF, ff, sh_fms[agf], v, tid & 63);
#endif
              if ((tid & 63) == 0) { sh_vrx[agf] = v.x; sh_vry[agf] = v.y; }
            }
          } else if (N <= G16) {
            solve3(std::integral_constant<int, 16>{});
          } else {
            solve3(std::integral_constant<int, 64>{});
          }
          WG_SYNC();
          if (failf != NOFAIL) v_orca = f2(sh_vrx[lane], sh_vry[lane]);
          if (wave0) PRIO(3, 0, 3, 3, 3, 3, 3, 3); else PRIO(1, 0, 2, 0, 0, 0, 0, 1);
        }
        }
      }
#ifdef CAGPU_WGTIME
      if (tid == 0) wg_lp3 = wall_clock64() - wg_lp3;  // the linearProgram3 pass (incl. its barrier), 10 ns ticks
#endif
      TICK(3);
      // ================= A2c: policy post-processing (env.py:305-323) and move (agent.py:192-241), one lane per agent
      TICK(0);
      if (wave0) {
        double spd = 0.0, dh = 0.0;
        if (query) {
          if (pol == CA_POL_RVO) {
            const float ts = static_cast<float>(p.rvo_dt);
            const F2 v = v_orca;
            // Agent::update: float position += v * timeStep; RVOPolicy.py:96-111
            const float npx = sh_fpx[lane] + v.x * ts, npy = sh_fpy[lane] + v.y * ts;
            const double dpx = static_cast<double>(npx) - r.px, dpy = static_cast<double>(npy) - r.py;
            const double ang = AB(4) ? dpy : atan2(dpy, dpx);
            TICK(13);
            const double nh = (ang < 0.0) ? ang + kTwoPi : ((ang == 0.0) ? 0.0 : ang);  // `% (2*pi)`, :102
            dh = wrap_pi(nh - r.heading);
            TICK(14);
            spd = k.inv_rvo_dt * sqrtd(dpx * dpx + dpy * dpy);  // RVOPolicy.py:106: 1/self.dt * norm
            if (fabs(dh) > kPi / 6) {
              dh = ((dh > 0.0) - (dh < 0.0)) * (kPi / 6);
              spd = 0.0;
            }
            if (k.s.rvo_heading_noise) dh = dh + k.s.rvo_heading_noise[i];  // RVOPolicy.py:118-119 (drawn by the caller)
          } else if (pol == CA_POL_NONCOOP) {  // NonCooperativePolicy.py:21
            const Ego eg = ego_frame(r.px, r.py, r.gx, r.gy, r.heading);
            spd = r.ps;
            dh = -eg.heading_ego;
          } else if (pol == CA_POL_STATIC) {  // StaticPolicy.py:21-23
            r.gx = r.px;
            r.gy = r.py;
            statics_dirty = true;
          } else if (k.ext) {
            const double e0 = k.ext[2 * i], e1 = k.ext[2 * i + 1];
            if (pol == CA_POL_EXTERNAL) {  // ExternalPolicy.py:14-16
              spd = e0;
              dh = e1;
            } else if (pol == CA_POL_LEARNING) {  // LearningPolicy.py:29-33
              dh = p.max_heading_change * (2. * e1 - 1.);
              spd = r.ps * e0;
            } else if (pol == CA_POL_LEARNING_GA3C || pol == CA_POL_GA3C_CADRL) {  // LearningPolicyGA3C.py:24-26,
              // GA3CCADRLPolicy.py:81-84 (index from cagpu_ga3c), network.py:7-16
              int q = static_cast<int>(e0);
              q = q < 0 ? 0 : (q > 10 ? 10 : q);
              const int hq = (q < 5) ? q - 2 : ((q - 5) % 3 - 1) * 2;  // heading index in units of pi/12
              const double s0 = (q < 5) ? 1.0 : ((q < 8) ? 0.5 : 0.0);
              spd = r.ps * s0;
              dh = (hq == -2) ? -kPi / 6 : (hq == -1) ? -kPi / 12 : (hq == 0) ? 0.0 : (hq == 1) ? kPi / 12 : kPi / 6;
            }
          }
        }
        TICK(4);
        const float a0f = static_cast<float>(spd), a1f = static_cast<float>(dh);  // float32 `all_actions`
        if (active && k.o.actions) reinterpret_cast<float2*>(k.o.actions)[i + ring_a] = make_float2(a0f, a1f);
        if (active && k.o.orca_vel)  // parity hook: the velocity rvo2 chose for this agent (RVOPolicy.py:93), 0 if not queried
          reinterpret_cast<float2*>(k.o.orca_vel)[i + ring_a] = rvo ? make_float2(v_orca.x, v_orca.y) : make_float2(0.f, 0.f);
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
              if (AB(8)) { sn = nh; cs = 1.0 - nh; } else sincos_heading(nh, sn, cs);
              r.px += a0 * cs * p.dt;
              r.py += a0 * sn * p.dt;
              r.vx = a0 * cs;
              r.vy = a0 * sn;
              r.heading = nh;
              if (dyn == CA_DYN_UNICYCLE) r.td = turning_dir_next(r.td, nh);
            } else if (k.s.ext_state) {  // a host-side Dynamics subclass integrated this agent (agent.py:214-220)
              const double* q = k.s.ext_state + 5 * i;
              const double npx = q[0], npy = q[1], nvx = q[2], nvy = q[3], nh = q[4];
              if (!(npx != npx || npy != npy || nvx != nvx || nvy != nvy || nh != nh)) {
                r.px = npx; r.py = npy; r.vx = nvx; r.vy = nvy; r.heading = nh;
              }
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
      }
      do_sense = active;
      TICK(5);
    }

    // ================= sensing passes: once, and a second time for envs that auto-reset in this step
    auto sense_pass = [&](const int pass) -> int {
      // ---- A: publish the post-move tile + ego frames (agent.py:329-349, Dynamics.py:24-41)
      Ego eg;
      eg.dist = 0.0; eg.prx = eg.pry = eg.orx = eg.ory = eg.heading_ego = 0.0;
      if (wave0) {
        sh_px[lane] = r.px; sh_py[lane] = r.py; sh_vx[lane] = r.vx; sh_vy[lane] = r.vy; sh_rad[lane] = r.rad;
        if (do_sense) {
          if (AB(16)) { eg.dist = r.px; eg.prx = 1.0; eg.pry = 0.0; eg.heading_ego = r.heading; } else eg = ego_frame(r.px, r.py, r.gx, r.gy, r.heading);
          sh_prx[lane] = eg.prx;
          sh_pry[lane] = eg.pry;
        }
        sh_sense[lane] = do_sense ? 1 : 0;
        if (RO) sh_q[lane] = 0;  // reused below: case index + 1 of an env that auto-resets in this step
      }
      WG_SYNC();
      TICK(6);

      // ---- P3: every (agent, other) pair: centre distance -> collision gap, sensor key, p_orth
      //      (env.py:458-512; OtherAgentsStatesSensor.py:76-107)
      PRIO(1, 0, 2, 1, 1, 2, 1, 1);
      if (NC != 0 && p.sort_mode != CA_SORT_TIME_TO_IMPACT) {
        // N compiled in: one item per UNORDERED pair {a, b} -- the float64 square root, the collision gap and the
        // horizon test are the same for both directions; only d - r_host - r_other (operand order) and p_orth (the
        // host's frame) are direction-specific.  Per env: the pairs (a, a + s mod N) for s = 1 .. (N-1)/2, for even N
        // the N/2 antipodal pairs, then N self entries: N (N + 1) / 2 items, 220 per 4-env tile = ONE round of 256.
        constexpr int NN = NC ? NC : 2, HALF = (NN - 1) / 2, PE = NN * (NN + 1) / 2;
        const int n_un = tile_envs * PE;
        DUP(2048)
#pragma unroll
        FOR_ITEMS_UPTO(w, n_un) {
          if (AB(32)) continue;
          const int le2 = div_small<PE>(w, 1.0f / static_cast<float>(PE));
          const int u = w - __mul24(le2, PE), eb = __mul24(le2, NN);
          if (!sh_sense[eb]) continue;  // sensing is decided per env
          if (u >= PE - NN) {  // self entry: never sensed, no gap
            const int a1 = u - (PE - NN);
            kmat[a1 * CS + eb + a1] = KEY_NONE;
            omat[a1 * CS + eb + a1] = 0.0;
            d2mat[a1 * CS + eb + a1] = 0.f;
            gmat[a1 * CS + eb + a1] = INFINITY;
            continue;
          }
          int a1, b1;
          if (u < NN * HALF) {
            const int sft = div_small<NN>(u, 1.0f / static_cast<float>(NN));
            a1 = u - __mul24(sft, NN);
            b1 = a1 + sft + 1;
            b1 -= (b1 >= NN) ? NN : 0;
          } else {
            a1 = u - NN * HALF;
            b1 = a1 + NN / 2;
          }
          const int ga = eb + a1, gb = eb + b1;
          const double ax = sh_px[ga], ay = sh_py[ga], ar = sh_rad[ga];
          const double bx = sh_px[gb], by = sh_py[gb], br = sh_rad[gb];
          const double rx = bx - ax, ry = by - ay;  // other - host for host a; exactly negated for host b
          const double d = sqrtd(rx * rx + ry * ry);
          int key_ab = KEY_NONE, key_ba = KEY_NONE;
          double po_ab = 0.0, po_ba = 0.0, d2_ab = 0.0, d2_ba = 0.0;
          const bool both = !p.ragged || (ar > 0.0 && br > 0.0);  // (an absent slot of a ragged batch has radius 0)
          if (both && !(d > p.sensing_horizon)) {
            d2_ab = d - ar - br;
            d2_ba = d - br - ar;
            // numpy scalar round(x, 2) bucket; ordering of k == ordering of k/100 (an integer: exact as int32)
            key_ab = static_cast<int>(fmin(fmax(rint(d2_ab * 100.0), -2.0e9), 2.0e9));
            key_ba = static_cast<int>(fmin(fmax(rint(d2_ba * 100.0), -2.0e9), 2.0e9));
            po_ab = rx * (-sh_pry[ga]) + ry * sh_prx[ga];
            po_ba = (-rx) * (-sh_pry[gb]) + (-ry) * sh_prx[gb];
          }
          kmat[b1 * CS + ga] = key_ab;
          omat[b1 * CS + ga] = po_ab;
          d2mat[b1 * CS + ga] = static_cast<float>(d2_ab);
          gmat[b1 * CS + ga] = both ? d - (ar + br) : INFINITY;
          kmat[a1 * CS + gb] = key_ba;
          omat[a1 * CS + gb] = po_ba;
          d2mat[a1 * CS + gb] = static_cast<float>(d2_ba);
          gmat[a1 * CS + gb] = both ? d - (br + ar) : INFINITY;
        }
      } else {
      DUP(2048)
#pragma unroll
      FOR_PAIR_ITEMS(w) {
        if (AB(32)) continue;
        const int ag = div_small<NC>(w, inv_n);
        const int j = w - __mul24(ag, N);
        if (!sh_sense[ag]) continue;
        const int eb = __mul24(div_small<NC>(ag, inv_n), N), aa = ag - eb;
        int key = KEY_NONE;
        double po = 0.0, d2o = 0.0, gap = INFINITY;
        if (j != aa) {
          const double hx = sh_px[ag], hy = sh_py[ag], hr = sh_rad[ag];
          const double ox = sh_px[eb + j], oy = sh_py[eb + j], orad = sh_rad[eb + j];
          const double rx = ox - hx, ry = oy - hy;
          const double d = sqrtd(rx * rx + ry * ry);
          const bool both = !p.ragged || (hr > 0.0 && orad > 0.0);  // (an absent slot of a ragged batch has radius 0)
          if (both) gap = d - (hr + orad);
          if (both && !(d > p.sensing_horizon)) {
            d2o = d - hr - orad;
            // numpy scalar round(x, 2) bucket; ordering of k == ordering of k/100 (an integer: exact as int32)
            key = static_cast<int>(fmin(fmax(rint(d2o * 100.0), -2.0e9), 2.0e9));
            po = rx * (-sh_pry[ag]) + ry * sh_prx[ag];
            if (p.sort_mode == CA_SORT_TIME_TO_IMPACT)  // sensor :96-104
              tmat[j * CS + ag] = time_to_impact(hx, hy, ox, oy, sh_vx[ag], sh_vy[ag], sh_vx[eb + j], sh_vy[eb + j],
                                                  hr + orad);
          }
        }
        kmat[j * CS + ag] = key;
        omat[j * CS + ag] = po;
        d2mat[j * CS + ag] = static_cast<float>(d2o);
        gmat[j * CS + ag] = gap;
      }
      }
      WG_SYNC();

      TICK(7);
      // ---- P4: rank the candidates of every agent and emit its rows (OtherAgentsStatesSensor.py:20-55,109-143)
      PRIO(1, 0, 1, 1, 0, 1, 0, 1);
      DUP(4096)
#pragma unroll
      FOR_PAIR_ITEMS(w) {
        if (AB(64)) continue;
        const int ag = div_small<NC>(w, inv_n);
        const int j = w - __mul24(ag, N);
        if (!sh_sense[ag]) continue;
        const int eb = __mul24(div_small<NC>(ag, inv_n), N), aa = ag - eb;
        const int kj = kmat[j * CS + ag];
        const double oj = omat[j * CS + ag];
        int rank = 0, cnt = 0, lt = 0, same = 0;  // lt / same: candidates in closer buckets / in this one (itself included)
        if (p.sort_mode == CA_SORT_TIME_TO_IMPACT) {  // key (-tti, -dist, p_orth), sensor :36-38
          const bool vj = kj != KEY_NONE;
          const double tj = vj ? tmat[j * CS + ag] : 0.0;
          for_n<8>(N, [&](const int q) {
            const int kq = kmat[q * CS + ag];
            const double oq = omat[q * CS + ag];
            const bool vq = kq != KEY_NONE;
            const double tq = vq ? tmat[q * CS + ag] : 0.0;
            const int lo = static_cast<int>(oq < oj) | (static_cast<int>(oq == oj) & static_cast<int>(q < j));
            const int lk = static_cast<int>(kq > kj) | (static_cast<int>(kq == kj) & lo);
            const int before = static_cast<int>(tq > tj) | (static_cast<int>(tq == tj) & lk);
            rank += static_cast<int>(vq) & static_cast<int>(vj) & before;
            cnt += static_cast<int>(vq);
          });
        } else {
          // Pass 1 ranks by the distance bucket alone (one 4-byte key per candidate, branch-free) and notes WHICH candidates
          // share this one's 1 cm bucket.  The (p_orth, index) tie-break -- an 8-byte load and two float64 compares -- then
          // visits only those: a lane pops the set bits of its own mask, the wave loops while any lane has one left (one or
          // two trips).  Walking all N candidates again whenever some lane of the wave held a tie -- in a dense crowd
          // always -- was the larger half of this phase at N = 50.
          // (the mask is gathered eight candidates at a time as m8 = 2 m8 + (kq == kj): a compare and an add-with-carry per
          // candidate like the rank itself, bit-reversed and shifted into place once per block -- a 64-bit shift by a
          // run-time q per candidate doubled the cost of this loop at N = 50)
          unsigned long long eq = 0ull;
          {
            int q0 = 0;
            for (; q0 + 8 <= N; q0 += 8) {
              unsigned m8 = 0u;
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int kq = kmat[(q0 + u) * CS + ag];
                rank += static_cast<int>(kq < kj);
                m8 = m8 + m8 + static_cast<unsigned>(kq == kj);
              }
              eq |= static_cast<unsigned long long>(__brev(m8) >> 24) << q0;  // (candidate q0 + u sits at bit 7 - u of m8)
            }
#pragma unroll
            for (int u = 0; u < 7; ++u)
              if (q0 + u < N) {
                const int kq = kmat[(q0 + u) * CS + ag];
                rank += static_cast<int>(kq < kj);
                eq |= (kq == kj) ? (1ull << (q0 + u)) : 0ull;
              }
          }
          same = __popcll(eq);  // (a key always equals itself)
          lt = rank;
          eq &= ~(1ull << j);
          if (kj == KEY_NONE) eq = 0ull;  // (not a candidate: its rank is never used)
          while (__any(eq != 0ull)) {
            if (eq != 0ull) {
              const int q = __ffsll(static_cast<long long>(eq)) - 1;
              eq &= eq - 1ull;
              const double oq = omat[q * CS + ag];
              rank += static_cast<int>(oq < oj) | (static_cast<int>(oq == oj) & static_cast<int>(q < j));
            }
          }
          if (p.sensing_horizon < INFINITY || p.ragged) {
            for_n<8>(N, [&](const int q) { cnt += static_cast<int>(kmat[q * CS + ag] != KEY_NONE); });
          } else {
            cnt = N - 1;  // every other agent of the env is sensed
          }
        }
        const int keep = cnt < p.obs_clip ? cnt : p.obs_clip;  // sensor :39
        float* row = (STAGE ? sh_obs : obs_tile) + __mul24(ag, W);
        if (j == aa) row[1] = static_cast<float>(keep);  // num_other_agents_observed
        for (int sl = j; sl < K; sl += N)                // zero the unfilled rows (sensor :112)
          if (sl >= keep) {
            float* z = row + 6 + 7 * sl;
            z[0] = z[1] = z[2] = z[3] = z[4] = z[5] = z[6] = 0.f;
          }
        const bool kept = (j != aa) && (kj != KEY_NONE) && (rank < keep);
        if (!kept) continue;
        // closest_last re-sorts the KEPT ones by (-key, p_orth), stable (sensor :41-43): the buckets in reverse, the order
        // inside a bucket as in the first pass.  No second ranking pass is needed for that: the kept ones of farther buckets
        // are keep - (closer ones, all kept) - (this bucket's kept ones = min(same, keep - lt)), and the place inside the
        // bucket is rank - lt.
        int slot = rank;
        if (p.sort_mode == CA_SORT_CLOSEST_LAST) {
          const int room = keep - lt;
          slot = room - (same < room ? same : room) + (rank - lt);
        }
        const double hx = sh_px[ag], hy = sh_py[ag], hr = sh_rad[ag], prx = sh_prx[ag], pry = sh_pry[ag];
        const double ox = sh_px[eb + j], oy = sh_py[eb + j], orad = sh_rad[eb + j];
        const double ovx = sh_vx[eb + j], ovy = sh_vy[eb + j];
        const double rx = ox - hx, ry = oy - hy;
        float* o7 = row + 6 + 7 * slot;
        o7[0] = static_cast<float>(rx * prx + ry * pry);
        o7[1] = static_cast<float>(rx * (-pry) + ry * prx);
        o7[2] = static_cast<float>(ovx * prx + ovy * pry);
        o7[3] = static_cast<float>(ovx * (-pry) + ovy * prx);
        o7[4] = static_cast<float>(orad);
        o7[5] = static_cast<float>(hr + orad);
        o7[6] = d2mat[j * CS + ag];
      }
      // ---- A3 (wave 0, while the other waves finish the pair items of P4): rewards + collision flag (env.py:394-456),
      // observation scalars
      if (wave0) PRIO(3, 0, 3, 3, 3, 3, 3, 3);
      if (wave0 && active) {
        if (k.mode == MODE_STEP && pass == 0) {
          double nearest = INFINITY;
          for_n<8>(N, [&](const int j) {
            const double g = gmat[j * CS + lane];
            nearest = (g < nearest) ? g : nearest;
          });
          const bool coll = nearest <= 0.0;  // some d <= r_i + r_j  <=>  min(d - (r_i + r_j)) <= 0
          double rw = p.reward_time_step;
          if (r.flags & CA_AT_GOAL) {
            if (!(r.flags & CA_WAS_AT_GOAL)) rw = p.reward_at_goal;
          } else if (!(r.flags & CA_WAS_IN_COLLISION)) {
            if (coll) {
              rw = p.reward_collision;
              r.flags |= CA_IN_COLLISION;
            } else if (hits_wall(k.map, r.px, r.py, r.rad)) {  // env.py:425-429, :494-506
              rw = p.reward_collision_wall;
              r.flags |= CA_IN_COLLISION;
            } else {
              if (nearest <= p.getting_close_range) rw = -0.1 - nearest / 2.0;
              if (fabs(static_cast<double>(r.act1)) > p.wiggly_threshold) rw += p.reward_wiggly;
            }
          }
          rw = fmin(fmax(rw, p.reward_min), p.reward_max);
          if (p.ragged && (r.flags & CA_ABSENT)) rw = 0.0;  // no agent in this slot: the zero padding of wrappers.py:143-173
          r.epr += rw;
          reward = static_cast<float>(rw);
          const bool d = (r.flags & (CA_AT_GOAL | CA_OUT_OF_TIME | CA_IN_COLLISION)) != 0;
          if (d) r.flags |= CA_DONE; else r.flags &= ~static_cast<uint32_t>(CA_DONE);
          sh_flag[lane] = r.flags;
          sh_r0[lane] = r.epr;
          sh_r1[lane] = r.t;
          sh_r2[lane] = r.t - r.slt;
        }
        if (do_sense) {
          float* row = (STAGE ? sh_obs : obs_tile) + __mul24(lane, W);
          const bool here = !(p.ragged && (r.flags & CA_ABSENT));
          row[0] = (here && (r.flags & CA_IS_LEARNING)) ? 1.f : 0.f;
          row[2] = here ? static_cast<float>(eg.dist) : 0.f;
          row[3] = here ? static_cast<float>(eg.heading_ego) : 0.f;
          row[4] = here ? static_cast<float>(r.ps) : 0.f;
          row[5] = static_cast<float>(r.rad);
        }
      }

      TICK(8);
      // No workgroup barrier here: A4 reads what A3 wrote (flags, episode scratch), and both run on the agent wave --
      // a wave-level fence is enough, and A4 overlaps with the other waves' second round of P4 instead of following it.
      if (wave0) wave_sync();

      TICK(9);
      // ---- A4 (wave 0): done / game over (env.py:514-553), auto-reset (vec_env.py:120-128), episode statistics
      bool need_second = false;
      if (k.mode == MODE_STEP && pass == 0) {
        do_sense = false;
        if (wave0 && active) {
          // AND / OR of the env's flag words, then bit tests: no short-circuit chains (they compile to one dependent
          // LDS round trip + branch per agent)
          uint32_t f_and = ~0u, f_or = 0u, learn_and = ~0u;
          for_n<8>(N, [&](const int j) {
            const uint32_t f = sh_flag[ebase + j];
            f_and &= f;
            f_or |= f;
            learn_and &= (f & CA_STILL_LEARNING) ? f : ~0u;  // learners only
          });
          const bool all_done = (f_and & CA_DONE) != 0, all_learning_done = (learn_and & CA_DONE) != 0;
          const bool any_coll = (f_or & CA_IN_COLLISION) != 0, all_goal = (f_and & CA_AT_GOAL) != 0;
          bool over = all_done;
          if (p.game_over_mode == CA_OVER_AGENT0) over = (sh_flag[ebase] & CA_DONE) != 0;
          else if (p.game_over_mode == CA_OVER_LEARNING_DONE) over = all_learning_done;
          k.o.rewards[i + ring_a] = reward;
          k.o.done[i + ring_a] = static_cast<uint8_t>((r.flags & CA_DONE) != 0);
          if (a == 0) k.o.game_over[e + ring_e] = static_cast<uint8_t>(over);
          if (over && k.table) {
            if (a == 0) {  // experiments/src/env_utils.py:56-87 reduced to counters, summed in agent order
              double tot_r = 0.0, ttg = 0.0, extra = 0.0;
              for_n<8>(N, [&](const int j) {  // (summed in agent order)
                tot_r += sh_r0[ebase + j];
                ttg += sh_r1[ebase + j];
                extra += sh_r2[ebase + j];
              });
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
            double h0 = 0.0;
            if (k.heading_seed) {  // test_cases.py:558-559 (training mode): uniform in [-pi, pi)
              const unsigned long long ge = static_cast<unsigned long long>(k.env_id_offset + e);
              h0 = -kPi + kTwoPi * gen::uniform_at(k.heading_seed, static_cast<unsigned>(ge), static_cast<unsigned>(ge >> 32),
                                                   static_cast<unsigned>(reset_cnt), static_cast<unsigned>(a));
            }
            reset_lane(r, k.table + (c * N + a) * 6, k.heading_seed != 0, h0, p);
            ep_step = 0;
            statics_dirty = true;
            if (RO) {
              if (a == 0) sh_q[lane] = static_cast<int>(c) + 1;  // the env's new case, at its first agent's slot
            } else {
              do_sense = true;
            }
            need_second = true;
          }
        }
      }
      const int again = __syncthreads_or(need_second ? 1 : 0);
#ifdef CAGPU_WGTIME
      if (tid == 0 && again) wg_info |= 1ull << 16;
#endif
      TICK(10);
      if (RO && again) {
        __syncthreads();  // (full fence: the copy below overwrites observation rows other threads stored in P4)
        // ---- RO: the observation of a freshly reset env is a pure function of its fixture case: it was computed once
        // (cagpu_reset on the whole table) and is copied here, instead of a second sensing pass for the tile
        for (int le2 = 0; le2 < tile_envs; ++le2) {  // workgroup-uniform: at most a few envs of a tile reset per step
          const int c1 = sh_q[le2 * N];
          if (!c1) continue;
          const float* src = k.reset_obs + static_cast<long>(c1 - 1) * N * W;
          const int base = le2 * N * W;
          for (int q = tid; q < N * W; q += NT) {
            const int a2 = q / W, col = q - a2 * W;
            float v = src[q];
            if (col == 0)  // (is_learning comes from the live flags; the row of an absent slot is all zeros: radius 0)
              v = ((sh_flag[le2 * N + a2] & CA_IS_LEARNING) && !(p.ragged && !(src[q + 5] > 0.f))) ? 1.f : 0.f;
            if (STAGE) sh_obs[base + q] = v;
            else obs_tile[base + q] = v;
          }
        }
        WG_SYNC();
      }
      if (RO || !again) {
        // ---- the tile's observation block leaves LDS as one contiguous, coalesced copy
        if (STAGE && !AB(128)) {
          const long total = tile_cnt * W;
          float* dst = obs_tile;
          if (k.mode == MODE_RESET && k.reset_mask) {
            for (long q = tid; q < total; q += NT) {
              const long ee = env0 + (q / W) / N;
              if (k.reset_mask[ee]) dst[q] = sh_obs[q];
            }
          } else if (((tile_base * W) & 3) == 0 && (total & 3) == 0) {
            const float4* src4 = reinterpret_cast<const float4*>(sh_obs);
            float4* dst4 = reinterpret_cast<float4*>(dst);
            for (long q = tid; q < (total >> 2); q += NT) dst4[q] = src4[q];
          } else {
            for (long q = tid; q < total; q += NT) dst[q] = sh_obs[q];
          }
        }
        return 0;
      }
      return 1;  // some env of the tile auto-reset: run the sensing pass once more for it
    };
    if (RO) {
      sense_pass(0);
    } else if (sense_pass(0)) {
      sense_pass(1);
    }
    if (MULTI || EXP(4)) WG_SYNC();  // the union is free again before the next step's ORCA view (n-step kernel only)
    TICK(11);
  };
  if (MULTI) {
    const int n_steps = (k.mode == MODE_STEP) ? k.n_steps : 1;
    for (int step = 0; step < n_steps; ++step) {
      one_step();
      obs_tile += k.ring_obs;  // (0 unless the caller keeps every step's outputs: cagpu_rollout_ring)
      ring_a += k.ring_agent;
      ring_e += k.ring_env;
    }
  } else {
    one_step();
  }

#ifdef CAGPU_ABLATE
  __syncthreads();
  if (tid < 16) atomicAdd(&g_prof[tid], sh_prof[tid]);
  if (tid < 16 && blockIdx.x < 1024) g_wgprof[blockIdx.x * 16 + tid] = sh_prof[tid];
#endif
#ifdef CAGPU_WGTIME
  unsigned long long wg_t1 = 0;
  if (tid == 0) wg_t1 = wall_clock64();
#endif
  // ---- store my agent.  The pointers are re-read from the kernarg segment here (laundered so the compiler does
  // not keep 19 pointer pairs alive in SGPRs across the whole kernel).
  if (active && k.mode != MODE_OBSERVE && (k.mode == MODE_STEP || !k.reset_mask || k.reset_mask[e])) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const KArgs ConstKArgs;
    ConstKArgs* ka = (ConstKArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
#else
    const KArgs* ka = &k;
#endif
    ka->s.pos_x[i] = r.px; ka->s.pos_y[i] = r.py; ka->s.vel_x[i] = r.vx; ka->s.vel_y[i] = r.vy;
    ka->s.heading[i] = r.heading; ka->s.time_remaining[i] = r.tr; ka->s.t[i] = r.t;
    ka->s.ep_reward[i] = r.epr;
    if (ka->s.turning_dir) ka->s.turning_dir[i] = r.td;
    if (statics_dirty) {
      ka->s.goal_x[i] = r.gx; ka->s.goal_y[i] = r.gy;
      ka->s.radius[i] = r.rad; ka->s.pref_speed[i] = r.ps; ka->s.slt[i] = r.slt;
    }
    reinterpret_cast<float2*>(ka->s.last_action)[i] = make_float2(r.act0, r.act1);
    ka->s.flags[i] = r.flags & ~static_cast<uint32_t>(CA_PLAN_VALID);  // (this kernel leaves no plan for the next step)
    ka->s.step_num[i] = r.step_num;
    if (a == 0) { ka->s.episode_step[e] = ep_step; ka->s.reset_count[e] = reset_cnt; }
  }
  // (Round 6, measured and dropped: the list of the agents the NEXT network query evaluates, packed here by every tile -- one
  //  epoch-tagged device-scope atomic per tile for its range + one arrival ticket -- so that cagpu_ga3c could skip its packing
  //  launch (6.9 us of a 143 us config-3 step).  1 366 tiles x 2 same-address atomics cost ~43 ns each, serialised: the step
  //  kernel went 34 -> 129 us (config 3: 146 -> 241 us per step, same box).  compact_kernel does the same with 80 workgroups.)
#ifdef CAGPU_WGTIME
  if (tid == 0 && blockIdx.x < 4096) {
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = g_wgtime + blockIdx.x * 8;
    o[0] = wg_t0; o[1] = wg_t1; o[2] = wall_clock64(); o[3] = wg_info; o[4] = hw; o[5] = xcc; o[6] = wg_lp3;
    o[7] = clock64() - wg_c0;  // shader-clock cycles over the workgroup's life (against o[2] - o[0] at 100 MHz: the clock)
#pragma unroll
    for (int q = 0; q < 16; ++q) g_wgphase[blockIdx.x * 16 + q] = wg_ph[q];
  }
#endif
}

#include "cagpu_pipe.inc"
#include "cagpu_big.inc"

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
  unsigned char* un = smem + static_cast<size_t>(BLOCK) * 5 * 4;
  float* dcol = reinterpret_cast<float*>(un) + lane;
  float4* Lcol = reinterpret_cast<float4*>(un + static_cast<size_t>(BLOCK) * N * 4) + lane;
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
  if (p->num_agents > big::NT_MAX) return fail(CA_EUNSUPPORTED, "cagpu: num_agents > 1024 is not supported (one thread per agent in the large-env kernel, 1024 threads per workgroup)%s");
  if (p->num_agents > 64 && !o->workspace)
    return fail(CA_EINVAL, "cagpu: num_agents > 64 runs the large-env kernel, which needs CaOut.workspace (cagpu_workspace_bytes(p) bytes)%s");
  if (p->max_obs < 0) return fail(CA_EINVAL, "cagpu: max_obs < 0%s");
  if (p->obs_clip < 0 || p->obs_clip > p->max_obs) return fail(CA_EINVAL, "cagpu: obs_clip must be in [0, max_obs]%s");
  if (p->sort_mode < CA_SORT_CLOSEST_FIRST || p->sort_mode > CA_SORT_TIME_TO_IMPACT)
    return fail(CA_EINVAL, "cagpu: unknown sort_mode (OtherAgentsStatesSensor.py:52 raises ValueError)%s");
  if (p->game_over_mode < 0 || p->game_over_mode > 2) return fail(CA_EINVAL, "cagpu: bad game_over_mode%s");
  if (!(p->dt > 0.0) || !(p->rvo_dt > 0.0)) return fail(CA_EINVAL, "cagpu: dt and rvo_dt must be > 0%s");
  if (!o->obs || !o->rewards || !o->done || !o->game_over) return fail(CA_EINVAL, "cagpu: NULL output pointer%s");
  const void* ptrs[] = {s->pos_x, s->pos_y, s->vel_x, s->vel_y, s->heading, s->goal_x, s->goal_y, s->radius,
                        s->pref_speed, s->time_remaining, s->t, s->slt, s->ep_reward, s->last_action, s->flags,
                        s->step_num, s->episode_step, s->reset_count, s->env_stats};
  for (const void* q : ptrs)
    if (!q) return fail(CA_EINVAL, "cagpu: NULL state pointer%s");
  return CA_OK;
}

// ---- launch plan.  Everything that selects a kernel instantiation or a tile geometry is decided here from the
// arguments and the device's CU count alone (no environment variables in the product build: -DCAGPU_KNOBS adds the
// experiment knobs of scratch/, parsed once; -DCAGPU_ABLATE also the in-kernel phase timers).
thread_local char g_last_kernel[160] = "";

int device_cus() {  // CU count of the current device, cached per device id
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int c = cache[dev].load(std::memory_order_relaxed);
  if (c <= 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cache[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}

struct Knobs {  // experiment overrides (-DCAGPU_ABLATE builds only); -1 / 0 = not set
  int tile = 0, nt = 0, stage = -1, no_nc = 0, rollout_fused = 0;
};
const Knobs& knobs() {
#if defined(CAGPU_ABLATE) || defined(CAGPU_KNOBS)
  static const Knobs kn = [] {
    Knobs q;
    if (const char* e = std::getenv("CAGPU_TILE")) q.tile = std::atoi(e);
    if (const char* e = std::getenv("CAGPU_NT")) q.nt = std::atoi(e);
    if (std::getenv("CAGPU_STAGE")) q.stage = 1;
    if (std::getenv("CAGPU_NOSTAGE")) q.stage = 0;
    if (std::getenv("CAGPU_NO_NC")) q.no_nc = 1;
    if (std::getenv("CAGPU_ROLLOUT_FUSED")) q.rollout_fused = 1;
    return q;
  }();
  return kn;
#else
  static const Knobs kn;
  return kn;
#endif
}

// n single-step launches in place of one n-step launch: the next launch writes the next slot of the caller's output ring
void ring_advance(KArgs& k) {
  k.o.obs += k.ring_obs;
  k.o.rewards += k.ring_agent;
  k.o.done += k.ring_agent;
  k.o.game_over += k.ring_env;
  if (k.o.actions) k.o.actions += 2 * k.ring_agent;
  if (k.o.orca_vel) k.o.orca_vel += 2 * k.ring_agent;
}

template <int NT, bool STAGE, int NC, bool MULTI, bool RO, int TE = 0>
int launch_main5(const KArgs& k, size_t total, hipStream_t st) {
  // per instantiation and device: raise the dynamic-LDS limit once, not on every launch
  static std::atomic<size_t> lds_limit[64];
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  std::atomic<size_t>& lim = lds_limit[dev_id & 63];
  size_t have = lim.load(std::memory_order_relaxed);
  if (have == 0) have = 48 * 1024;
  if (total > have) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ca_kernel<NT, STAGE, NC, MULTI, RO, TE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(total));
    if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: hipFuncSetAttribute: %s", hipGetErrorString(e));
    lim.store(total, std::memory_order_relaxed);
  }
  const int tile_envs = k.tile_envs;
  const unsigned grid = static_cast<unsigned>((k.p.num_envs + tile_envs - 1) / tile_envs);
  std::snprintf(g_last_kernel, sizeof(g_last_kernel), "ca_kernel<%d, %s, %d, %s, %s, %d> grid=%u lds=%zu tile_envs=%d",
                NT, STAGE ? "true" : "false", NC, MULTI ? "true" : "false", RO ? "true" : "false", TE, grid, total,
                tile_envs);
  hipLaunchKernelGGL((ca_kernel<NT, STAGE, NC, MULTI, RO, TE>), dim3(grid), dim3(NT), total, st, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
}

template <int NT, bool STAGE, int NC, bool MULTI, int TE = 0>
int launch_main4(const KArgs& k, size_t total, hipStream_t st) {
  if (k.mode == MODE_STEP && k.table && k.reset_obs) return launch_main5<NT, STAGE, NC, MULTI, true, TE>(k, total, st);
  return launch_main5<NT, STAGE, NC, MULTI, false, TE>(k, total, st);
}

template <int NT, bool STAGE, int NC, int TE = 0>
int launch_main3(const KArgs& k, size_t total, hipStream_t st) {
  if (k.mode == MODE_STEP && k.n_steps > 1) return launch_main4<NT, STAGE, NC, true, TE>(k, total, st);
  return launch_main4<NT, STAGE, NC, false, TE>(k, total, st);
}

template <int NT, bool STAGE>
int launch_main2(const KArgs& k, size_t total, hipStream_t st) {
  if constexpr (NT <= 256) {  // the 512-thread geometry exists for large N only
    if (k.p.num_agents == 10 && !knobs().no_nc) {  // N and the tile size compiled in
      if (k.tile_envs == ROW / 10) return launch_main3<NT, STAGE, 10>(k, total, st);
      if (k.tile_envs == 4) return launch_main3<NT, STAGE, 10, 4>(k, total, st);
    }
#ifndef CAGPU_FAST
    if (k.p.num_agents == 20 && k.tile_envs == ROW / 20 && !knobs().no_nc)  // BASELINE config 3 (4096 x 20)
      return launch_main3<NT, STAGE, 20>(k, total, st);
#endif
  }
#ifdef CAGPU_FAST  // scratch builds: only the N = 10 instantiations are compiled
  return fail(CA_EUNSUPPORTED, "cagpu: CAGPU_FAST experiment build supports num_agents == 10 only%s");
#else
  return launch_main3<NT, STAGE, 0>(k, total, st);
#endif
}

template <int NT>
int launch_main(const KArgs& k, hipStream_t st) {
  const int N = k.p.num_agents, W = 6 + 7 * k.p.max_obs;
  const int cs = k.col_stride;
  const size_t un_orca = lds_orca_bytes(N, cs);
  const int tti = k.p.sort_mode == CA_SORT_TIME_TO_IMPACT ? 1 : 0;
  size_t un_sense = lds_sense_bytes(N, W, 1, tti, cs);
  bool stage = true;
  size_t total = lds_fixed_bytes() + (un_orca > un_sense ? un_orca : un_sense);
  // Staging the tile's observation block in LDS (one coalesced copy-out) wins while every workgroup of the launch is
  // resident at once; for larger batches the smaller footprint without it (5 instead of 3 workgroups per CU at
  // N = 10) hides more latency: 145 vs 167 us at 32768 envs, 32.9 vs 30.7 us at 4096 (profiles/r01_kernel_geometry.md).
  const int n_cu = device_cus();
  const long wgs = (static_cast<long>(k.p.num_envs) + k.tile_envs - 1) / k.tile_envs;
  // (a fourth staged workgroup per CU fits the LDS on paper at N = 10 but measured 44 us at 5120 envs against 35 us for
  // the unstaged layout, so the staged layout is only kept up to three per CU)
  const long per_cu = static_cast<long>((160 * 1024) / total);
  const long resident_staged = (per_cu < 3 ? per_cu : 3) * n_cu;
  bool crowded = wgs > resident_staged;
  if (knobs().stage == 1) crowded = false;
#ifdef CAGPU_FAST
  crowded = true;
#endif
  if (total > 64 * 1024 || crowded || knobs().stage == 0) {  // give up the staging area
    un_sense = lds_sense_bytes(N, W, 0, tti, cs);
    stage = false;
    total = lds_fixed_bytes() + (un_orca > un_sense ? un_orca : un_sense);
  }
  if (total > 160 * 1024) return fail(CA_EUNSUPPORTED, "cagpu: num_agents too large for the 160 KiB LDS tile%s");
  // The fused n-step kernel (<= 128 VGPRs: four 4-wave workgroups per CU) only pays while the whole launch is resident
  // at once; otherwise n launches of the single-step kernel are faster (136 vs 158 us / step at 32768 envs) and give
  // bit-identical results (envs never interact; tests/test_gpu_parity.py::test_rollout_equals_repeated_steps).
  const long lds_cap = static_cast<long>((160 * 1024) / total);
  const long reg_cap = (N == 20) ? 2 : 4;  // (ca_kernel's __launch_bounds__)
  const long fused_resident = (lds_cap < reg_cap ? lds_cap : reg_cap) * n_cu;
  if (k.mode == MODE_STEP && k.n_steps > 1 && wgs > fused_resident && !knobs().rollout_fused) {
    KArgs k1 = k;
    k1.n_steps = 1;
    for (int i = 0; i < k.n_steps; ++i) {
#ifdef CAGPU_FAST
      const int rc = launch_main2<256, false>(k1, total, st);
#else
      const int rc = stage ? launch_main2<256, true>(k1, total, st) : launch_main2<256, false>(k1, total, st);
#endif
      if (rc) return rc;
      ring_advance(k1);
    }
    return CA_OK;
  }
#ifdef CAGPU_FAST
  return launch_main2<NT, false>(k, total, st);
#else
  return stage ? launch_main2<NT, true>(k, total, st) : launch_main2<NT, false>(k, total, st);
#endif
}

// The software-pipelined kernel (cagpu_pipe.inc): the caller handed over CaState.next_action, the agent count has an
// instantiation, the sensor sorts closest_first, an attached fixture table comes with its reset observations, and the
// grid suits its 4-env tiles.
bool pipe_eligible(const KArgs& k) {
  const int n = k.p.num_agents;
  if (!k.s.next_action || k.stage_obs) return false;
  if (k.s.rvo_collab || k.s.rvo_heading_noise || k.s.ext_state) return false;  // (per-step inputs belong to the step that consumes them)
#ifdef CAGPU_FAST
  if (n != 10) return false;
#endif
  if (!(n == 10 || n == 8 || n == 6 || n == 5 || n == 4 || n == 3 || n == 2)) return false;  // (the instantiations)
  if (k.mode != MODE_STEP && k.mode != pipe::MODE_PLAN) return false;
  if (k.p.sort_mode != CA_SORT_CLOSEST_FIRST) return false;
  if (k.table && !k.reset_obs) return false;
  // grid: one round of resident workgroups (4 per CU), or at least two -- in between (1.5 rounds at 6144 envs) ca_kernel's
  // 6-env tiles fit the device better: 21.3 vs 22.9 us per step; from 8192 envs on this kernel is ahead again (28.9 vs 33.0
  // us; 32768 envs: 85.1 vs 87.8, fused rollout 65.7 vs 87.5 us per step; profiles/r03_kernel_geometry.md)
  if (n != 10) return true;
  const long wgs = (static_cast<long>(k.p.num_envs) + 3) / 4, cap = 4L * device_cus();
  return wgs <= cap || wgs >= 2 * cap;
}

// the epoch tags of the GA3C-CADRL packing counters (ga3c::tagged_add): no two packings of a process share a tag
uint32_t next_pack_epoch() {
  static std::atomic<uint32_t> epoch_source{static_cast<uint32_t>(std::chrono::steady_clock::now().time_since_epoch().count())};
  return epoch_source.fetch_add(1, std::memory_order_relaxed);
}

#ifndef CAGPU_PIPE_YIELD_T
#define CAGPU_PIPE_YIELD_T 2   // (A/B builds: 0 = off)
#endif
template <int NC, int TE, bool MULTI>
int launch_pipe2(const KArgs& k0, hipStream_t st) {
  using G = pipe::Geo<NC, TE>;
  static_assert(G::LDS <= 48 * 1024 && (NC != 10 || G::LDS <= 40 * 1024), "three (metric geometry: four) workgroups per CU");
  static_assert(TE <= 32 && NC * TE <= 64 && NC >= 2 && NC <= 10, "one agent wave per half; lp3_wave8 holds at most 9 lines");
  KArgs k = k0;
  const unsigned grid = static_cast<unsigned>((k.p.num_envs + TE - 1) / TE);
  k.inv_h_f = 1.0f / static_cast<float>(k.p.rvo_time_horizon);   // (x86 float division: correctly rounded, like the device's `/`)
  k.inv_dt_f = 1.0f / static_cast<float>(k.p.rvo_dt);
  // progress-fair priorities (PIPE_PRIO): an n-step launch whose workgroups are all resident at once and share their CUs
  k.yield_t = 0;
  if (MULTI && CAGPU_PIPE_YIELD_T > 0) {
    const long cus = device_cus();
    const long per_cu = (160 * 1024) / static_cast<long>(G::LDS) < 4 ? (160 * 1024) / static_cast<long>(G::LDS) : 4;
    if (grid > cus && grid <= per_cu * cus) k.yield_t = CAGPU_PIPE_YIELD_T;
  }
  std::snprintf(g_last_kernel, sizeof(g_last_kernel), "ca_pipe_kernel<%d, %d, %s> grid=%u lds=%zu mode=%d%s", NC, TE,
                MULTI ? "true" : "false", grid, static_cast<size_t>(G::LDS), k.mode, k.yield_t > 0 ? " fair" : "");
  if (MULTI && k.yield_t > 0) hipLaunchKernelGGL((pipe::ca_pipe_kernel<NC, TE, MULTI, MULTI>), dim3(grid), dim3(pipe::PNT), G::LDS, st, k);
  else hipLaunchKernelGGL((pipe::ca_pipe_kernel<NC, TE, MULTI, false>), dim3(grid), dim3(pipe::PNT), G::LDS, st, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
}

template <int NC, int TE>
int launch_pipe1(const KArgs& k, hipStream_t st) {
  if (k.mode == MODE_STEP && k.n_steps > 1) return launch_pipe2<NC, TE, true>(k, st);
  return launch_pipe2<NC, TE, false>(k, st);
}

int launch_pipe(const KArgs& k, hipStream_t st) {
  switch (k.p.num_agents) {
    // 4-env tiles at EVERY batch size (1024 workgroups at the metric's 4096 envs, 4 per CU).  Smaller tiles for small batches
    // -- 1 / 2 / 3 envs per tile so that a 1024 / 2048 / 3072-env batch is still one full round of 1024 resident workgroups --
    // were built and measured in round 4 and are SLOWER: 12.41 vs 11.89 us per step at 1024 envs (BASELINE configs[1]), 13.35
    // vs 12.83 at 2048, 14.69 vs 14.05 at 3072 (profiles/r04_kernel_geometry.md).  A workgroup's duration is its serial
    // chain (a tile alone on a CU still needs ~9.5 us), not its pair work, so four times as many workgroups only add their
    // fixed costs and contend for the issue slots the chains need.
    // Round 6, the same question in ring mode (2- and 1-env tiles for grids of at most one 4-env workgroup per CU, same box,
    // 1024 x 10): ring of 64 8.10 -> 8.08 / 7.99 us per step, ring of 20 9.40 -> 9.40 / 9.46, 2000-step rollout 6.73 -> 6.57 / 6.45,
    // one launch per step 11.78 -> 11.92 / 12.18: below the 0.5 us bar in every mode a caller sees; not built
    // (profiles/r06_kernel_geometry.md).
    case 10: return launch_pipe1<10, 4>(k, st);
#ifndef CAGPU_FAST
    case 8: return launch_pipe1<8, 8>(k, st);
    case 6: return launch_pipe1<6, 10>(k, st);
    case 5: return launch_pipe1<5, 12>(k, st);
    case 4: return launch_pipe1<4, 16>(k, st);
    case 3: return launch_pipe1<3, 21>(k, st);
    case 2: return launch_pipe1<2, 32>(k, st);
#endif
    default: return fail(CA_EUNSUPPORTED, "cagpu: no pipelined instantiation for this agent count%s");
  }
}

// Envs with more than 64 agents: the one-thread-per-agent kernel of cagpu_big.inc over the caller's workspace; as many
// workgroups as the workspace holds, each walking envs e = blockIdx, blockIdx + grid, ...; a rollout is n launches.
int launch_big(const KArgs& k0, hipStream_t st) {
  KArgs k = k0;
  const int N = k.p.num_agents;
  const size_t per = big::ws_bytes_per_wg(N);
  long wgs = static_cast<long>(k.o.workspace_bytes / per);
  if (wgs < 1) return fail(CA_EINVAL, "cagpu: CaOut.workspace is smaller than one workgroup's share (cagpu_workspace_bytes)%s");
  if (wgs > k.p.num_envs) wgs = k.p.num_envs;
  const int n_steps = (k.mode == MODE_STEP) ? k.n_steps : 1;
  k.n_steps = 1;
  const int nt = big::threads_for(N);
  const size_t lds = big::lds_bytes(nt);
  std::snprintf(g_last_kernel, sizeof(g_last_kernel), "ca_big_kernel grid=%ld threads=%d ws_per_wg=%zu mode=%d", wgs, nt, per,
                k.mode);
  if (nt == 1024) {  // 88 KB of LDS: above the default dynamic limit
    static std::atomic<bool> raised[64];
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    if (!raised[dev_id & 63].load(std::memory_order_relaxed)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&big::ca_big_kernel<1024>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: hipFuncSetAttribute: %s", hipGetErrorString(e));
      raised[dev_id & 63].store(true, std::memory_order_relaxed);
    }
  }
  for (int s = 0; s < n_steps; ++s) {
    unsigned char* wsp = static_cast<unsigned char*>(k.o.workspace);
    if (nt == 256) hipLaunchKernelGGL(big::ca_big_kernel<256>, dim3(static_cast<unsigned>(wgs)), dim3(256), lds, st, k, wsp, per);
    else if (nt == 512) hipLaunchKernelGGL(big::ca_big_kernel<512>, dim3(static_cast<unsigned>(wgs)), dim3(512), lds, st, k, wsp, per);
    else hipLaunchKernelGGL(big::ca_big_kernel<1024>, dim3(static_cast<unsigned>(wgs)), dim3(1024), lds, st, k, wsp, per);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
    ring_advance(k);
  }
  return CA_OK;
}

// Workgroup size (measured on MI355X at 4096 envs x 10 agents, profiles/r01_kernel_geometry.md): 256 threads for both the
// single-step kernel (30.7 us; 128 -> 39, 384 / 512 -> 40) and the n-step rollout kernel (22.2 us / step; 128 -> 28.8).
// The rollout kernel only reaches that since the build disables machine LICM (build_native.py): hoisted loop invariants
// had cost it 217 VGPRs (2 waves / SIMD: the 683 workgroups of 256 threads no longer fit at once) instead of 139.
int launch_any(const KArgs& k0, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  KArgs k = k0;
  const int N = k.p.num_agents;
#ifndef CAGPU_NOPIPE
  if (pipe_eligible(k)) return launch_pipe(k, st);
#endif
  if (k.mode != pipe::MODE_PLAN && N > 64) return launch_big(k, st);
  if (k.mode == pipe::MODE_PLAN) return fail(CA_EUNSUPPORTED, "cagpu_plan: needs CaState.next_action, num_agents in {2, 3, 4, 5, 6, 8, 10}, closest_first sorting and a grid of at most 4 x CUs or at least 8 x CUs tiles%s");
  k.tile_envs = ROW / N;
  k.col_stride = (N > 32) ? (CAGPU_CSPAD ? (N | 1) : N) : CS_ROW;  // single-env tiles: only the N columns in use (see ca_kernel)
  // N = 10: tiles of 4 envs instead of 6 while that still gives at most 4 workgroups per CU (all co-resident, evenly
  // spread): 27.3 vs 30.9 us at 4096 envs, 23.3 vs 27.6 at 2048; beyond that the 6-env tile wins
  // (profiles/r01_kernel_geometry.md)
  if (N == 10 && k.mode == MODE_STEP && !knobs().tile) {
    if ((static_cast<long>(k.p.num_envs) + 3) / 4 <= 4L * device_cus()) k.tile_envs = 4;
  }
  if (knobs().tile >= 1 && knobs().tile < k.tile_envs) k.tile_envs = knobs().tile;
  // N > 32: the tile is a single env whose N^2 pair items (and N wave-wide linear programs) keep 8 waves busy, and
  // its LDS footprint allows only one or two workgroups per CU anyway
  int nt = (N > 32) ? 512 : 256;
  if (knobs().nt) nt = knobs().nt;
#ifdef CAGPU_FAST
  return launch_main<256>(k, st);
#else
  if (nt <= 256) return launch_main<256>(k, st);
  return launch_main<512>(k, st);
#endif
}

int pick_block(int N) { return N <= 64 ? 64 : (N <= 128 ? 128 : 256); }

template <int BLOCK>
int launch_orca(const OrcaArgs& k, hipStream_t st) {
  const int N = k.num_agents;
  const size_t total = static_cast<size_t>(BLOCK) * (5 * 4 + N * 4 + 2 * (N > 1 ? N - 1 : 1) * 16);
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


// ---------------------------------------------------------------- parity hook: the device's own libm-dependent operations
// (cagpu_debug_libm).  The step's results differ from a CPU run of the same algorithm only through these: atan2 (ROCm's
// ocml, vs glibc's on the host), the heading's sin / cos (sincos_heading above, a short-range kernel) and the lean
// divide / square root sequences (divq / sqrtq / divd / sqrtd above).
__global__ void debug_libm_kernel(const int n, const int op, const double* a, const double* b, double* o0, double* o1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b ? b[i] : 0.0;
  double r0 = 0.0, r1 = 0.0;
  switch (op) {
    case 0: r0 = atan2(x, y); break;                                   // atan2(a, b)
    case 1: sincos_heading(x, r0, r1); break;                          // (sin a, cos a), a in [-pi, pi]
    case 2: r0 = divq(static_cast<float>(x), static_cast<float>(y)); r1 = static_cast<float>(x) / static_cast<float>(y); break;
    case 3: r0 = sqrtq(static_cast<float>(x)); r1 = sqrtf(static_cast<float>(x)); break;
    case 4: r0 = divd(x, y); r1 = x / y; break;
    case 5: r0 = sqrtd(x); r1 = sqrt(x); break;
    case 6: { const Ego e = ego_frame(0.0, 0.0, x, y, 0.0); r0 = e.heading_ego; r1 = e.dist; } break;  // goal (a, b) seen from the origin
    default: break;
  }
  o0[i] = r0;
  if (o1) o1[i] = r1;
}

// cagpu_debug_copy8: a streaming copy with the step kernels' 8-bytes-per-lane access shape (counter calibration)
__global__ __launch_bounds__(256) void copy8_kernel(const long n, const double* __restrict__ src, double* __restrict__ dst) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

}  // namespace

extern "C" {

int cagpu_version(void) { return CAGPU_VERSION; }

const char* cagpu_last_error(void) { return g_err; }

const char* cagpu_last_kernel(void) { return g_last_kernel; }

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
                     int32_t n_steps, void* stream, const CaMap* map = nullptr, const bool ring = false,
                     const int64_t snapshot_delta = 0, const bool query_snapshot = false) {
  int rc = check_params(p, s, o);
  if (rc) return rc;
  if (n_steps < 1) return fail(CA_EINVAL, "cagpu: n_steps must be >= 1%s");
  if ((n_steps > 1 || ring) && (s->rvo_collab || s->rvo_heading_noise || s->ext_state))
    return fail(CA_EINVAL, "cagpu: CaState.rvo_collab / rvo_heading_noise / ext_state are inputs of ONE step (the caller draws / "
                           "integrates them per step): not accepted by a multi-step call%s");
  KArgs k;
  std::memset(&k, 0, sizeof(k));
  k.p = *p; k.s = *s; k.o = *o; k.ext = ext;
  if (ar) {
    if (!ar->table || ar->n_cases < 1) return fail(CA_EINVAL, "cagpu: bad CaAutoReset%s");
    k.table = ar->table; k.n_cases = ar->n_cases; k.env_id_offset = ar->env_id_offset; k.case_stride = ar->case_stride;
    k.heading_seed = ar->heading_seed;
    k.reset_obs = ar->heading_seed ? nullptr : ar->reset_obs;  // (a reset observation depends on the heading)
    k.reset_plan = ar->heading_seed ? nullptr : ar->reset_plan;
  }
  if (map && map->static_bits) {
    if (map->rows < 1 || map->cols < 1 || !(map->cell > 0.0)) return fail(CA_EINVAL, "cagpu: bad CaMap%s");
    k.map = *map;
  }
  k.n_steps = n_steps; k.mode = MODE_STEP;
  k.inv_rvo_dt = 1.0 / p->rvo_dt;
  {
    static std::atomic<int> seq{0};
    k.launch_seq = seq.fetch_add(1, std::memory_order_relaxed) + 1;
  }
  if (ring) {
    k.ring_agent = static_cast<int64_t>(p->num_envs) * p->num_agents;
    k.ring_obs = k.ring_agent * (6 + 7 * p->max_obs);
    k.ring_env = p->num_envs;
    // the in-kernel snapshot is the pipelined n-step kernel's (launch_pipe with n_steps > 1): every other form of the call
    // leaves it to the caller (cagpu_ring_snapshots says which, from the same arguments)
#ifdef CAGPU_NOPIPE   // (a build whose launcher never picks the pipelined kernel must not promise its snapshot either)
    const bool can = false;
#else
    const bool can = n_steps > 1 && pipe_eligible(k);
#endif
    if (query_snapshot) return can ? 1 : 0;
    if (snapshot_delta != 0 && !can)
      return fail(CA_EUNSUPPORTED, "cagpu_rollout_ring: snapshot_delta needs the pipelined n-step kernel (cagpu_ring_snapshots() == 1 for these arguments)%s");
    k.snap_delta = snapshot_delta;
  }
#ifdef CAGPU_ABLATE
  if (const char* ab = std::getenv("CAGPU_ABLATE")) k.ablate = std::atoi(ab);
#endif
  return launch_any(k, stream);
}

int cagpu_step(const CaParams* p, const CaState* s, const CaOut* o, const double* ext_actions, const CaAutoReset* ar,
               void* stream) {
  return step_impl(p, s, o, ext_actions, ar, 1, stream);
}

int cagpu_step_map(const CaParams* p, const CaState* s, const CaOut* o, const double* ext_actions, const CaAutoReset* ar,
                   const CaMap* map, void* stream) {
  return step_impl(p, s, o, ext_actions, ar, 1, stream, map);
}

int cagpu_laserscan(const CaParams* p, const CaState* s, const CaMap* map, const CaScan* scan, void* stream) {
  if (!p || !s || !map || !scan) return fail(CA_EINVAL, "cagpu_laserscan: NULL argument%s");
  if (p->num_envs < 1 || p->num_agents < 1 || p->num_agents > 256) return fail(CA_EINVAL, "cagpu_laserscan: bad sizes%s");
  if (map->rows < 1 || map->cols < 1 || !(map->cell > 0.0)) return fail(CA_EINVAL, "cagpu_laserscan: bad CaMap%s");
  if (!scan->hist || !scan->out || scan->num_beams < 2 || scan->num_to_store < 1 || scan->num_ranges < 1 ||
      scan->num_ranges > 255)
    return fail(CA_EINVAL, "cagpu_laserscan: bad CaScan%s");
  if (!s->pos_x || !s->pos_y || !s->heading || !s->radius || !s->step_num) return fail(CA_EINVAL, "cagpu_laserscan: NULL state pointer%s");
  if (!(scan->range_res / map->cell + 0.2 < SCAN_PAD - 1))
    return fail(CA_EUNSUPPORTED, "cagpu_laserscan: range_res above 6 cells is not supported (LDS bitmap border)%s");
  ScanArgs k;
  std::memset(&k, 0, sizeof(k));
  k.p = *p; k.s = *s; k.m = *map; k.sc = *scan;
  const int N = p->num_agents;
  const size_t total = align16(scan_grid_words(map->rows, map->cols) * 4) + static_cast<size_t>(N) * (4 * 8 + 2 * 8 + 3 * 4 + 2 * 8 + 2 * 8 + 3 * 4 + 4 * 4) +
                       static_cast<size_t>(scan->num_beams) * 16 + 256 * 4 + 32 + static_cast<size_t>(HIST_AG) * SCAN_NT;
  if (total > 160 * 1024) return fail(CA_EUNSUPPORTED, "cagpu_laserscan: map too large for the LDS bitmap%s");
  if (total > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(total));
    if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(scan_kernel, dim3(static_cast<unsigned>(p->num_envs)), dim3(SCAN_NT), total,
                     static_cast<hipStream_t>(stream), k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
}

int cagpu_ga3c(const CaParams* p, const CaState* s, const float* obs, const CaNet* net, double* ext_actions, float* logits,
               void* stream) {
  if (!p || !s || !net || !ext_actions) return fail(CA_EINVAL, "cagpu_ga3c: NULL argument%s");
  if (!obs) {  // fused sensing: the kernel computes the observation rows it needs from the state
    if (p->num_agents > 32 || p->sort_mode == CA_SORT_TIME_TO_IMPACT)
      return fail(CA_EUNSUPPORTED, "cagpu_ga3c: fused sensing (obs == NULL) needs num_agents <= 32 and closest_first / closest_last sorting%s");
    if (p->obs_clip < 0 || p->obs_clip > p->max_obs) return fail(CA_EINVAL, "cagpu_ga3c: obs_clip must be in [0, max_obs]%s");
    const void* q[] = {s->pos_x, s->pos_y, s->vel_x, s->vel_y, s->heading, s->goal_x, s->goal_y, s->radius, s->pref_speed};
    for (const void* x : q)
      if (!x) return fail(CA_EINVAL, "cagpu_ga3c: fused sensing needs the state arrays%s");
  }
  if (p->num_envs < 1 || p->num_agents < 1 || p->max_obs < 0) return fail(CA_EINVAL, "cagpu_ga3c: bad sizes%s");
  if (!s->flags) return fail(CA_EINVAL, "cagpu_ga3c: NULL state pointer%s");
  const void* w[] = {net->lstm_kernel, net->lstm_bias, net->layer1_kernel, net->layer1_bias, net->layer2_kernel,
                     net->layer2_bias, net->fc1_kernel, net->fc1_bias, net->logits_kernel, net->logits_bias,
                     net->input_mean, net->input_std};
  for (const void* q : w)
    if (!q || (reinterpret_cast<uintptr_t>(q) & 15)) return fail(CA_EINVAL, "cagpu_ga3c: NULL or not 16-byte aligned weight pointer%s");
  if (!net->packed || (reinterpret_cast<uintptr_t>(net->packed) & 15))
    return fail(CA_EINVAL, "cagpu_ga3c: CaNet.packed is NULL or not 16-byte aligned (fill it once per checkpoint with cagpu_ga3c_pack)%s");
  ga3c::Args k;
  std::memset(&k, 0, sizeof(k));
  k.obs = obs; k.flags = s->flags;
  k.pos_x = s->pos_x; k.pos_y = s->pos_y; k.vel_x = s->vel_x; k.vel_y = s->vel_y; k.heading = s->heading;
  k.goal_x = s->goal_x; k.goal_y = s->goal_y; k.radius = s->radius; k.pref_speed = s->pref_speed;
  k.N = p->num_agents; k.K = p->max_obs; k.obs_clip = p->obs_clip; k.sort_mode = p->sort_mode; k.ragged = p->ragged;
  k.sensing_horizon = p->sensing_horizon;
  k.B = static_cast<long>(p->num_envs) * p->num_agents;
  k.W = 6 + 7 * p->max_obs;
  k.net = *net; k.ext = ext_actions; k.logits = logits;
  k.rows = net->rows_scratch;
  if (k.rows) {
    if (k.B >= (1L << 31)) return fail(CA_EUNSUPPORTED, "cagpu_ga3c: more than 2^31 agents with rows_scratch%s");
    // the packing's two counters are tagged with this call's epoch (compact_kernel): nothing to clear, whatever an earlier
    // launch or the caller's allocation left in the scratch; rows_scratch must hold num_envs * num_agents + 6 words
    const uint32_t epoch = next_pack_epoch();
    hipLaunchKernelGGL(ga3c::compact_kernel, dim3(static_cast<unsigned>((k.B + 4 * ga3c::CP_NT - 1) / (4 * ga3c::CP_NT))),
                       dim3(ga3c::CP_NT), 0, static_cast<hipStream_t>(stream), s->flags, k.B, net->rows_scratch, net->agent_net,
                       net->net_index, epoch);
  }
  static_assert(ga3c::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
#if defined(CAGPU_GA3C_SOLO)   // timing experiment (scratch/ga3c_phases.py): one workgroup per CU, a wave has its SIMD to itself
  constexpr size_t GA3C_LDS = 100 * 1024;
#else
  constexpr size_t GA3C_LDS = ga3c::LDS_BYTES;
#endif
  static thread_local bool lds_raised[16] = {false};
  int dev_id = 0;
  (void)hipGetDevice(&dev_id);
  hipError_t e = hipSuccess;
  if (!lds_raised[dev_id & 15]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ga3c::ga3c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(GA3C_LDS));
    if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: hipFuncSetAttribute: %s", hipGetErrorString(e));
    lds_raised[dev_id & 15] = true;
  }
  k.slots = 2 * device_cus();
  k.force_tile = 0;
#if defined(CAGPU_KNOBS)
  if (const char* ft = std::getenv("CAGPU_GA3C_TILE")) k.force_tile = std::atoi(ft);
#endif
  // the tile height (64 / 48 / 32 rows) is chosen on the device from the number of live rows: the grid covers the worst
  // case (32-row tiles); workgroups beyond the last tile leave at once
  const unsigned grid = static_cast<unsigned>((k.B + 31) / 32);
  hipLaunchKernelGGL(ga3c::ga3c_kernel, dim3(grid), dim3(ga3c::NT), GA3C_LDS, static_cast<hipStream_t>(stream), k);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
}

static int generate_impl(int64_t num_cases, int32_t num_agents, int32_t n_min, int32_t n_max, const double* side_ranges,
                         int32_t n_ranges, double side_lo, double side_hi, double speed_lo, double speed_hi, double radius_lo,
                         double radius_hi, uint64_t seed, double* cases, int32_t* counts, int32_t* status, void* stream) {
  if (num_cases < 1 || num_agents < 1 || num_agents > 4096) return fail(CA_EINVAL, "cagpu_generate_cases: bad sizes%s");
  if (!cases) return fail(CA_EINVAL, "cagpu_generate_cases: NULL cases%s");
  if (!(side_lo > 0.0) || !(side_hi >= side_lo) || !(speed_lo > 0.0) || !(speed_hi >= speed_lo) || !(radius_lo > 0.0) ||
      !(radius_hi >= radius_lo))
    return fail(CA_EINVAL, "cagpu_generate_cases: bounds must be positive and ordered%s");
  gen::Args a;
  std::memset(&a, 0, sizeof(a));
  a.cases = cases; a.status = status; a.C = num_cases; a.N = num_agents;
  a.side_lo = side_lo; a.side_hi = side_hi; a.speed_lo = speed_lo; a.speed_hi = speed_hi;
  a.radius_lo = radius_lo; a.radius_hi = radius_hi; a.seed = seed;
  a.max_attempts = 20000;  // the reference loops until a sample is accepted (its square / circle grows 1 % per retry)
  a.counts = counts;
  if (n_max > 0) {
    if (n_min < 1 || n_min > n_max || n_max > num_agents)
      return fail(CA_EINVAL, "cagpu_generate_cases_ragged: need 1 <= n_min <= n_max <= max_agents%s");
    a.n_min = n_min; a.n_max = n_max;
  }
  if (n_ranges > 0) {
    if (n_ranges > 8 || !side_ranges) return fail(CA_EINVAL, "cagpu_generate_cases_ragged: at most 8 side ranges%s");
    for (int r = 0; r < n_ranges; ++r) {
      for (int k = 0; k < 4; ++k) a.ranges[r][k] = side_ranges[r * 4 + k];
      if (!(a.ranges[r][2] > 0.0) || !(a.ranges[r][3] >= a.ranges[r][2]))
        return fail(CA_EINVAL, "cagpu_generate_cases_ragged: side ranges must be positive and ordered%s");
    }
    a.n_ranges = n_ranges;
    // the reference asserts that some entry holds the drawn count (test_cases.py:241)
    for (int n = (n_max > 0 ? n_min : num_agents); n <= (n_max > 0 ? n_max : num_agents); ++n) {
      bool held = false;
      for (int r = 0; r < n_ranges; ++r) held = held || (a.ranges[r][0] <= n && n < a.ranges[r][1]);
      if (!held) return fail(CA_EINVAL, "cagpu_generate_cases_ragged: an agent count in [n_min, n_max] has no side range%s");
    }
  }
  hipLaunchKernelGGL(gen::generate_kernel, dim3(static_cast<unsigned>((num_cases + 63) / 64)), dim3(64), 0,
                     static_cast<hipStream_t>(stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
}

int cagpu_generate_cases(int64_t num_cases, int32_t num_agents, double side_lo, double side_hi, double speed_lo,
                         double speed_hi, double radius_lo, double radius_hi, uint64_t seed, double* cases, int32_t* status,
                         void* stream) {
  return generate_impl(num_cases, num_agents, 0, 0, nullptr, 0, side_lo, side_hi, speed_lo, speed_hi, radius_lo, radius_hi,
                       seed, cases, nullptr, status, stream);
}

int cagpu_generate_cases_ragged(int64_t num_cases, int32_t max_agents, int32_t n_min, int32_t n_max,
                                const double* side_ranges, int32_t n_ranges, double speed_lo, double speed_hi,
                                double radius_lo, double radius_hi, uint64_t seed, double* cases, int32_t* counts,
                                int32_t* status, void* stream) {
  if (n_ranges < 1) return fail(CA_EINVAL, "cagpu_generate_cases_ragged: need at least one side range%s");
  if (n_max < 1) return fail(CA_EINVAL, "cagpu_generate_cases_ragged: need 1 <= n_min <= n_max <= max_agents%s");
  return generate_impl(num_cases, max_agents, n_min, n_max, side_ranges, n_ranges, 1.0, 1.0, speed_lo, speed_hi, radius_lo,
                       radius_hi, seed, cases, counts, status, stream);
}

int cagpu_rollout(const CaParams* p, const CaState* s, const CaOut* o, const double* ext_actions, const CaAutoReset* ar,
                  int32_t n_steps, void* stream) {
  return step_impl(p, s, o, ext_actions, ar, n_steps, stream);
}

int cagpu_rollout_ring(const CaParams* p, const CaState* s, const CaOut* o, const double* ext_actions, const CaAutoReset* ar,
                       int32_t n_steps, int64_t snapshot_delta, void* stream) {
  return step_impl(p, s, o, ext_actions, ar, n_steps, stream, nullptr, true, snapshot_delta);
}

int cagpu_ring_snapshots(const CaParams* p, const CaState* s, const CaOut* o, const CaAutoReset* ar, int32_t n_steps) {
  return step_impl(p, s, o, nullptr, ar, n_steps, nullptr, nullptr, true, 0, true);
}

int cagpu_plan(const CaParams* p, const CaState* s, void* stream) {
  if (!p || !s) return fail(CA_EINVAL, "cagpu_plan: NULL params/state%s");
  if (p->num_agents > 10) return fail(CA_EUNSUPPORTED, "cagpu_plan: no pipelined step kernel for more than 10 agents per env%s");
  CaOut o;
  std::memset(&o, 0, sizeof(o));
  o.obs = reinterpret_cast<float*>(1); o.rewards = o.obs; o.done = reinterpret_cast<uint8_t*>(1); o.game_over = o.done;  // (not touched in this mode)
  int rc = check_params(p, s, &o);
  if (rc) return rc;
  if (!s->next_action) return fail(CA_EINVAL, "cagpu_plan: CaState.next_action is NULL%s");
  KArgs k;
  std::memset(&k, 0, sizeof(k));
  k.p = *p; k.s = *s;
  k.n_steps = 1; k.mode = pipe::MODE_PLAN;
  k.inv_rvo_dt = 1.0 / p->rvo_dt;
  return launch_any(k, stream);
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

#ifdef CAGPU_STEPTIME
int cagpu_debug_steptime(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pipe::g_steptime), sizeof(unsigned long long) * 1024 * 66);
  return 0;
}
#endif
#ifdef CAGPU_PIPETIME
int cagpu_debug_pipetime(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pipe::g_pipetime), sizeof(unsigned long long) * 1024 * 32);
  return 0;
}
#endif
#ifdef CAGPU_WGTIME
int cagpu_debug_wgtime(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgtime), sizeof(unsigned long long) * 4096 * 8);
  return 0;
}
int cagpu_debug_wgphase(unsigned int* out) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgphase), sizeof(unsigned int) * 4096 * 16);
  return 0;
}
#endif
#ifdef CAGPU_ABLATE
int cagpu_debug_wgprof(unsigned long long* out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgprof), sizeof(unsigned long long) * 1024 * 16);
  return 0;
}
int cagpu_debug_prof(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
  return 0;
}
#endif

uint64_t cagpu_ga3c_packed_bytes(void) { return static_cast<uint64_t>(ga3c::PK_TOTAL) * sizeof(ga3c::u32x4); }

int cagpu_ga3c_pack(const CaNet* net, void* packed, uint64_t bytes, void* stream) {
  if (!net || !packed) return fail(CA_EINVAL, "cagpu_ga3c_pack: NULL argument%s");
  if (bytes < cagpu_ga3c_packed_bytes() || (reinterpret_cast<uintptr_t>(packed) & 15))
    return fail(CA_EINVAL, "cagpu_ga3c_pack: the buffer must hold cagpu_ga3c_packed_bytes() bytes, 16-byte aligned%s");
  if (!net->lstm_kernel || !net->layer1_kernel || !net->layer2_kernel || !net->fc1_kernel)
    return fail(CA_EINVAL, "cagpu_ga3c_pack: NULL weight pointer%s");
  ga3c::u32x4* out = static_cast<ga3c::u32x4*>(packed);
  struct { const float* w; int row0, k_real, nkb, at; } jobs[4] = {
      {net->lstm_kernel, 7, 64, 2, ga3c::PK_LSTM},     // rows 0..6 (x_t) and the bias: split by the kernel itself
      {net->layer1_kernel, 4, 64, 2, ga3c::PK_L1},     // rows 0..3 (host) stay float32
      {net->layer2_kernel, 0, 256, 8, ga3c::PK_L2},
      {net->fc1_kernel, 0, 256, 8, ga3c::PK_FC1}};
  for (const auto& j : jobs) {
    const int threads = j.nkb * 16 * 64;
    hipLaunchKernelGGL(ga3c::pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), j.w,
                       j.row0, j.k_real, j.nkb, out + j.at, j.at == ga3c::PK_LSTM ? 1 : 0);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu_ga3c_pack: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
}

uint64_t cagpu_workspace_bytes(const CaParams* p) {
  if (!p || p->num_agents <= 64 || p->num_agents > big::NT_MAX || p->num_envs < 1) return 0;
  // two resident workgroups per CU keep the device busy (one above 256 agents: 8 / 16 waves and up to 88 KB of LDS each);
  // more only cost memory (a share is 15 MB at 512 agents, 63 MB at 1024)
  long wgs = (p->num_agents <= 256 ? 2L : 1L) * device_cus();
  if (wgs < 1) wgs = 512;
  if (wgs > p->num_envs) wgs = p->num_envs;
  return static_cast<uint64_t>(wgs) * big::ws_bytes_per_wg(p->num_agents);
}

int cagpu_device_faults(uint32_t* faults, int32_t clear) {
  if (!faults) return fail(CA_EINVAL, "cagpu_device_faults: NULL pointer%s");
  unsigned int v = 0u;
  hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_fault), sizeof(v));  // (synchronises the device)
  if (e == hipSuccess && clear && v) {
    const unsigned int z = 0u;
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_fault), &z, sizeof(z));
  }
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu_device_faults: %s", hipGetErrorString(e));
  *faults = v;
  return CA_OK;
}

int cagpu_device_faults_async(uint32_t* host_dst, void* stream) {
  if (!host_dst) return fail(CA_EINVAL, "cagpu_device_faults_async: NULL pointer%s");
  hipError_t e = hipMemcpyFromSymbolAsync(host_dst, HIP_SYMBOL(g_fault), sizeof(unsigned int), 0, hipMemcpyDeviceToHost,
                                          static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu_device_faults_async: %s", hipGetErrorString(e));
  return CA_OK;
}

int cagpu_debug_libm(int32_t op, int32_t n, const double* a, const double* b, double* out0, double* out1) {
  if (n < 1 || !a || !out0 || op < 0 || op > 6) return fail(CA_EINVAL, "cagpu_debug_libm: bad arguments%s");
  double* d = nullptr;
  const size_t sz = sizeof(double) * static_cast<size_t>(n);
  if (hipMalloc(reinterpret_cast<void**>(&d), 4 * sz) != hipSuccess) return fail(CA_ELAUNCH, "cagpu_debug_libm: hipMalloc failed%s");
  hipError_t e = hipMemcpy(d, a, sz, hipMemcpyHostToDevice);
  if (e == hipSuccess && b) e = hipMemcpy(d + n, b, sz, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(debug_libm_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, op, d, b ? d + n : nullptr, d + 2 * n,
                       d + 3 * n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out0, d + 2 * n, sz, hipMemcpyDeviceToHost);
  if (e == hipSuccess && out1) e = hipMemcpy(out1, d + 3 * n, sz, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu_debug_libm: %s", hipGetErrorString(e));
  return CA_OK;
}

int cagpu_debug_copy8(int64_t n, const double* src, double* dst, void* stream) {
  if (n < 1 || !src || !dst) return fail(CA_EINVAL, "cagpu_debug_copy8: bad arguments%s");
  hipLaunchKernelGGL(copy8_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<long>(n), src, dst);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CA_ELAUNCH, "cagpu: kernel launch failed: %s", hipGetErrorString(e));
  return CA_OK;
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
