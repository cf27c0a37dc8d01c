// oracle/orca_ref.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// CPU restatement, in C `float` arithmetic, of the ORCA velocity computation that the reference
// reaches through `rvo2.PyRVOSimulator.doStep()` (call sites:
// gym_collision_avoidance/envs/policies/RVOPolicy.py:25-28,46,70-74,86-90,93,96).
//
// PARITY UNPINNED for this file: the arithmetic lives in the third-party `mit-acl/Python-RVO2`
// git submodule (.gitmodules:1-4; a fork of sybrenstuvel/Python-RVO2 wrapping the UNC "RVO2
// Library" v2.0.x, C++), whose directory is EMPTY in /root/reference and whose pinned commit is
// unknown (no .git).  There is no network, so the library cannot be fetched, and the reference's
// tests hold no golden vector for it (tests/test_collision_avoidance.py:60-80 only checks that a
// PNG exists).  What follows restates the published algorithm -- van den Berg, Guy, Lin, Manocha,
// "Reciprocal n-body collision avoidance" (ORCA), and the RVO2 v2.0.x reference implementation's
// operation order (Agent::computeNeighbors / computeNewVelocity / linearProgram1-3 / update,
// RVO_EPSILON = 1e-5f, Vector2 division as multiply-by-reciprocal) -- as specified in
// SURVEY.md Appendix B.  It is self-pinned: tests/golden/ records its outputs, and its behaviour
// is anchored on the reference's own call sites (it drives the unmodified RVOPolicy.py through
// the `rvo2` module built from oracle/rvo2_module/).
//
// Fork delta: `setAgentCollabCoeff` (RVOPolicy.py:86-90) replaces upstream's constant 0.5 in
// `line.point = velocity + 0.5f * u`; with Config.RVO_COLLAB_COEFF = 0.5 (config.py:85) it is
// identical to upstream ORCA.
//
// Build with -ffp-contract=off (no FMA contraction): upstream is built for baseline x86-64.
#ifndef ORACLE_ORCA_REF_H_
#define ORACLE_ORCA_REF_H_

#include <cmath>
#include <cstddef>
#include <vector>

namespace orca_ref {

static const float kEps = 0.00001f;  // RVO_EPSILON

struct Vec {
  float x, y;
};
struct HalfPlane {  // an ORCA line: permitted side is to the LEFT of `dir` through `pt`
  Vec pt, dir;
};

static inline Vec mk(float x, float y) { Vec v = {x, y}; return v; }
static inline Vec add(Vec a, Vec b) { return mk(a.x + b.x, a.y + b.y); }
static inline Vec sub(Vec a, Vec b) { return mk(a.x - b.x, a.y - b.y); }
static inline Vec scl(float s, Vec a) { return mk(s * a.x, s * a.y); }   // s * v  (also v * s)
static inline float dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y; }
static inline float cross(Vec a, Vec b) { return a.x * b.y - a.y * b.x; }  // det(a, b)
static inline float sq(float a) { return a * a; }
static inline float len(Vec a) { return std::sqrt(dot(a, a)); }
// v / s is evaluated as v * (1/s) in the library's vector type.
static inline Vec divs(Vec a, float s) { const float inv = 1.0f / s; return mk(a.x * inv, a.y * inv); }
static inline Vec unit(Vec a) { return divs(a, len(a)); }

struct Body {  // what one simulated agent carries into a step
  Vec pos, vel, pref;
  float radius, max_speed, collab;
};

// Neighbour list of `self`: every other agent with distSq < rangeSq, ascending by distSq,
// insertion with strict `<` so ties keep visit (= index) order, capped at max_nb
// (Agent::insertAgentNeighbor).  Visit order is index order, which is what the library's
// kd-tree yields while the agent count does not exceed its leaf size (10).
// g_tie_reverse (tests only, ca_oracle_set_tie_order): exactly tied distSq entries in REVERSE visit order -- the other order
// a kd-tree build of the library could visit them in once an env holds more than MAX_LEAF_SIZE = 10 agents (upstream's tree
// permutes its agent array, and keeps the permutation from step to step).  tests/test_orca_semantics.py uses it to bound what
// the visit order of tied neighbours can change at all on the N = 20 / 50 fixtures.
static int g_tie_reverse = 0;
static inline void neighbours(const Body* a, size_t n, size_t self, float range_sq, size_t max_nb,
                              std::vector<std::pair<float, size_t> >& out) {
  out.clear();
  if (max_nb == 0) return;
  for (size_t j = 0; j < n; ++j) {
    if (j == self) continue;
    const float d2 = dot(sub(a[self].pos, a[j].pos), sub(a[self].pos, a[j].pos));
    if (d2 < range_sq) {
      if (out.size() < max_nb) out.push_back(std::make_pair(d2, j));
      size_t i = out.size() - 1;
      while (i != 0 && (d2 < out[i - 1].first || (g_tie_reverse && d2 == out[i - 1].first))) {
        out[i] = out[i - 1];
        --i;
      }
      out[i] = std::make_pair(d2, j);
      if (out.size() == max_nb) range_sq = out.back().first;
    }
  }
}

// The half-plane agent `me` must respect because of agent `ot` (Agent::computeNewVelocity, agent part).
static inline HalfPlane half_plane(const Body& me, const Body& ot, float inv_horizon, float time_step) {
  const Vec rp = sub(ot.pos, me.pos);
  const Vec rv = sub(me.vel, ot.vel);
  const float d2 = dot(rp, rp);
  const float R = me.radius + ot.radius;
  const float R2 = sq(R);
  HalfPlane h;
  Vec u;
  if (d2 > R2) {
    const Vec w = sub(rv, scl(inv_horizon, rp));
    const float w2 = dot(w, w);
    const float dp1 = dot(w, rp);
    if (dp1 < 0.0f && sq(dp1) > R2 * w2) {  // closest point is on the cut-off disc
      const float wl = std::sqrt(w2);
      const Vec uw = divs(w, wl);
      h.dir = mk(uw.y, -uw.x);
      u = scl(R * inv_horizon - wl, uw);
    } else {  // closest point is on one of the legs
      const float leg = std::sqrt(d2 - R2);
      if (cross(rp, w) > 0.0f) {
        h.dir = divs(mk(rp.x * leg - rp.y * R, rp.x * R + rp.y * leg), d2);
      } else {
        const Vec t = divs(mk(rp.x * leg + rp.y * R, -rp.x * R + rp.y * leg), d2);
        h.dir = mk(-t.x, -t.y);
      }
      const float dp2 = dot(rv, h.dir);
      u = sub(scl(dp2, h.dir), rv);
    }
  } else {  // already overlapping: get out within one time step
    const float inv_dt = 1.0f / time_step;
    const Vec w = sub(rv, scl(inv_dt, rp));
    const float wl = len(w);
    const Vec uw = divs(w, wl);
    h.dir = mk(uw.y, -uw.x);
    u = scl(R * inv_dt - wl, uw);
  }
  h.pt = add(me.vel, scl(me.collab, u));
  return h;
}

// 1-D program on line `k` subject to lines [0,k) and the speed disc (linearProgram1).
static inline bool lp1(const std::vector<HalfPlane>& L, size_t k, float radius, Vec opt, bool dir_opt, Vec& res) {
  const float dp = dot(L[k].pt, L[k].dir);
  const float disc = sq(dp) + sq(radius) - dot(L[k].pt, L[k].pt);
  if (disc < 0.0f) return false;
  const float sd = std::sqrt(disc);
  float t_lo = -dp - sd;
  float t_hi = -dp + sd;
  for (size_t i = 0; i < k; ++i) {
    const float den = cross(L[k].dir, L[i].dir);
    const float num = cross(L[i].dir, sub(L[k].pt, L[i].pt));
    if (std::fabs(den) <= kEps) {
      if (num < 0.0f) return false;
      continue;
    }
    const float t = num / den;
    if (den >= 0.0f) t_hi = std::min(t_hi, t);
    else t_lo = std::max(t_lo, t);
    if (t_lo > t_hi) return false;
  }
  if (dir_opt) {
    if (dot(opt, L[k].dir) > 0.0f) res = add(L[k].pt, scl(t_hi, L[k].dir));
    else res = add(L[k].pt, scl(t_lo, L[k].dir));
  } else {
    const float t = dot(L[k].dir, sub(opt, L[k].pt));
    if (t < t_lo) res = add(L[k].pt, scl(t_lo, L[k].dir));
    else if (t > t_hi) res = add(L[k].pt, scl(t_hi, L[k].dir));
    else res = add(L[k].pt, scl(t, L[k].dir));
  }
  return true;
}

// 2-D program (linearProgram2): returns the index of the first line that made it infeasible, or L.size().
#ifdef ORCA_REF_STATS  // scratch/lp_stats.cpp only: how often the incremental program needs which 1-D programme
struct LpStats {
  long queries, no_violation_at_start, lp1_calls, lp1_not_flagged_at_start, flagged_at_start, lines, infeasible;
  long hist_calls[16];
  long flagged_margin[8], surprise_margin[8], query_surprise_margin[8];  // the same with "nearly violated" lines flagged too
  long has_infeasible_line, fail_at_first_infeasible, fail_elsewhere;  // 1-D programmes that are infeasible on their own
};
static const float kLpStatsMargins[8] = {0.0f, 0.02f, 0.05f, 0.1f, 0.2f, 0.3f, 0.5f, 1.0f};
static LpStats g_lp_stats;
static std::vector<int> g_lp_log;  // per query: number of lines linearProgram3 acted on + 256 x (lines from the failing one on); -1: linearProgram2 was feasible
#endif
static inline size_t lp2(const std::vector<HalfPlane>& L, float radius, Vec opt, bool dir_opt, Vec& res) {
  if (dir_opt) res = scl(radius, opt);  // opt * radius (commutative per component)
  else if (dot(opt, opt) > sq(radius)) res = scl(radius, unit(opt));
  else res = opt;
#ifdef ORCA_REF_STATS
  unsigned v0 = 0;
  long calls = 0, first_inf = -1;
  unsigned vm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!dir_opt) {
    for (int k = 0; k < 8; ++k) {
      for (size_t i = 0; i < L.size(); ++i)
        if (cross(L[i].dir, sub(L[i].pt, res)) > -kLpStatsMargins[k]) vm[k] |= 1u << i;
      g_lp_stats.flagged_margin[k] += __builtin_popcount(vm[k]);
    }
    for (size_t i = 0; i < L.size(); ++i)
      if (cross(L[i].dir, sub(L[i].pt, res)) > 0.0f) v0 |= 1u << i;
    for (size_t i = 0; i < L.size() && first_inf < 0; ++i) {  // first line violated at the start whose 1-D programme is infeasible
      Vec tmp = res;
      if (cross(L[i].dir, sub(L[i].pt, res)) > 0.0f && !lp1(L, i, radius, opt, false, tmp)) first_inf = static_cast<long>(i);
    }
    g_lp_stats.has_infeasible_line += (first_inf >= 0);
    g_lp_stats.queries += 1;
    g_lp_stats.lines += static_cast<long>(L.size());
    g_lp_stats.no_violation_at_start += (v0 == 0);
    g_lp_stats.flagged_at_start += __builtin_popcount(v0);
  }
#endif
  for (size_t i = 0; i < L.size(); ++i) {
    if (cross(L[i].dir, sub(L[i].pt, res)) > 0.0f) {
#ifdef ORCA_REF_STATS
      if (!dir_opt) {
        ++calls;
        g_lp_stats.lp1_calls += 1;
        g_lp_stats.lp1_not_flagged_at_start += !((v0 >> i) & 1u);
        for (int k = 0; k < 8; ++k)
          if (!((vm[k] >> i) & 1u)) { g_lp_stats.surprise_margin[k] += 1; sm[k] = 1; }
      }
#endif
      const Vec keep = res;
      if (!lp1(L, i, radius, opt, dir_opt, res)) {
        res = keep;
#ifdef ORCA_REF_STATS
        if (!dir_opt) {
          g_lp_stats.fail_at_first_infeasible += (static_cast<long>(i) == first_inf);
          g_lp_stats.fail_elsewhere += (static_cast<long>(i) != first_inf);
          g_lp_stats.infeasible += 1;
          g_lp_stats.hist_calls[calls < 15 ? calls : 15] += 1;
          for (int k = 0; k < 8; ++k) g_lp_stats.query_surprise_margin[k] += sm[k];
        }
#endif
        return i;
      }
    }
  }
#ifdef ORCA_REF_STATS
  if (!dir_opt) {
    g_lp_stats.hist_calls[calls < 15 ? calls : 15] += 1;
    for (int k = 0; k < 8; ++k) g_lp_stats.query_surprise_margin[k] += sm[k];
  }
#endif
  return L.size();
}

// Fallback (linearProgram3, no obstacle lines): minimise the maximum penetration.
static inline void lp3(const std::vector<HalfPlane>& L, size_t begin, float radius, Vec& res) {
  float depth = 0.0f;
  std::vector<HalfPlane> P;
#ifdef ORCA_REF_STATS
  g_lp_log.push_back(static_cast<int>((L.size() - begin) << 8));  // (bits 8 ..: lines from the failing one on)
#endif
  for (size_t i = begin; i < L.size(); ++i) {
    if (cross(L[i].dir, sub(L[i].pt, res)) > depth) {
#ifdef ORCA_REF_STATS
      g_lp_log.back() += 1;
      if ((g_lp_log.back() & 0xFF) >= 2) g_lp_log.back() |= 1 << (16 + static_cast<int>(i - begin));  // (bits 16 ..: which later lines acted)
#endif
      P.clear();
      for (size_t j = 0; j < i; ++j) {
        HalfPlane h;
        const float D = cross(L[i].dir, L[j].dir);
        if (std::fabs(D) <= kEps) {
          if (dot(L[i].dir, L[j].dir) > 0.0f) continue;
          h.pt = scl(0.5f, add(L[i].pt, L[j].pt));
        } else {
          h.pt = add(L[i].pt, scl(cross(L[j].dir, sub(L[i].pt, L[j].pt)) / D, L[i].dir));
        }
        h.dir = unit(sub(L[j].dir, L[i].dir));
        P.push_back(h);
      }
      const Vec keep = res;
      if (lp2(P, radius, mk(-L[i].dir.y, L[i].dir.x), true, res) < P.size()) res = keep;
      depth = cross(L[i].dir, sub(L[i].pt, res));
    }
  }
}

// New velocity of agent `self` (Agent::computeNeighbors + computeNewVelocity).
static inline Vec new_velocity(const Body* a, size_t n, size_t self, float neighbor_dist, size_t max_nb,
                               float time_horizon, float time_step) {
  std::vector<std::pair<float, size_t> > nb;
  neighbours(a, n, self, sq(neighbor_dist), max_nb, nb);
  const float inv_h = 1.0f / time_horizon;
  std::vector<HalfPlane> L;
  L.reserve(nb.size());
  for (size_t i = 0; i < nb.size(); ++i) L.push_back(half_plane(a[self], a[nb[i].second], inv_h, time_step));
  Vec v;
  const size_t fail = lp2(L, a[self].max_speed, a[self].pref, false, v);
#ifdef ORCA_REF_STATS
  if (!(fail < L.size())) g_lp_log.push_back(-1);
#endif
  if (fail < L.size()) lp3(L, fail, a[self].max_speed, v);
  return v;
}

// Agent::update position rule: position += newVelocity * timeStep (float).
static inline Vec advance(Vec pos, Vec v, float time_step) { return add(pos, scl(time_step, v)); }

}  // namespace orca_ref

#endif  // ORACLE_ORCA_REF_H_
