// scratch/lp_stats.cpp -- how many 1-D programmes (linearProgram1) does the incremental ORCA programme really need on
// the benchmark workload?  Builds the CPU oracle with -DORCA_REF_STATS and exports the counters.
//   g++ -O2 -fPIC -std=c++14 -ffp-contract=off -DORCA_REF_STATS -shared -o /tmp/libca_oracle_stats.so scratch/lp_stats.cpp
#include "../oracle/ca_oracle.cpp"
extern "C" void lp_stats(long* out) {
  const orca_ref::LpStats& s = orca_ref::g_lp_stats;
  out[0] = s.queries; out[1] = s.no_violation_at_start; out[2] = s.lp1_calls; out[3] = s.lp1_not_flagged_at_start;
  out[4] = s.flagged_at_start; out[5] = s.lines; out[6] = s.infeasible;
  for (int i = 0; i < 16; ++i) out[7 + i] = s.hist_calls[i];
  out[47] = s.has_infeasible_line; out[48] = s.fail_at_first_infeasible; out[49] = s.fail_elsewhere;
  for (int i = 0; i < 8; ++i) { out[23 + i] = s.flagged_margin[i]; out[31 + i] = s.surprise_margin[i]; out[39 + i] = s.query_surprise_margin[i]; }
}
extern "C" long lp_log(int* out, long cap, int clear) {
  const long n = static_cast<long>(orca_ref::g_lp_log.size());
  for (long i = 0; i < n && i < cap; ++i) out[i] = orca_ref::g_lp_log[i];
  if (clear) orca_ref::g_lp_log.clear();
  return n;
}
