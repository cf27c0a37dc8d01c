#!/usr/bin/env python3
"""scratch/lp_stats.py -- statistics of the incremental ORCA programme on the benchmark workload (CPU oracle built with
-DORCA_REF_STATS, see scratch/lp_stats.cpp): how many 1-D programmes linearProgram2 really calls, how many lines are
violated at its starting point, and how often a line that was not flagged there is violated later ("surprise"), for a
range of flagging margins."""
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ca_oracle as orc
orc.LIB_PATH = '/tmp/libca_oracle_stats.so'
orc._lib = None
table = np.load('gym_collision_avoidance_amd/data/test_cases.npz')['n10']
E = 256
o = orc.Oracle(orc.default_params(E, 10))
o.s['policy'][:] = orc.POL_RVO
o.reset(table[np.arange(E) % 500])
o.rollout(table, 400)   # steady state
lib = C.CDLL('/tmp/libca_oracle_stats.so')
buf = (C.c_long * 50)()
lib.lp_stats(buf); a0 = np.array(buf[:])
o.rollout(table, 400)
lib.lp_stats(buf); a = np.array(buf[:]) - a0
q = a[0]
print("queries", q, "= %.3f of the agent-steps" % (q / (E * 10 * 400)))
print("no violation at start: %.3f" % (a[1] / q))
print("lines/query %.2f  flagged at start/query %.2f  lp1 calls/query %.2f  (not flagged at start: %.3f/query)" % (a[5] / q, a[4] / q, a[2] / q, a[3] / q))
print("infeasible %.4f" % (a[6] / q))
print("hist of lp1 calls per query:", np.round(a[7:23] / q, 3))
for k, m in enumerate([0.0, 0.02, 0.05, 0.1, 0.2, 0.3, 0.5, 1.0]):
    print("margin %.2f: flagged/query %.2f  surprise lines/query %.4f  queries with a surprise %.4f" % (m, a[23 + k] / q, a[31 + k] / q, a[39 + k] / q))
print("queries with a 1-D programme that is infeasible on its own: %.4f; linearProgram2 fails AT the first such line: %.4f, at another line: %.4f" % (a[47] / q, a[48] / q, a[49] / q))
