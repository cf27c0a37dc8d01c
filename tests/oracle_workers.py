"""Module-level (picklable) workers that run the CPU oracle on a slice of a batch in their own process: the oracle with the
numpy GA3C-CADRL network is the slow side of the free-running outcome comparison (tests/test_gpu_bench_geometry.py)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def ga3c_episodes(args):
    """cases [E, N, 6] -> (final flags [E*N], over [E] bool, episode length [E]) of E GA3C-CADRL scenes run free for at most T
    steps (no auto-reset; K = 19, closest_last: BASELINE configs[2])"""
    cases, T = args
    from oracle import ca_oracle as orc
    E, N = cases.shape[:2]
    o = orc.Oracle(orc.default_params(E, N, max_obs=19, sort_mode=orc.SORT_CLOSEST_LAST))
    o.set_policies(orc.POL_GA3C_CADRL)
    o.reset(cases)
    over, length = np.zeros(E, bool), np.zeros(E)
    for _ in range(T):
        o.step()
        length += ~over
        over |= o.game_over.astype(bool)
        if over.all():
            break
    return o.s["flags"].copy(), over, length
