#!/usr/bin/env python3
"""oracle/gen_presets.py -- TEST / DATA INFRASTRUCTURE.  Records the reference's hand-written scenario presets
(`preset_testCases(n)` without full_test_suite, gym_collision_avoidance/envs/test_cases.py:626-897) by IMPORTING the
unmodified reference and calling the function, and writes them as data to gym_collision_avoidance_amd/data/presets.npz
(key "n<agents>_<index>" = float64 [N, 6] = px, py, gx, gy, pref_speed, radius) -- the same way oracle/gen_golden.py
converts the 500-case pickles.  Only runs in the build container (the reference does not travel); the output is committed.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("CA_REFERENCE_ROOT", "/root/reference")


def main():
    os.environ["GYM_CONFIG_PATH"] = os.path.join(HERE, "golden_configs.py")
    os.environ["GYM_CONFIG_CLASS"] = "Bench10"
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path[:0] = [os.path.join(HERE, "stubs"), os.path.join(HERE, "_build"), REF]
    import warnings
    warnings.filterwarnings("ignore")
    from gym_collision_avoidance.envs import test_cases as tc
    out = {}
    for n in range(1, 21):
        try:
            cases = tc.preset_testCases(n)
        except Exception:  # noqa: BLE001 -- the reference defines presets for some agent counts only
            continue
        for i, c in enumerate(cases):
            out["n%d_%d" % (n, i)] = np.asarray(c, dtype=np.float64)
        print("n = %d: %d presets of %s agents" % (n, len(cases), sorted({np.asarray(c).shape[0] for c in cases})))
    np.savez_compressed(os.path.join(REPO, "gym_collision_avoidance_amd", "data", "presets.npz"), **out)
    # golden vectors for the seeded builders (tests/test_host_logic.py::test_scenario_builders_match_the_reference)
    gold = {}
    np.random.seed(3)
    gold["huge_seed3_12_10"] = tc.make_testcase_huge(1, 12, 10)
    np.random.seed(0)
    agents = tc.cadrl_test_case_to_agents(tc.preset_testCases(6)[0], policies="noncoop")
    tc.formation(agents, "C")
    gold["formation_C_seed0"] = np.array([[*a.pos_global_frame, *a.goal_global_frame, a.heading_global_frame] for a in agents])
    crazy = tc.get_testcase_crazy("noncoop")
    gold["crazy"] = np.array([[*a.pos_global_frame, *a.goal_global_frame, a.pref_speed, a.radius, a.heading_global_frame]
                              for a in crazy])
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "builders.npz"), **gold)


if __name__ == "__main__":
    main()
