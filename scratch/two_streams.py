import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gym_collision_avoidance_amd import _native as nat, core
N = 10
table = np.load("gym_collision_avoidance_amd/data/test_cases.npz")["n10"]
def make(E, off):
    sim = core.BatchedSim(core.make_params(E, N))
    sim.set_plugins(nat.POL_RVO); sim.set_fixture_table(table, env_id_offset=off, case_stride=4096); sim.reset_from_table()
    return sim
for parts in (1, 2, 4):
    E = 4096 // parts
    sims = [make(E, i * E) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def run(n):
        for _ in range(n):
            for s, st in zip(sims, streams):
                with torch.cuda.stream(st):
                    s.step()
    run(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("parts", parts, "us per 4096-env step: %.2f" % (dt / 2000 * 1e6), "M agent-steps/s: %.0f" % (4096 * N * 2000 / dt / 1e6))
