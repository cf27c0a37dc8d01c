import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import golden_util as gu
from tests.test_gpu_parity import _pair, _upload, F64
from gym_collision_avoidance_amd import _native as nat
N, E = 10, 256
table = gu.fixtures(N)
o, g = _pair(E, N)
g.set_plugins(nat.POL_RVO)
cases = table[np.arange(E) % 500]
o.reset(cases); g.reset(cases)
np.set_printoptions(precision=17)
for t in range(12):
    o.step(); g.step()
    gh = g.state["heading"].cpu().numpy().reshape(-1)
    dh = np.abs(gh - o.s["heading"])
    i = int(dh.argmax())
    print("step", t, "agent", i, "herr", dh[i], "o.head", o.s["heading"][i], "g.head", gh[i],
          "act o", o.actions.reshape(-1,2)[i], "act g", g.actions.cpu().numpy().reshape(-1,2)[i],
          "pos err", abs(g.state["pos_x"].cpu().numpy().reshape(-1)[i]-o.s["pos_x"][i]),
          "vel", o.s["vel_x"][i], o.s["vel_y"][i], "flags", o.s["flags"][i])
