import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import golden_util as gu
from tests.test_gpu_parity import _pair, _upload, F64
from gym_collision_avoidance_amd import _native as nat
from oracle import ca_oracle as orc
N, E = 10, 256
table = gu.fixtures(N)
o, g = _pair(E, N)
g.set_plugins(nat.POL_RVO)
cases = table[np.arange(E) % 500]
o.reset(cases); g.reset(cases)
for n in F64:
    d = np.abs(g.state[n].cpu().numpy().reshape(-1) - o.s[n]).max()
    print("reset", n, d)
mx = {}
for t in range(60):
    _upload(o, g)
    o.step(); g.step()
    for n in F64:
        d = np.abs(g.state[n].cpu().numpy().reshape(-1) - o.s[n])
        if d.max() > mx.get(n, (0,))[0]:
            mx[n] = (d.max(), t, int(d.argmax()))
    da = np.abs(g.actions.cpu().numpy().reshape(-1, 2) - o.actions.reshape(-1, 2))
    if da.max() > mx.get("act", (0,))[0]:
        mx["act"] = (da.max(), t, int(da.argmax()))
for k, v in mx.items():
    print(k, v)
# free-running divergence growth
o.reset(cases); g.reset(cases)
for t in range(40):
    o.step(); g.step()
    d = np.abs(g.state["pos_x"].cpu().numpy().reshape(-1) - o.s["pos_x"])
    dh = np.abs(g.state["heading"].cpu().numpy().reshape(-1) - o.s["heading"])
    da = np.abs(g.actions.cpu().numpy().reshape(-1, 2) - o.actions.reshape(-1, 2))
    print(t, "pos", d.max(), "head", dh.max(), "act", da.max(0), "n_bad", (d > 1e-9).sum())
