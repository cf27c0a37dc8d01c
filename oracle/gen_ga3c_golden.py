#!/usr/bin/env python3
"""oracle/gen_ga3c_golden.py -- TEST INFRASTRUCTURE.  Golden logits of the GA3C-CADRL network from the checkpoint's OWN graph.

The reference runs the TF1 graph stored in `network_01900000.meta` (GA3C_CADRL/network.py:43-74: import_meta_graph + restore;
:24-41: sess.run('Softmax:0', {'X:0': x})).  TensorFlow is not installed here, so oracle/tf_graph_exec.py decodes that
GraphDef (protobuf wire format) and executes the nodes it finds, with the variables of the checkpoint's data file.  This
script feeds it 1 024 observation rows -- 768 taken from 20-agent episodes the GA3C-CADRL policy itself drives (the CPU
oracle, closest_last ordering, K = 19: the rows the policy really sees, from 0 to 19 other agents) and 256 random ones --
and commits X, logits_p/BiasAdd and Softmax as tests/golden/ga3c_graph.npz.  oracle/ga3c_ref.py (the hand restatement) and
the HIP kernel (cagpu_ga3c) are held to these vectors: a misreading of the .meta shared by both would show up here.

Only works in the build container (/root/reference holds the checkpoint); the output is committed.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
REF = os.environ.get("CA_REFERENCE_ROOT", "/root/reference")
CKPT = os.path.join(REF, "gym_collision_avoidance", "envs", "policies", "GA3C_CADRL", "checkpoints")
SHIPPED = {"IROS18": "network_01900000", "run-20190727_015942-jzuhlntn": "network_01490000",
           "run-20190727_192048-qedrf08y": "network_01900000"}


def episode_rows(n_rows, seed=0):
    """observation rows [n, 139] float32 of GA3C-CADRL agents mid-episode (oracle, N = 20, K = 19, closest_last)"""
    from oracle import ca_oracle as orc
    table = np.load(os.path.join(REPO, "gym_collision_avoidance_amd", "data", "test_cases.npz"))["n20"]
    E, N = 24, 20
    o = orc.Oracle(orc.default_params(E, N, max_obs=19, sort_mode=orc.SORT_CLOSEST_LAST))
    o.set_policies(orc.POL_GA3C_CADRL)
    o.reset(table[:E])
    rows, rng = [], np.random.default_rng(seed)
    for t in range(60):
        if t % 4 == 0:
            live = (o.view("flags") & orc.DONE) == 0
            r = o.obs.reshape(-1, o.W)[live.reshape(-1)].astype(np.float32)
            rows.append(r[rng.permutation(len(r))[:n_rows // 12]])
        o.step()
    rows = np.concatenate(rows)[:n_rows]
    # the policy also meets short neighbour lists (small envs, clipped sensors): truncate some rows to 0 .. 6 others
    for i in range(0, len(rows), 5):
        k = int(rng.integers(0, 7))
        rows[i, 1] = min(rows[i, 1], k)
        rows[i, 6 + 7 * k:] = 0.0
    return rows


def random_rows(n, seed=1):
    rng = np.random.default_rng(seed)
    K = 19
    obs = np.zeros((n, 6 + 7 * K), np.float32)
    num = rng.integers(0, K + 1, size=n)
    obs[:, 1] = num
    obs[:, 2] = rng.uniform(0.1, 12.0, n)
    obs[:, 3] = rng.uniform(-np.pi, np.pi, n)
    obs[:, 4] = rng.uniform(0.5, 1.5, n)
    obs[:, 5] = rng.uniform(0.2, 0.8, n)
    oth = np.stack([rng.uniform(-8, 8, (n, K)), rng.uniform(-8, 8, (n, K)), rng.uniform(-1.5, 1.5, (n, K)),
                    rng.uniform(-1.5, 1.5, (n, K)), rng.uniform(0.2, 0.8, (n, K)), rng.uniform(0.4, 1.6, (n, K)),
                    rng.uniform(0.0, 10.0, (n, K))], axis=-1).astype(np.float32)
    oth *= (np.arange(K)[None, :] < num[:, None])[..., None]
    obs[:, 6:] = oth.reshape(n, 7 * K)
    return obs


def main():
    from oracle import tf_graph_exec as tg
    obs = np.concatenate([episode_rows(768), random_rows(256)]).astype(np.float32)
    assert obs.shape == (1024, 139)
    x = obs[:, 1:].copy()      # GA3CCADRLPolicy.py:69-75: every state but is_learning; 138 = the placeholder's width
    out = {"obs": obs, "X": x}
    for run, name in SHIPPED.items():
        (logits, softmax), ex = tg.predict(os.path.join(CKPT, run, name), x)
        key = run.replace("-", "_")
        out["logits_" + key], out["softmax_" + key] = logits.astype(np.float32), softmax.astype(np.float32)
        print("%s/%s: logits %s, |logits| max %.2f, ops executed: %s" % (run, name, logits.shape, np.abs(logits).max(),
                                                                          " ".join(sorted(ex.executed))))
    out["ops_executed"] = np.array(sorted(ex.executed))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "ga3c_graph.npz"), **out)


if __name__ == "__main__":
    main()
