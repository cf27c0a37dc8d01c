"""tests/golden/ga3c_graph.npz: logits of the GA3C-CADRL network produced by EXECUTING the checkpoint's own TF1 graph
(oracle/tf_graph_exec.py decodes `network_*.meta` and runs its nodes; oracle/gen_ga3c_golden.py) for 1 024 observation
rows and all three shipped checkpoints.  The hand restatement oracle/ga3c_ref.py (CPU, here) and the HIP kernel cagpu_ga3c
(GPU, below) are both held to them: a misreading of the graph shared by the two would not survive."""
import os

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden", "ga3c_graph.npz")
RUNS = {"IROS18": "network_01900000", "run-20190727_015942-jzuhlntn": "network_01490000",
        "run-20190727_192048-qedrf08y": "network_01900000"}


def _weights(run):
    return os.path.join(REPO, "gym_collision_avoidance_amd", "data", "ga3c_cadrl", run, RUNS[run] + ".npz")


def _check(got, want):
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=3e-4)     # |logits| reach 90: float32 summation order
    srt = np.sort(want, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-3
    assert clear.mean() > 0.95
    assert np.array_equal(np.argmax(got, axis=1)[clear], np.argmax(want, axis=1)[clear])


@pytest.mark.parametrize("run", list(RUNS))
def test_numpy_restatement_matches_the_checkpoints_own_graph(run):
    from oracle.ga3c_ref import GA3CNet
    z = np.load(GOLD)
    assert {"MatMul", "Split", "Sigmoid", "Tanh", "Select", "TensorArrayReadV3", "Enter", "Exit", "Softmax"} <= set(z["ops_executed"])
    net = GA3CNet(_weights(run))
    x = net.policy_vector(z["obs"])
    assert np.array_equal(x, z["X"])
    key = run.replace("-", "_")
    _check(net.logits(x), z["logits_" + key])
    np.testing.assert_allclose(net.predict_p(x), z["softmax_" + key], rtol=0, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("run", list(RUNS))
def test_hip_network_matches_the_checkpoints_own_graph(run):
    torch = pytest.importorskip("torch")
    from gym_collision_avoidance_amd import _native as nat, core
    z = np.load(GOLD)
    E, N, K = 64, 16, 19
    g = core.BatchedSim(core.make_params(E, N, max_obs=K, sort_mode=nat.SORT_CLOSEST_LAST))
    g.set_plugins(nat.POL_GA3C_CADRL)
    g.obs.copy_(torch.from_numpy(z["obs"].reshape(E, N, 6 + 7 * K)))
    g.load_ga3c(_weights(run), keep_logits=True)
    ext = g.ga3c()
    torch.cuda.synchronize()
    key = run.replace("-", "_")
    want = z["logits_" + key]
    got = g.ga3c_logits.cpu().numpy().reshape(-1, 11)
    _check(got, want)
    assert np.array_equal(ext.cpu().numpy().reshape(-1, 2)[:, 0], np.argmax(got, axis=1))
