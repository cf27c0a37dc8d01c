#!/usr/bin/env python3
"""Convert the reference's GA3C-CADRL TensorFlow checkpoints into the .npz files shipped under
gym_collision_avoidance_amd/data/ga3c_cadrl/ (data conversion, like gen_golden.py's fixture tables; no TensorFlow).

    python oracle/extract_ga3c_weights.py            # the three checkpoints of the reference tree
    python oracle/extract_ga3c_weights.py <checkpoint prefix> <out.npz>

The reader itself is part of the package (envs/policies/GA3C_CADRL/network.py: `read_checkpoint`), so that
`policy.initialize_network(checkpt_dir=<a reference checkpoint directory>)` also works on the raw TF files.  This
script additionally checks that the input-normalisation constants the package hard-codes (= config.py:93-149) occur
byte-for-byte in the graph (`.meta`).
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402

from gym_collision_avoidance_amd.envs.policies.GA3C_CADRL import network  # noqa: E402

REF = "/root/reference/gym_collision_avoidance/envs/policies/GA3C_CADRL/checkpoints"
SHIPPED = [("IROS18", "network_01900000"), ("run-20190727_015942-jzuhlntn", "network_01490000"),
           ("run-20190727_192048-qedrf08y", "network_01900000")]


def convert(prefix, out):
    w = network.read_checkpoint(prefix)
    meta = open(prefix + ".meta", "rb").read()
    assert meta.find(w["input_mean"].tobytes()) >= 0 and meta.find(w["input_std"].tobytes()) >= 0, \
        "normalisation constants not found in %s.meta" % prefix
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, **w)
    print("%s -> %s (%d bytes): %s" % (prefix, out, os.path.getsize(out),
                                       ", ".join("%s%s" % (k, list(v.shape)) for k, v in w.items())))


if __name__ == "__main__":
    if len(sys.argv) > 2:
        convert(sys.argv[1], sys.argv[2])
    else:
        for d, n in SHIPPED:
            convert(os.path.join(REF, d, n), os.path.join(network.DATA_DIR, d, n + ".npz"))
