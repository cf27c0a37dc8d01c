"""GA3C-CADRL network: the action table, the weight files and a TensorFlow-free checkpoint reader
(reference: gym_collision_avoidance/envs/policies/GA3C_CADRL/network.py).

The reference restores a TF1 graph and runs it in a tf.Session (network.py:43-74); here the graph is the fused
matrix-core kernel of csrc/cagpu_ga3c.inc and this module only gets its weights to the device:

  * `Actions`               -- the 11 discrete actions (network.py:7-16);
  * `read_checkpoint(p)`    -- reads a TF "V2" checkpoint (`p.index` + `p.data-00000-of-00001`) without TensorFlow, so the
                               reference's own checkpoint directories keep working;
  * `load_weights(p)`       -- `p.npz` (the shipped conversions under data/ga3c_cadrl/) or a TF checkpoint prefix.
"""
import os
import struct

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))),
                        "data", "ga3c_cadrl")
NUM_OTHER_AGENTS, INPUT_LENGTH = 19, 138

# variable name in the checkpoint -> key used by core.BatchedSim.load_ga3c / include/cagpu.h CaNet
VARIABLES = {"rnn/lstm_cell/kernel": "lstm_kernel", "rnn/lstm_cell/bias": "lstm_bias",
             "layer1/kernel": "layer1_kernel", "layer1/bias": "layer1_bias",
             "layer2/kernel": "layer2_kernel", "layer2/bias": "layer2_bias",
             "fullyconnected1/kernel": "fc1_kernel", "fullyconnected1/bias": "fc1_bias",
             "logits_p/kernel": "logits_p_kernel", "logits_p/bias": "logits_p_bias",
             "logits_v/kernel": "logits_v_kernel", "logits_v/bias": "logits_v_bias"}


class Actions(object):
    """[speed factor, delta heading]: full speed x 5 headings, half speed x 3, stopped x 3 (network.py:7-16)."""

    def __init__(self):
        s6, s12 = np.pi / 6, np.pi / 12
        self.actions = np.array([[1, -s6], [1, -s12], [1, 0], [1, s12], [1, s6], [0.5, -s6], [0.5, 0], [0.5, s6],
                                 [0, -s6], [0, 0], [0, s6]], dtype=np.float64)
        self.num_actions = len(self.actions)


def input_normalisation():
    """`Const` / `Const_1` of the graph = avg / std of Config.STATE_INFO_DICT (config.py:93-149) for
    [num_other_agents, dist_to_goal, heading_ego_frame, pref_speed, radius] + 19 x other_agents_states."""
    mean = np.array([0, 0, 0, 1, 0.5] + [0, 0, 0, 0, 0.5, 0, 1] * NUM_OTHER_AGENTS, dtype=np.float32)
    std = np.array([1, 5, 3.14, 1, 1] + [5, 5, 1, 1, 1, 5, 1] * NUM_OTHER_AGENTS, dtype=np.float32)
    return mean, std


# ---------------------------------------------------------------- TF checkpoint ("tensor bundle") reader
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _block(buf, off, size):
    """(key, value) entries of one table block: shared-prefix key compression, restart array skipped"""
    if buf[off + size] != 0:
        raise ValueError("compressed checkpoint index blocks are not supported")
    blk = buf[off:off + size]
    end = len(blk) - 4 - 4 * struct.unpack("<I", blk[-4:])[0]
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(blk, pos)
        fresh, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + bytes(blk[pos:pos + fresh])
        pos += fresh
        yield key, bytes(blk[pos:pos + vlen])
        pos += vlen


def _fields(buf):
    """(field number, value) pairs of a protobuf message (varint, 32/64-bit and length-delimited wire types)"""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        wire = tag & 7
        if wire == 0:
            v, pos = _varint(buf, pos)
        elif wire == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            v = struct.unpack("<I", buf[pos:pos + 4])[0]
            pos += 4
        elif wire == 1:
            v = struct.unpack("<Q", buf[pos:pos + 8])[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        yield tag >> 3, v


def read_checkpoint_index(path):
    """`<prefix>.index` -> {variable name: {dtype, shape, shard, offset, size}} (BundleEntryProto per key)"""
    buf = open(path, "rb").read()
    footer = buf[-48:]
    if footer[-8:] != struct.pack("<Q", 0xdb4775248b80fb57):
        raise ValueError("%s is not a TensorFlow checkpoint index" % path)
    pos = 0
    _, pos = _varint(footer, pos)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isz, pos = _varint(footer, pos)
    out = {}
    for _, handle in _block(buf, ioff, isz):
        boff, p = _varint(handle, 0)
        bsz, p = _varint(handle, p)
        for key, val in _block(buf, boff, bsz):
            if not key:
                continue  # the bundle header
            e = {"dtype": 0, "shape": [], "shard": 0, "offset": 0, "size": 0}
            for f, v in _fields(val):
                if f == 1:
                    e["dtype"] = v
                elif f == 2:
                    e["shape"] = [dict(_fields(d)).get(1, 0) for f2, d in _fields(v) if f2 == 2]
                elif f == 3:
                    e["shard"] = v
                elif f == 4:
                    e["offset"] = v
                elif f == 5:
                    e["size"] = v
            name = key.decode()
            out[name[:-2] if name.endswith(":0") else name] = e
    return out


def read_checkpoint(prefix):
    """TF checkpoint prefix -> {short name: float32 array} for the inference variables (optimizer slots dropped)"""
    index = read_checkpoint_index(prefix + ".index")
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    out = {}
    for name, short in VARIABLES.items():
        e = index[name]
        if e["dtype"] != 1 or e["shard"] != 0:
            raise ValueError("variable %s: expected a float32 tensor in shard 0, got %r" % (name, e))
        out[short] = np.frombuffer(data, dtype="<f4", count=e["size"] // 4, offset=e["offset"]).reshape(e["shape"]).copy()
    out["input_mean"], out["input_std"] = input_normalisation()
    return out


_cache = {}


def load_weights(path):
    """`path`: '<...>.npz', or a prefix that has '<prefix>.npz' or '<prefix>.index' next to it.  Cached per path: a
    500-case suite builds thousands of policy objects that all name the same checkpoint."""
    if path not in _cache:
        _cache[path] = _load_weights(path)
    return _cache[path]


def _load_weights(path):
    if path.endswith(".npz") or os.path.exists(path + ".npz"):
        with np.load(path if path.endswith(".npz") else path + ".npz") as z:
            return {k: z[k] for k in z.files}
    if os.path.exists(path + ".index"):
        return read_checkpoint(path)
    raise FileNotFoundError("no GA3C-CADRL weights at %s(.npz|.index)" % path)
