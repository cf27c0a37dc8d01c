"""RCCL, once, before an 8-GPU node exists for this repository: the N > 1 code path of bench.py and sharding.py on a process
group of ONE rank with backend "nccl" (= RCCL on ROCm) on the one MI355X of the test box -- init_process_group(device_id=),
barrier, the float64 all_reduce(MAX) of the block table, reduce_episode_stats' all_reduce(SUM), the all_gather of per-rank
times, broadcast, destroy.  Each case runs in its own process under a time-out (a hung rendezvous must not take the suite
down).  What this cannot show: xGMI traffic between two devices -- the driver's 8-GPU run does."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_ENV_DROP = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GYM_CONFIG_CLASS", "GYM_CONFIG_PATH")

_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from gym_collision_avoidance_amd.sharding import gather_episode_stats, reduce_episode_stats, shard_env_ids
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29713")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.barrier()
# the simulator's own counters, after real episodes
import bench
sim, table, N, K = bench.build_workload("rvo10", 512, dev, rank=0, world=1)
sim.rollout(400)
local = sim.episode_stats()
assert float(local[0]) > 256, local
tot = reduce_episode_stats(local, force=True)            # all_reduce(SUM) of float64[8] on RCCL
assert tot.data_ptr() != local.data_ptr() and torch.equal(tot, local), (tot, local)
assert torch.equal(reduce_episode_stats(local), local)  # (world 1 without force: untouched, no collective)
allg = gather_episode_stats(local)
assert allg.shape == (1, 8) and torch.equal(allg[0], local)
outs = [torch.empty_like(local)]
dist.all_gather(outs, local)                             # the per-rank table of the bench line
assert torch.equal(outs[0], local)
blk = torch.tensor([[1.5, 2.5], [0.25, 7.0]], dtype=torch.float64, device=dev)
ref = blk.clone()
dist.all_reduce(blk, op=dist.ReduceOp.MAX)               # the block table's MAX over ranks
assert torch.equal(blk, ref)
nb = torch.tensor([17], dtype=torch.int64, device=dev)
dist.broadcast(nb, 0)
assert int(nb.item()) == 17
assert shard_env_ids(0, 1, 512) == (0, 512)
dist.barrier()
torch.cuda.synchronize(dev)
dist.destroy_process_group()
print("rccl one-rank ok")
"""


def _env():
    return {k: v for k, v in os.environ.items() if k not in _ENV_DROP}


def test_sharding_collectives_on_a_one_rank_rccl_group(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % REPO)
    r = subprocess.run([sys.executable, str(script)], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "rccl one-rank ok" in out, out[-3000:]


def test_bench_force_dist_drives_the_multi_gpu_path_on_rccl():
    """`bench.py --force-dist`: N = 1, but the process group is initialised (nccl) and every collective of the N > 1 path
    runs -- the line says so (`distributed`, `ranks_seen`, `stats_allreduce_us`) and its statistics equal the local ones
    (bench.py exits non-zero otherwise)"""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--force-dist",
           "--no-cpu-baseline", "--no-extras", "--min-timed-seconds", "0.05"]
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["distributed"] == {"backend": "nccl", "world_size": 1} and d["n_gpus"] == 1 and d["ranks_seen"] == 1
    assert d["stats_allreduce_us"] is not None and d["stats_allreduce_us"] > 0 and len(d["per_rank_event_ms_per_step"]) == 1
    assert d["config"]["launch_mode"] == "lookahead-20" and d["roofline"]["steps_per_launch"] == 20
    assert d["metric"] == "agent-steps/sec at 4096 envs x 10 agents (RVO)" and d["episode_stats"]["episodes"] > 0
    assert d["timed_blocks"]["blocks"] >= 2 and d["value"] > 0


def _check_two_rank_line(d, backend):
    assert d["distributed"] == {"backend": backend, "world_size": 2} and d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert len(d["per_rank_event_ms_per_step"]) == 2 and all(t > 0 for t in d["per_rank_event_ms_per_step"])
    assert d["stats_allreduce_us"] is not None and d["stats_allreduce_us"] > 0
    per = d["episode_stats_per_rank"]
    assert len(per) == 2 and per[0] != per[1], per          # two different shards ...
    tot = [d["episode_stats"][k] for k in ("episodes", "collision_episodes", "all_at_goal_episodes", "stuck_episodes", "sum_steps",
                                           "sum_total_reward", "sum_time_to_goal", "sum_extra_time_to_goal")]
    for q in range(8):                                      # ... whose counters sum to the all-reduced ones
        assert abs(per[0][q] + per[1][q] - tot[q]) <= 1e-9 * max(1.0, abs(tot[q])), (q, per, tot)
    assert d["config"]["envs_per_gpu"] == 4096 and "8192 envs in all" in d["config"]["workload"]
    assert d["scaling"] == "weak" and d["value"] > 0 and d["provenance"]["lib_sha256"]


def test_two_ranks_sharing_the_one_device_over_gloo():
    """the two-rank logic of bench.py on the 1-GPU box: `--gpus 2 --share-device --backend gloo` (both ranks on cuda:0; not a
    scaling number) -- the same assertions the RCCL form below makes on a multi-GPU node"""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
           "--no-extras", "--min-timed-seconds", "0.05", "--share-device", "--backend", "gloo"]
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    _check_two_rank_line(json.loads(lines[0]), "gloo")


def test_two_ranks_on_two_devices_over_rccl():
    """Arms itself wherever the box has >= 2 devices (the 1-GPU lease of this repository's rounds has one: skipped there, with
    the reason logged): `bench.py --gpus 2` starts two `nccl` ranks on two devices exactly as the driver's launcher form does
    and the line must show both of them -- ranks_seen, one device time per rank, the all-gathered shard counters summing to the
    all-reduced episode statistics, two different shards (rank r owns global envs [r E, (r + 1) E): sharding.py:24-38)."""
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("two-rank RCCL test needs >= 2 visible devices, this box has %d (it arms itself on a multi-GPU node)" % n_dev)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
           "--no-extras", "--min-timed-seconds", "0.05"]
    env = _env()
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    _check_two_rank_line(json.loads(lines[0]), "nccl")
