"""Env-axis sharding across the GPUs of a node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests).

Envs never interact (nothing in collision_avoidance_env.py:156-575 couples two env instances), so the step path has
NO collective: rank r owns the contiguous global env ids [r*E_local, (r+1)*E_local).  The only exchange is the
episode-statistics record (experiments/src/env_utils.py:56-87 reduced to 8 counters): one all-reduce(sum) of a
float64[8] per reporting interval -- latency-bound (64 bytes), kept off the step path.
"""
import torch
import torch.distributed as dist


def shard_env_ids(rank, world_size, envs_per_rank):
    """-> (env_id_offset, case_stride) for CaAutoReset: global id of this shard's env 0, and the global number of
    envs (env g's k-th reset loads fixture case (g + k*stride) % n_cases, so shards replay disjoint case streams
    exactly as one big single-GPU batch would)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    return rank * envs_per_rank, world_size * envs_per_rank


def reduce_episode_stats(stats, world_size=None, force=False):
    """all-reduce(sum) of the per-shard float64[8] counters (core.STAT_NAMES).  No-op for a single process -- unless
    `force` and a process group exists (a one-rank group: the collective still runs, on RCCL for backend nccl)."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world_size > 1 or (force and dist.is_available() and dist.is_initialized()):
        stats = stats.clone()
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats


def gather_episode_stats(stats):
    """all-gather of the per-shard counters -> [world, 8] (per-GPU breakdown for reports)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return stats.unsqueeze(0)
    out = [torch.empty_like(stats) for _ in range(dist.get_world_size())]
    dist.all_gather(out, stats)
    return torch.stack(out)
