"""Sharding of a batch of independent games over ranks (one process per GPU).

Games are independent objects (the reference runs one `RiichiEnv` per game, `riichienv-python/src/env.rs:74-118`), so
the batch is partitioned by global game index with no data-path collective: rank r owns global games
[r*B, (r+1)*B).  Wall seeds and policy keys are functions of the GLOBAL index (`RmjConfig.game_offset`), therefore a
game's trajectory does not depend on how many ranks the batch was split over.  `torch.distributed` is used only for
the barrier and to combine the per-rank counters of a measurement.
"""
from __future__ import annotations

MASK64 = 0xFFFFFFFFFFFFFFFF


def shard_offset(rank: int, games_per_rank: int) -> int:
    """Global index of the first game owned by `rank` (weak scaling: every rank owns `games_per_rank` games)."""
    if rank < 0 or games_per_rank <= 0:
        raise ValueError("rank must be >= 0 and games_per_rank > 0")
    return rank * games_per_rank


def game_seed(base_seed: int, global_game: int) -> int:
    """Seed of global game g when no per-game seed array is given (rmj_create: base_seed + game_offset + local)."""
    return (int(base_seed) + int(global_game)) & MASK64


def owner_of(global_game: int, games_per_rank: int) -> tuple[int, int]:
    """(rank, local index) of a global game."""
    return global_game // games_per_rank, global_game % games_per_rank


def reduce_measurement(dist, wall_s: float, steps: float, device=None) -> tuple[float, float]:
    """Whole-job (wall, steps) of a timed region: MAX of the ranks' wall time, SUM of the env.steps they made.
    `dist` is torch.distributed (initialised, any backend) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(wall_s), float(steps)
    import torch

    t = torch.tensor([wall_s, steps], dtype=torch.float64, device=device)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(tmax[0]), float(t[1])
