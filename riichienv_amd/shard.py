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


def splitmix64(x: int) -> int:
    """state/wall.rs:83-88."""
    z = (int(x) + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def game_seed(base_seed: int, global_game: int) -> int:
    """Episode seed of global game g when no per-game seed array is given (rmj_create): splitmix64(base_seed + g).

    The reference keys the wall of a game's k-th hand by splitmix64(seed + k) (state/wall.rs:38): with CONSECUTIVE
    episode seeds game g's k-th hand would deal the wall of game g+k's first hand, so the default seeds of a batch are
    decorrelated first (the reference's own default is OS entropy per env, env.rs:107-110).  Explicit per-game seeds
    (`seeds=`) are used as given, like RiichiEnv(seed=...)."""
    return splitmix64((int(base_seed) + int(global_game)) & MASK64)


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


def gather_measurement(dist, rank: int, wall_s: float, steps: float, device=None) -> dict:
    """What a multi-rank bench line reports beyond reduce_measurement, from ONE all_gather of (rank, wall, steps): the ranks that
    took part (`ranks_seen`, in gather order - a proof that the reduction saw N processes), each rank's own rate
    (`per_rank_value`), and the whole-job figures MAX(wall) / SUM(steps)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {"wall": float(wall_s), "steps": float(steps), "ranks_seen": [int(rank)], "per_rank_value": [float(steps) / max(float(wall_s), 1e-12)],
                "per_rank_wall_s": [float(wall_s)]}
    import torch

    mine = torch.tensor([float(rank), float(wall_s), float(steps)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    rows = [[float(x) for x in p.cpu()] for p in parts]
    return {"wall": max(r[1] for r in rows), "steps": sum(r[2] for r in rows), "ranks_seen": [int(r[0]) for r in rows],
            "per_rank_value": [r[2] / max(r[1], 1e-12) for r in rows], "per_rank_wall_s": [r[1] for r in rows]}
