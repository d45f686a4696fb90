"""N>1 path on CPU: two gloo ranks each own one shard of a batch of games (riichienv_amd/shard.py), run the
oracle on it with global-index seeds / policy keys, and the union must equal the one-rank run over the whole batch
(a game's trajectory does not depend on the number of ranks); the measurement reduction is MAX(wall), SUM(steps)."""
import os
import socket
import zlib

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle
from riichienv_amd import abi, shard

GAMES_PER_RANK = 6
STEPS = 400
BASE_SEED = 77
POLICY_SEED = 0xC0FFEE


def _run_shard(rank, games_per_rank, mode):
    """K lock-step env.steps of this rank's games; returns per-game (global index, step_count, scores, log digest)."""
    off = shard.shard_offset(rank, games_per_rank)
    out = []
    for local in range(games_per_rank):
        g = off + local
        assert shard.owner_of(g, games_per_rank) == (rank, local)
        game = oracle.Game(game_mode=mode, seed=shard.game_seed(BASE_SEED, g))
        game.reset()
        made = 0
        for _ in range(STEPS):
            _, _, done = game.status()
            if done:
                game.reset()
                continue
            game.step(game.random_actions(POLICY_SEED, g))
            made += 1
        v = game.peek()
        log = game.log()
        out.append((g, made, game.step_count, tuple(int(v.players[p].score) for p in range(4)), len(log), zlib.crc32("\n".join(log).encode())))
    return out


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = _run_shard(rank, GAMES_PER_RANK, mode)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        wall, steps = shard.reduce_measurement(dist, 1.0 + rank, float(sum(m[1] for m in mine)))
        gm = shard.gather_measurement(dist, rank, 1.0 + rank, float(sum(m[1] for m in mine)))
        dist.barrier()
        if rank == 0:
            q.put((gathered, wall, steps, gm))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", [2, 5])
def test_two_rank_shards_equal_one_rank(mode):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    gathered, wall, steps, gm = q.get()
    two = [x for part in gathered for x in part]
    one = _run_shard(0, 2 * GAMES_PER_RANK, mode)        # the whole batch on one rank
    assert two == one                                    # incl. the CRC of every game's MJAI log
    assert [x[0] for x in two] == list(range(2 * GAMES_PER_RANK))
    assert wall == 2.0                                   # MAX over ranks
    assert steps == float(sum(x[1] for x in one))        # SUM over ranks
    assert steps > 0
    # the fields a multi-rank bench line carries to show that N ranks took part (bench.py: ranks_seen, per_rank_value)
    assert gm["ranks_seen"] == [0, 1] and gm["wall"] == 2.0 and gm["steps"] == steps
    per = [float(sum(x[1] for x in part)) for part in gathered]
    assert gm["per_rank_value"] == [per[0] / 1.0, per[1] / 2.0] and gm["per_rank_wall_s"] == [1.0, 2.0]


def test_shard_helpers():
    assert shard.shard_offset(0, 65536) == 0 and shard.shard_offset(7, 65536) == 7 * 65536
    assert shard.game_seed(2**64 - 1, 2) == shard.splitmix64(1)          # wraps like the device's uint64 sum
    assert shard.splitmix64(0) == 0xE220A8397B1DCDAF                      # splitmix64 known answer (first output of seed 0)
    # decorrelated default seeds: no two (game, hand) pairs of a batch share a wall key seed + hand_index
    keys = {(shard.game_seed(77, g) + k) & shard.MASK64 for g in range(4096) for k in range(32)}
    assert len(keys) == 4096 * 32
    assert shard.owner_of(65536 * 3 + 5, 65536) == (3, 5)
    with pytest.raises(ValueError):
        shard.shard_offset(-1, 4)
    assert shard.reduce_measurement(None, 0.5, 10) == (0.5, 10.0)
    assert shard.gather_measurement(None, 0, 0.5, 10) == {"wall": 0.5, "steps": 10.0, "ranks_seen": [0], "per_rank_value": [20.0], "per_rank_wall_s": [0.5]}
    assert abi.NO_ACTION == 0xFFFFFFFFFFFFFFFF
