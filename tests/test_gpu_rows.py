"""GPU: one and two games per wave (what small batches run as: RMJ_ROWS / the batch-size rule of rmj_create) against the oracle -
the rest of the suite pins RMJ_ROWS=4 (tests/conftest.py)."""
import os

import pytest

from riichienv_amd import vecenv
from riichienv_amd.shard import game_seed
from tests.test_gpu_step import _compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows", ["1", "2", ""])
@pytest.mark.parametrize("mode,greedy", [(2, False), (5, False), (2, True)])
def test_rollouts_with_fewer_games_per_wave(rows, mode, greedy, monkeypatch):
    from oracle import oracle

    if rows:
        monkeypatch.setenv("RMJ_ROWS", rows)
    else:
        monkeypatch.delenv("RMJ_ROWS", raising=False)      # the default rule: 2 560 games and fewer -> one game per wave
    n, seed, pseed, rate = 203, 61 + mode, 17, 96          # (a ragged last wave)
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=4096)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    total = 0
    for k in (1, 1, 250, 40):                              # per-step launches and fused rollouts
        if greedy:
            env.step_greedy(pseed, k, auto_reset=True, call_rate_256=rate)
        else:
            env.step_random(pseed, k, auto_reset=True)
        for g, o in enumerate(games):
            for _ in range(k):
                if o.status()[2]:
                    o.reset()
                    continue
                o.step(o.greedy_actions(pseed, g, rate) if greedy else o.random_actions(pseed, g))
        total += k
        _compare(env, games, range(n), total)
    for g in range(0, n, 29):
        assert env.mjai_log(g) == games[g].log(), g
    env.close()
