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
        monkeypatch.delenv("RMJ_ROWS", raising=False)      # the default rule: 3 584 games and fewer -> one game per wave
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


def _unlisted_seats_are_clean(env):
    """device-only invariant of a publication: a seat that is not to act (or any seat of a finished game) has no list and an empty mask row"""
    import numpy as np

    act, _, dn = env.status()
    _, cnt = env.legal()
    mask = env.mask().reshape(env.n, 4, -1).sum(axis=2)
    idle = ~(((act[:, None] >> np.arange(4)[None, :]) & 1).astype(bool)) | dn.astype(bool)[:, None]
    bad = np.argwhere(idle & ((cnt != 0) | (mask != 0)))
    assert len(bad) == 0, bad[:8].tolist()


@pytest.mark.parametrize("mode,k,steps,rate", [(5, 505, 6000, -1), (5, 505, 5998, -1), (3, 511, 6000, -1), (4, 701, 4000, 160), (2, 505, 3000, -1), (2, 701, 3000, 200)])
def test_last_step_of_a_fused_rollout_rewrites_every_row(mode, k, steps, rate):
    """Quiet steps of a fused rollout publish no mask rows; its last step rewrites all four - also when that step is a round end whose first
    list the full path writes (pass 2 of an inline-response rollout: a 3P dealer's list of more than 16 entries).  Found by the round-4
    soak (profiles/r04_parity_soak_final.log): the full path took the caller's flags there, which carry neither STEP_F_QUIET nor _ALLROWS."""
    seed, pseed = 7000 + 131 * k + mode, 0xA5A5 + 977 * k
    env = vecenv.VecRiichiEnv(512, game_mode=mode, seed=seed, event_ring=64)
    env.reset()
    if rate >= 0:
        env.step_greedy(pseed, steps, auto_reset=True, call_rate_256=rate)
    else:
        env.step_random(pseed, steps, auto_reset=True)
    _unlisted_seats_are_clean(env)
    env.close()
