"""Oracle feature encoder vs the reference's placement KATs (observation/encode.rs:632-803: relative seat order of
the broadcast channels; docs/FEATURE_ENCODING.md:8-82 channel map; observation/mod.rs:105 -> ch 70-73 are zero)."""
import numpy as np

from oracle import oracle
from tests.env_adapters import OracleEnv
from tests.scenarios import setup, tiles


def test_encode_channel_placement():
    env = OracleEnv(game_mode=2, seed=3)

    def mut(v):
        for p, s in enumerate([31000, 22000, 27000, 20000]):
            v.players[p].score = s
        v.honba = 3
        v.riichi_sticks = 2
        v.round_wind = 1
        v.oya = 2
        v.kyoku_idx = 2
        v.players[1].riichi_declared = 1
        v.players[3].n_discards = 2
        v.players[3].discards[0] = 4
        v.players[3].discards[1] = 108

    setup(env, hands=[tiles("123m456p789s1122z"), None, None, None], drawn_tile=None, mutate=mut)
    for pid in range(4):
        a = env.g.encode(pid)
        assert a.shape == (74, 34)
        rel = [(pid + i) % 4 for i in range(4)]
        sc = [31000, 22000, 27000, 20000]
        for c in range(4):  # scores in relative order (encode.rs:632-700)
            assert np.allclose(a[39 + c], min(sc[rel[c]], 100000) / 100000.0)
            assert np.allclose(a[43 + c], min(sc[rel[c]], 30000) / 30000.0)
            assert np.allclose(a[26 + c], (2 if rel[c] == 3 else 0) / 24.0)
        assert (a[31 + rel.index(1)] == 1).all() and a[31:35].sum() == 34  # riichi of seat 1 in relative position
        assert np.allclose(a[37], 0.3) and np.allclose(a[38], 0.4)
        assert a[35].sum() == 1 and a[35, 28] == 1  # round wind S
        assert a[36, 27 + (pid + 4 - 2) % 4] == 1
        rank = sum(s > sc[pid] for s in sc)
        assert (a[49 + rank] == 1).all() and a[49:53].sum() == 34
        assert np.allclose(a[53], 2 / 8.0) and np.allclose(a[54], (1 * 4 + 2) / 7.0)
        assert a[70:74].sum() == 0
    a0 = env.g.encode(0)
    assert a0[0, 0] == 1 and a0[1, 27] == 1 and a0[2, 27] == 0  # hand 11z -> count>=2
    assert a0[47].sum() > 0 and (a0[48] == 1).all()  # 13-tile tenpai hand: waits channel
    a3 = env.g.encode(3)
    assert a3[10, 27] == 1 and a3[11, 1] == 1  # own discards, most recent first
    assert a0[14 + 2 * 4, 27] == 1  # seat 3 is kamicha (relative 3) of seat 0
