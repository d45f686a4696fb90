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


def _ext_env(discards=None, melds=None):
    """make_obs of observation/encode.rs:593-621: hands 0,4,..,48 (+seat), given discards / melds, nothing else."""
    from riichienv_amd import abi

    env = OracleEnv(game_mode=2, seed=5)

    def mut(v):
        for p in range(4):
            d = (discards or [[]] * 4)[p]
            v.players[p].n_discards = len(d)
            for i, t in enumerate(d):
                v.players[p].discards[i] = t
            ms = (melds or [[]] * 4)[p]
            v.players[p].n_melds = len(ms)
            for i, (mt, ts) in enumerate(ms):
                m = v.players[p].melds[i]
                m.meld_type, m.n_tiles, m.opened, m.from_who, m.called_tile = mt, len(ts), int(mt != abi.MELD_ANKAN), 0, -1
                for k, t in enumerate(ts):
                    m.tiles[k] = t
        v.n_dora = 0

    setup(env, hands=[[4 * k + p for k in range(13)] for p in range(4)], drawn_tile=None, mutate=mut)
    return env


def test_extended_relative_order_kats():
    """observation/encode.rs:632-803 (unit tests of the extended blocks), channel offsets of python.rs:1271-1296."""
    from riichienv_amd import abi

    env = _ext_env(discards=[[0], [], [36], []])
    e0, e2 = env.g.encode_extended(0), env.g.encode_extended(2)
    assert e0.shape == (215, 34)
    assert e0[74, 0] > 0 and e0[76, 9] > 0 and e2[74, 9] > 0 and e2[76, 0] > 0 and e0[74, 0] == e2[74, 9] == 1.0
    env = _ext_env(discards=[[0, 4], [8], [12, 16, 20], []])
    e0, e2 = env.g.encode_extended(0), env.g.encode_extended(2)
    assert abs(e0[78 + 3, 0] - e2[78 + 2 * 4 + 3, 0]) < 1e-6 and np.allclose(e0[78 + 3], 2 / 18.0)
    assert (e0[78 + 4] == 0.5).all() and abs(e0[78, 0] - 0.5) > 1e-6
    assert (e2[78 + 4] == 0.5).all() and abs(e2[78, 0] - 0.5) > 1e-6
    env = _ext_env(melds=[[], [(abi.MELD_ANKAN, [0, 1, 2, 3])], [], []])
    e0, e3 = env.g.encode_extended(0), env.g.encode_extended(3)
    assert e0[94 + 1, 0] == 1 and e0[94, 0] == 0 and e3[94 + 2, 0] == 1 and e3[94 + 1, 0] == 0
    env = _ext_env(melds=[[], [], [(abi.MELD_CHI, [0, 4, 8])], []])
    e0, e1 = env.g.encode_extended(0), env.g.encode_extended(1)
    assert e0[98 + 40, 0] == 1 and e0[98 + 41, 1] == 1 and e1[98 + 20, 0] == 1 and e1[98 + 21, 1] == 1
    d = [[0], [4], [8], [12]]
    env = _ext_env(discards=d)
    for pid in range(4):
        assert env.g.encode_extended(pid)[74, d[pid][0] // 4] > 0
        assert (env.g.encode_extended(pid)[:74] == env.g.encode(pid)).all()


def test_extended_decay_and_scalars():
    env = _ext_env(discards=[[0, 1, 36, 2], [], [], []])
    e = env.g.encode_extended(0)
    import ctypes

    libm = ctypes.CDLL("libm.so.6")                     # f32::exp of the reference = the platform's expf
    libm.expf.restype, libm.expf.argtypes = ctypes.c_float, [ctypes.c_float]
    w = [np.float32(libm.expf(float(np.float32(-0.2) * np.float32(a)))) for a in (3, 2, 0, 1)]
    assert e[74, 0] == np.float32(np.float32(w[0] + w[1]) + w[2])      # accumulated in turn order
    assert e[74, 9] == w[3]
    assert np.allclose(e[189], 13 / 34.0) and (e[192] == 0).all() and (e[193] == 0).all()
    assert (e[194:197] == 0).all() and (e[206:215] == 0).all()


def _ext_env_3p(discards=None, melds=None):
    """make_obs of observation_3p/encode.rs:621-654: sanma hands 1m 9m 1-9p 1-2s (+ seat), given discards / melds, nothing else"""
    from riichienv_amd import abi

    env = OracleEnv(game_mode=5, seed=5)

    def mut(v):
        for p in range(3):
            d = (discards or [[]] * 3)[p]
            v.players[p].n_discards = len(d)
            for i, t in enumerate(d):
                v.players[p].discards[i] = t
            ms = (melds or [[]] * 3)[p]
            v.players[p].n_melds = len(ms)
            for i, (mt, ts, called) in enumerate(ms):
                m = v.players[p].melds[i]
                m.meld_type, m.n_tiles, m.opened, m.from_who, m.called_tile = mt, len(ts), int(mt != abi.MELD_ANKAN), 0, called
                for k, t in enumerate(ts):
                    m.tiles[k] = t
        v.n_dora = 0

    base = [0, 32, 36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76]
    setup(env, hands=[[t + p for t in base] for p in range(3)] + [None], drawn_tile=None, mutate=mut)
    return env


def test_extended_relative_order_kats_3p():
    """observation_3p/encode.rs:667-791 (the five unit tests of the 3P extended blocks: test_discard_decay_relative_order_3p, test_shanten_relative_order_3p,
    test_ankan_relative_order_3p, test_fuuro_relative_order_3p, test_self_channel_always_first_3p); channel offsets of observation_3p/python.rs:1117-1138 (the 4P
    ones: decay 74, shanten 78, ankan 94, fuuro 98), columns = compact 3P tile indices (1m 0, 9m 1, 1p 2, ... 1s 11)."""
    from riichienv_amd import abi

    env = _ext_env_3p(discards=[[0], [36], []])
    e0, e1 = env.g.encode_extended(0), env.g.encode_extended(1)
    assert e0.shape == (215, 27)
    assert e0[74 + 0, 0] > 0 and e0[74 + 1, 2] > 0             # seat 0: itself, then seat 1
    assert e1[74 + 0, 2] > 0 and e1[74 + 2, 0] > 0             # seat 1: itself, seat 0 is two seats on
    env = _ext_env_3p(discards=[[0, 32], [36], [40, 44, 48]])
    e0, e1 = env.g.encode_extended(0), env.g.encode_extended(1)
    assert abs(e0[78 + 3, 0] - e1[78 + 2 * 4 + 3, 0]) < 1e-6   # seat 0's turn count, seen by itself and by seat 1
    assert (e0[78 + 4] == 0.5).all() and abs(e0[78, 0] - 0.5) > 1e-6
    env = _ext_env_3p(melds=[[], [(abi.MELD_ANKAN, [36, 37, 38, 39], -1)], []])
    e0, e2 = env.g.encode_extended(0), env.g.encode_extended(2)
    assert e0[94 + 1, 2] == 1 and e0[94 + 0, 2] == 0 and e2[94 + 2, 2] == 1 and e2[94 + 1, 2] == 0
    env = _ext_env_3p(melds=[[], [], [(abi.MELD_PON, [36, 37, 38], 36)]])
    e0, e1 = env.g.encode_extended(0), env.g.encode_extended(1)
    assert e0[98 + 40, 2] == 1 and e1[98 + 20, 2] == 1
    d = [[0], [36], [72]]
    env = _ext_env_3p(discards=d)
    for pid, col in enumerate((0, 2, 11)):
        assert env.g.encode_extended(pid)[74, col] > 0
