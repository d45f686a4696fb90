"""Oracle pinning of the auxiliary encoders: encode_yaku_possibility against the unit tests of
yaku_checker.rs:419-466 and against expectations derived by hand from its rules (yaku_checker.rs:27-412),
encode_kawa_overview against observation/python.rs:881-925 (incl. the red-five id / column quirks) and the 3P shapes of
tests/env/test_sanma.py:282-292, encode_furiten_ron_possibility (all ones, observation/mod.rs:105)."""
import numpy as np
import pytest

from tests.aux_cases import CASES, apply_case
from tests.env_adapters import OracleEnv
from tests.scenarios import setup

ALL = set()


def _env(name, mode=2, make=OracleEnv):
    env = make(game_mode=mode, seed=5)
    setup(env, hands=[[4 * k + p for k in range(13)] for p in range(4)], drawn_tile=None,
          mutate=lambda v: apply_case(v, CASES[name], mode >= 3))
    return env


def _zeros(y, seat):
    return {int(i) for i in np.where(y[seat, :, 0] == 0)[0]}


# seat -> yaku indices that must be 0.0 (everything else 1.0); derived by hand from yaku_checker.rs
EXPECT = {
    "no_melds": {},
    "pon_1m": {0: {0, 9, 12, 15, 19}},                              # yaku_checker.rs:424-432 (tanyao impossible)
    "pon_5m": {0: {9, 12, 13, 14, 15, 16, 17, 19}},                 # :434-442 (tanyao stays), :457-465 (toitoi possible)
    "chi_123m": {0: {0, 8, 9, 12, 13, 14, 15, 19}},                 # :447-455 (toitoi impossible)
    "white_dead": {0: {1, 11}, 1: {11}},
    "white_set_and_dead": {0: {0, 9, 13, 15, 17, 19}},
    "winds": {0: {5}, 1: {5}, 2: {5}, 3: {4, 5}},
    "one_suit": {0: {0, 8, 9, 12, 13, 14, 15, 19}, 1: {0, 7, 9, 12, 13, 15, 17, 19},
                 2: {0, 6, 7, 8, 9, 12, 13, 14, 15, 16, 17, 19}, 3: {0, 9, 13, 15, 17, 19}},
    "dragons": {0: {2, 10, 11, 15}, 1: {11}, 2: {0, 9, 13, 15, 17, 19}},
    "kokushi_dead": {0: {15}},
    "outside": {0: {0, 8, 9, 12, 13, 14, 15, 19}, 1: {8, 9, 12, 13, 14, 15, 16, 17, 19}, 2: {0, 9, 13, 15, 17, 19},
                3: {0, 6, 7, 8, 9, 12, 13, 14, 15, 19}},
}


@pytest.mark.parametrize("name", sorted(EXPECT))
def test_yaku_possibility_expectations(name):
    y = _env(name).g.encode_yaku_possibility()
    assert y.shape == (4, 21, 2) and (y[:, :, 0] == y[:, :, 1]).all() and set(np.unique(y)) <= {0.0, 1.0}
    for seat in range(4):
        assert _zeros(y, seat) == set(EXPECT[name].get(seat, set())), (name, seat, _zeros(y, seat))


def test_kawa_overview_channels_and_red_quirks():
    k = _env("kawa").g.encode_kawa_overview()
    assert k.shape == (4, 7, 34) and set(np.unique(k)) <= {0.0, 1.0}
    want = np.zeros((4, 7, 34), np.float32)
    want[0, 0:4, 0] = 1          # four 1m: "at least 1..4 discarded"
    want[0, 0, 1] = 1            # 2m
    want[0, 0, 5] = 1            # tile id 20 is a 6m ...
    want[0, 4, 5] = 1            # ... and the reference's "red 5m" id, marked in column 5 (python.rs:902-916)
    for t in (24, 28, 16, 52, 88):
        want[1, 0, t // 4] = 1
    want[1, 5, 14] = 1           # id 24 -> channel 5, column 5 + 9
    want[1, 6, 23] = 1           # id 28 -> channel 6, column 5 + 18; the real reds 16 / 52 / 88 set nothing
    want[2, 0:3, 8] = 1
    want[2, 0:2, 33] = 1
    want[3, 0:2, 33] = 1
    assert (k == want).all(), np.argwhere(k != want)


def test_furiten_ron_possibility_is_all_ones():
    for name in ("kawa", "outside"):
        f = _env(name).g.encode_furiten_ron_possibility()
        assert f.shape == (4, 21) and (f == 1.0).all()


def test_sanma_shapes_and_compact_columns():
    env = OracleEnv(game_mode=5, seed=5)
    case = dict(discards=[[0, 32, 36, 24, 28, 20], [33, 34, 132], [], []], melds=[[], [(1, [108, 109, 110])], [], []], oya=1)
    setup(env, hands=[[4 * k + p for k in range(13)] for p in range(4)], drawn_tile=None, mutate=lambda v: apply_case(v, case, True))
    k = env.g.encode_kawa_overview()
    y = env.g.encode_yaku_possibility()
    f = env.g.encode_furiten_ron_possibility()
    assert k.shape == (3, 7, 27) and y.shape == (3, 21, 2) and f.shape == (3, 21)   # tests/env/test_sanma.py:282-292
    want = np.zeros((3, 7, 27), np.float32)
    want[0, 0, 0] = want[0, 0, 1] = want[0, 0, 2] = 1     # 1m, 9m, 1p -> compact 0, 1, 2; ids 24 / 28 / 20 are 2m-8m: no column
    want[0, 5, 6] = want[0, 6, 15] = 1                     # observation_3p/python.rs:789-801; channel 4 never set
    want[1, 0:2, 1] = 1
    want[1, 0, 33 - 7] = 1
    assert (k == want).all(), np.argwhere(k != want)
    # seat winds with oya = 1 in a three-seat game: seat 1 = E; its pon of E is a set -> round and seat wind stay possible
    assert _zeros(y, 1) == {0, 9, 13, 15, 17, 19} and _zeros(y, 0) == set() and _zeros(y, 2) == set()
    assert (f == 1.0).all()
