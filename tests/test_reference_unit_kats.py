"""The five small unit tests of the reference's riichienv-core/src/tests.rs that no scenario covered yet: test_agari_standard (:9),
test_basic_pinfu (:23), test_chiitoitsu (:53), test_kokushi (:65) - win shapes over type histograms - and
test_seeded_shuffle_changes_between_rounds (:155): with a fixed episode seed, consecutive rounds deal different walls (the reference
compares the wall digests; here the walls themselves, the digest being a function of the wall and a salt)."""
import numpy as np
import pytest

from riichienv_amd.shard import game_seed

SHAPES = {
    "agari_standard / basic_pinfu: 123m 456m 789m 123p 11s": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 18, 18],
    "chiitoitsu: pairs of 0 2 4 6 8 10 12": [t for t in (0, 2, 4, 6, 8, 10, 12) for _ in range(2)],
    "kokushi: thirteen terminals and honors + 1m": [0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33, 0],
}


def _counts():
    c = np.zeros((len(SHAPES), 34), np.uint8)
    for i, types in enumerate(SHAPES.values()):
        for t in types:
            c[i, t] += 1
    return c


def test_win_shapes_oracle():
    from oracle import oracle

    ag, _, _ = oracle.agari_counts(_counts())
    assert ag.tolist() == [1, 1, 1]
    # one tile short: 13-tile hands wait on exactly the missing tile (pair wait 1s; chiitoi single; kokushi 13-sided minus nothing)
    c = _counts()
    c[0, 18] -= 1
    c[1, 12] -= 1
    c[2, 0] -= 1
    ag, tp, w = oracle.agari_counts(c)
    assert ag.tolist() == [0, 0, 0] and tp.tolist() == [1, 1, 1]
    assert (int(w[0]) >> 18) & 1 and int(w[1]) == 1 << 12 and int(w[2]) == sum(1 << t for t in (0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33))


@pytest.mark.gpu
def test_win_shapes_gpu():
    from oracle import oracle
    from riichienv_amd import vecenv

    for c in (_counts(), _counts() - np.eye(34, dtype=np.uint8)[[18, 12, 0]]):
        got = vecenv.agari_counts(c)
        ref = oracle.agari_counts(c)
        assert all((np.asarray(a) == np.asarray(b)).all() for a, b in zip(got, ref))


def _walls(make, rounds=3):
    """walls of consecutive rounds of one game with a fixed seed: exhaust nothing - restart rounds through reset() of the same seed index"""
    return make(rounds)


def test_seeded_walls_differ_between_rounds_oracle():
    from oracle import oracle

    o = oracle.Game(game_mode=2, seed=42)
    walls = []
    for _ in range(3):          # the constructor dealt hand_index 0; every reset() deals the next one (quirk Q1: reset does not reseed)
        o.reset()
        v = o.peek()
        walls.append(bytes(v.wall[: v.wall_len]))
    assert len(set(walls)) == 3
    o2 = oracle.Game(game_mode=2, seed=42)
    o2.reset()
    v2 = o2.peek()
    assert bytes(v2.wall[: v2.wall_len]) == walls[0]       # and the same seed deals the same walls again


@pytest.mark.gpu
def test_seeded_walls_differ_between_rounds_gpu():
    from oracle import oracle
    from riichienv_amd import vecenv

    env = vecenv.VecRiichiEnv(2, game_mode=2, seed=42)
    o = oracle.Game(game_mode=2, seed=game_seed(42, 0))
    walls = []
    for _ in range(3):
        env.reset()
        o.reset()
        v, ov = env.peek(0), o.peek()
        assert bytes(v.wall[: v.wall_len]) == bytes(ov.wall[: ov.wall_len])
        walls.append(bytes(v.wall[: v.wall_len]))
    assert len(set(walls)) == 3
    env.close()
