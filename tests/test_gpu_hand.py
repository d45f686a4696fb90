"""GPU parity (kernel gate): batched hand math through the C-ABI vs the oracle and vs the
reference's golden fixtures."""
import json
import os

import numpy as np
import pytest

from riichienv_amd import abi

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("name", ["agari_4p.json", "agari_3p.json"])
def test_eval_hands_fixtures(golden_dir, name):
    from oracle import oracle
    from riichienv_amd import vecenv

    cases = _load(golden_dir, name)
    hcs = [abi.hand_case_from_fixture(c) for c in cases]
    got = vecenv.eval_hands(hcs)
    ref = oracle.eval_hands(hcs)
    for i, (c, r, o) in enumerate(zip(cases, got, ref)):
        e = c["expected"]
        assert bool(r.is_win) == e["is_win"], (name, i)
        assert r.han == e["han"] and r.fu == e["fu"], (name, i, r.han, r.fu, e)
        assert list(r.yaku[: r.n_yaku]) == e["yaku"], (name, i, list(r.yaku[: r.n_yaku]), e["yaku"])
        for f in ("is_win", "yakuman", "has_win_shape", "n_yaku", "han", "fu", "ron_agari", "tsumo_agari_oya",
                  "tsumo_agari_ko", "waits", "is_tenpai", "is_agari"):
            assert getattr(r, f) == getattr(o, f), (name, i, f)


def test_eval_hands_perturbed(golden_dir):
    """Fixture hands with perturbed conditions / win tiles: GPU vs oracle (bit-exact, incl. non-wins)."""
    from oracle import oracle
    from riichienv_amd import vecenv

    rng = np.random.default_rng(7)
    cases = _load(golden_dir, "agari_4p.json")
    hcs = []
    for c in cases:
        for _ in range(3):
            hc = abi.hand_case_from_fixture(c)
            for k in ("tsumo", "riichi", "double_riichi", "ippatsu", "haitei", "houtei", "rinshan", "chankan",
                      "tsumo_first_turn"):
                if rng.random() < 0.3:
                    setattr(hc, k, int(rng.integers(0, 2)))
            hc.player_wind = int(rng.integers(0, 4))
            hc.round_wind = int(rng.integers(0, 4))
            hc.honba = int(rng.integers(0, 4))
            if rng.random() < 0.3:
                hc.win_tile = int(rng.integers(0, 136))
            hcs.append(hc)
    got = vecenv.eval_hands(hcs)
    ref = oracle.eval_hands(hcs)
    for i, (r, o) in enumerate(zip(got, ref)):
        for f in ("is_win", "yakuman", "has_win_shape", "n_yaku", "han", "fu", "ron_agari", "tsumo_agari_oya",
                  "tsumo_agari_ko", "waits", "is_tenpai", "is_agari"):
            assert getattr(r, f) == getattr(o, f), (i, f, getattr(r, f), getattr(o, f))
        assert list(r.yaku[: r.n_yaku]) == list(o.yaku[: o.n_yaku]), i


def _random_hands(rng, n, size):
    counts = np.zeros((n, 34), np.uint8)
    for i in range(n):
        pool = np.repeat(np.arange(34), 4)
        if i % 3 == 0:  # bias towards one or two suits so that tenpai/agari shapes appear
            lo = int(rng.integers(0, 3)) * 9
            pool = np.repeat(np.concatenate([np.arange(lo, lo + 9), np.arange(27, 34)]), 4)
        pick = rng.choice(pool, size=size, replace=False)
        np.add.at(counts[i], pick, 1)
    return counts


@pytest.mark.parametrize("size", [13, 14])
def test_agari_counts_random(golden_dir, size):
    from oracle import oracle
    from riichienv_amd import vecenv

    rng = np.random.default_rng(size)
    neg = np.array([c["counts_34"] for c in _load(golden_dir, "hands_negative.json")], dtype=np.uint8)
    counts = np.concatenate([_random_hands(rng, 20000, size), neg])
    ag, tp, w = vecenv.agari_counts(counts)
    oag, otp, ow = oracle.agari_counts(counts)
    assert (ag == oag).all()
    assert (tp == otp).all()
    assert (w == ow).all()


def _million_hands(seed, n, size):
    """n hands of `size` tiles, vectorised: the `size` smallest of 136 random keys; two thirds of the hands draw from one
    suit + honors or from two suits only, so that tenpai and complete shapes are frequent."""
    rng = np.random.default_rng(seed)
    keys = rng.random((n, 136), dtype=np.float32)
    style = rng.integers(0, 3, size=n)
    suit = rng.integers(0, 3, size=n)
    tile_suit = np.minimum(np.arange(136) // 36, 3)
    drop1 = (style == 1)[:, None] & (tile_suit[None, :] != suit[:, None]) & (tile_suit[None, :] != 3)   # one suit + honors
    drop2 = (style == 2)[:, None] & ((tile_suit[None, :] == suit[:, None]) | (tile_suit[None, :] == 3))  # the two other suits
    keys[drop1 | drop2] = 2.0
    idx = np.argpartition(keys, size, axis=1)[:, :size] // 4
    return (idx[:, :, None] == np.arange(34, dtype=idx.dtype)[None, None, :]).sum(axis=1).astype(np.uint8)


def test_agari_counts_million_hands():
    """is_agari / is_tenpai / waits of 10^6 random hands (13 and 14 tiles), bit-exact vs the oracle."""
    from oracle import oracle
    from riichienv_amd import vecenv

    seen_tenpai = seen_agari = 0
    for chunk in range(10):
        counts = _million_hands(1000 + chunk, 100000, 13 + (chunk & 1))
        assert counts.max() <= 4 and (counts.sum(axis=1) == 13 + (chunk & 1)).all()
        ag, tp, w = vecenv.agari_counts(counts)
        oag, otp, ow = oracle.agari_counts(counts)
        assert (ag == oag).all() and (tp == otp).all() and (w == ow).all(), chunk
        seen_tenpai += int(otp.sum())
        seen_agari += int(oag.sum())
    assert seen_tenpai > 2000 and seen_agari > 100, (seen_tenpai, seen_agari)


def test_score_table():
    from oracle import oracle
    from riichienv_amd import vecenv

    han, fu, oya, tsumo, honba, npl = np.meshgrid(np.arange(0, 70), [20, 25, 30, 40, 50, 70, 110], [0, 1], [0, 1],
                                                  [0, 3], [3, 4], indexing="ij")
    args = [x.ravel() for x in (han, fu, oya, tsumo, honba, npl)]
    assert (vecenv.calculate_score(*args) == oracle.calculate_score(*args)).all()


def test_shanten_kats_and_random():
    """Row A7: rmj_shanten (tables generated from first principles) vs the reference's KATs and vs the oracle's
    plain enumeration, 4P and 3P, for 13/14-tile hands and for hands with melds (10/11/7/8/4/5 tiles)."""
    from oracle import oracle
    from riichienv_amd import vecenv
    from tests.test_oracle_shanten import KATS, counts_of

    c = np.array([counts_of(h) for h, _, _ in KATS])
    s4 = vecenv.shanten(c, sanma=False)
    s3 = vecenv.shanten(c, sanma=True)
    for i, (h, e4, e3) in enumerate(KATS):
        assert s4[i] == e4, (h, s4[i], e4)
        if e3 is not None:
            assert s3[i] == e3, (h, s3[i], e3)
    rng = np.random.default_rng(99)
    hands = np.concatenate([_random_hands(rng, 400, size) for size in (13, 14, 10, 11, 7, 8, 4, 5, 1, 2)])
    assert (vecenv.shanten(hands, False) == oracle.shanten(hands, False)).all()
    # sanma hands: no 2m-8m
    sh = hands.copy()
    sh[:, 1:8] = 0
    assert (vecenv.shanten(sh, True) == oracle.shanten(sh, True)).all()
    assert (vecenv.shanten(sh, False) == oracle.shanten(sh, False)).all()


def _ukeire_hands(rng, n, sanma):
    """hands of 13/14 (and a few 10/11/7/8) tiles drawn from a real wall, plus random visible counts"""
    types = [t for t in range(34) if not (sanma and 1 <= t <= 7)]
    wall = np.repeat(np.array(types), 4)
    hands = np.zeros((n, 34), np.uint8)
    vis = np.zeros((n, 34), np.uint8)
    for i in range(n):
        k = [13, 14, 13, 14, 10, 11, 7, 8, 4, 5, 1, 2][i % 12]
        p = rng.permutation(len(wall))
        if i % 3 == 0:   # flush-ish hands: closer to tenpai
            suit = rng.integers(1, 3)
            pool = np.array([j for j in range(len(wall)) if 9 * suit <= wall[j] < 9 * suit + 9 or wall[j] >= 27])
            p = pool[rng.permutation(len(pool))]
        for t in wall[p[:k]]:
            hands[i, t] += 1
        for t in wall[p[k:k + int(rng.integers(0, 60))]]:
            vis[i, t] += 1
    return hands, vis


@pytest.mark.parametrize("sanma", [False, True])
def test_effective_tiles_and_ukeire_parity(sanma):
    """Row A7 (rest): rmj_effective_tiles / rmj_best_ukeire against the oracle on random hands and the reference KATs."""
    from oracle import oracle
    from riichienv_amd import vecenv
    from tests.test_oracle_shanten import H_4P_13, UKEIRE_KATS, _types, _vis

    rng = np.random.default_rng(77 + int(sanma))
    hands, vis = _ukeire_hands(rng, 1800, sanma)   # (round 4: the 4P walk evaluates every pair as one entry of a pre-merged pair of vectors)
    assert (vecenv.effective_tiles(hands, sanma=sanma) == oracle.effective_tiles(hands, sanma=sanma)).all()
    assert (vecenv.best_ukeire(hands, vis, sanma=sanma) == oracle.best_ukeire(hands, vis, sanma=sanma)).all()
    for hand, v, sm, want in UKEIRE_KATS:
        if sm == sanma:
            assert vecenv.best_ukeire([_types(hand)], [_vis(v)], sanma=sm)[0] == want
    if not sanma:
        assert vecenv.effective_tiles([_types(H_4P_13)])[0] == 1
    with pytest.raises(ValueError):
        vecenv.effective_tiles([_types(H_4P_13[:-1])], sanma=sanma)


def test_hand_kats_from_reference_unit_tests_gpu():
    """The tests.rs KATs of tests/hand_kats.py through rmj_eval_hands, and identical to the oracle's full result."""
    from oracle import oracle
    from riichienv_amd import vecenv
    from tests.hand_kats import HAND_KATS, check

    cases = [abi.hand_case_from_fixture(c) for _, c, _ in HAND_KATS]
    res = vecenv.eval_hands(cases)
    ref = oracle.eval_hands(cases)
    for (name, _, want), r, o in zip(HAND_KATS, res, ref):
        check(name, r, want)
        assert (r.is_win, r.han, r.fu, list(r.yaku[: r.n_yaku]), r.ron_agari, r.tsumo_agari_oya, r.tsumo_agari_ko) == \
               (o.is_win, o.han, o.fu, list(o.yaku[: o.n_yaku]), o.ron_agari, o.tsumo_agari_oya, o.tsumo_agari_ko), name


def test_shanten_and_ukeire_equal_the_reference_tables():
    """rmj_shanten / rmj_effective_tiles / rmj_best_ukeire against the answers of the REFERENCE's nyanten lookup
    (tests/golden/shanten_vectors.json, scripts/gen_shanten_vectors.py): 10^5 sampled hands per variant, every len/3 class."""
    from riichienv_amd import vecenv
    from tests.shanten_sampler import sample_hand, sample_hands, sample_visible
    from tests.test_oracle_shanten_golden import GOLD, expected_shanten

    for tag, sanma in (("4p", False), ("3p", True)):
        hands = sample_hands(GOLD["seed"], GOLD["n_shanten"], sanma)
        got = vecenv.shanten(hands, sanma=sanma)
        exp = expected_shanten(tag)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (tag, bad[:5], hands[bad[:1]], got[bad[:5]], exp[bad[:5]])
        n = GOLD["n_ukeire"]
        hands = np.array([sample_hand(GOLD["seed"] + 1, i, sanma) for i in range(n)], dtype=np.uint8)
        vis = np.array([sample_visible(GOLD["seed"] + 1, i, hands[i]) for i in range(n)], dtype=np.uint8)
        eff_exp = np.array(GOLD[f"effective_tiles_{tag}"])
        uke_exp = np.array(GOLD[f"best_ukeire_{tag}"])
        k = eff_exp >= 0
        assert (vecenv.effective_tiles(hands[k], sanma=sanma) == eff_exp[k]).all(), tag
        k = uke_exp >= 0
        assert (vecenv.best_ukeire(hands[k], vis[k], sanma=sanma) == uke_exp[k]).all(), tag


def test_empty_batches():
    """n = 0 is a valid batch for every batched hand entry point (nothing launched, RMJ_OK)"""
    from riichienv_amd import vecenv

    assert vecenv.eval_hands([]) == []
    z = np.zeros((0, 34), np.uint8)
    ag, tp, w = vecenv.agari_counts(z)
    assert ag.shape == (0,) and tp.shape == (0,) and w.shape == (0,)
    assert vecenv.shanten(z).shape == (0,) and vecenv.shanten(z, True).shape == (0,)
    e = np.zeros(0, np.uint8)
    assert vecenv.calculate_score(e, e, e, e, np.zeros(0, np.uint32), e).shape[0] == 0
