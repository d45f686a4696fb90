"""Pins the ORACLE hand mathematics against the reference's own golden vectors:
riichienv-core/benches/data/agari_{4p,3p}.json (816 + 402 cases), hands_negative.json (200)
and the score table of riichienv-core/tests/agari_correctness.rs:288-326.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle
from riichienv_amd import abi


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("name", ["agari_4p.json", "agari_3p.json"])
def test_agari_fixtures(golden_dir, name):
    cases = _load(golden_dir, name)
    res = oracle.eval_hands([abi.hand_case_from_fixture(c) for c in cases])
    for i, (c, r) in enumerate(zip(cases, res)):
        e = c["expected"]
        assert r.is_agari == 1, (name, i)
        assert bool(r.is_win) == e["is_win"], (name, i)
        assert r.han == e["han"], (name, i, list(r.yaku[: r.n_yaku]), e)
        assert r.fu == e["fu"], (name, i)
        # the fixtures are in the reference's emission order: pin order, not only the set
        assert list(r.yaku[: r.n_yaku]) == e["yaku"], (name, i)


def test_negative_hands(golden_dir):
    cases = _load(golden_dir, "hands_negative.json")
    counts = np.array([c["counts_34"] for c in cases], dtype=np.uint8)
    ag, tp, waits = oracle.agari_counts(counts)
    L = oracle.lib()
    for i, c in enumerate(cases):
        total = int(counts[i].sum())
        assert total in (13, 14)
        assert ag[i] == 0, i  # negative = not a winning shape
        if total == 13:
            assert bool(tp[i]) == c["is_tenpai"], i
            assert bool(L.orc_is_tenpai_free(counts[i].ctypes.data)) == c["is_tenpai"], i


SCORE_ROWS = [  # riichienv-core/tests/agari_correctness.rs:288-326
    (1, 30, 0, 0, 0, 4, 1000, 0, 0), (1, 30, 0, 1, 0, 4, 0, 500, 300), (3, 30, 1, 0, 0, 4, 5800, 0, 0),
    (5, 0, 0, 0, 0, 4, 8000, 0, 0), (5, 0, 1, 1, 0, 4, 0, 0, 4000), (5, 0, 0, 1, 0, 4, 0, 4000, 2000),
    (6, 0, 0, 0, 0, 4, 12000, 0, 0), (8, 0, 0, 0, 0, 4, 16000, 0, 0), (11, 0, 0, 0, 0, 4, 24000, 0, 0),
    (13, 0, 0, 0, 0, 4, 32000, 0, 0), (13, 0, 1, 0, 0, 4, 48000, 0, 0), (13, 0, 0, 1, 0, 4, 0, 16000, 8000),
    (13, 0, 1, 1, 0, 4, 0, 0, 16000), (26, 0, 0, 0, 0, 4, 64000, 0, 0), (26, 0, 1, 0, 0, 4, 96000, 0, 0),
    (26, 0, 0, 1, 0, 4, 0, 32000, 16000), (26, 0, 1, 1, 0, 4, 0, 0, 32000), (26, 0, 0, 0, 2, 4, 64600, 0, 0),
    (26, 0, 1, 1, 2, 4, 0, 200, 32200), (39, 0, 0, 0, 0, 4, 96000, 0, 0), (39, 0, 1, 1, 0, 4, 0, 0, 48000),
    (52, 0, 0, 0, 0, 4, 128000, 0, 0), (65, 0, 0, 0, 0, 4, 160000, 0, 0), (13, 0, 0, 0, 0, 3, 32000, 0, 0),
    (13, 0, 0, 1, 0, 3, 0, 16000, 8000), (26, 0, 0, 0, 0, 3, 64000, 0, 0), (26, 0, 1, 1, 0, 3, 0, 0, 32000),
    # riichienv-core/src/tests.rs:78-90 (no kiriage mangan)
    (4, 30, 0, 1, 0, 4, 0, 3900, 2000),
]


def test_score_rows():
    r = np.array(SCORE_ROWS)
    out = oracle.calculate_score(r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5])
    assert (out[:, 1] == r[:, 6]).all()
    assert (out[:, 2] == r[:, 7]).all()
    assert (out[:, 3] == r[:, 8]).all()


def test_tid_to_mjai():
    # parser.rs:301-334; README.md:94 sample, tests/test_mjai_parity.py:12-18 ('5mr')
    assert oracle.tid_to_mjai(16) == "5mr" and oracle.tid_to_mjai(52) == "5pr" and oracle.tid_to_mjai(88) == "5sr"
    assert oracle.tid_to_mjai(0) == "1m" and oracle.tid_to_mjai(17) == "5m" and oracle.tid_to_mjai(35) == "9m"
    assert oracle.tid_to_mjai(36) == "1p" and oracle.tid_to_mjai(107) == "9s"
    assert [oracle.tid_to_mjai(108 + 4 * i) for i in range(7)] == list("ESWNPFC")


def test_hand_kats_from_reference_unit_tests():
    """riichienv-core/src/tests.rs:93-148, 312-372, 1510-1592 (yakuman ids, win shape without yaku, kazoe cap)."""
    from tests.hand_kats import HAND_KATS, check

    res = oracle.eval_hands([abi.hand_case_from_fixture(c) for _, c, _ in HAND_KATS])
    for (name, _, want), r in zip(HAND_KATS, res):
        check(name, r, want)


def _counts(types):
    c = np.zeros(34, np.uint8)
    for t in types:
        c[t] += 1
    return c


UNIT_SHAPES = [  # riichienv-core/src/tests.rs:9-75: test_agari_standard :9-20 / test_basic_pinfu :23-50 (123m 456m 789m 123p 11s), test_chiitoitsu :53-62, test_kokushi :65-75
    ("standard / pinfu shape", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 18, 18]),
    ("chiitoitsu", [t for t in (0, 2, 4, 6, 8, 10, 12) for _ in range(2)]),
    ("kokushi", [0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33, 0]),
]


def test_unit_shapes_of_the_reference_are_wins():
    """tests.rs:9-75 (is_agari / is_chiitoitsu / is_kokushi on three textbook hands); the same hands without their last tile are tenpai and wait on it.
    state/game_mode.rs:68-74, 77-90 (test_game_mode_config_four_player: 25 000 points to start with; test_sanma_excluded_tiles): a sanma deal holds
    no 2m-8m, a 4P game starts at 25 000."""
    counts = np.array([_counts(h) for _, h in UNIT_SHAPES] + [_counts(h[:-1]) for _, h in UNIT_SHAPES], dtype=np.uint8)
    ag, tp, waits = oracle.agari_counts(counts)
    n = len(UNIT_SHAPES)
    assert list(ag[:n]) == [1] * n
    for i, (name, h) in enumerate(UNIT_SHAPES):
        assert tp[n + i] == 1 and (int(waits[n + i]) >> h[-1]) & 1, name
    g3 = oracle.Game(game_mode=5, seed=3)
    v = g3.peek()
    tiles = list(v.wall[: v.wall_len]) + [t for p in range(3) for t in v.players[p].hand[: v.players[p].hand_len]]
    assert len(tiles) == 108 and all(t < 4 or t >= 32 for t in tiles) and {t // 4 for t in tiles} == set(range(34)) - set(range(1, 8))
    g4 = oracle.Game(game_mode=2, seed=3)
    assert [g4.peek().players[p].score for p in range(4)] == [25000] * 4


def _group_residues(tiles):
    """sizes mod 3 of the connected groups of a hand (r4_group_residues, csrc/rmj_step4.hip.h): tiles of one suit within two ranks chain, honors only
    with their own type"""
    key = lambda t: t + 7 * ((t >= 9) + (t >= 18)) if t < 27 else 48 + 3 * (t - 27)   # noqa: E731  (r4_key)
    ks = sorted(key(t) for t in tiles)
    sizes, cur = [], 1
    for a, b in zip(ks, ks[1:]):
        if b - a > 2:
            sizes.append(cur)
            cur = 1
        else:
            cur += 1
    sizes.append(cur)
    return sum(1 for s in sizes if s % 3 == 1), sum(1 for s in sizes if s % 3 == 2), len(sizes)


@pytest.mark.parametrize("n", [4, 5, 7, 8, 10, 11, 13, 14])
def test_group_residue_bound_never_rules_out_a_tenpai(n):
    """The bound the fused rollouts put in front of the table shanten (round 6, journal r06 section 5): a hand of 3m + 1 tiles with shanten <= 0 has
    groups of sizes {1} or {2, 2} mod 3, one of 3m + 2 tiles {2}, {1, 1} or {2, 2, 1}; seven pairs need six pairs, thirteen orphans twelve kinds - and
    (the riichi bound's gate) twelve kinds need twelve groups.  Checked against the oracle's shanten (shanten.rs:163-261) on random hands and on hands
    built from sets."""
    rng = np.random.default_rng(600 + n)
    deck = np.repeat(np.arange(34), 4)
    hands = [list(rng.permutation(deck)[:n]) for _ in range(6000)]
    while len(hands) < 12000:     # near-tenpai hands: sets first, then loose tiles
        cnt, tiles = np.zeros(34, int), []
        while len(tiles) < n - 3:
            if rng.random() < 0.5:
                s, r = int(rng.integers(3)), int(rng.integers(7))
                ts = [9 * s + r, 9 * s + r + 1, 9 * s + r + 2]
            else:
                ts = [int(rng.integers(34))] * 3
            if all(cnt[t] < 4 for t in ts) and cnt[ts[0]] + (3 if ts[0] == ts[1] else 1) <= 4:
                for t in ts:
                    cnt[t] += 1
                    tiles.append(t)
        while len(tiles) < n:
            t = int(rng.integers(34))
            if cnt[t] < 4:
                cnt[t] += 1
                tiles.append(t)
        hands.append(tiles[:n])
    counts = np.zeros((len(hands), 34), np.uint8)
    for i, h in enumerate(hands):
        for t in h:
            counts[i, t] += 1
    assert counts.max() <= 4
    sh = oracle.shanten(counts)
    allowed = {(1, 0), (0, 2)} if n % 3 == 1 else {(0, 1), (2, 0), (1, 2)}
    n_low = 0
    for i, h in enumerate(hands):
        r1, r2, groups = _group_residues(h)
        kinds = sum(1 for t in (0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33) if counts[i, t])
        if kinds >= 12:
            assert groups >= 12
        if sh[i] <= 0:
            n_low += 1
            ok = (r1, r2) in allowed or (n >= 13 and (int((counts[i] >= 2).sum()) >= 6 or kinds >= 12))
            assert ok, (h, int(sh[i]), r1, r2)
    assert n_low > 3000
