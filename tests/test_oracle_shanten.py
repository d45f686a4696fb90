"""Pins the ORACLE shanten on the reference's KATs (tests/test_shanten.py:4-114, README.md:245-273)."""
import numpy as np

from oracle import oracle
from tests.scenarios import tiles


def counts_of(s):
    c = np.zeros(34, np.uint8)
    for t in tiles(s):
        c[t // 4] += 1
    return c


KATS = [  # (hand, 4P, 3P)
    ("1111m111122233z", 1, 2), ("111m111z222z333z44z", -1, -1), ("123456789p11222z", -1, -1),
    ("111m123456789s11z", -1, -1), ("19m19p19s1234567z", 0, 0), ("111m999m123p789s1z", 0, 0),
    ("1199m1199p1199s1z", 0, 0), ("11m99m123p456s111z", 0, 0), ("111m999m123p13s7z", 1, 1),
    ("11119999m22345s", 1, 2), ("1111m9m1234567z", 3, 3), ("111m999m111p11z", -1, -1),
    ("111m123456789p1z", 0, 0), ("999m111222333z1p", 0, 0), ("11m99m11p99p11s99s1z", 0, 0),
    ("111999m111999p1z", 0, 0), ("19m147p258s12345z", 5, 5),
    ("123m456p789s11z", -1, None), ("123m456p78s11z", 0, None),  # README.md:249-255
]


def test_shanten_kats():
    c = np.array([counts_of(h) for h, _, _ in KATS])
    s4 = oracle.shanten(c, sanma=False)
    s3 = oracle.shanten(c, sanma=True)
    for i, (h, e4, e3) in enumerate(KATS):
        assert s4[i] == e4, (h, s4[i], e4)
        if e3 is not None:
            assert s3[i] == e3, (h, s3[i], e3)


def _types(ts):
    c = np.zeros(34, np.uint8)
    for t in ts:
        c[t] += 1
    return c


# shanten.rs:628-785 (unit tests of calculate_effective_tiles_with_discard / calculate_best_ukeire, 4P and 3P)
H_4P_13 = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 27, 27]          # 123m 456m 789m 12p 11z: tenpai on 3p
H_4P_14 = H_4P_13 + [28]
H_3P_14 = [9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 27, 27, 28]
H_4P_SHANPON = [0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 11, 29, 29, 28]  # 123m 456m 789m 33p 332z
H_3P_SHANPON = [9, 10, 11, 12, 13, 14, 15, 16, 17, 20, 20, 29, 29, 28]


def _vis(pairs):
    v = np.zeros(34, np.uint8)
    for t, n in pairs:
        v[t] = n
    return v


UKEIRE_KATS = [  # (hand types, visible, sanma, expected best_ukeire)
    (H_4P_14, [], False, 4), (H_3P_14, [], True, 4), (H_4P_SHANPON, [], False, 4), (H_4P_SHANPON, [(11, 1)], False, 3),
    (H_4P_SHANPON, [(11, 3)], False, 2), (H_3P_SHANPON, [(20, 3)], True, 2),
]


def test_effective_tiles_kats():
    assert oracle.effective_tiles([_types(H_4P_13)])[0] == 1
    assert oracle.effective_tiles([_types(H_4P_14)])[0] >= 1
    assert oracle.effective_tiles([_types(H_3P_14)], sanma=True)[0] >= 1
    assert oracle.effective_tiles([_types(H_4P_13[:-1])])[0] == 0xFFFFFFFF      # 3n hand: the reference panics
    assert oracle.effective_tiles([_types(H_3P_14[:-2])], sanma=True)[0] == 0xFFFFFFFF


def test_best_ukeire_kats():
    for hand, vis, sanma, want in UKEIRE_KATS:
        assert oracle.best_ukeire([_types(hand)], [_vis(vis)], sanma=sanma)[0] == want, (hand, vis, sanma)
