"""Hand-evaluation KATs transcribed from riichienv-core/src/tests.rs (shared by the oracle and the GPU tests)."""


def case(tiles, win, tsumo=False, riichi=False, ippatsu=False, player_wind=0, round_wind=0, dora=(), melds=()):
    cond = {k: False for k in ("tsumo", "riichi", "double_riichi", "ippatsu", "haitei", "houtei", "rinshan", "chankan", "tsumo_first_turn")}
    cond.update(tsumo=tsumo, riichi=riichi, ippatsu=ippatsu, player_wind=player_wind, round_wind=round_wind, honba=0)
    return {"tiles_136": sorted(tiles), "melds": list(melds), "win_tile_136": win, "dora_indicators": list(dora), "ura_indicators": [],
            "conditions": cond}


def t34(types):
    """tile types -> distinct 136-ids (k-th copy of a type gets id 4*type + k)"""
    seen = {}
    out = []
    for t in types:
        k = seen.get(t, 0)
        seen[t] = k + 1
        out.append(4 * t + k)
    return out


# (name, case, checks) ; checks: dict of expected result fields / predicates
_tsuu = t34([27] * 3 + [28] * 3 + [29] * 3 + [30] * 3 + [31, 31])
_ryuu = t34([19, 20, 21, 23, 23, 23, 25, 25, 25, 32, 32, 32, 19, 19])
_dais = t34([27] * 3 + [28] * 3 + [29] * 3 + [30] * 3 + [0, 0])
_m84 = [0, 1, 2, 60, 64, 72, 73, 74, 76, 80, 96, 100, 104]           # 111m 78p 11123s 789s (tests.rs:312-372)
_kazoe = [0, 1, 4, 5, 8, 9, 12, 16, 20, 24, 28, 32, 17]               # 112233 4 5r 6789 m + 5m (tests.rs:1510-1592)

_south = [0, 1, 2, 4, 5, 6, 8, 9, 10, 32]                              # 111m 222m 333m 9m + pon of South (test_riichienv_hora.py:63-100)

HAND_KATS = [
    # tests/env/test_riichienv_hora.py:63-100: North seat in a South round rons 9m: round wind, toitoi, sanankou, honitsu
    ("south_round_toitoi", case(_south, 33, player_wind=3, round_wind=1,
                                melds=[{"meld_type": "pon", "tiles": [112, 113, 114], "opened": True, "from_who": 0}]),
     {"is_win": 1, "yaku": [11, 21, 22, 27]}),
    # tests.rs:93-108 tsuuiisou (id 39; also daisuushii 50 -> >= 13 han)
    ("tsuuiisou", case(_tsuu[:-1], _tsuu[-1]), {"is_win": 1, "yakuman": 1, "has": [39], "min_han": 13}),
    # tests.rs:110-130 ryuuiisou (id 40)
    ("ryuuiisou", case(_ryuu[:-1], _ryuu[-1]), {"is_win": 1, "yakuman": 1, "has": [40], "min_han": 13}),
    # tests.rs:132-148 daisuushii (id 50, double)
    ("daisuushii", case(_dais[:-1], _dais[-1]), {"is_win": 1, "yakuman": 1, "has": [50], "min_han": 26}),
    # tests.rs:312-372 (Match 84): 6p completes the shape without yaku, 9p is junchan
    ("match84_6p", case(_m84, 56, player_wind=2, round_wind=0), {"is_win": 0, "shape": 1, "han": 0}),
    ("match84_9p", case(_m84, 68, player_wind=2, round_wind=0), {"is_win": 1, "min_han": 3}),
    # tests.rs:1510-1592 kazoe: 14 han without a yakuman yaku -> han reported raw, payments of a single yakuman
    ("kazoe_cap", case(_kazoe, 18, tsumo=True, riichi=True, ippatsu=True, player_wind=1, round_wind=0, dora=[0]),
     {"is_win": 1, "yakuman": 0, "min_han": 14, "tsumo_oya": 16000, "tsumo_ko": 8000}),
]


def check(name, r, want):
    ids = list(r.yaku[: r.n_yaku])
    for k, v in want.items():
        if k == "yaku":
            assert ids == v, (name, ids)
        elif k == "has":
            assert all(y in ids for y in v), (name, ids)
        elif k == "min_han":
            assert r.han >= v, (name, r.han)
        elif k == "shape":
            assert r.has_win_shape == v, name
        elif k == "tsumo_oya":
            assert r.tsumo_agari_oya == v, (name, r.tsumo_agari_oya)
        elif k == "tsumo_ko":
            assert r.tsumo_agari_ko == v, (name, r.tsumo_agari_ko)
        else:
            assert getattr(r, k) == v, (name, k, getattr(r, k))
