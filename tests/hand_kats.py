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


# ---- the reference's Python evaluator tests (hand strings written out as ids: parse_hand hands out copies in order and skips
# the red copy of a five unless the digit is 0, parser.rs:43-97)
def _ids(s):
    out, digits, used = [], [], {}
    for ch in s:
        if ch.isdigit():
            digits.append(int(ch))
            continue
        su = "mpsz".index(ch)
        for d in digits:
            if d == 0:
                out.append(su * 36 + 16)
                continue
            t = su * 9 + d - 1
            k = used.get(t, 1 if (su < 3 and d == 5) else 0)
            out.append(4 * t + k)
            used[t] = k + 1
        digits = []
    return out


def _pon(ids):
    return {"meld_type": "pon", "tiles": ids, "opened": True, "from_who": 0}


def _chi(ids):
    return {"meld_type": "chi", "tiles": ids, "opened": True, "from_who": 0}


_akas = [8, 12, 16, 48, 52, 56, 80, 84, 88]                             # 345m 456p 345s with the three red fives
_text = _ids("123m456p789s111z2z")                                       # test_agari_calculator.py:41-98

HAND_KATS += [
    # tests/test_core.py:46-111 (yaku ids by scenario; the win tile is a further copy of its type)
    ("core_tanyao", case(_ids("234m234p234s66m88s"), 22), {"is_win": 1, "has": [12], "min_han": 1}),
    ("core_pinfu", case(_ids("123m456p789s23p99m"), 37), {"is_win": 1, "has": [14], "min_han": 1}),
    ("core_yakuhai_white", case(_ids("123m456p78s88m"), 104, melds=[_pon([124, 125, 126])]), {"is_win": 1, "has": [7], "min_han": 1}),
    ("core_honitsu", case(_ids("123m567m111m33z22z"), 114), {"is_win": 1, "has": [27], "min_han": 3}),
    ("core_red_dora_pinfu", case(_ids("234m067p678s34m22z"), 17), {"is_win": 1, "has": [14], "min_han": 2}),
    ("core_no_honroutou", case(_ids("11s22z"), 74, melds=[_pon([124, 125, 126]), _chi([84, 88, 92]), _chi([24, 28, 32])]),
     {"is_win": 1, "has": [7], "lacks": [24, 31]}),
    # tests/test_core.py:114-148: three red fives + tanyao = 4 han, the aka id once
    ("core_three_aka", case(_akas + [92, 93, 94, 64], 65), {"is_win": 1, "yakuman": 0, "has": [32, 12], "once": [32], "han": 4}),
    # tests/test_core.py:151-187: red fives alone are no yaku
    ("core_only_aka_fails", case(_akas + [104, 105, 106, 108], 109, player_wind=1, round_wind=1), {"is_win": 0}),
    # tests/test_agari_calculator.py:4-38: a riichi hand whose wait is not two-sided has no pinfu
    ("calc_no_pinfu", case([12, 17, 21, 68, 68, 80, 80, 83, 96, 104, 120, 120, 122], 100, riichi=True, player_wind=3, round_wind=0),
     {"lacks": [14]}),
    # tests/test_agari_calculator.py:41-98: 123m456p789s EEE S, win on S, by seat wind and win type
    ("calc_text_oya_ron", case(_text, 113), {"is_win": 1, "han": 2, "fu": 40, "ron_agari": 3900, "tsumo_agari_oya": 0, "tsumo_agari_ko": 0}),
    ("calc_text_south_tsumo", case(_text, 113, tsumo=True, player_wind=1),
     {"is_win": 1, "han": 2, "fu": 40, "ron_agari": 0, "tsumo_agari_oya": 1300, "tsumo_agari_ko": 700}),
    ("calc_text_east_tsumo", case(_text, 113, tsumo=True, player_wind=0),
     {"is_win": 1, "han": 3, "fu": 40, "ron_agari": 0, "tsumo_agari_oya": 0, "tsumo_agari_ko": 2600}),
    ("calc_text_west_tsumo", case(_text, 113, tsumo=True, player_wind=2),
     {"is_win": 1, "han": 2, "fu": 40, "ron_agari": 0, "tsumo_agari_oya": 1300, "tsumo_agari_ko": 700}),
    ("calc_text_north_tsumo", case(_text, 113, tsumo=True, player_wind=3),
     {"is_win": 1, "han": 2, "fu": 40, "ron_agari": 0, "tsumo_agari_oya": 1300, "tsumo_agari_ko": 700}),
    # riichienv-core/src/tests.rs:263-272: 111222333m 444p 1s is tenpai on 1s (waits bit 18)
    ("rs_is_tenpai", case([0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, 72], 73), {"is_tenpai": 1, "waits_has": [18]}),
    # tests/test_agari_calculator.py:100-141: an open hand whose only han is a red five is no win (yaku shibari)
    ("calc_yaku_shibari", case([4, 8, 52, 56, 60, 76, 77, 92, 96, 100], 0, melds=[_chi([16, 20, 24])]), {"is_win": 0}),
]


def check(name, r, want):
    ids = list(r.yaku[: r.n_yaku])
    for k, v in want.items():
        if k == "yaku":
            assert ids == v, (name, ids)
        elif k == "has":
            assert all(y in ids for y in v), (name, ids)
        elif k == "waits_has":
            assert all((int(r.waits) >> t) & 1 for t in v), (name, hex(int(r.waits)))
        elif k == "lacks":
            assert not any(y in ids for y in v), (name, ids)
        elif k == "once":
            assert all(ids.count(y) == 1 for y in v), (name, ids)
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
