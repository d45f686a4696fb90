"""riichienv_amd.hand, host side (no GPU): the hand-text parser of parser.rs:9-300 and HandEvaluator.hand_from_text / to_text
(src/riichienv/hand.py), with the parsing expectations of the reference's tests/test_core.py:7-44."""
import pytest

from riichienv_amd.compat import MeldType
from riichienv_amd.hand import Conditions, HandEvaluator, Wind, parse_hand, parse_tile


def test_hand_parsing_of_test_core():
    """tests/test_core.py:7-44 (the parts that need no evaluation)"""
    hand = HandEvaluator.hand_from_text("123m456p789s111z2z")
    assert len(hand.tiles_136) == 13 and hand.to_text() == "123m456p789s1112z"
    red = HandEvaluator.hand_from_text("055m456p789s1122z")
    assert 16 in red.tiles_136 and red.to_text() == "055m456p789s1122z"
    melded = HandEvaluator.hand_from_text("123m456p789s2z(p1z0)")
    assert len(melded.tiles_136) == 10 and len(melded.melds) == 1 and melded.melds[0].meld_type == MeldType.Pon
    assert melded.to_text() == "123m456p789s2z(p1z0)"
    with pytest.raises(ValueError, match="Hand must have 13 tiles"):
        HandEvaluator.hand_from_text("123m456p789s111z")
    with pytest.raises(ValueError, match="Hand must have 14 tiles"):     # a kan asks for one more tile
        HandEvaluator.hand_from_text("123m456p78s2z(k1z0)")


def test_tile_manager_and_melds():
    """parser.rs:9-41 (copies are handed out in order; a plain five avoids the red copy while it can), :148-300 (meld syntax)"""
    assert parse_hand("1m1m1m1m")[0] == [0, 1, 2, 3]
    assert parse_hand("5555m")[0] == [17, 18, 19, 16] and parse_hand("0555m")[0] == [16, 17, 18, 19]
    with pytest.raises(ValueError, match="No more copies"):
        parse_hand("11111m")
    with pytest.raises(ValueError, match="No more copies"):
        parse_hand("00m")
    with pytest.raises(ValueError, match="Pending digits"):
        parse_hand("123")
    tiles, melds = parse_hand("055m(p5z1)(k2z)(s3p2)(123s0)(k0s1)")
    assert tiles == [16, 17, 18]
    assert [(m.meld_type, m.tiles, m.opened) for m in melds] == [
        (MeldType.Pon, [124, 125, 126], True), (MeldType.Ankan, [112, 113, 114, 115], False), (MeldType.Kakan, [44, 45, 46, 47], True),
        (MeldType.Chi, [72, 76, 80], True), (MeldType.Daiminkan, [88, 89, 90, 91], True)]
    assert parse_hand("(p0m1)")[1][0].tiles == [16, 17, 18]              # a red five named in a pon
    assert parse_hand("0m(p5m1)")[1][0].tiles == [17, 18, 19]            # ... or left to the standing tiles
    with pytest.raises(ValueError, match="Chi meld requires 3 digits"):
        parse_hand("(12m0)")
    with pytest.raises(ValueError, match="Invalid suit in meld"):
        parse_hand("(p1x0)")
    assert (parse_tile("2z"), parse_tile("0p"), parse_tile("5p"), parse_tile("1m")) == (112, 52, 53, 0)
    for bad, msg in (("12m", "exactly one tile"), ("", "No tile found"), ("(p1z0)", "meld syntax")):
        with pytest.raises(ValueError, match=msg):
            parse_tile(bad)


def test_conditions_defaults():
    c = Conditions()
    assert (c.tsumo, c.riichi, c.player_wind, c.round_wind, c.honba, c.kita_count, c.is_sanma, c.num_players) == (False, False, 0, 0, 0, 0, False, 4)
    assert [int(w) for w in (Wind.East, Wind.South, Wind.West, Wind.North)] == [0, 1, 2, 3]


def test_the_package_answers_to_the_reference_names():
    """src/riichienv/__init__.py: `import riichienv as rv` users find the same names on the package"""
    import riichienv_amd as rv

    for name in ("RiichiEnv", "Action", "ActionType", "Observation", "Meld", "MeldType", "Phase", "GameRule", "Conditions", "HandEvaluator",
                 "HandEvaluator3P", "Wind", "WinResult", "calculate_score", "calculate_shanten", "calculate_shanten_3p", "check_riichi_candidates",
                 "parse_hand", "parse_tile", "MjaiReplay", "MjSoulReplay", "Kyoku", "WinResultContext", "convert"):
        assert getattr(rv, name) is not None, name
    assert rv.convert.tid_to_mjai(16) == "5mr" and rv.parse_tile("0p") == 52 and int(rv.Wind.North) == 3
    # src/riichienv/action.py:5-17 (PascalCase aliases), game_mode.py, consts.py
    assert rv.ActionType.Discard is rv.ActionType.DISCARD and rv.ActionType.KyushuKyuhai is rv.ActionType.KYUSHU_KYUHAI
    assert len(list(rv.ActionType)) == 12 and int(rv.GameType.SAN_HANCHAN) == 5
    assert (rv.consts.N_TILE_TYPES_4P, rv.consts.N_TILE_TYPES_3P, rv.consts.N_TILES_4P, rv.consts.N_TILES_3P) == (34, 27, 136, 108)
    with pytest.raises(AttributeError):
        rv.no_such_name


def test_yaku_table():
    """yaku.rs:35-127: one entry per yaku id the evaluator can emit; id = Mahjong Soul's id; Tenhou's indices"""
    import json
    import os

    import riichienv_amd as rv

    allk = rv.get_all_yaku()
    assert len(allk) == 49 and len({y.id for y in allk}) == 49 and all(y.id == y.mjsoul_id for y in allk)
    assert [y.id for y in allk] == sorted(y.id for y in allk) and rv.get_yaku_by_id(46) is None and rv.get_yaku_by_id(0) is None
    for i, en, tenhou in ((1, "Menzen Tsumo", 0), (2, "Riichi", 1), (30, "Ippatsu", 2), (14, "Pinfu", 7), (12, "Tanyao", 8), (25, "Chiitoitsu", 22),
                          (31, "Dora", 52), (33, "Ura Dora", 53), (32, "Aka Dora", 54), (42, "Kokushi Musou", 47), (50, "Dai Suusi", 49)):
        y = rv.get_yaku_by_id(i)
        assert (y.name_en, y.tenhou_id) == (en, tenhou), i
    assert rv.get_yaku_by_id(2).name == "立直" and "Riichi" in repr(rv.get_yaku_by_id(2))
    # every id in the reference's golden agari cases has an entry
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    seen = set()
    for f in ("agari_4p.json", "agari_3p.json"):
        with open(os.path.join(gold, f)) as fh:
            data = json.load(fh)
        for case in (data if isinstance(data, list) else data.get("cases", [])):
            exp = case.get("expected", case)
            seen.update(int(y) for y in exp.get("yaku", []))
    assert seen and all(rv.get_yaku_by_id(i) is not None for i in seen), sorted(seen)
