"""riichienv_amd.hand on the GPU: the reference's Python-level hand tests, transcribed onto the same names - tests/test_core.py
(calc_from_text, yaku scenarios, aka dora, tenhou), tests/test_agari_calculator.py, tests/test_calculate_score.py and the
parse_hand + calculate_shanten(_3p) KATs of tests/test_shanten.py.  Every call is a batch of one through rmj_eval_hands /
rmj_calculate_score / rmj_shanten."""
import pytest

from riichienv_amd.compat import Meld, MeldType
from riichienv_amd.hand import (Conditions, HandEvaluator, HandEvaluator3P, Wind, calculate_score, calculate_shanten, calculate_shanten_3p,
                                check_riichi_candidates, parse_hand, parse_tile)

pytestmark = pytest.mark.gpu


def test_core_calc_from_text_and_yaku_scenarios():
    """tests/test_core.py:26-30, 46-111"""
    res = HandEvaluator.calc_from_text("123m456p789s111z22z")
    assert res.is_win and res.han > 0
    scenarios = [("234m234p234s66m88s", "6m", lambda y: 12 in y), ("123m456p789s23p99m", "1p", lambda y: 14 in y),
                 ("123m456p78s88m(p5z0)", "9s", lambda y: 7 in y), ("123m567m111m33z22z", "2z", lambda y: 27 in y),
                 ("234m067p678s34m22z", "5m", lambda y: 14 in y),
                 ("11s22z(p5z0)(456s0)(789m0)", "1s", lambda y: 7 in y and 24 not in y and 31 not in y)]
    for hand, win, check in scenarios:
        res = HandEvaluator.hand_from_text(hand).calc(parse_hand(win)[0][0], conditions=Conditions())
        assert check(res.yaku), (hand, res.yaku)


def test_core_aka_dora_and_tenhou():
    """tests/test_core.py:114-187, 218-235"""
    res = HandEvaluator(sorted([8, 12, 16, 48, 52, 56, 80, 84, 88, 92, 93, 94, 64])).calc(65, [], Conditions(), [])
    assert res.is_win and not res.yakuman and 12 in res.yaku and res.yaku.count(32) == 1 and res.han == 4
    only_aka = HandEvaluator(sorted([8, 12, 16, 48, 52, 56, 80, 84, 88, 104, 105, 106, 108]))
    assert not only_aka.calc(109, [], Conditions(player_wind=Wind.South, round_wind=Wind.South), []).is_win
    standing = [48, 56, 12, 49, 24, 8, 16, 0, 25, 53, 26, 52, 44]
    res = HandEvaluator(sorted(standing + [0]), []).calc(0, dora_indicators=[], ura_indicators=[], conditions=Conditions(tsumo=True, tsumo_first_turn=True))
    assert res.is_win and 35 in res.yaku


def test_agari_calculator():
    """tests/test_agari_calculator.py:4-139"""
    cond = Conditions(tsumo=False, riichi=True, player_wind=Wind.North, round_wind=Wind.East)
    res = HandEvaluator([12, 17, 21, 68, 68, 80, 80, 83, 96, 104, 120, 120, 122], []).calc(win_tile=100, dora_indicators=[], conditions=cond, ura_indicators=[])
    assert 14 not in res.yaku
    hand = HandEvaluator.hand_from_text("123m456p789s111z2z")
    win = parse_tile("2z")
    rows = [(Conditions(), 2, 0, 0, 3900), (Conditions(tsumo=True, player_wind=Wind.South), 2, 1300, 700, 0),
            (Conditions(tsumo=True, player_wind=Wind.East), 3, 0, 2600, 0), (Conditions(tsumo=False, player_wind=Wind.East), 2, 0, 0, 3900),
            (Conditions(tsumo=True, player_wind=Wind.West), 2, 1300, 700, 0), (Conditions(tsumo=True, player_wind=Wind.North), 2, 1300, 700, 0)]
    for c, han, oya, ko, ron in rows:
        r = hand.calc(win, conditions=c)
        assert r.is_win and (r.han, r.fu, r.tsumo_agari_oya, r.tsumo_agari_ko, r.ron_agari) == (han, 40, oya, ko, ron), c
    shibari = HandEvaluator([4, 8, 52, 56, 60, 76, 77, 92, 96, 100], [Meld(MeldType.Chi, [16, 20, 24], True)])
    assert not shibari.calc(0, dora_indicators=[], conditions=Conditions()).is_win


def test_calculate_score():
    """tests/test_calculate_score.py; 3P: tests/env/test_sanma.py:468-477"""
    s = calculate_score(4, 30, False, True, 0)
    assert (s.pay_tsumo_oya, s.pay_tsumo_ko, s.total) == (3900, 2000, 7900)
    s3 = calculate_score(4, 30, False, True, 0, 3)
    assert (s3.pay_tsumo_oya, s3.pay_tsumo_ko, s3.total) == (3900, 2000, 5900)


def test_shanten_kats_through_the_api():
    """tests/test_shanten.py:4-114"""
    both = [("1111m111122233z", 1, 2), ("111m111z222z333z44z", -1, -1), ("123456789p11222z", -1, -1), ("111m123456789s11z", -1, -1),
            ("19m19p19s1234567z", 0, 0), ("111m999m123p789s1z", 0, 0), ("1199m1199p1199s1z", 0, 0), ("11m99m123p456s111z", 0, 0),
            ("111m999m123p13s7z", 1, 1), ("11119999m22345s", 1, 2), ("1111m9m1234567z", 3, 3), ("111m999m111p11z", -1, -1),
            ("111m123456789p1z", 0, 0), ("999m111222333z1p", 0, 0), ("11m99m11p99p11s99s1z", 0, 0), ("111999m111999p1z", 0, 0),
            ("19m147p258s12345z", 5, 5)]
    for text, s4, s3 in both:
        tiles, _ = parse_hand(text)
        assert (calculate_shanten(tiles), calculate_shanten_3p(tiles)) == (s4, s3), text


def test_waits_tenpai_and_riichi_candidates():
    """HandEvaluator.is_tenpai / get_waits (hand_evaluator.rs:178-213) and check_riichi_candidates (:263-284)"""
    h = HandEvaluator.hand_from_text("123m456p789s1112z")
    assert h.is_tenpai() and h.get_waits() == [28]
    assert not HandEvaluator.hand_from_text("159m159p159s1234z").is_tenpai()
    wide = HandEvaluator.hand_from_text("2345666m456p789s")          # 1-4-7m and 2-5m shapes around the triplet
    assert wide.get_waits() == [0, 1, 3, 4, 6]
    t14, _ = parse_hand("123m4569p789s1112z")                          # two loose tiles, 9p and 2z: discarding either leaves a tanki wait on the other
    assert sorted(t // 4 for t in check_riichi_candidates(t14)) == [17, 28]
    assert check_riichi_candidates(parse_hand("159m159p159s12345z")[0]) == []
    t11, _ = parse_hand("4569p789s1112z")                              # the same hand with 123m called away
    assert sorted(t // 4 for t in check_riichi_candidates(t11)) == [17, 28]
    done, _ = parse_hand("123m456p789s11122z")                         # a complete hand: every discard leaves a tenpai hand
    assert len(check_riichi_candidates(done)) == 14
    # 3P: the evaluator of the sanma variant scores with two payers
    r3 = HandEvaluator3P.hand_from_text("111m456p789s111z2z").calc(parse_tile("2z"), conditions=Conditions(tsumo=True, player_wind=Wind.South))
    assert r3.is_win and r3.tsumo_agari_oya > 0 and r3.ron_agari == 0


def test_win_context_create_calculator_and_calculate():
    """replay/mod.rs:2160-2179 on the reference's real log: WinResultContext.create_calculator() + calculate() give what the batch
    evaluation of the same contexts gives (and the payments of the log, tests/win_context_util.check_points)"""
    from riichienv_amd.replay import evaluate_win_contexts
    from tests.win_context_util import check_points, contexts_with_deltas

    rows = contexts_with_deltas()
    evaluate_win_contexts([c for _, c, _ in rows])
    for k, c, h in rows:
        r = c.calculate(c.create_calculator())
        a = c.actual
        assert (r.is_win, r.han, r.fu, r.yaku, r.ron_agari, r.tsumo_agari_oya, r.tsumo_agari_ko) == (
            bool(a.is_win), a.han, a.fu, list(a.yaku[: a.n_yaku]), a.ron_agari, a.tsumo_agari_oya, a.tsumo_agari_ko)
        check_points(k, c, h, r)
        assert [y.id for y in r.yaku_list()] == r.yaku
    k, c, _ = rows[0]
    no_riichi = c.calculate(c.create_calculator(), Conditions(tsumo=c.conditions["tsumo"], player_wind=c.conditions["player_wind"], round_wind=c.conditions["round_wind"]))
    assert 2 not in no_riichi.yaku
