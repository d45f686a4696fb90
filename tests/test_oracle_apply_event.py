"""Row N1 on the oracle: apply_mjai_event restated (state/event_handler.rs, state_3p/event_handler.rs), pinned on the
flows of the reference's tests/env/test_apply_event.py."""
import pytest

from oracle import oracle
from riichienv_amd import abi
from tests.apply_events_util import CHI_TEHAIS, TEHAIS_3P, TEHAIS_4P, start_kyoku


def types_of(game, pid):
    return {abi.unpack_action(a)[0] for a in game.legal(pid)}


def feed(game, events):
    for ev in events:
        game.apply_event(ev)


def active(game):
    return game.status()[0]


def test_mjai_to_tid():
    assert [abi.mjai_to_tid(s) for s in ("1m", "5m", "5mr", "0p", "9s", "E", "C", "1z", "7z")] == [0, 17, 16, 52, 104, 108, 132, 108, 132]
    with pytest.raises(ValueError):
        abi.mjai_to_tid("?")


def test_4p_tsumo_dahai_pon_flow():
    g = oracle.Game(game_mode=0)
    feed(g, [{"type": "start_game"}, start_kyoku(TEHAIS_4P)])
    assert active(g) == 0                       # nobody acts before the first tsumo
    v = g.peek()
    assert v.wall_len == 84 and v.drawable_count == 70 and v.current_player == 0xFF
    assert list(v.players[1].hand[:13]) == sorted(abi.mjai_to_tid(t) for t in TEHAIS_4P[1])
    g.apply_event({"type": "tsumo", "actor": 0, "pai": "4p"})
    assert active(g) == 1 and abi.DISCARD in types_of(g, 0) and g.legal(1) == []
    g.apply_event({"type": "dahai", "actor": 0, "pai": "1m", "tsumogiri": False})
    assert active(g) & 2 and {abi.PON, abi.PASS} <= types_of(g, 1)       # P1 holds a pair of 1m
    assert not (active(g) & 1)
    g.apply_event({"type": "pon", "actor": 1, "target": 0, "pai": "1m", "consumed": ["1m", "1m"]})
    assert active(g) == 2 and abi.DISCARD in types_of(g, 1)
    v = g.peek()
    assert v.players[1].n_melds == 1 and v.players[1].hand_len == 11 and v.players[1].melds[0].from_who == -1
    discards = {abi.unpack_action(a)[1] // 4 for a in g.legal(1) if abi.unpack_action(a)[0] == abi.DISCARD}
    assert 0 not in discards                    # kuikae: the called type is forbidden


def test_4p_dahai_without_reaction_and_next_tsumo():
    g = oracle.Game(game_mode=0)
    feed(g, [{"type": "start_game"}, start_kyoku(TEHAIS_4P), {"type": "tsumo", "actor": 0, "pai": "4p"},
             {"type": "dahai", "actor": 0, "pai": "4p", "tsumogiri": True}])
    # 4p: P1 holds 4p5p6p.. -> chi possible for the next seat; everyone else silent
    assert (active(g) & ~2) == 0
    g.apply_event({"type": "tsumo", "actor": 1, "pai": "6s"})
    assert active(g) == 2 and abi.DISCARD in types_of(g, 1)


def test_chi_kuikae_forbids_called_and_other_side_tile():
    g = oracle.Game(game_mode=0)
    feed(g, [{"type": "start_game"}, start_kyoku(CHI_TEHAIS), {"type": "tsumo", "actor": 0, "pai": "4z"},
             {"type": "dahai", "actor": 0, "pai": "3m", "tsumogiri": False}])
    assert abi.CHI in types_of(g, 1)
    g.apply_event({"type": "chi", "actor": 1, "target": 0, "pai": "3m", "consumed": ["4m", "5m"]})
    t34 = {abi.unpack_action(a)[1] // 4 for a in g.legal(1) if abi.unpack_action(a)[0] == abi.DISCARD}
    assert 2 not in t34 and 5 not in t34 and t34


def test_reach_flow_and_terminal_events():
    g = oracle.Game(game_mode=0)
    feed(g, [{"type": "start_game"}, start_kyoku(TEHAIS_4P), {"type": "tsumo", "actor": 0, "pai": "1m"},
             {"type": "reach", "actor": 0}])
    assert g.peek().players[0].riichi_stage
    feed(g, [{"type": "dahai", "actor": 0, "pai": "1m", "tsumogiri": False}, {"type": "reach_accepted", "actor": 0}])
    v = g.peek()
    assert v.players[0].riichi_declared and v.players[0].score == 24000 and v.riichi_sticks == 1
    g.apply_event({"type": "dora", "dora_marker": "3p"})
    assert g.peek().n_dora == 2
    g.apply_event({"type": "hora", "actor": 0, "target": 0})
    assert g.status()[2] == 1


def test_3p_pon_and_kita():
    g = oracle.Game(game_mode=5)
    feed(g, [{"type": "start_game"}, start_kyoku(TEHAIS_3P), {"type": "tsumo", "actor": 0, "pai": "3z"}])
    assert active(g) == 1 and abi.DISCARD in types_of(g, 0)
    v = g.peek()
    assert v.wall_len == 108 - 39 - 1 and v.drawable_count == 108 - 39 - 14 - 1
    g.apply_event({"type": "dahai", "actor": 0, "pai": "1p", "tsumogiri": False})
    assert {abi.PON, abi.PASS} <= types_of(g, 1)
    g.apply_event({"type": "pon", "actor": 1, "target": 0, "pai": "1p", "consumed": ["1p", "1p"]})
    assert active(g) == 2
    g2 = oracle.Game(game_mode=5)
    hands = [list(h) for h in TEHAIS_3P]
    hands[0][12] = "4z"                          # give P0 a North tile
    feed(g2, [{"type": "start_game"}, start_kyoku(hands), {"type": "tsumo", "actor": 0, "pai": "3z"}, {"type": "kita", "actor": 0}])
    v = g2.peek()
    assert v.players[0].n_kita == 1 and v.players[0].kita[0] // 4 == 30 and v.current_player == 0xFF


def _pass_observations_of_seat1(events):
    """Walk a log through apply_event(replay=True); whenever seat 1 is offered a claim and the log goes on without it taking one (its implicit
    Pass), record whether Ron was among the offers - what KyokuStepIterator yields as the seat's Pass steps."""
    g = oracle.Game(game_mode=0)
    out = []
    for ev in events:
        act, ph, _ = g.status()
        claims = ev.get("type") in ("hora", "pon", "chi", "daiminkan") and ev.get("actor") == 1
        if ph == abi.WAIT_RESPONSE and (act >> 1) & 1 and not claims:
            out.append(abi.RON in types_of(g, 1))
        g.apply_event(ev, replay=True)
    return out


def test_doujun_furiten_resets_after_own_discard():
    """tests/env/test_apply_event.py:535-574: seat 1 passes the Ron on seat 0's 3m, draws and discards: the same-turn furiten is gone
    and seat 2's 3m is offered again."""
    from tests.apply_events_util import furiten_log

    assert _pass_observations_of_seat1(furiten_log(False)) == [True, True]


def test_riichi_furiten_persists_after_own_discard():
    """tests/env/test_apply_event.py:576-632: in riichi the missed Ron is permanent: the second 3m produces no offer at all."""
    from tests.apply_events_util import furiten_log

    assert _pass_observations_of_seat1(furiten_log(True)) == [True]


_REACH_TEHAI = ["1m", "2m", "3m", "4m", "5m", "6m", "7m", "8m", "9m", "1p", "2p", "3p", "1s"]


@pytest.mark.parametrize("mode,npl,wall,live", [(2, 4, 84, 70), (5, 3, 69, 55)])
def test_start_kyoku_rewinds_the_wall_and_the_tile_count(mode, npl, wall, live):
    """riichienv-core/src/tests.rs:576-668 (4P), 707-795 (3P): start_kyoku rewinds the wall to before the dealer's draw and resets
    drawable_count whatever an earlier round left; the first tsumo takes one tile."""
    g = oracle.Game(game_mode=mode)
    assert g.peek().wall_len == wall - 1                      # the constructor's round has drawn already
    v = g.peek()
    v.drawable_count = 1                                       # a depleted earlier round
    g.poke(v)
    tehais = [[f"{1 + p}p"] * 13 for p in range(npl)]
    sk = start_kyoku(tehais, oya=1, scores=[25000] * npl if npl == 4 else [35000] * npl)
    sk["kyoku"], sk["dora_marker"] = 2, "1p"
    feed(g, [sk])
    v = g.peek()
    assert (v.wall_len, v.drawable_count, v.needs_tsumo, v.drawn_tile) == (wall, live, 1, -1)
    g.apply_event({"type": "tsumo", "actor": 1, "pai": "5p"})
    v = g.peek()
    assert (v.wall_len, v.drawable_count) == (wall - 1, live - 1)


@pytest.mark.parametrize("mode,npl", [(2, 4), (5, 3)])
def test_replay_start_kyoku_offers_reach(mode, npl):
    """tests.rs:669-706 (4P), 796-834 (3P), issue #198: a reach-eligible tenpai right after a replayed start_kyoku is offered Riichi
    although the earlier round ended with drawable_count = 1.  (3P: the manzu 2-8 of the 4P fixture are replaced by pinzu / souzu.)"""
    g = oracle.Game(game_mode=mode)
    v = g.peek()
    v.drawable_count = 1
    g.poke(v)
    tehai = _REACH_TEHAI if npl == 4 else ["1p", "2p", "3p", "4p", "5p", "6p", "7p", "8p", "9p", "1s", "2s", "3s", "9m"]
    sk = start_kyoku([tehai] + [["1z"] * 13 for _ in range(npl - 1)], oya=0, scores=[25000] * npl if npl == 4 else [35000] * npl)
    sk["kyoku"], sk["dora_marker"] = 2, "9s"
    feed(g, [sk, {"type": "tsumo", "actor": 0, "pai": "E"}])
    assert abi.RIICHI in types_of(g, 0)
