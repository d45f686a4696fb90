"""The API-surface half of the reference's tests/env/test_sanma.py, transcribed against the drop-in shim (riichienv_amd.compat: one game on the GPU behind
RiichiEnv's names) - the ledger (tests/REFERENCE_TESTS.md, round 6) listed these as not restated anywhere: the state-machine half of that file lives in
tests/scenarios.py (sc_3p_*), run on the oracle and on the GPU.  Every test names the reference test it restates (file:lines).

Not transcribed: TestSanmaSerialization and test_to_dict (tests/env/test_sanma.py:336-388: base64 / dict round trips of the PyO3 Observation object - the shim's
Observation is rebuilt from device state, it has no wire format)."""
import math
import struct

import pytest

pytestmark = pytest.mark.gpu
W3 = 27     # N_TILE_TYPES_3P (riichienv/consts.py)


def _env(seed=42, game_mode="3p-red-half", **kw):
    """_create_sanma_env (tests/env/test_sanma.py:25-29)"""
    from riichienv_amd.compat import RiichiEnv

    env = RiichiEnv(game_mode=game_mode, seed=seed, **kw)
    return env, env.reset()


def _pass_all(env, obs):
    from riichienv_amd.compat import Action, ActionType, Phase

    while env.phase == Phase.WaitResponse and not env.is_done:
        obs = env.step({p: Action(ActionType.PASS) for p in env.active_players})
    return obs


def _play_one_turn(env, obs):
    """tests/env/test_sanma.py:32-41"""
    from riichienv_amd.compat import Action, ActionType

    pid = env.current_player
    obs = env.step({pid: Action(ActionType.DISCARD, tile=obs[pid].hand[-1])})
    return _pass_all(env, obs)


def test_observation_fields_and_sizes():
    """test_observation_fields :212-221, test_encode_shape :227-232, test_encode_extended_shape :233-238, test_mask_size :239-245,
    test_mask_has_legal_actions :246-251, test_find_action :252-262"""
    _, obs = _env()
    o = obs[0]
    assert o.player_id == 0
    assert len(o.hands) == 3 and len(o.melds) == 3 and len(o.discards) == 3 and len(o.scores) == 3 and len(o.riichi_declared) == 3
    assert o.action_space_size == 60
    assert len(o.encode()) == 74 * W3 * 4
    assert len(o.encode_extended()) == 215 * W3 * 4
    mask = o.mask()
    assert len(mask) == o.action_space_size and sum(mask) > 0
    aid = next(i for i, v in enumerate(mask) if v == 1)
    assert o.find_action(aid) is not None


def test_auxiliary_encoder_shapes():
    """test_encode_discard_history_decay :263-275, test_encode_shanten_efficiency :276-281, test_encode_fuuro_overview :294-299,
    test_encode_ankan_overview :300-305, test_encode_action_availability :306-311, test_encode_riichi_sutehais :312-317, test_encode_last_tedashis :318-323,
    test_encode_pass_context :324-329, test_encode_discard_candidates :330-335 (+ yaku possibility / kawa overview, :282-293)"""
    env, obs = _env()
    o = obs[0]
    assert len(o.encode_shanten_efficiency()) == 3 * 4 * 4
    assert len(o.encode_yaku_possibility()) == 3 * 21 * 2 * 4
    assert len(o.encode_kawa_overview()) == 3 * 7 * W3 * 4
    assert len(o.encode_fuuro_overview()) == 3 * 4 * 5 * W3 * 4
    assert len(o.encode_ankan_overview()) == 3 * W3 * 4
    assert len(o.encode_action_availability()) == 11 * 4
    assert len(o.encode_riichi_sutehais()) == 2 * 3 * 4
    assert len(o.encode_last_tedashis()) == 2 * 3 * 4
    assert len(o.encode_pass_context()) == 3 * 4
    assert len(o.encode_discard_candidates()) == 5 * 4
    for _ in range(3):
        if env.is_done:
            break
        obs = _play_one_turn(env, obs)
    if not env.is_done:
        assert len(obs[env.current_player].encode_discard_history_decay()) == 3 * W3 * 4


def test_encoded_values_are_finite():
    """test_encode_values_finite :582-590, test_encode_extended_values_finite :591-598"""
    _, obs = _env()
    for enc in (obs[0].encode(), obs[0].encode_extended()):
        for v in struct.unpack(f"<{len(enc) // 4}f", enc):
            assert not math.isnan(v) and abs(v) < 1e10


def test_points_and_ranks():
    """test_points_basic :390-395, test_ranks :396-402"""
    env, _ = _env()
    assert env.points("basic") == [40.0, 0.0, -40.0]
    ranks = env.ranks()
    assert len(ranks) == 3 and set(ranks) == {1, 2, 3}


def test_seeds():
    """test_seeded_determinism :501-507, test_different_seeds_differ :508-513"""
    a, _ = _env(seed=123)
    b, _ = _env(seed=123)
    assert a.hands == b.hands and a.wall == b.wall
    c, _ = _env(seed=1)
    d, _ = _env(seed=2)
    assert c.hands != d.hands


def test_mjai_events_of_a_sanma_round():
    """test_mjai_events_visible_to_all :514-531, test_start_kyoku_scores :532-544"""
    from riichienv_amd.compat import Action, ActionType

    env, obs = _env()
    sk = next(e for e in env.mjai_log if e["type"] == "start_kyoku")
    assert sk["scores"] == [35000, 35000, 35000]
    obs = env.step({0: Action(ActionType.DISCARD, tile=obs[0].hand[-1])})
    obs = _pass_all(env, obs)
    o = obs[env.current_player]
    assert o.events[1]["type"] == "start_kyoku" and len(o.events[1]["tehais"]) == 3


def test_select_action_from_mjai():
    """test_select_action_from_mjai_discard :546-554, test_select_action_from_mjai_pass :555-580"""
    from riichienv_amd.compat import Action, ActionType
    from riichienv_amd.convert import tid_to_mjai

    env, obs = _env()
    o = obs[0]
    act = o.select_action_from_mjai({"type": "dahai", "pai": tid_to_mjai(o.hand[0]), "actor": 0})
    assert act is not None and act.action_type == ActionType.DISCARD
    h = env.hands
    h[0] = [36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84, 88]
    h[1] = sorted([37, 38, 49, 53, 57, 61, 65, 69, 73, 77, 81, 85, 89])
    env.hands = h
    env.current_player = 0
    env.active_players = [0]
    env.drawn_tile = 88
    obs = env.step({0: Action(ActionType.DISCARD, tile=36)})
    assert 1 in obs                                     # seat 1 holds two 1p: Pon is offered
    act = obs[1].select_action_from_mjai({"type": "none"})
    assert act is not None and act.action_type == ActionType.PASS


def test_a_full_sanma_round_by_the_reference_recipe():
    """test_play_full_round :478-500: Tsumo when offered, else the last tile; everybody passes"""
    from riichienv_amd.compat import Action, ActionType

    env, obs = _env(seed=7)
    turns = 0
    while not env.is_done and turns < 200:
        pid = env.current_player
        legals = obs[pid].legal_actions()
        tsumo = next((a for a in legals if a.action_type == ActionType.TSUMO), None)
        obs = env.step({pid: tsumo if tsumo else Action(ActionType.DISCARD, tile=obs[pid].hand[-1])})
        obs = _pass_all(env, obs)
        turns += 1
    assert env.is_done or turns >= 200
