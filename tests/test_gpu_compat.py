"""The reference-named scalar API (riichienv_amd.compat) over the HIP path: the README loop and a few of the
reference's Python tests, written the way the reference writes them."""
import json

import pytest

pytestmark = pytest.mark.gpu


def test_readme_loop_random_agent():
    """README.md:54-63: obs = env.reset(); while not env.done(): actions = {pid: agent.act(obs)}; obs = env.step(actions)."""
    from riichienv_amd.compat import RandomAgent, RiichiEnv

    agent = RandomAgent(seed=7)
    env = RiichiEnv(game_mode="4p-red-single", seed=42)
    obs_dict = env.reset()
    steps = 0
    while not env.done():
        actions = {pid: agent.act(obs) for pid, obs in obs_dict.items()}
        obs_dict = env.step(actions)
        steps += 1
        assert steps < 2000
    log = env.mjai_log
    assert log[0]["type"] == "start_game" and log[-1]["type"] == "end_game"
    assert sum(env.scores()) + 1000 * env.riichi_sticks == 100000
    assert sorted(env.ranks()) == [1, 2, 3, 4]
    assert len(env.points("basic")) == 4
    with pytest.raises(ValueError):
        env.points("nope")


def test_observation_surface():
    from riichienv_amd.compat import ActionType, Phase, RiichiEnv

    env = RiichiEnv(game_mode=0, seed=1)
    obs = env.reset()
    assert list(obs.keys()) == [0] and env.phase == Phase.WaitAct and env.current_player == 0
    o = obs[0]
    assert len(o.hand) == 14 and o.hands[1] == [] and len(o.mask()) == 82 and len(o.encode()) == 74 * 34 * 4
    assert all(a.action_type in (ActionType.DISCARD, ActionType.RIICHI, ActionType.TSUMO, ActionType.ANKAN,
                                 ActionType.KYUSHU_KYUHAI) for a in o.legal_actions())
    ev = [json.loads(s) for s in o.new_events()]
    assert [e["type"] for e in ev] == ["start_game", "start_kyoku", "tsumo"]
    assert ev[1]["tehais"][1] == ["?"] * 13 and ev[1]["tehais"][0] != ["?"] * 13  # per-seat masking
    a = o.legal_actions()[0]
    assert o.find_action(a.encode()) is not None and json.loads(a.to_mjai())["type"] == "dahai"
    with pytest.raises(ValueError):
        env.reset(scores=[1, 2, 3])  # env.rs:815-823


def test_illegal_discard_like_reference():
    """tests/env/test_illegal_actions.py:5-58 through the compat API."""
    from riichienv_amd.compat import Action, ActionType, Phase, RiichiEnv

    env = RiichiEnv(game_mode="4p-red-east", seed=42)
    env.reset()
    assert env.current_player == 0 and env.phase == Phase.WaitAct
    p0 = env.hands[0]
    bad = next(t for t in range(136) if t not in p0)
    assert env.step({0: Action(ActionType.DISCARD, tile=bad)}) == {}
    assert not env.done()
    ry = [e for e in env.mjai_log if e["type"] == "ryukyoku"][-1]
    assert "Error: Illegal Action" in ry["reason"] and ry["deltas"] == [-12000, 4000, 4000, 4000]
    assert env.scores() == [13000, 29000, 29000, 29000]
    assert env.oya == 0 and env.honba == 1 and len(env.hands[0]) == 14
    assert env.step({}) == {}  # quirk Q9: last_error stays set


def test_claim_priority_like_reference():
    """tests/env/rule_validation/test_claim_priority.py through setters (env.hands = ..., env.drawn_tile = ...)."""
    from riichienv_amd.compat import Action, ActionType, Phase, RiichiEnv

    env = RiichiEnv(seed=1, game_mode=0)
    env.reset()
    env.hands = [sorted([57] + list(range(0, 12)) + [100]), [62, 65] + list(range(108, 119)),
                 [56, 58] + list(range(120, 130)) + [131], [12, 16, 19, 21, 48, 59, 64, 77, 81, 89, 104, 130, 133]]
    env.active_players = [0]
    env.current_player = 0
    env.drawn_tile = 100
    env.step({0: Action(ActionType.DISCARD, tile=57)})
    assert env.phase == Phase.WaitResponse and env.active_players == [1, 2]
    env.step({1: Action(ActionType.CHI, tile=57, consume_tiles=[62, 65]), 2: Action(ActionType.PON, tile=57, consume_tiles=[56, 58])})
    assert env.phase == Phase.WaitAct and env.active_players == [2]
    assert env.mjai_log[-1]["type"] == "pon"


def test_sanma_readme_loop_and_surface():
    """3-player mode through the reference-named API: Observation3P surface (60-way space, 74 x 27 features) + loop."""
    from riichienv_amd.compat import ActionType, RandomAgent, RiichiEnv

    agent = RandomAgent(seed=3)
    env = RiichiEnv(game_mode="3p-red-half", seed=11)
    obs_dict = env.reset()
    o = obs_dict[0]
    assert env.num_players == 3 and len(o.hands) == 3 and len(o.mask()) == 60 and o.action_space_size == 60
    assert len(o.encode()) == 74 * 27 * 4 and len(o.encode_extended()) == 215 * 27 * 4
    a = o.legal_actions()[0]
    assert o.find_action(a.encode_3p()) is not None
    steps = 0
    seen_kita = False
    while not env.done():
        actions = {pid: agent.act(ob) for pid, ob in obs_dict.items()}
        seen_kita = seen_kita or any(x.action_type == ActionType.KITA for x in actions.values())
        obs_dict = env.step(actions)
        steps += 1
        assert steps < 6000
    assert len(env.scores()) == 3 and sorted(env.ranks()) == [1, 2, 3] and len(env.points("basic")) == 3
    assert sum(env.scores()) + 1000 * env.riichi_sticks == 105000
    with pytest.raises(ValueError):
        env.points("ouza-normal")
    with pytest.raises(ValueError):
        env.reset(scores=[1, 2, 3, 4])


def test_observe_event_flow_like_reference():
    """tests/env/test_apply_event.py (full-information variants) through compat.observe_event + select_action_from_mjai."""
    from riichienv_amd.compat import ActionType, RiichiEnv
    from tests.apply_events_util import TEHAIS_4P, start_kyoku

    env = RiichiEnv(game_mode=0)
    assert env.observe_event({"type": "start_game"}, 1) is None
    assert env.observe_event(start_kyoku(TEHAIS_4P), 1) is None
    assert env.observe_event({"type": "tsumo", "actor": 0, "pai": "4p"}, 1) is None      # another seat's draw
    obs = env.observe_event({"type": "dahai", "actor": 0, "pai": "1m", "tsumogiri": False}, 1)
    assert obs is not None and {ActionType.PON, ActionType.PASS} <= {a.action_type for a in obs.legal_actions()}
    sel = obs.select_action_from_mjai({"type": "pon", "actor": 1, "target": 0, "pai": "1m", "consumed": ["1m", "1m"]})
    assert sel is not None and sel.action_type == ActionType.PON
    assert obs.select_action_from_mjai({"type": "none"}).action_type == ActionType.PASS
    obs = env.observe_event({"type": "pon", "actor": 1, "target": 0, "pai": "1m", "consumed": ["1m", "1m"]}, 1)
    assert obs is not None and any(a.action_type == ActionType.DISCARD for a in obs.legal_actions())
    d = obs.select_action_from_mjai({"type": "dahai", "pai": "5s", "tsumogiri": False})
    assert d is not None and d.tile // 4 == 22
    assert env.observe_event({"type": "hora", "actor": 0, "target": 0}, 1) is None


def _masked(tehais, my_seat):
    return [h if i == my_seat else ["?"] * 13 for i, h in enumerate(tehais)]


def test_observe_event_masked_stream_like_reference():
    """tests/env/test_apply_event.py:104-233 with the other seats' hands masked ("?"), the way a bot sees the stream."""
    from riichienv_amd.compat import ActionType, RiichiEnv
    from tests.apply_events_util import TEHAIS_3P, TEHAIS_4P, start_kyoku

    env = RiichiEnv(game_mode=0)
    env.observe_event({"type": "start_game"}, 1)
    assert env.observe_event(start_kyoku(_masked(TEHAIS_4P, 1)), 1) is None
    assert env.observe_event({"type": "tsumo", "actor": 0, "pai": "?"}, 1) is None
    obs = env.observe_event({"type": "dahai", "actor": 0, "pai": "1m", "tsumogiri": True}, 1)
    assert obs is not None and {ActionType.PON, ActionType.PASS} <= {a.action_type for a in obs.legal_actions()}
    obs = env.observe_event({"type": "tsumo", "actor": 1, "pai": "6s"}, 1)      # after everybody passed
    assert obs is not None and any(a.action_type == ActionType.DISCARD for a in obs.legal_actions())
    assert len(obs.hand) == 14

    env = RiichiEnv(game_mode=0)
    env.observe_event({"type": "start_game"}, 0)
    env.observe_event(start_kyoku(_masked(TEHAIS_4P, 0)), 0)
    obs = env.observe_event({"type": "tsumo", "actor": 0, "pai": "4p"}, 0)
    assert obs is not None
    assert env.observe_event({"type": "dahai", "actor": 0, "pai": "4p", "tsumogiri": True}, 0) is None
    assert env.observe_event({"type": "tsumo", "actor": 1, "pai": "?"}, 0) is None
    obs = env.observe_event({"type": "dahai", "actor": 1, "pai": "4s", "tsumogiri": True}, 0)
    assert obs is None or obs.legal_actions()

    env = RiichiEnv(game_mode="3p-red-half")
    env.observe_event({"type": "start_game"}, 1)
    env.observe_event(start_kyoku(_masked(TEHAIS_3P, 1)), 1)
    env.observe_event({"type": "tsumo", "actor": 0, "pai": "?"}, 1)
    obs = env.observe_event({"type": "dahai", "actor": 0, "pai": "1p", "tsumogiri": False}, 1)
    assert obs is not None and ActionType.PON in {a.action_type for a in obs.legal_actions()}


def test_feature_block_accessors_match_reference_shapes():
    """tests/env/test_apply_event.py:test_encode_shanten_efficiency_handles_quad_draw (len 64 / 48) and the block layout."""
    import numpy as np

    from riichienv_amd.compat import RiichiEnv

    for mode, npl, w in ((0, 4, 34), ("3p-red-half", 3, 27)):
        env = RiichiEnv(game_mode=mode, seed=5)
        o = env.reset()[0]
        assert len(o.encode_shanten_efficiency()) == 4 * 4 * npl        # 64 (4P) / 48 (3P) bytes
        eff = np.frombuffer(o.encode_shanten_efficiency(), np.float32).reshape(npl, 4)
        assert (eff[1:, :3] == 0.5).all() and 0 <= eff[0, 0] <= 1
        assert len(o.encode_discard_history_decay()) == npl * w * 4 and len(o.encode_fuuro_overview()) == npl * 20 * w * 4
        assert len(o.encode_action_availability()) == 44 and len(o.encode_discard_candidates()) == 20
        assert len(o.encode_last_tedashis()) == (npl - 1) * 12 and len(o.encode_pass_context()) == 12
        cand = np.frombuffer(o.encode_discard_candidates(), np.float32)
        assert abs(cand[0] - 14 / 34.0) < 1e-7


def test_observation_events_are_the_delta_like_reference():
    """tests/test_observation_serialization.py:85-120, tests/env/test_riichienv.py:42-45: Observation.events holds the
    seat's NEW events since its previous observation (dicts), not the whole log."""
    from riichienv_amd.compat import Action, ActionType, Phase, RiichiEnv

    env = RiichiEnv(seed=9)
    obs = env.reset()
    dealer = obs[0]
    assert [e["type"] for e in dealer.events] == ["start_game", "start_kyoku", "tsumo"]
    obs = env.step({0: Action(ActionType.DISCARD, tile=dealer.hand[0])})
    for pid in (1, 2, 3):
        while env.phase == Phase.WaitResponse:
            obs = env.step({p: Action(ActionType.PASS) for p in env.active_players})
        obs = env.step({pid: Action(ActionType.DISCARD, tile=obs[pid].hand[-1])})   # tsumogiri: nobody's hand changes shape
    while env.phase == Phase.WaitResponse:
        obs = env.step({p: Action(ActionType.PASS) for p in env.active_players})
    assert [e["type"] for e in obs[0].events] == ["dahai", "tsumo"] * 4


def test_apply_event_records_the_callers_log():
    """apply_and_log (riichienv-python/src/env.rs:52-72): applied events are pushed into mjai_log (the caller's event, extra
    fields included) and, masked per seat, into the seats' logs that Observation.events / new_events draw from."""
    import os

    from riichienv_amd.compat import RiichiEnv
    from riichienv_amd.replay import load_mjai_jsonl

    log = load_mjai_jsonl(os.path.join(os.path.dirname(__file__), "golden", "126_204_0_mjai.jsonl"))
    env = RiichiEnv(game_mode="4p-red-half", seed=1)
    env.reset()                                   # leaves reset's own events in the device log: start_game restarts the logs
    cut = next(i for i, e in enumerate(log) if e["type"] == "dahai") + 1
    for e in log[:cut]:
        env.apply_event(e)
    assert env.mjai_log == log[:cut]
    first = log[cut - 1]
    assert first["type"] == "dahai"
    # seat 1 sees the dealer's first discard: its events since the start hold the masked start_kyoku and tsumo
    ev = env.get_observation(1).events
    assert [x["type"] for x in ev][:2] == ["start_game", "start_kyoku"] and ev[-1] == first
    sk = ev[1]
    assert sk["tehais"][1] == log[1]["tehais"][1] and all(h == ["?"] * 13 for i, h in enumerate(sk["tehais"]) if i != 1)
    ts = [x for x in ev if x["type"] == "tsumo"]
    assert ts and all(x["pai"] == "?" for x in ts if x["actor"] != 1)
    assert env.get_observation(1).events == []    # the cursor moved (state/mod.rs:211-218)
    o = env.observe_event(log[cut], 2)            # observe_event = apply + get_observation (env.rs:895-948): cursor of seat 2
    assert env.mjai_log == log[: cut + 1]
    assert env.get_observation(2).events == [] and (o is None or o.events[-1]["type"] == log[cut]["type"])


def test_reset_starts_a_fresh_game_like_reference():
    """tests/env/test_reset_game.py:9-53: reset() without arguments restores scores, round wind, dealer, honba and deposits;
    a scores list of the wrong length is a ValueError."""
    from riichienv_amd.compat import RiichiEnv

    env = RiichiEnv(seed=42)
    env.reset(scores=[30000, 20000, 40000, 10000])
    assert env.scores() == [30000, 20000, 40000, 10000]
    env.reset()
    assert env.scores() == [25000] * 4
    env.reset(round_wind=1, oya=2, honba=3, kyotaku=5)
    assert (env.round_wind, env.oya, env.honba, env.riichi_sticks) == (1, 2, 3, 5)
    env.reset()
    assert (env.round_wind, env.oya, env.honba, env.riichi_sticks) == (0, 0, 0, 0)
    for bad in ([25000] * 3, []):
        with pytest.raises(ValueError, match="does not match"):
            env.reset(scores=bad)
    e3 = RiichiEnv(seed=42, game_mode="3p-red-east")
    e3.reset(scores=[40000, 30000, 35000])
    assert e3.scores() == [40000, 30000, 35000]
    e3.reset()
    assert e3.scores() == [35000] * 3
    with pytest.raises(ValueError, match="does not match"):
        e3.reset(scores=[35000] * 4)


def test_apply_and_observe_event_log_like_reference():
    """tests/env/test_apply_event_mjai_log.py (issue #184): start_game through apply_event / observe_event clears the
    constructor's log, later events accumulate in order with their fields (also the caller's extra ones), 4P and 3P."""
    from riichienv_amd.compat import RiichiEnv
    from tests.apply_events_util import TEHAIS_3P, TEHAIS_4P, start_kyoku

    for mode, tehais in (("4p-red-half", TEHAIS_4P), ("3p-red-half", TEHAIS_3P)):
        for feed in ("apply", "observe"):
            env = RiichiEnv(game_mode=mode)
            assert len(env.mjai_log) > 0                                   # the constructor's own round
            push = env.apply_event if feed == "apply" else (lambda e: env.observe_event(e, 0))
            push({"type": "start_game", "names": ["A", "B", "C", "D"][: len(tehais)]})
            assert len(env.mjai_log) == 1 and env.mjai_log[0]["names"][0] == "A"
            sk = start_kyoku(tehais)
            events = [sk, {"type": "tsumo", "actor": 0, "pai": "5p"}, {"type": "dahai", "actor": 0, "pai": "5p", "tsumogiri": True}]
            for e in events:
                push(e)
            log = env.mjai_log
            assert [e["type"] for e in log] == ["start_game", "start_kyoku", "tsumo", "dahai"]
            assert log[2] == {"type": "tsumo", "actor": 0, "pai": "5p"} and log[3]["tsumogiri"] is True
            assert len(log[1]["tehais"]) == len(tehais)
            push({"type": "start_game"})                                   # a second start_game resets the log again
            assert len(env.mjai_log) == 1 and env.mjai_log[0]["type"] == "start_game"


def test_ranks_and_points_kats():
    """tests/env/test_env_ranks_points.py:7-60: ranks (ties by seat) and the preset point rules."""
    from riichienv_amd.compat import RiichiEnv

    env = RiichiEnv(seed=42)
    env.reset(scores=[30000, 20000, 40000, 10000])
    assert env.ranks() == [2, 3, 1, 4]
    for oya, sc in ((0, [25000] * 4), (1, [25000] * 4), (0, [30000, 30000, 20000, 20000])):
        env.reset(oya=oya, scores=sc)
        assert env.ranks() == [1, 2, 3, 4]
    env.reset(oya=0, scores=[35000, 25000, 25000, 15000])
    assert env.points("basic") == [60, 10, -10, -60]
    env.reset(oya=0, scores=[40000, 30000, 20000, 10000])
    assert env.points("ouza-tyoujyo") == [100, 40, -40, -100] and env.points("ouza-normal") == [50, 20, -20, -50]
    with pytest.raises(ValueError, match="Unknown preset rule: nonexistent"):
        env.points("nonexistent")


def test_state_attributes_of_the_reference_env():
    """The state getters / setters the reference's tests reach for (riichienv-python/src/env.rs:134-622; _riichienv.pyi:762-863):
    per-seat flags as lists, last_discard, current_claims, pending_kan, agari_results, game_mode / game_type / seed."""
    from riichienv_amd.compat import ActionType, GameType, Phase, RandomAgent, RiichiEnv

    env = RiichiEnv(game_mode="4p-red-half", seed=11, skip_mjai_logging=True)
    obs = env.reset()
    assert not env.is_done and env.game_mode() == 2 and env.game_type == GameType.YON_HANCHAN and env.seed == 11 and env.skip_mjai_logging
    for name in ("riichi_stage", "double_riichi_declared", "missed_agari_riichi", "missed_agari_doujun", "ippatsu_cycle"):
        assert getattr(env, name) == [False] * 4, name
    assert env.nagashi_eligible == [True] * 4 and env.score_deltas == [0] * 4 and env.forbidden_discards == [[], [], [], []]
    assert env.last_discard is None and env.pending_kan is None and env.current_claims == {} and env.riichi_pending_acceptance is None
    assert env.pending_kan_dora_count == 0 and not env.is_rinshan_flag and env.agari_results == {}
    # setters write through to the device record
    env.missed_agari_doujun = [False, True, False, False]
    env.forbidden_discards = [[], [], [4, 8], []]
    env.riichi_pending_acceptance = 2
    env.pending_kan_dora_count = 1
    assert env.missed_agari_doujun == [False, True, False, False] and env.forbidden_discards[2] == [4, 8]
    assert env.riichi_pending_acceptance == 2 and env.pending_kan_dora_count == 1
    env.riichi_pending_acceptance = None
    env.pending_kan_dora_count = 0
    env.missed_agari_doujun = [False] * 4
    env.forbidden_discards = [[], [], [], []]
    # play until somebody is offered a claim: current_claims = the published lists of the answering seats
    agent = RandomAgent(seed=3)
    seen = False
    for _ in range(400):
        if env.done():
            break
        if env.phase == Phase.WaitResponse:
            cl = env.current_claims
            assert set(cl) == set(env.active_players) == set(obs) and env.last_discard is not None
            for p, acts in cl.items():
                assert [(a.action_type, a.tile) for a in acts] == [(a.action_type, a.tile) for a in obs[p].legal_actions()]
                assert any(a.action_type == ActionType.PASS for a in acts)
            seen = True
            break
        obs = env.step({p: agent.act(o) for p, o in obs.items()})
    assert seen
    assert env.player_event_counts == [0, 0, 0, 0] or all(c >= 0 for c in env.player_event_counts)


def test_quickstart_example_runs():
    """examples/quickstart.py: the four ways in (reference loop, batched rollout, policy loop on the GPU, hand math + logs)"""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "quickstart.py")
    spec = importlib.util.spec_from_file_location("quickstart", path)
    q = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(q)
    scores, ranks = q.reference_loop()
    assert len(scores) == 4 and sorted(ranks) == [1, 2, 3, 4]
    total, _, points, rows = q.batched_rollout(1024, 100)
    assert 1024 * 190 <= total <= 1024 * 200 and len(points[0]) == 4 and rows > 0   # (a game waits at most a few steps for its restart)
    shape, steps = q.policy_loop(512, 10)
    assert shape == (74, 34) and steps > 0
    hand, n_dec, first = q.hands_and_logs()
    assert hand[:4] == (2, 40, 1300, 700) and n_dec > 20 and len(first) == 3


@pytest.mark.parametrize("mode,rule_name", [("4p-red-half", "tenhou"), ("3p-red-half", "mjsoul")])
def test_scalar_env_against_the_oracle_under_winning_play(mode, rule_name):
    """The drop-in loop of the reference (obs = env.step({pid: action})) with the reference's objects - Observation.legal_actions(),
    .new_events(), .hand, env.scores(), env.mjai_log - against the oracle playing the same game: the greedy policy picks the
    actions (riichi, kans, kita, wins of every kind), every observation of every step is compared, the whole log at the end."""
    from oracle import oracle
    from riichienv_amd import abi
    from riichienv_amd.compat import Action, GameRule, RiichiEnv

    rule = GameRule.default_mjsoul() if rule_name == "mjsoul" else GameRule.default_tenhou()
    env = RiichiEnv(game_mode=mode, seed=4242, rule=rule)
    o = oracle.Game(game_mode=env._mode, seed=4242, rule_bits=rule.bits() | abi.RULE_REFERENCE_RNG)   # (the shim deals the reference's seed -> wall by default)
    o.reset()
    obs = env.reset()
    cursor = [0] * 4
    steps = wins = 0
    while not env.done():
        oa, _, od = o.status()
        assert not od and sorted(obs.keys()) == [s for s in range(4) if (oa >> s) & 1], steps
        v = o.peek()
        for pid, ob in obs.items():
            assert [a._pack() for a in ob.legal_actions()] == o.legal(pid), (steps, pid)
            log = o.log(pid)
            assert ob.new_events() == log[cursor[pid]:], (steps, pid, ob.new_events()[:3], log[cursor[pid]:][:3])
            cursor[pid] = len(log)
            assert list(ob.hand) == [int(t) for t in v.players[pid].hand[: v.players[pid].hand_len]], (steps, pid)
            assert bytes(ob.mask()) == bytes(bytearray(o.mask(pid))[: len(ob.mask())]), (steps, pid)
        acts = [int(x) for x in o.greedy_actions(77, 0, 64)]
        obs = env.step({pid: Action._from_packed(acts[pid]) for pid in obs})
        o.step(acts)
        steps += 1
        assert steps < 6000
    assert o.status()[2]
    assert [json.dumps(e, sort_keys=True, separators=(",", ":")) for e in env.mjai_log] == \
        [json.dumps(json.loads(s), sort_keys=True, separators=(",", ":")) for s in o.log()]
    v = o.peek()
    assert list(env.scores())[: env.num_players] == [v.players[p].score for p in range(env.num_players)]
    wins = sum(1 for e in env.mjai_log if e["type"] == "hora")
    assert wins >= 3 and any(e["type"] == "reach_accepted" for e in env.mjai_log)


def test_observe_event_3p_like_reference():
    """tests/env/test_apply_event.py:292-458 (class TestApplyEvent3P: test_start_game_returns_none :298-302, test_start_kyoku_returns_none :303-308,
    test_tsumo_returns_obs_for_actor :309-319, test_tsumo_returns_none_for_other_player :333-339, test_dahai_returns_obs_when_pon_available :340-352,
    test_dahai_returns_none_when_no_reaction :353-363, test_hora_returns_none :364-370, test_ryukyoku_returns_none :371-377, test_tsumo_after_opponent_discard
    :378-388, test_pon_then_discard_flow :389-401, test_kita_returns_none_for_other :402-422, test_multi_turn_sequence :423-458) and the 4P twins
    test_start_game_returns_none :98-102, test_multi_turn_sequence :254-291: masked streams through compat.observe_event in a sanma game."""
    from riichienv_amd.compat import ActionType, RiichiEnv
    from tests.apply_events_util import TEHAIS_3P, TEHAIS_4P, start_kyoku

    def env_for(seat, tehais=TEHAIS_3P, mode="3p-red-half"):
        e = RiichiEnv(game_mode=mode)
        assert e.observe_event({"type": "start_game"}, seat) is None
        assert e.observe_event(start_kyoku(_masked(tehais, seat)), seat) is None
        return e

    e = env_for(0)
    obs = e.observe_event({"type": "tsumo", "actor": 0, "pai": "3z"}, 0)
    assert obs is not None and any(a.action_type == ActionType.DISCARD for a in obs.legal_actions())
    assert e.observe_event({"type": "dahai", "actor": 0, "pai": "3z", "tsumogiri": True}, 0) is None      # no reaction to the own discard
    assert e.observe_event({"type": "tsumo", "actor": 1, "pai": "?"}, 0) is None
    obs = e.observe_event({"type": "dahai", "actor": 1, "pai": "4s", "tsumogiri": True}, 0)
    assert obs is None or len(obs.legal_actions()) > 0
    e = env_for(1)
    assert e.observe_event({"type": "tsumo", "actor": 0, "pai": "?"}, 1) is None
    obs = e.observe_event({"type": "dahai", "actor": 0, "pai": "1p", "tsumogiri": False}, 1)
    assert obs is not None and {ActionType.PON, ActionType.PASS} <= {a.action_type for a in obs.legal_actions()}
    obs = e.observe_event({"type": "pon", "actor": 1, "target": 0, "pai": "1p", "consumed": ["1p", "1p"]}, 1)
    assert obs is not None and any(a.action_type == ActionType.DISCARD for a in obs.legal_actions())
    e = env_for(2)
    e.observe_event({"type": "tsumo", "actor": 0, "pai": "?"}, 2)
    obs = e.observe_event({"type": "dahai", "actor": 0, "pai": "2p", "tsumogiri": False}, 2)
    assert obs is None or len(obs.legal_actions()) > 0
    e = env_for(1)
    e.observe_event({"type": "tsumo", "actor": 0, "pai": "?"}, 1)
    e.observe_event({"type": "dahai", "actor": 0, "pai": "3z", "tsumogiri": True}, 1)
    obs = e.observe_event({"type": "tsumo", "actor": 1, "pai": "5z"}, 1)
    assert obs is not None and len(obs.legal_actions()) > 0
    e = env_for(1)
    e.observe_event({"type": "tsumo", "actor": 0, "pai": "?"}, 1)
    res = e.observe_event({"type": "kita", "actor": 0}, 1)
    assert res is None or len(res.legal_actions()) > 0
    for ev in ({"type": "hora", "actor": 0, "target": 0}, {"type": "ryukyoku"}):
        assert env_for(0).observe_event(ev, 0) is None
    # the 4P twin of the multi-turn flow (tests/env/test_apply_event.py:254-291)
    e = env_for(0, TEHAIS_4P, "default")
    assert e.observe_event({"type": "tsumo", "actor": 0, "pai": "4p"}, 0) is not None
    assert e.observe_event({"type": "dahai", "actor": 0, "pai": "4p", "tsumogiri": True}, 0) is None
    assert e.observe_event({"type": "tsumo", "actor": 1, "pai": "?"}, 0) is None
    obs = e.observe_event({"type": "dahai", "actor": 1, "pai": "4s", "tsumogiri": True}, 0)
    assert obs is None or len(obs.legal_actions()) > 0


def test_non_action_events_never_return_an_observation():
    """tests/env/test_apply_event.py:469-483 (TestApplyEventConsistency.test_non_action_events_always_none), 4P and 3P, every seat"""
    from riichienv_amd.compat import RiichiEnv
    from tests.apply_events_util import TEHAIS_3P, TEHAIS_4P, start_kyoku

    for mode, tehais in (("default", TEHAIS_4P), ("3p-red-half", TEHAIS_3P)):
        for pid in range(len(tehais)):
            env = RiichiEnv(game_mode=mode)
            for ev in ({"type": "start_game"}, start_kyoku(_masked(tehais, pid)), {"type": "dora", "dora_marker": "3p"},
                       {"type": "hora", "actor": 0, "target": 0}, {"type": "ryukyoku"}):
                assert env.observe_event(ev, pid) is None, (mode, pid, ev["type"])


def test_apply_event_log_covers_calls_reach_and_round_end():
    """tests/env/test_apply_event_mjai_log.py: test_events_accumulate_in_order :75-89 / :208-221, test_start_game_preserves_custom_fields :90-96,
    test_pon_event_logged :97-116 / :222-246, test_chi_event_logged :117-142, test_reach_events_logged :143-156, test_full_round_event_count :157-176,
    test_end_game_logged :177-196, test_start_game_clears_constructor_log :65-74 / :200-207, test_observe_event_logs_events_4p :250-273, test_observe_event_logs_events_3p :274-296,
    test_observe_event_clears_on_start_game :297-311, test_tsumo_log_contains_pai_field :315-325, test_dahai_log_contains_tsumogiri_field :326-338,
    test_start_kyoku_log_contains_tehais :339-348, test_multiple_start_game_resets_log :349-361 - everything apply_event / observe_event is given shows up in
    mjai_log, in order, with its fields."""
    from riichienv_amd.compat import RiichiEnv
    from tests.apply_events_util import CHI_TEHAIS, TEHAIS_3P, TEHAIS_4P, start_kyoku

    env = RiichiEnv(game_mode="default")
    assert len(env.mjai_log) > 0
    round4 = [{"type": "start_game", "names": ["A", "B", "C", "D"]}, start_kyoku(TEHAIS_4P),
              {"type": "tsumo", "actor": 0, "pai": "5p"}, {"type": "dahai", "actor": 0, "pai": "5p", "tsumogiri": True},
              {"type": "tsumo", "actor": 1, "pai": "6s"}, {"type": "dahai", "actor": 1, "pai": "6s", "tsumogiri": True},
              {"type": "tsumo", "actor": 2, "pai": "4z"}, {"type": "dahai", "actor": 2, "pai": "4z", "tsumogiri": True},
              {"type": "tsumo", "actor": 3, "pai": "2z"}, {"type": "dahai", "actor": 3, "pai": "2z", "tsumogiri": True}]
    for ev in round4:
        env.apply_event(ev)
    log = env.mjai_log
    assert len(log) == len(round4) and [e["type"] for e in log] == [e["type"] for e in round4]
    assert log[0]["names"] == ["A", "B", "C", "D"] and log[2]["pai"] == "5p" and log[3]["tsumogiri"] is True and len(log[1]["tehais"]) == 4
    # pon, then the caller's discard
    env = RiichiEnv(game_mode="default")
    for ev in ({"type": "start_game"}, start_kyoku(TEHAIS_4P), {"type": "tsumo", "actor": 0, "pai": "4p"},
               {"type": "dahai", "actor": 0, "pai": "1m", "tsumogiri": False},
               {"type": "pon", "actor": 1, "target": 0, "pai": "1m", "consumed": ["1m", "1m"]}, {"type": "dahai", "actor": 1, "pai": "5s", "tsumogiri": False}):
        env.apply_event(ev)
    pon = next(e for e in env.mjai_log if e["type"] == "pon")
    assert (pon["actor"], pon["target"], pon["pai"], pon["consumed"]) == (1, 0, "1m", ["1m", "1m"])
    assert [e["type"] for e in env.mjai_log][-2:] == ["pon", "dahai"]
    # chi
    env = RiichiEnv(game_mode="default")
    for ev in ({"type": "start_game"}, start_kyoku(CHI_TEHAIS), {"type": "tsumo", "actor": 0, "pai": "4z"},
               {"type": "dahai", "actor": 0, "pai": "3m", "tsumogiri": False}, {"type": "chi", "actor": 1, "target": 0, "pai": "3m", "consumed": ["4m", "5m"]}):
        env.apply_event(ev)
    chi = next(e for e in env.mjai_log if e["type"] == "chi")
    assert (chi["actor"], chi["consumed"]) == (1, ["4m", "5m"])
    # reach / reach_accepted, and the end of the round and of the game
    env = RiichiEnv(game_mode="default")
    for ev in ({"type": "start_game"}, start_kyoku(TEHAIS_4P), {"type": "tsumo", "actor": 0, "pai": "5p"}, {"type": "reach", "actor": 0},
               {"type": "dahai", "actor": 0, "pai": "1m", "tsumogiri": False}, {"type": "reach_accepted", "actor": 0}):
        env.apply_event(ev)
    assert {"reach", "reach_accepted"} <= {e["type"] for e in env.mjai_log}
    env = RiichiEnv(game_mode="default")
    for ev in ({"type": "start_game"}, start_kyoku(TEHAIS_4P), {"type": "tsumo", "actor": 0, "pai": "5p"}, {"type": "hora", "actor": 0, "target": 0},
               {"type": "end_kyoku"}, {"type": "end_game"}):
        env.apply_event(ev)
    assert [e["type"] for e in env.mjai_log][-2:] == ["end_kyoku", "end_game"]
    # 3P through observe_event
    env = RiichiEnv(game_mode="3p-red-half")
    for ev in ({"type": "start_game"}, start_kyoku(TEHAIS_3P), {"type": "tsumo", "actor": 0, "pai": "3z"}, {"type": "dahai", "actor": 0, "pai": "3z", "tsumogiri": True}):
        env.observe_event(ev, 0)
    assert [e["type"] for e in env.mjai_log] == ["start_game", "start_kyoku", "tsumo", "dahai"] and len(env.mjai_log[1]["tehais"]) == 3
    env.observe_event({"type": "start_game"}, 0)
    assert [e["type"] for e in env.mjai_log] == ["start_game"]


def test_new_events_like_reference():
    """tests/env/test_riichienv.py:129-196 (test_new_events): with the other seats' hands emptied (no claims), one go-around shows each seat exactly the events since
    its previous observation: 3 at the start (the own tsumo unmasked), then 5 / 7 / 9 for seats 1 / 2 / 3 (each ending in the seat's own tsumo), and seat 0 sees
    the eight events dahai 0 ... tsumo 0 when its turn comes back."""
    import json

    from riichienv_amd.compat import Action, ActionType, Phase, RiichiEnv

    env = RiichiEnv(seed=9)
    obs = env.reset()
    h = env.hands
    h[1], h[2], h[3] = [], [], []
    env.hands = h
    assert env.phase == Phase.WaitAct
    first = obs[0].new_events()
    assert len(first) == 3 and json.loads(first[2])["pai"] != "?"
    p0 = []

    def collect(o):
        if 0 in o:
            p0.extend(json.loads(e) for e in o[0].new_events())

    obs = env.step({0: Action(ActionType.DISCARD, tile=obs[0].hand[0])})
    collect(obs)
    assert env.phase == Phase.WaitAct and 1 in obs and 0 not in obs
    for pid, want in ((1, 5), (2, 7), (3, 9)):
        new = [json.loads(e) for e in obs[pid].new_events()]
        assert len(new) == want and new[-1]["type"] == "tsumo" and new[-1]["actor"] == pid
        obs = env.step({pid: Action(ActionType.DISCARD, tile=obs[pid].hand[0])})
        collect(obs)
        if env.phase == Phase.WaitResponse:
            obs = env.step({p: Action(ActionType.PASS) for p in env.active_players})
            collect(obs)
    assert env.phase == Phase.WaitAct and 0 in obs and 3 not in obs
    assert [(e["type"], e["actor"]) for e in p0] == [("dahai", 0), ("tsumo", 1), ("dahai", 1), ("tsumo", 2), ("dahai", 2), ("tsumo", 3), ("dahai", 3), ("tsumo", 0)]
    assert [e["pai"] == "?" for e in p0 if e["type"] == "tsumo"] == [True, True, True, False]
