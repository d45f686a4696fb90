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
