"""Row N1 on the GPU: rmj_apply_events against the oracle's apply_mjai_event.  Event streams = the reference-test flows
and whole MJAI logs of finished random rollouts (each game is fed its own log, one event per launch)."""
from riichienv_amd.shard import game_seed
import json

import numpy as np
import pytest

from riichienv_amd import abi
from tests.apply_events_util import CHI_TEHAIS, TEHAIS_3P, TEHAIS_4P, start_kyoku
from tests.parity_util import diff_dict, fmt_action, normalize_view

pytestmark = pytest.mark.gpu


def _norm(v):
    d = normalize_view(v)
    for p in d["players"]:
        for m in p["melds"]:
            m["tiles"] = sorted(m["tiles"])     # the device stores meld tiles sorted (documented deviation of this path)
    d.pop("wall_seed", None)
    d.pop("hand_index", None)
    return d


def _compare(env, games, step_no):
    act, ph, dn = env.status()
    legal, cnt = env.legal()
    mask = env.mask()
    for g, o in enumerate(games):
        oa, op, od = o.status()
        assert (act[g], ph[g], dn[g]) == (oa, op, od), (g, step_no, (act[g], ph[g], dn[g]), (oa, op, od))
        d = diff_dict(_norm(env.peek(g)), _norm(o.peek()))
        assert not d, (g, step_no, d[:8])
        for s in range(4):
            if (oa >> s) & 1 and not od:
                ol = o.legal(s)
                gl = [int(x) for x in legal[g, s, : cnt[g, s]]]
                assert gl == ol, (g, step_no, s, [fmt_action(a) for a in gl], [fmt_action(a) for a in ol])
                assert (mask[g, s] == o.mask(s)).all(), (g, step_no, s)
            else:
                assert cnt[g, s] == 0


def _run_streams(mode, streams, prepare=None):
    from oracle import oracle
    from riichienv_amd import vecenv

    n = len(streams)
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=1)
    games = [oracle.Game(game_mode=mode, seed=game_seed(1, g)) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    if prepare:
        for g, o in enumerate(games):
            v = o.peek()
            prepare(v)
            o.poke(v)
            env.poke(g, v)
    for k in range(max(len(s) for s in streams)):
        evs = [s[k] if k < len(s) else None for s in streams]
        env.apply_events(evs)
        for o, ev in zip(games, evs):
            if ev is not None:
                o.apply_event(ev)
        _compare(env, games, k)


def test_reference_flows_4p():
    pre = [{"type": "start_game"}]
    flows = [
        pre + [start_kyoku(TEHAIS_4P), {"type": "tsumo", "actor": 0, "pai": "4p"}, {"type": "dahai", "actor": 0, "pai": "1m", "tsumogiri": False},
               {"type": "pon", "actor": 1, "target": 0, "pai": "1m", "consumed": ["1m", "1m"]}, {"type": "dahai", "actor": 1, "pai": "5s", "tsumogiri": False}],
        pre + [start_kyoku(CHI_TEHAIS), {"type": "tsumo", "actor": 0, "pai": "4z"}, {"type": "dahai", "actor": 0, "pai": "3m", "tsumogiri": False},
               {"type": "chi", "actor": 1, "target": 0, "pai": "3m", "consumed": ["4m", "5m"]}, {"type": "dahai", "actor": 1, "pai": "1z", "tsumogiri": False}],
        pre + [start_kyoku(TEHAIS_4P, oya=2), {"type": "tsumo", "actor": 2, "pai": "1m"}, {"type": "reach", "actor": 2},
               {"type": "dahai", "actor": 2, "pai": "1z", "tsumogiri": False}, {"type": "reach_accepted", "actor": 2}, {"type": "dora", "dora_marker": "3p"},
               {"type": "tsumo", "actor": 3, "pai": "5mr"}, {"type": "hora", "actor": 3, "target": 3}, {"type": "end_kyoku"}],
    ]
    _run_streams(0, flows)


def test_reference_flows_3p():
    pre = [{"type": "start_game"}]
    hands = [list(h) for h in TEHAIS_3P]
    hands[0][12] = "4z"
    flows = [
        pre + [start_kyoku(TEHAIS_3P), {"type": "tsumo", "actor": 0, "pai": "3z"}, {"type": "dahai", "actor": 0, "pai": "1p", "tsumogiri": False},
               {"type": "pon", "actor": 1, "target": 0, "pai": "1p", "consumed": ["1p", "1p"]}, {"type": "dahai", "actor": 1, "pai": "3z", "tsumogiri": False}],
        pre + [start_kyoku(hands), {"type": "tsumo", "actor": 0, "pai": "3z"}, {"type": "kita", "actor": 0}, {"type": "tsumo", "actor": 0, "pai": "9s"},
               {"type": "dahai", "actor": 0, "pai": "9s", "tsumogiri": True}, {"type": "ryukyoku"}],
    ]
    _run_streams(5, flows)


@pytest.mark.parametrize("mode", [2, 5])
def test_replay_own_logs(mode):
    """Logs of random rollouts (calls, kans, riichi, kita, wins, draws, several kyoku) fed back through apply_events."""
    from oracle import oracle

    streams = []
    for g in range(24):
        o = oracle.Game(game_mode=mode, seed=500 + g)
        o.reset()
        for _ in range(260 + 20 * g):
            if o.status()[2]:
                break
            o.step(o.random_actions(77, g))
        streams.append([json.loads(x) for x in o.log()])
    _run_streams(mode, streams)


@pytest.mark.parametrize("mode,rate", [(2, 64), (5, 32)])
def test_replay_logs_of_winning_play(mode, rate):
    """Logs of the greedy policy (riichi declared and accepted, ippatsu, kan dora, Ron / Tsumo / multi-Ron settlements, renchan, games
    that end) fed back through apply_events: a RandomAgent's logs hold next to none of these."""
    from oracle import oracle

    streams, kinds = [], set()
    for g in range(24):
        o = oracle.Game(game_mode=mode, seed=900 + g)
        o.reset()
        for _ in range(300 + 25 * g):
            if o.status()[2]:
                break
            o.step([int(x) for x in o.greedy_actions(31, g, rate)])
        streams.append([json.loads(x) for x in o.log()])
        kinds |= {e["type"] for e in streams[-1]}
    assert {"hora", "reach", "reach_accepted", "end_kyoku"} <= kinds, kinds
    _run_streams(mode, streams)


@pytest.mark.parametrize("mode,npl", [(2, 4), (5, 3)])
def test_start_kyoku_after_a_depleted_round(mode, npl):
    """riichienv-core/src/tests.rs:576-834 on the device: start_kyoku rewinds the wall and resets drawable_count (left at 1 by the
    poke), the first tsumo takes one tile, a reach-eligible tenpai is offered Riichi (issue #198); every event compared with the
    oracle."""
    def deplete(v):
        v.drawable_count = 1

    score = [25000] * 4 if npl == 4 else [35000] * 3
    # (the Rust tests deal thirteen copies of one tile to a seat; the device counts types in 3-bit fields, so playable hands here)
    plain = [["1p", "4p", "7p", "1s", "4s", "7s", "E", "S", "W", "N", "P", "F", "C"], ["2p", "5p", "8p", "2s", "5s", "8s", "E", "S", "W", "N", "P", "F", "C"],
             ["3p", "6p", "9p", "3s", "6s", "9s", "E", "S", "W", "N", "P", "F", "C"], ["1p", "5p", "9p", "1s", "5s", "9s", "E", "S", "W", "N", "P", "F", "9m"]]
    sk1 = start_kyoku(plain[:npl], oya=1, scores=score)
    sk1["kyoku"], sk1["dora_marker"] = 2, "1p"
    tehai = (["1m", "2m", "3m", "4m", "5m", "6m", "7m", "8m", "9m", "1p", "2p", "3p", "1s"] if npl == 4 else
             ["1p", "2p", "3p", "4p", "5p", "6p", "7p", "8p", "9p", "1s", "2s", "3s", "9m"])
    sk2 = start_kyoku([tehai] + [["1z"] * 13 for _ in range(npl - 1)], oya=0, scores=score)
    sk2["kyoku"], sk2["dora_marker"] = 2, "9s"
    _run_streams(mode, [[sk1, {"type": "tsumo", "actor": 1, "pai": "5p"}], [sk2, {"type": "tsumo", "actor": 0, "pai": "E"}]], prepare=deplete)
