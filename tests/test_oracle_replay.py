"""Pins the ORACLE state machine on the reference's real-log fixture tests/data/126_204_0_mjai.jsonl
(1 hanchan, 12 kyoku): every logged decision must be in the oracle's legal list, and the oracle's emitted
MJAI events (tsumo/dahai/calls/dora/reach_accepted/hora deltas + ura markers/ryukyoku deltas) must equal the log."""
import json
import os

import pytest

from oracle import oracle
from tests.replay_util import ReplayDriver, build_wall, comparable, load_log, split_kyoku

WIND = {"E": 0, "S": 1, "W": 2, "N": 3}


def replay_all(make_env, golden_dir):
    events = load_log(os.path.join(golden_dir, "126_204_0_mjai.jsonl"))
    kyokus = split_kyoku(events)
    assert len(kyokus) == 12  # tests/test_mjai_replay.py:78-100
    n_hora = 0
    for k, kev in enumerate(kyokus):
        sk = kev[0]
        env = make_env()
        env.reset(wall=build_wall(kev), oya=sk["oya"], round_wind=WIND[sk["bakaze"]], scores=sk["scores"], honba=sk["honba"],
                  kyotaku=sk["kyotaku"])
        ReplayDriver(env).run_kyoku(kev)
        got = [json.loads(s) for s in env.log()]
        assert got[0]["type"] == "start_game"
        got = got[1:]
        end = next(i for i, e in enumerate(got) if e["type"] == "end_kyoku")
        got = got[: end + 1]
        assert [comparable(e) for e in got] == [comparable(e) for e in kev], k
        n_hora += sum(e["type"] == "hora" for e in got)
    assert n_hora == 9
    # tests/test_mjai_replay.py: kyoku 0 end scores
    k0 = kyokus[0]
    hora = next(e for e in k0 if e["type"] == "hora")
    assert [a + b for a, b in zip(k0[0]["scores"], hora["deltas"])] != []  # deltas present


def test_oracle_replays_reference_log(golden_dir):
    class Env(oracle.Game):
        pass

    replay_all(lambda: Env(game_mode=2, seed=1), golden_dir)
