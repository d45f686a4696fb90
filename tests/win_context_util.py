"""Shared by the CPU (oracle) and GPU tests of the win-context replay: the expectations a real log carries for each of
its hora events, derived from the log's own deltas (the reference emits no han / fu / yaku in hora events)."""
import json
import os

from riichienv_amd.replay import MjaiReplay

HERE = os.path.dirname(os.path.abspath(__file__))
LOG = os.path.join(HERE, "golden", "126_204_0_mjai.jsonl")


def contexts_with_deltas():
    """[(kyoku, context, hora event)] of the reference's real hanchan log (tests/data/126_204_0_mjai.jsonl)."""
    r = MjaiReplay.from_jsonl(LOG)
    out = []
    for k in r.take_kyokus():
        horas = [e for e in k.mjai_events if e.get("type") == "hora"]
        ctxs = list(k.take_win_result_contexts())
        assert len(ctxs) == len(horas)
        out += [(k, c, h) for c, h in zip(ctxs, horas)]
    return out


def check_points(k, c, hora, actual):
    """The payments in the log's deltas must be the evaluator's points (+ honba): state/mod.rs:750-893 (tsumo),
    :945-1142 (ron).  The iterator passes honba = 0 (replay/mod.rs:2040), so honba is added here."""
    d = hora["deltas"]
    oya = k.ju
    assert actual.is_win, (k.chang, k.ju, k.ben, c.seat)
    if c.conditions["tsumo"]:
        for i in range(len(d)):
            if i == c.seat:
                continue
            pay = actual.tsumo_agari_oya if i == oya else actual.tsumo_agari_ko
            assert -d[i] == pay + 100 * k.ben, (k.chang, k.ju, i, d, pay)
    else:
        tgt = hora["target"]
        hb = 100 * (len(d) - 1) * k.ben          # 300 per honba with four players, 200 with three
        assert -d[tgt] == actual.ron_agari + hb, (k.chang, k.ju, d, actual.ron_agari)
        assert d[c.seat] >= actual.ron_agari + hb     # + riichi deposits
    assert bool(hora.get("ura_markers")) == bool(c.ura_indicators) or not c.conditions["riichi"]


def synthetic_log():
    """A hand-made round exercising the flag bookkeeping of the iterator (replay/mod.rs:1741-2036): double riichi with
    ippatsu broken by a pon, a kakan robbed by Ron (chankan), the tile count."""
    t13 = ["1m", "2m", "3m", "4m", "5m", "6m", "7m", "8m", "9m", "1p", "1p", "E", "E"]
    return [
        {"type": "start_game"},
        {"type": "start_kyoku", "bakaze": "S", "kyoku": 2, "honba": 1, "kyotaku": 0, "oya": 1, "scores": [25000] * 4,
         "dora_marker": "3s", "tehais": [["2p", "3p", "2s", "3s", "5s", "5s", "6s", "7s", "8s", "W", "W", "W", "N"], t13,
                                         ["9s", "9s", "4p", "5p", "6p", "7p", "8p", "9p", "P", "P", "F", "F", "C"],
                                         ["1s", "1s", "4m", "2p", "2p", "6m", "6m", "7m", "8m", "S", "S", "S", "C"]]},
        {"type": "tsumo", "actor": 1, "pai": "C"},
        {"type": "reach", "actor": 1},
        {"type": "dahai", "actor": 1, "pai": "C", "tsumogiri": True},
        {"type": "reach_accepted", "actor": 1},
        {"type": "tsumo", "actor": 2, "pai": "1s"},
        {"type": "dahai", "actor": 2, "pai": "1s", "tsumogiri": True},
        {"type": "pon", "actor": 3, "target": 2, "pai": "1s", "consumed": ["1s", "1s"]},
        {"type": "dahai", "actor": 3, "pai": "C", "tsumogiri": False},
        {"type": "tsumo", "actor": 0, "pai": "1p"},
        {"type": "dahai", "actor": 0, "pai": "N", "tsumogiri": False},
        {"type": "tsumo", "actor": 1, "pai": "9s"},
        {"type": "dahai", "actor": 1, "pai": "9s", "tsumogiri": True},
        {"type": "tsumo", "actor": 2, "pai": "N"},
        {"type": "dahai", "actor": 2, "pai": "N", "tsumogiri": True},
        {"type": "tsumo", "actor": 3, "pai": "1s"},
        {"type": "kakan", "actor": 3, "pai": "1s", "consumed": ["1s", "1s", "1s"]},
        {"type": "hora", "actor": 0, "target": 3, "deltas": [0, 0, 0, 0], "ura_markers": []},
        {"type": "end_kyoku"},
        {"type": "end_game"},
    ]


def write_jsonl(path, events):
    with open(path, "w") as f:
        for e in events:
            f.write(json.dumps(e) + "\n")
