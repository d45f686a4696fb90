"""Event streams for the MJAI-ingestion tests (row N1): the flows of the reference's tests/env/test_apply_event.py
(full-information variants: every hand is given, this build does not ingest masked "?" tiles) and a stream generator
that replays a finished rollout's own MJAI log back through apply_event."""
import json

TEHAIS_4P = [
    ["1m", "2m", "3m", "4m", "5m", "6m", "7m", "8m", "9m", "1p", "2p", "3p", "4p"],
    ["1m", "1m", "4p", "5p", "6p", "7p", "8p", "9p", "1s", "2s", "3s", "4s", "5s"],
    ["1z", "2z", "3z", "4z", "5z", "6z", "7z", "1s", "2s", "3s", "7s", "8s", "9s"],
    ["4s", "5s", "6s", "7s", "8s", "9s", "4m", "5m", "6m", "7m", "8m", "9m", "1z"],
]
TEHAIS_3P = [
    ["1m", "9m", "1p", "2p", "3p", "4p", "5p", "6p", "7s", "8s", "9s", "1z", "2z"],
    ["1p", "1p", "7p", "8p", "9p", "1s", "2s", "3s", "4s", "5s", "6s", "3z", "4z"],
    ["5z", "6z", "7z", "7s", "8s", "9s", "7p", "8p", "9p", "1m", "9m", "1z", "2z"],
]
CHI_TEHAIS = [
    ["1s", "2s", "3s", "4s", "5s", "6s", "7s", "8s", "9s", "1z", "2z", "3z", "3m"],
    ["4m", "5m", "6m", "1p", "2p", "3p", "4p", "5p", "6p", "7p", "8p", "9p", "1z"],
    ["1z", "2z", "3z", "4z", "5z", "6z", "7z", "1s", "2s", "3s", "7s", "8s", "9s"],
    ["4s", "5s", "6s", "7s", "8s", "9s", "7m", "8m", "9m", "7m", "8m", "9m", "2z"],
]


def start_kyoku(tehais, oya=0, scores=None):
    np_ = len(tehais)
    return {"type": "start_kyoku", "bakaze": "E", "dora_marker": "2p", "kyoku": 1, "honba": 0, "kyotaku": 0, "oya": oya,
            "scores": scores or [25000 if np_ == 4 else 35000] * np_, "tehais": tehais}


def log_to_events(log_lines):
    """A game's own MJAI log (JSON strings) as an event stream; hora/ryukyoku/end_* carry no state payload."""
    return [json.loads(x) for x in log_lines]


# tests/env/test_apply_event.py:513-530 (TestReplayFuriten): seat 1 is tenpai on 2m / 3m (123456789s + 111m 2m), seats 0 and 2 hold 3m
FURITEN_TEHAIS = [["3m", "3m", "5m", "6m", "7m", "8m", "9m", "1p", "2p", "3p", "4p", "5p", "6p"],
                  ["1s", "2s", "3s", "4s", "5s", "6s", "7s", "8s", "9s", "1m", "1m", "1m", "2m"],
                  ["1z", "2z", "3z", "4z", "5z", "6z", "7z", "7p", "8p", "9p", "3m", "8s", "9s"],
                  ["4s", "5s", "6s", "7s", "8s", "9s", "4m", "5m", "6m", "7m", "8m", "9m", "1z"]]


def _t(actor, pai):
    return {"type": "tsumo", "actor": actor, "pai": pai}


def _d(actor, pai, tsumogiri):
    return {"type": "dahai", "actor": actor, "pai": pai, "tsumogiri": tsumogiri}


def furiten_log(riichi):
    """The two logs of TestReplayFuriten (tests/env/test_apply_event.py:535-632)"""
    sk = {"type": "start_kyoku", "bakaze": "E", "kyoku": 1, "honba": 0, "kyoutaku": 0, "oya": 0, "scores": [25000] * 4, "dora_marker": "2p",
          "tehais": FURITEN_TEHAIS}
    if not riichi:
        body = [_t(0, "7p"), _d(0, "3m", False), _t(1, "4z"), _d(1, "4z", True), _t(2, "5z"), _d(2, "3m", False), _t(3, "3z"), _d(3, "3z", True)]
    else:
        body = [_t(0, "7p"), _d(0, "7p", True), _t(1, "4z"), {"type": "reach", "actor": 1}, _d(1, "4z", True), {"type": "reach_accepted", "actor": 1},
                _t(2, "5z"), _d(2, "5z", True), _t(3, "2z"), _d(3, "2z", True), _t(0, "1z"), _d(0, "3m", False), _t(1, "6z"), _d(1, "6z", True),
                _t(2, "7z"), _d(2, "3m", False), _t(3, "3z"), _d(3, "3z", True)]
    return [{"type": "start_game"}, sk] + body + [{"type": "ryukyoku", "reason": "yao9"}, {"type": "end_kyoku"}, {"type": "end_game"}]
