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
