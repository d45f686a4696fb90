"""Host-side MJAI replay API: MjaiReplay.from_jsonl / num_rounds / take_kyokus / events / grp_features
against the expectations of the reference's own tests (tests/test_mjai_replay.py:11-104) and its real hanchan log."""
import gzip
import json
import os

import pytest

from riichienv_amd.replay import MjaiReplay

HERE = os.path.dirname(__file__)


@pytest.fixture
def sample():  # tests/test_mjai_replay.py:11-37
    tehai = ["1s", "1s", "1s", "2s", "3s", "4s", "5s", "6s", "7s", "8s", "9s", "9s", "9s"]
    return [
        {"type": "start_game", "names": ["A", "B", "C", "D"], "id": "test_game"},
        {"type": "start_kyoku", "bakaze": "E", "kyoku": 1, "honba": 0, "kyoutaku": 0, "oya": 0,
         "scores": [25000, 25000, 25000, 25000], "dora_marker": "1m", "tehais": [tehai] * 4},
        {"type": "tsumo", "actor": 0, "pai": "2m"},
        {"type": "dahai", "actor": 0, "pai": "2m", "tsumogiri": True},
        {"type": "ryukyoku", "reason": "test"},
        {"type": "end_kyoku"},
        {"type": "end_game"},
    ]


def test_jsonl_plain_and_gzip(tmp_path, sample):  # tests/test_mjai_replay.py:40-76
    plain, gz = tmp_path / "t.jsonl", tmp_path / "t.jsonl.gz"
    plain.write_text("".join(json.dumps(e) + "\n" for e in sample))
    with gzip.open(gz, "wt") as f:
        f.write("".join(json.dumps(e) + "\n" for e in sample))
    for path in (plain, gz):
        r = MjaiReplay.from_jsonl(str(path))
        assert r.num_rounds() == 1
        k = list(r.take_kyokus())
        assert len(k) == 1 and len(k[0].events()) == 4 and k[0].events()[0]["name"] == "NewRound"
        f = k[0].grp_features()
        assert set(f) >= {"chang", "ju", "ben", "liqibang", "scores", "end_scores", "delta_scores", "wliqi"}
        assert f["scores"] == f["end_scores"] == [25000] * 4 and f["delta_scores"] == [0] * 4
    with pytest.raises(ValueError):
        MjaiReplay.from_jsonl(str(plain), rule="nope")


def test_real_hanchan_log():  # tests/test_mjai_replay.py:79-102 on tests/data/126_204_0_mjai.jsonl
    r = MjaiReplay.from_jsonl(os.path.join(HERE, "golden", "126_204_0_mjai.jsonl"))
    assert r.num_rounds() == 12
    k = list(r.take_kyokus())
    assert len(k) == 12
    f0 = k[0].grp_features()
    assert f0["scores"] == [25000] * 4 and f0["end_scores"] == [21000, 22000, 23000, 34000]
    assert f0["delta_scores"] == [-4000, -3000, -2000, 9000] and "wliqi" in f0
    for i in range(11):
        assert k[i + 1].grp_features()["scores"] == k[i].grp_features()["end_scores"]
    # the last round has no successor: its end scores come from the hora / ryukyoku deltas and the accepted deposits
    last = k[-1].grp_features()
    assert sum(last["end_scores"]) + 1000 * 0 <= 100000 and len(last["delta_scores"]) == 4


def test_wliqi_and_double_ron_end_scores():
    """mjai_replay.rs:412-418 (double riichi = riichi on the first discard with no call before) and :581-599 (the first
    hora starts from the scores minus accepted deposits, later hora events add up)"""
    tehai = ["1s"] * 13
    log = [
        {"type": "start_kyoku", "bakaze": "S", "kyoku": 3, "honba": 1, "kyotaku": 2, "oya": 2, "scores": [30000, 20000, 25000, 25000],
         "dora_marker": "1m", "tehais": [tehai] * 4},
        {"type": "tsumo", "actor": 2, "pai": "2m"}, {"type": "reach", "actor": 2},
        {"type": "dahai", "actor": 2, "pai": "2m", "tsumogiri": True}, {"type": "reach_accepted", "actor": 2},
        {"type": "tsumo", "actor": 3, "pai": "3m"}, {"type": "dahai", "actor": 3, "pai": "3m", "tsumogiri": True},
        {"type": "pon", "actor": 1, "target": 3, "pai": "3m", "consumed": ["3m", "3m"]},
        {"type": "reach", "actor": 1}, {"type": "dahai", "actor": 1, "pai": "4m", "tsumogiri": False},
        {"type": "hora", "actor": 2, "target": 1, "deltas": [0, -8000, 11300, 0]},
        {"type": "hora", "actor": 0, "target": 1, "deltas": [2000, -2000, 0, 0]},
        {"type": "end_kyoku"},
    ]
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rmj_wliqi_test.jsonl")
    with open(path, "w") as f:
        f.write("".join(json.dumps(e) + "\n" for e in log))
    k = list(MjaiReplay.from_jsonl(path).take_kyokus())[0]
    f = k.grp_features()
    assert (f["chang"], f["ju"], f["ben"], f["liqibang"]) == (1, 2, 1, 2)
    assert f["wliqi"] == [False, False, True, False]                 # seat 1 declared after a call and not on a first discard
    assert f["end_scores"] == [32000, 10000, 25000 + 11300 - 1000, 25000]   # seat 1's riichi was never accepted: no deposit
    names = [e["name"] for e in k.events()]
    assert names == ["NewRound", "DealTile", "DiscardTile", "DealTile", "DiscardTile", "ChiPengGang", "DiscardTile", "Hule"]
    assert len(k.events()[-1]["data"]["hules"]) == 2                 # consecutive hora events are one Hule action
