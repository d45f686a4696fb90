"""Row N2, Mahjong Soul records (replay/mjsoul_replay.rs:20-688): MjSoulReplay reads the action records into the same Kyoku
objects as MjaiReplay, and the WinResultContextIterator's wall-dependent half (replay/mod.rs:1634-1724: indicators recomputed
from `paishan`, ura indicators from the wall, indicators of open kans pending until the discard, replacement draws marked by
their `doras` list) reconstructs every win.  The reference ships no MjSoul record: the records come from games the oracle
played (tests/mjsoul_util.py), and what pins the reader is independent of it - the payments of every win computed from the
reconstructed context must be the deltas the game's MJAI log recorded, and indicators / ura indicators derived from the wall
must be the ones the log's dora events / hora events carry.  The evaluator here is the oracle (CPU test)."""
import gzip
import json

import pytest

from riichienv_amd import abi
from riichienv_amd.replay import MjaiReplay, MjSoulReplay, mjsoul_tile_to_mjai
from tests.mjsoul_util import play_logged_game, to_mjsoul_rounds
from tests.win_context_util import check_points, write_jsonl


def _oracle_eval(ctxs):
    from oracle import oracle

    ctxs = list(ctxs)
    if ctxs:
        for c, r in zip(ctxs, oracle.eval_hands([c.hand_case() for c in ctxs])):
            c.actual = r
    return ctxs


def _replacement_draw_wins(events):
    """per hora event: was it a tsumo on the replacement draw of a kan / kita"""
    out, hist = [], []
    for e in events:
        if e["type"] == "hora":
            prev = [x for x in hist if x["type"] not in ("dora", "reach", "reach_accepted", "hora")]
            out.append(e["actor"] == e["target"] and len(prev) >= 2 and prev[-1]["type"] == "tsumo" and
                       prev[-2]["type"] in ("ankan", "kakan", "daiminkan", "kita"))
        hist.append(e)
    return out


def test_tile_names_and_schema_defaults():
    assert [mjsoul_tile_to_mjai(t) for t in ("1m", "0m", "5p", "0s", "1z", "4z", "7z")] == ["1m", "5mr", "5p", "5sr", "E", "N", "C"]
    t13 = ["1m"] * 13
    rounds = [[
        {"name": "NewRound", "data": {"scores": [25000] * 4, "dora_indicators": ["3s", "0p"], "doras": ["9m"], "tiles0": t13 + ["2m"],
                                      "tiles1": t13, "tiles2": t13, "tiles3": t13, "chang": 1, "ju": 2, "honba": 3, "liqibang": 1}},
        {"name": "DiscardTile", "data": {"seat": 0, "tile": "2m", "is_liqi": True, "is_wliqi": True}},
        {"name": "DealTile", "data": {"seat": 1, "tile": "0p", "dora_marker": "7z", "left_tile_count": 68}},
        {"name": "ChiPengGang", "data": {"seat": 2, "type": 2, "tiles": ["1m", "1m", "1m", "1m"], "froms": [2, 2, 2, 1]}},
        {"name": "ChiPengGang", "data": {"seat": 3, "type": 9, "tiles": ["1m", "2m", "3m"], "froms": [3, 3, 2]}},
        {"name": "AnGangAddGang", "data": {"seat": 1, "type": 3, "tiles": "5p"}},
        {"name": "AnGangAddGang", "data": {"seat": 1, "type": 2, "tiles": "0s"}},
        {"name": "BaBei", "data": {"seat": 0}},
        {"name": "dora", "data": {"dora_marker": "1z"}},
        {"name": "LiuJu", "data": {}},
        {"name": "SomethingNew", "data": {"x": 1}},
        {"name": "Hule", "data": {"hules": [{"seat": 1, "hu_tile": "0p", "zimo": True, "count": 3, "fu": 40,
                                             "fans": [{"id": 1, "val": 1}, {"id": 31, "val": 2}, {"id": 7, "val": 0}, {"id": 8}],
                                             "hand": [], "li_doras": ["2z"], "ura_dora_indicators": ["3z"], "yiman": False,
                                             "point_rong": 0, "point_zimo_qin": 2000, "point_zimo_xian": 1000}]}},
    ], [{"name": "NewRound", "data": {"scores": [24000, 26000, 25000, 25000], "dora_marker": "1p", "tiles0": t13, "tiles1": t13,
                                      "tiles2": t13, "tiles3": t13, "chang": 1, "ju": 3, "ben": 0, "liqibang": 0, "left_tile_count": 69}}]]
    r = MjSoulReplay.from_dict({"head": {}, "data": rounds})
    assert r.num_rounds() == 2
    k, k2 = list(r.take_kyokus())
    # dora_indicators win over doras, honba is the alias of ben, the tile count defaults to 70 (mjsoul_replay.rs:464-493)
    assert k.doras == ["3s", "5pr"] and k.ben == 3 and k.left_tile_count == 70 and (k.chang, k.ju, k.liqibang) == (1, 2, 1)
    assert k2.doras == ["1p"] and k2.left_tile_count == 69 and k.end_scores == k2.scores and k2.end_scores == k2.scores
    assert k.wliqi == [True, False, False, False] and len(k.hands[0]) == 14
    names = [a["name"] for a in k.actions]
    assert names == ["Other", "DiscardTile", "DealTile", "ChiPengGang", "ChiPengGang", "AnGangAddGang", "AnGangAddGang", "BaBei", "Dora",
                     "LiuJu", "Other", "Hule"]
    a = k.actions
    assert a[1]["doras"] is None and a[2]["doras"] == ["C"] and a[2]["left_tile_count"] == 68        # dora_marker fallback
    assert a[3]["meld_type"] == "Daiminkan" and a[4]["meld_type"] == "Chi"                            # unknown code -> Chi
    assert a[5]["meld_type"] == "Ankan" and a[6]["meld_type"] == "Kakan" and a[6]["tiles"] == ["5sr"]
    assert a[9]["lj_type"] == 0 and a[9]["tiles"] == []
    h = a[11]["hules"][0]
    assert h["fans"] == [1, 31] and h["li_doras"] == [abi.mjai_to_tid("W", True)] and h["hu_tile"] == 52 and h["zimo"]
    with pytest.raises(ValueError):
        MjSoulReplay.from_dict({"head": {}})
    with pytest.raises(ValueError):
        MjSoulReplay.from_dict("x")


@pytest.mark.parametrize("mode,seed", [(2, 1), (2, 2), (2, 5), (5, 2), (5, 3)])
def test_records_of_played_games_reconstruct_every_win(mode, seed, tmp_path):
    events, walls = play_logged_game(mode, seed)
    horas = [e for e in events if e["type"] == "hora"]
    assert horas
    p = tmp_path / "g.jsonl"
    write_jsonl(p, events)
    ka = list(MjaiReplay.from_jsonl(str(p)).take_kyokus())
    gz = tmp_path / "g.json.gz"
    with gzip.open(gz, "wt") as f:
        json.dump({"rounds": to_mjsoul_rounds(events, walls)}, f)
    r = MjSoulReplay.from_json(str(gz))
    kb = list(r.take_kyokus())
    assert r.num_rounds() == len(ka)
    for i, (a, b) in enumerate(zip(ka, kb)):
        assert (a.scores, a.chang, a.ju, a.ben, a.liqibang, a.wliqi) == (b.scores, b.chang, b.ju, b.ben, b.liqibang, b.wliqi), i
        if i + 1 < len(ka):
            assert a.end_scores == b.end_scores, i
        assert [sorted(h) for h in a.hands] != [] and len(b.hands) == len(a.hands)
    # every win: same evaluator inputs as the MJAI path, indicators from the wall = indicators of the log, payments = deltas
    ca = [(k, c) for k in ka for c in k.take_win_result_contexts()]
    cb = [(k, c) for k in kb for c in k.take_win_result_contexts()]
    assert len(ca) == len(cb) == len(horas)
    repl = _replacement_draw_wins(events)
    _oracle_eval([c for _, c in cb])
    for (k, c1), (k2, c2), h, rin in zip(ca, cb, horas, repl):
        assert (c1.seat, c1.agari_tile, sorted(c1.tiles)) == (c2.seat, c2.agari_tile, sorted(c2.tiles))
        assert [(m["meld_type"], sorted(m["tiles"])) for m in c1.melds] == [(m["meld_type"], sorted(m["tiles"])) for m in c2.melds]
        assert c1.dora_indicators == c2.dora_indicators, (k.chang, k.ju, c1.dora_indicators, c2.dora_indicators)
        assert c1.ura_indicators == c2.ura_indicators
        d1, d2 = dict(c1.conditions), dict(c2.conditions)
        assert d2["rinshan"] == rin and not d1["rinshan"]     # (an MJAI draw carries nothing that marks a replacement draw)
        assert {x: d1[x] for x in d1 if x not in ("haitei", "rinshan")} == {x: d2[x] for x in d2 if x not in ("haitei", "rinshan")}
        assert d2["haitei"] == (d1["haitei"] and not rin)
        check_points(k, c2, h, c2.actual)


def test_verify_counts_mismatches():
    events, walls = play_logged_game(2, 1)
    plain = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls))
    ctxs = _oracle_eval([c for k in plain.take_kyokus() for c in k.take_win_result_contexts()])
    exp = {i: dict(count=c.actual.han, fu=c.actual.fu, fans=list(c.actual.yaku[: c.actual.n_yaku])) for i, c in enumerate(ctxs)}
    n = len(ctxs)
    assert n >= 3
    # the same games without the wall (indicators from the records' own `doras` lists, ura indicators from li_doras)
    for with_wall in (True, False):
        r = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls, with_paishan=with_wall, expectations=exp))
        got = r.verify(evaluate=_oracle_eval)
        assert got == (n, 0), (with_wall, got)
    exp[0]["fu"] += 10
    exp[1]["fans"] = exp[1]["fans"] + [1] if 1 not in exp[1]["fans"] else [y for y in exp[1]["fans"] if y != 1]
    assert MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls, expectations=exp)).verify(evaluate=_oracle_eval)[1] >= 1
