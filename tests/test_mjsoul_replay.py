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


def _oracle_tenpai(cases):
    from oracle import oracle

    return [bool(r.is_tenpai) for r in oracle.eval_hands(list(cases))] if cases else []


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
    plain = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls), tenpai=_oracle_tenpai)
    ctxs = _oracle_eval([c for k in plain.take_kyokus() for c in k.take_win_result_contexts()])
    exp = {i: dict(count=c.actual.han, fu=c.actual.fu, fans=list(c.actual.yaku[: c.actual.n_yaku])) for i, c in enumerate(ctxs)}
    n = len(ctxs)
    assert n >= 3
    # the same games without the wall (indicators from the records' own `doras` lists, ura indicators from li_doras)
    for with_wall in (True, False):
        r = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls, with_paishan=with_wall, expectations=exp), tenpai=_oracle_tenpai)
        got = r.verify(evaluate=_oracle_eval)
        assert got == (n, 0), (with_wall, got)
    exp[0]["fu"] += 10
    exp[1]["fans"] = exp[1]["fans"] + [1] if 1 not in exp[1]["fans"] else [y for y in exp[1]["fans"] if y != 1]
    assert MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls, expectations=exp), tenpai=_oracle_tenpai).verify(evaluate=_oracle_eval)[1] >= 1


def _records_with_points(events, walls):
    """the records of a played game whose Hule entries carry what Mahjong Soul writes: yaku ids, the yakuman flag and the
    points without honba (evaluated by the oracle from the reconstructed contexts)"""
    plain = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls), tenpai=_oracle_tenpai)
    ctxs = _oracle_eval([c for k in plain.take_kyokus() for c in k.take_win_result_contexts()])
    exp = {i: dict(count=c.actual.han, fu=c.actual.fu, fans=list(c.actual.yaku[: c.actual.n_yaku]), yiman=bool(c.actual.yakuman),
                   point_rong=c.actual.ron_agari, point_zimo_qin=c.actual.tsumo_agari_oya, point_zimo_xian=c.actual.tsumo_agari_ko)
           for i, c in enumerate(ctxs)}
    return to_mjsoul_rounds(events, walls, expectations=exp)


def _per_round_records(rounds):
    """every round as the last round of its own one-round record: the walker is checked on all rounds, not only the final one"""
    return [[r] for r in rounds]


@pytest.mark.parametrize("mode,seed", [(2, 1), (2, 2), (2, 5), (2, 7), (5, 2), (5, 3), (5, 4)])
def test_game_end_scores_of_played_games(mode, seed):
    """from_dict's game_end_scores (mjsoul_replay.rs:259-339: the last round replayed through apply_log_action): the scores the
    walker ends with are the scores the oracle's game ended with; and, round by round, the start scores of the next round."""
    events, walls, final = play_logged_game(mode, seed, with_scores=True)
    rounds = _records_with_points(events, walls)
    r = MjSoulReplay.from_dict({"data": rounds}, tenpai=_oracle_tenpai)
    ks = list(r.take_kyokus())
    assert ks[-1].game_end_scores == final and ks[0].game_end_scores == final and ks[-1].end_scores == final
    f = ks[0].take_grp_features()
    order = sorted(range(len(final)), key=lambda i: (-final[i], i))
    assert [f["final_ranks"][s] for s in order] == list(range(len(final)))
    assert f["round_initial_scores"] == ks[0].scores and f["round_end_scores"] == ks[1].scores
    assert f["round_delta_scores"] == [b - a for a, b in zip(ks[0].scores, ks[1].scores)]
    assert [len(f[f"player{i}_initial_hand_tids"]) for i in range(len(final))] == [14 if i == ks[0].ju else 13 for i in range(len(final))]
    # each round alone: one batch over all of them (from_dicts), end of round i = start of round i + 1
    singles = MjSoulReplay.from_dicts(_per_round_records(rounds), tenpai=_oracle_tenpai)
    ends = [x.rounds[0].game_end_scores for x in singles]
    kinds = set()
    for i in range(len(ks) - 1):
        assert ends[i] == ks[i + 1].scores, (i, ends[i], ks[i + 1].scores, [a["name"] for a in ks[i].actions][-2:])
        kinds.add(ks[i].actions[-1]["name"])
    assert ends[-1] == final
    assert "Hule" in kinds


def test_grp_features_of_an_mjai_kyoku_have_no_game_end_scores(tmp_path):
    """mjai_replay.rs:266: MJAI rounds carry no game_end_scores - final_ranks falls back to the round's own end ranks"""
    events, walls = play_logged_game(2, 1)
    p = tmp_path / "g.jsonl"
    write_jsonl(p, events)
    k = next(iter(MjaiReplay.from_jsonl(str(p)).take_kyokus()))
    f = k.take_grp_features()
    assert k.game_end_scores is None and f["final_ranks"] == f["round_end_ranks"]
    assert f["round_delta_ranks"] == [e - s for s, e in zip(f["round_initial_ranks"], f["round_end_ranks"])]


def _round(scores, actions, oya=0, ben=0, liqibang=0, hands=None):
    n = len(scores)
    t13 = ["1m", "4m", "7m", "1p", "4p", "7p", "1s", "4s", "7s", "1z", "2z", "3z", "4z"]          # far from tenpai
    d = dict(scores=list(scores), dora_marker="1s", chang=0, ju=oya, ben=ben, liqibang=liqibang)
    for i in range(4):
        d[f"tiles{i}"] = list((hands or {}).get(i, t13 + ["1m"] * (i == oya))) if i < n else []
    return [{"name": "NewRound", "data": d}] + actions


def _hule(seat, zimo, **kw):
    h = dict(seat=seat, hu_tile="1m", zimo=zimo, count=1, fu=30, fans=[{"id": y, "val": 1} for y in kw.pop("fans", [1])], hand=[],
             yiman=False, point_rong=0, point_zimo_qin=0, point_zimo_xian=0)
    h.update(kw)
    return h


def _end(actions, **kw):
    r = MjSoulReplay.from_dict([_round(kw.pop("scores", [25000] * 4), actions, **kw)], tenpai=_oracle_tenpai)
    return r.rounds[0].game_end_scores


def test_log_walker_rules_read_from_the_reference():
    """apply_log_action's rules that the played games do not reach, as hand-made records (state/event_handler.rs:332-891,
    state_3p/event_handler.rs:365-840)"""
    D = lambda s, t, **kw: {"name": "DiscardTile", "data": dict(seat=s, tile=t, **kw)}        # noqa: E731
    T = lambda s, t: {"name": "DealTile", "data": dict(seat=s, tile=t)}                        # noqa: E731
    P = lambda s, t, frm: {"name": "ChiPengGang", "data": dict(seat=s, type=1, tiles=[t, t, t], froms=[s, s, frm])}   # noqa: E731
    H = lambda *hs: {"name": "Hule", "data": {"hules": list(hs)}}                              # noqa: E731
    # a Ron on the riichi discard voids the deposit; of a double Ron only the first winner takes honba and the sticks
    got = _end([D(0, "1m", is_liqi=True), H(_hule(1, False, point_rong=2000), _hule(2, False, point_rong=3900))], ben=2, liqibang=1)
    assert got == [25000 - 2000 - 600 - 3900, 25000 + 2000 + 600 + 1000, 25000 + 3900, 25000]
    # the deposit stands once the next draw confirms it; a tsumo pays qin / xian + 100 per honba each
    got = _end([D(0, "1m", is_liqi=True), T(1, "9s"), H(_hule(1, True, point_zimo_qin=2000, point_zimo_xian=1000))], ben=1)
    assert got == [25000 - 1000 - 2100, 25000 + 2100 + 1100 + 1100 + 1000, 25000 - 1100, 25000 - 1100]
    # a second riichi flag of the same seat does not cost another deposit (4P: `if !riichi_declared`)
    got = _end([D(0, "1m", is_liqi=True), T(1, "9s"), D(1, "9s"), T(0, "2s"), D(0, "2s", is_liqi=True), T(1, "3s"), {"name": "LiuJu", "data": {}}])
    assert got == [24000, 25000, 25000, 25000]
    # abortive draw: the pending deposit is taken with four players, left alone with three
    assert _end([D(0, "1m", is_liqi=True), {"name": "LiuJu", "data": {}}]) == [24000, 25000, 25000, 25000]
    assert _end([D(0, "1m", is_liqi=True), {"name": "LiuJu", "data": {}}], scores=[35000] * 3) == [35000] * 3
    # pao: the third dragon pon makes the discarder liable - a yakuman tsumo is paid by that seat alone (+ all the honba)
    dragons = [D(0, "5z"), P(1, "5z", 0), D(1, "9s"), T(2, "6z"), D(2, "6z"), P(1, "6z", 2), D(1, "8s"), T(2, "7z"), D(2, "7z"), P(1, "7z", 2), D(1, "7s"), T(2, "1s"),
               D(2, "1s"), T(3, "2s"), D(3, "2s"), T(0, "3s"), D(0, "3s"), T(1, "1m")]
    tsumo = H(_hule(1, True, yiman=True, fans=[37], point_zimo_qin=16000, point_zimo_xian=8000))
    assert _end(dragons + [tsumo], ben=1) == [25000, 25000 + 32000 + 300, 25000 - 32000 - 300, 25000]
    # ... with a second, unrelated yakuman the other half is split as a normal tsumo (oya half, the others a quarter each)
    two = H(_hule(1, True, yiman=True, fans=[37, 39], point_zimo_qin=32000, point_zimo_xian=16000))
    assert _end(dragons + [two]) == [25000 - 16000, 25000 + 64000, 25000 - 32000 - 8000, 25000 - 8000]
    # ... and a Ron is shared by the liable seat and the discarder, honba on the liable seat
    ron = dragons[:-1] + [T(1, "4s"), D(1, "4s"), T(2, "5s"), D(2, "5s"), T(3, "1m"), D(3, "1m"), H(_hule(1, False, yiman=True, fans=[37], point_rong=32000))]
    assert _end(ron, ben=2) == [25000, 25000 + 32000 + 600, 25000 - 16000 - 600, 25000 - 16000]
    # nagashi mangan: only terminal / honor discards and none of them called; every eligible seat is paid a mangan tsumo
    base = [D(0, "1m"), T(1, "5s"), D(1, "5s"), T(2, "5p"), D(2, "5p"), T(3, "4s"), D(3, "4s"), {"name": "NoTile", "data": {}}]
    assert _end(base) == [25000 + 12000, 21000, 21000, 21000]
    called = [D(0, "1m"), P(1, "1m", 0), D(1, "5s"), T(2, "5p"), D(2, "5p"), T(3, "4s"), D(3, "4s"), {"name": "NoTile", "data": {}}]
    hands = {1: ["1m", "1m", "2p", "3p", "4p", "5s", "6s", "7s", "5s", "2s", "3s", "4s", "9p"]}
    got = _end(called, hands=hands)           # nobody is eligible: seat 1 (1m pon + three runs + 9p) is the only tenpai hand
    assert got == [24000, 28000, 24000, 24000]
    # three players: a Hule flagged zimo on another seat's turn is a Ron on the last discard (state_3p/event_handler.rs:598-611)
    got = _end([D(0, "1p"), H(_hule(2, True, point_rong=8000, point_zimo_qin=4000, point_zimo_xian=2000))], scores=[35000] * 3, ben=1)
    assert got == [35000 - 8000 - 200, 35000, 35000 + 8000 + 200]
    got = _end([D(0, "1p"), {"name": "DealTile", "data": dict(seat=1, tile="9s")}, H(_hule(1, True, point_zimo_qin=4000, point_zimo_xian=2000))],
               scores=[35000] * 3, ben=1)
    assert got == [35000 - 4100, 35000 + 4100 + 2100, 35000 - 2100]


_MJAI_KEYS = {"start_kyoku": ["bakaze", "kyoku", "honba", "kyotaku", "oya", "scores", "dora_marker", "tehais"], "tsumo": ["actor", "pai"],
              "dahai": ["actor", "pai", "tsumogiri"], "reach": ["actor"], "reach_accepted": ["actor"], "chi": ["actor", "target", "pai", "consumed"],
              "pon": ["actor", "target", "pai", "consumed"], "daiminkan": ["actor", "target", "pai", "consumed"], "ankan": ["actor", "consumed"],
              "kakan": ["actor", "pai", "consumed"], "kita": ["actor"], "dora": ["dora_marker"], "hora": ["actor", "target"]}


def _norm_mjai(events):
    """the fields a record can carry; a discard of a tile with the drawn tile's NAME counts as tsumogiri (records hold names)"""
    out, drawn = [], {}
    for e in events:
        d = {k: e.get(k) for k in _MJAI_KEYS.get(e["type"], [])}
        d["type"] = e["type"]
        if "consumed" in d:
            d["consumed"] = sorted(d["consumed"])
        if e["type"] == "start_kyoku":
            d["tehais"] = [sorted(h) for h in e["tehais"][: len(e["scores"])]]
        if e["type"] == "tsumo":
            drawn[e["actor"]] = e["pai"]
        elif e["type"] == "dahai":
            d["tsumogiri"] = drawn.get(e["actor"]) == e["pai"]
            drawn[e["actor"]] = None
        elif e["type"] in ("chi", "pon", "daiminkan", "ankan", "kakan", "kita"):
            drawn[e["actor"]] = None
        out.append(d)
    return out


@pytest.mark.parametrize("mode,seed", [(2, 1), (2, 5), (2, 7), (5, 2), (5, 4)])
def test_records_convert_back_to_the_mjai_log_they_came_from(mode, seed):
    """MjSoulReplay.to_mjai (the input of ReplayBatch for Mahjong Soul records): the MJAI log of an oracle-played game, written
    as a record and read back, is the same event stream - deals, draws, discards with their tsumogiri flag, calls with their
    exact tiles (red fives), riichi declaration / acceptance (none after a Ron on the riichi discard), wins with their targets.
    Indicators of open kans: the record lists them on the discard, so they follow it (the log reveals them just before)."""
    events, walls = play_logged_game(mode, seed)
    r = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls), tenpai=_oracle_tenpai)
    a, b = _norm_mjai(events), _norm_mjai(r.to_mjai())
    strip = lambda xs: [x for x in xs if x["type"] != "dora"]      # noqa: E731
    assert strip(a) == strip(b)

    def dora_positions(xs):
        pos, k = [], 0
        for x in xs:
            if x["type"] == "dora":
                pos.append((x["dora_marker"], k))
            else:
                k += 1
        return pos

    pa, pb = dora_positions(a), dora_positions(b)
    assert [m for m, _ in pa] == [m for m, _ in pb] and len(pa) >= 1
    assert all(kb - ka in (0, 1) for (_, ka), (_, kb) in zip(pa, pb))
    # the reference's own step iterator never reveals the listed indicators
    quiet = r.to_mjai(reveal_listed_doras=False)
    assert not any(e["type"] == "dora" for e in quiet) and len(quiet) == len(strip(b))
