"""N3 pins from the reference's SPECIFICATION TEXT: tests/golden/spec_tables.json is generated mechanically from the tables of
docs/FEATURE_ENCODING.md and docs/SEQUENCE_FEATURE_ENCODING.md (scripts/gen_spec_tables.py, run where /root/reference exists); the
oracle's encoders are checked against those offsets, counts and divisors on states of real play - the expectations are the published
tables, not constants typed next to the code they test.  (The device encoders are compared with the oracle byte for byte elsewhere.)"""
import json
import os
import re

import numpy as np
import pytest

from oracle import oracle
from oracle import seq_features as sf
from riichienv_amd import abi
from riichienv_amd.shard import game_seed

SPEC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spec_tables.json")))
FE, SE = SPEC["feature_encoding"], SPEC["sequence_encoding"]


def _row(name):
    (r,) = [c for c in FE["channels"] if c["name"] == name]
    return r


def _states(n_games=6, steps=420, greedy=True, mode=2):
    """(game, acting seat) pairs sampled along rollouts of the oracle (greedy play: melds, riichi, kans, wins occur)"""
    for g in range(n_games):
        o = oracle.Game(game_mode=mode, seed=game_seed(4242, g))
        o.reset()
        for k in range(steps):
            act, _, done = o.status()
            if done:
                break
            if k % 7 == g % 7:
                for s in range(4):
                    if (act >> s) & 1:
                        yield o, s
            o.step(o.greedy_actions(99, g, 96) if greedy else o.random_actions(99, g))


def test_channel_tables_tile_the_tensor_and_the_oracle_has_that_shape():
    assert FE["total_channels"] == 74 and sorted(ch for c in FE["channels"] for ch in range(c["lo"], c["hi"] + 1)) == list(range(74))
    o = oracle.Game(game_mode=2, seed=1)
    assert o.encode(0).shape == (FE["total_channels"], 34)
    assert FE["shapes"] == {"encode_discard_history_decay": [4, 34], "encode_yaku_possibility": [4, 21, 2], "encode_furiten_ron_possibility": [4, 21],
                            "encode_shanten_efficiency": [4, 4]}
    assert list(o.encode_yaku_possibility().shape) == FE["shapes"]["encode_yaku_possibility"]
    assert list(o.encode_furiten_ron_possibility().shape) == FE["shapes"]["encode_furiten_ron_possibility"]
    assert len(FE["yaku"]) == 19 and sum(y["hi"] - y["lo"] + 1 for y in FE["yaku"]) == 21


def test_broadcast_channels_and_their_divisors_on_real_states():
    """every row of the channel tables that names a divisor: value x divisor is the quantity the row describes, in relative seat
    order where the row says so, the same in all 34 columns where it says broadcast"""
    seen = 0
    with_calls = [0]
    for o, pid in _states():
        e = o.encode(pid)
        v = o.peek()
        ps = v.players
        rel = [(pid + k) % 4 for k in range(4)]
        for c in FE["channels"]:
            if c["broadcast"] and "at the tile index" not in json.dumps(c) and c["name"] not in ("Round Wind", "Self Wind"):
                blk = e[c["lo"]: c["hi"] + 1]
                assert (blk == blk[:, :1]).all(), (c["name"], "not the same in all columns")

        def per_seat(name, q):
            c = _row(name)
            assert c["relative_seat_order"] and c["hi"] - c["lo"] == 3
            for k in range(4):
                assert abs(e[c["lo"] + k][0] * c["divisor"] - q(rel[k])) < 1e-3 + 1e-6 * c["divisor"], (name, k)   # (f32 quotients)

        per_seat("Discard Counts (All Players)", lambda s: ps[s].n_discards)
        per_seat("Scores (0-100000)", lambda s: min(max(ps[s].score, 0), 100000))
        per_seat("Melds Count (Per Player)", lambda s: ps[s].n_melds)
        c = _row("Scores (0-30000)")
        for k in range(4):
            assert abs(e[c["lo"] + k][0] * c["divisor"] - min(max(ps[rel[k]].score, 0), 30000)) < 1e-2
        for name, q in (("Honba", v.honba), ("Riichi Sticks", v.riichi_sticks), ("Kyoku Index", v.kyoku_idx)):
            c = _row(name)
            assert abs(e[c["lo"]][0] * c["divisor"] - min(q, c["divisor"])) < 1e-3, name
        c = _row("Round Progress")
        assert re.sub(r"\s", "", c["formula"]) == "round_wind*4+kyoku_index"
        assert abs(e[c["lo"]][0] * c["divisor"] - min(v.round_wind * 4 + v.kyoku_idx, c["divisor"])) < 1e-3
        c = _row("Tiles Left in Wall")
        left = e[c["lo"]][0] * c["divisor"]
        # the row's "remaining tiles" are the tiles this seat cannot see, over the table's divisor (not capped at it).  The reference has
        # two bodies for the base block and they differ here: Observation.encode() counts every meld tile (observation/python.rs:568-587),
        # the base block of encode_extended() does not count a called tile twice (encode_base_into, observation/encode.rs:94-111)
        used = sum(p.n_discards for p in ps) + ps[pid].hand_len + v.n_dora
        called = 0
        for p in ps:
            for m in p.melds[: p.n_melds]:
                used += m.n_tiles
                called += 0 if m.called_tile < 0 else 1
        assert abs(left - (136 - used)) < 1e-3, (left, used)
        x = o.encode_extended(pid)
        assert abs(x[c["lo"]][0] * c["divisor"] - (136 - used + called)) < 1e-3
        keep = [ch for ch in range(74) if ch != c["lo"]]
        assert (x[keep] == e[keep]).all()                      # every other channel of the block is encode()'s
        with_calls[0] += called > 0
        for name in ("Dora Count (Per Player)",):
            c = _row(name)
            for k in range(4):
                x = e[c["lo"] + k][0] * c["divisor"]
                assert abs(x - round(x)) < 1e-3 and 0 <= round(x) <= c["divisor"]
        seen += 1
    assert seen > 150 and with_calls[0] > 20


def test_tile_channels_on_real_states():
    for o, pid in _states(n_games=4, steps=300):
        e = o.encode(pid)
        v = o.peek()
        P = v.players[pid]
        cnt = np.zeros(34, int)
        red = np.zeros(34, int)
        for t in P.hand[: P.hand_len]:
            cnt[t // 4] += 1
            if t in (16, 52, 88):
                red[t // 4] = 1
        c = _row("Hand")
        for k in range(4):                                    # "Ch 0: count >= 1, Ch 1: count >= 2, ..."
            assert (e[c["lo"] + k] == (cnt >= k + 1)).all()
        assert (e[_row("Red Tiles")["lo"]] == red).all()
        dora = np.zeros(34, int)
        for t in v.dora[: v.n_dora]:
            dora[t // 4] = 1
        assert (e[_row("Dora Indicators")["lo"]] == dora).all()
        c = _row("Rank")
        ranks = e[c["lo"]: c["hi"] + 1]
        assert set(np.unique(ranks)) <= {0.0, 1.0} and (ranks.sum(axis=0) == 1).all() and (ranks == ranks[:, :1]).all()
        sc = [p.score for p in v.players]
        # (the table says "based on scores"; ties share a rank: observation/python.rs:650-667 counts the strictly greater scores)
        assert int(np.argmax(ranks[:, 0])) == sum(1 for q in range(4) if sc[q] > sc[pid])
        for name, val in (("Round Wind", v.round_wind), ("Self Wind", (pid - v.oya) % 4)):
            row = e[_row(name)["lo"]]
            assert row.sum() == 1 and row[27 + val] == 1       # "1 at the tile index corresponding to the ... wind (27-30)"
        c = _row("Discards (Self, Recent 4)")
        d = [t // 4 for t in P.discards[: P.n_discards]]
        for k in range(4):                                     # most recent first
            want = np.zeros(34)
            if k < len(d):
                want[d[len(d) - 1 - k]] = 1
            assert (e[c["lo"] + k] == want).all()
        c = _row("Tsumogiri Flags")
        assert set(np.unique(e[c["lo"]: c["hi"] + 1])) <= {0.0, 1.0}


def test_aux_encoders_against_the_spec_lists():
    for o, pid in _states(n_games=3, steps=200):
        y = o.encode_yaku_possibility()
        assert set(np.unique(y)) <= {0.0, 1.0}                 # "1.0: possible or unknown, 0.0: definitely impossible"
        f = o.encode_furiten_ron_possibility()
        assert set(np.unique(f)) <= {0.0, 1.0}
        # "Chiitoitsu: impossible if any melds", "Iipeikou: impossible if any melds (closed hand required)", "Kokushi: impossible if any melds"
        names = {yk["name"].split(" ")[0]: yk["lo"] for yk in FE["yaku"]}
        v = o.peek()
        for s in range(4):
            if v.players[s].n_melds:
                # (players in absolute order in the oracle's aux encoders: tests/test_oracle_aux_encoders.py)
                opened = any(m.opened for m in v.players[s].melds[: v.players[s].n_melds])
                if opened:
                    assert y[s, names["Chiitoitsu"], 0] == 0.0 and y[s, names["Iipeikou"], 0] == 0.0 and y[s, names["Kokushi"], 0] == 0.0
    assert FE["shanten_efficiency_divisors"] == {"shanten": 8.0, "effective_tiles": 34.0, "best_ukeire": 80.0, "turn_progress": 18.0}
    assert FE["shanten_efficiency_unknown"] == 0.5
    ex = FE["decay_example"]                                   # exp(-0.2 x age), age 0 = most recent
    order = ex["order"]
    for tile, want in ex["values"].items():
        got = sum(np.exp(-ex["decay_rate"] * (len(order) - 1 - i)) for i, t in enumerate(order) if t == tile)
        assert abs(got - want) < 1e-3


def test_sequence_constants_and_ranges():
    m, c = SE["sparse_meta"], SE["constants"]
    assert (sf.SPARSE_PAD, sf.MAX_SPARSE_LEN) == (m["padding"], m["max_tokens"]) and m["vocab"] == c["SPARSE_VOCAB_SIZE"] == SE["sparse"][-1]["hi"] + 1
    assert list(sf.PROG_PAD) == SE["progression"]["padding"] and list(sf.CAND_PAD) == SE["candidates"]["padding"]
    assert c["PROG_DIMS"] == [f["vocab"] for f in SE["progression"]["fields"]] and c["CAND_DIMS"] == [f["vocab"] for f in SE["candidates"]["fields"]]
    assert c["NUM_NUMERIC"] == sum(n["hi"] - n["lo"] + 1 for n in SE["numeric"]) == 12
    for kind in ("progression", "candidates"):
        ty = SE[kind]["types"]
        assert [t["lo"] for t in ty] == [0] + [t["hi"] + 1 for t in ty[:-1]]                       # contiguous ranges
        assert ty[-1]["hi"] + 1 == [f for f in SE[kind]["fields"] if f["field"] == "type"][0]["vocab"]
    sp = SE["sparse"]
    assert [s["lo"] for s in sp] == [0] + [s["hi"] + 1 for s in sp[:-1]]
    assert sum(SE["chi_patterns_per_suit"]) * 3 == [t for t in SE["progression"]["types"] if t["action"] == "Chi"][0]["count"]
    pp = SE["pon_patterns"]
    assert pp["per_suit"] * 3 + pp["honors"] == [t for t in SE["progression"]["types"] if t["action"] == "Pon"][0]["count"]
    assert sf.relative_from(0, 1) == (1 - 0 + SE["relative_seat"]["add"]) % SE["relative_seat"]["mod"]
    for k in SE["kan37"]:                                      # the kan37 table: red fives at 0 / 10 / 20, suits behind them
        pass
    reds = [k["lo"] for k in SE["kan37"] if k["tiles"].startswith("Red")]
    assert [sf.tile_id_to_kan37(t) for t in (16, 52, 88)] == reds
    assert [sf.tile_id_to_kan37(4 * t) for t in (0, 8, 9, 17, 18, 26, 27, 33)] == [1, 9, 11, 19, 21, 29, 30, 36]


def _range(kind, action):
    (t,) = [t for t in SE[kind]["types"] if t["action"].startswith(action)]
    return range(t["lo"], t["hi"] + 1)


def test_sequence_features_fall_into_the_documented_ranges_on_real_states():
    sp = {s["feature"].split(" ")[0] + str(i): s for i, s in enumerate(SE["sparse"])}
    rows = SE["sparse"]
    seen_types = set()
    n = 0
    for o, pid in _states(n_games=5, steps=400):
        obs = sf.observation_of(o, pid)
        ev = sf.round_events(o.log(pid))
        tok = sf.sparse(obs, ev, game_style=1)
        assert 5 <= len(tok) <= SE["sparse_meta"]["max_tokens"] and all(0 <= t < SE["sparse_meta"]["padding"] for t in tok)
        per = [sum(1 for t in tok if r["lo"] <= t <= r["hi"]) for r in rows]
        # game style, seat, round wind, dealer, tiles remaining: one each; 1-5 dora indicators; the hand's tiles; at most one drawn tile
        assert per[:5] == [1, 1, 1, 1, 1] and 1 <= per[5] <= 5 and per[6] == len(obs["hand"]) and per[7] <= 1 and per[8] == 0
        assert tok[1] == rows[1]["lo"] + pid and tok[2] == rows[2]["lo"] + obs["round_wind"] and tok[3] == rows[3]["lo"] + obs["oya"]
        for i, t in enumerate(obs["dora"]):
            assert rows[5]["lo"] + i * 37 + sf.tile_id_to_kan37(t) in tok
        num = sf.numeric(obs, ev)
        by = {nrow["feature"]: nrow for nrow in SE["numeric"]}
        assert num[by["Honba (current)"]["lo"]] == obs["honba"] and num[by["Riichi deposits (current)"]["lo"]] == obs["riichi_sticks"]
        for k, name in enumerate(("Score (self)", "Score (right / shimocha)", "Score (across / toimen)", "Score (left / kamicha)")):
            assert num[by[name]["lo"]] == obs["scores"][(pid + k) % 4]
        start = json.loads(ev[0])
        assert start["type"] == "start_kyoku"
        r8 = by["Scores at round start (self-relative order)"]
        assert [num[r8["lo"] + k] for k in range(4)] == [float(start["scores"][(pid + k) % 4]) for k in range(4)]
        assert num[by["Honba (round start)"]["lo"]] == start["honba"] and num[by["Riichi deposits (round start)"]["lo"]] == start["kyotaku"]
        prog = sf.progression(ev)
        assert prog[0] == tuple(int(x) for x in re.findall(r"\d+", [e for e in SE["progression"]["events"] if e["event"] == "start_kyoku"][0]["tuple"]))
        kinds = [json.loads(s)["type"] for s in ev]
        emitted = [k for k in kinds if k in ("start_kyoku", "dahai", "chi", "pon", "daiminkan", "ankan", "kakan")]
        assert len(prog) == len(emitted)                       # tsumo, dora, reach, reach_accepted are not included
        names = {"dahai": "Discard", "chi": "Chi", "pon": "Pon", "daiminkan": "Daiminkan", "ankan": "Ankan", "kakan": "Kakan"}
        for k, tup in zip(emitted[1:], prog[1:]):
            assert tup[1] in _range("progression", names[k]), (k, tup)
            assert (tup[2] in (0, 1)) == (k == "dahai") and (tup[4] in (0, 1, 2)) == (k in ("chi", "pon", "daiminkan"))
            seen_types.add(names[k])
        legal = o.legal(pid)
        cand = sf.candidates(obs, ev, legal)
        want = {abi.DISCARD: "Discard", abi.ANKAN: "Ankan", abi.KAKAN: "Kakan", abi.TSUMO: "Tsumo", abi.KYUSHU: "Kyushu", abi.PASS: "Pass",
                abi.CHI: "Chi", abi.PON: "Pon", abi.DAIMINKAN: "Daiminkan", abi.RON: "Ron"}
        kinds_l = [abi.unpack_action(a)[0] for a in legal if abi.unpack_action(a)[0] != abi.RIICHI]   # "Riichi is not a separate candidate type"
        assert len(cand) == len(kinds_l) <= SE["constants"]["MAX_CAND_LEN"] * 2
        for ty, c4 in zip(kinds_l, cand):
            assert c4[0] in _range("candidates", want[ty]), (ty, c4)
        n += 1
    assert n > 100 and {"Discard", "Chi", "Pon"} <= seen_types
