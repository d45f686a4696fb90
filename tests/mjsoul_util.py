"""Mahjong Soul records for the reader tests.  The reference ships no MjSoul record, so rounds are produced here: the oracle
plays full games with a policy that heads for tenpai (discard = a tile whose removal keeps the shanten lowest, every win /
riichi / kan / kita taken, calls sometimes), the walls of all rounds are kept, and the MJAI log is rewritten into the action
records mjsoul_replay.rs deserialises (RawAction, mjsoul_replay.rs:59-137) the way Mahjong Soul writes them: the dealer's first
draw inside tiles<oya>, replacement draws carrying the `doras` list, indicators of open kans on the following discard, the wall
as `paishan`."""
import json

import numpy as np

from riichienv_amd import abi
from riichienv_amd.shard import game_seed

_HON = "ESWNPFC"


def mjai_to_mjsoul(name):
    if name in _HON:
        return f"{_HON.index(name) + 1}z"
    if name.endswith("r"):
        return "0" + name[1]
    return name


def tid_to_mjsoul(t):
    t34 = t // 4
    if t in (16, 52, 88):
        return "0" + "mps"[t34 // 9]
    return f"{t34 % 9 + 1}{'mpsz'[t34 // 9]}"


def greedy_actions(o, rng, sanma, call_rate=0.25):
    """The policy of the played games: every win / riichi / kan / kita that is offered, a pon or chi a quarter of the time, a
    discard that keeps the shanten of the rest lowest (ties by `rng`), else pass.  Returns the four packed actions."""
    from oracle import oracle

    npl = 3 if sanma else 4
    act = o.status()[0]
    acts = [abi.NO_ACTION] * 4
    v = o.peek()
    for s in range(npl):
        if not (act >> s) & 1:
            continue
        legal = o.legal(s)
        if not legal:
            continue
        kinds = {}
        for a in legal:
            kinds.setdefault(abi.unpack_action(a)[0], []).append(a)
        pick = None
        for ty in (abi.TSUMO, abi.RON, abi.RIICHI, abi.ANKAN, abi.KAKAN, abi.DAIMINKAN, abi.KITA):
            if ty in kinds:
                pick = kinds[ty][0]
                break
        if pick is None and (abi.PON in kinds or abi.CHI in kinds) and rng.random() < call_rate:
            pick = (kinds.get(abi.PON) or kinds.get(abi.CHI))[0]
        if pick is None and abi.DISCARD in kinds:
            hand = list(v.players[s].hand[: v.players[s].hand_len])
            cands = kinds[abi.DISCARD]
            cnt = np.zeros((len(cands), 34), dtype=np.uint8)
            for i, a in enumerate(cands):
                rest = list(hand)
                rest.remove(abi.unpack_action(a)[1])
                for t in rest:
                    cnt[i, t // 4] += 1
            sh = np.asarray(oracle.shanten(cnt, sanma))
            best = np.flatnonzero(sh == sh.min())
            pick = cands[int(best[int(rng.integers(len(best)))])]
        if pick is None:
            pick = kinds.get(abi.PASS, legal)[0]
        acts[s] = pick
    return acts


def play_logged_game(mode, seed, rule=abi.RULE_MJSOUL, max_steps=8000, with_scores=False):
    """(MJAI events, walls): walls[i] = the 136-ids of round i's wall in draw order (= MjSoul's paishan); with_scores adds the
    scores the oracle's game ended with"""
    from oracle import oracle

    sanma = mode >= 3
    npl = 3 if sanma else 4
    o = oracle.Game(game_mode=mode, seed=game_seed(9090, seed), rule_bits=rule)
    o.reset()
    rng = np.random.default_rng(seed)
    walls, last_hand_index = [], None

    def note_wall():
        nonlocal last_hand_index
        v = o.peek()
        if v.hand_index != last_hand_index:
            last_hand_index = v.hand_index
            walls.append(list(v.wall[: v.wall_len])[::-1])

    note_wall()
    for _ in range(max_steps):
        if o.status()[2]:
            break
        acts = greedy_actions(o, rng, sanma)
        o.step(acts)
        note_wall()
    assert o.status()[2], "the game did not end"
    events = [json.loads(s) for s in o.log()]
    n_rounds = sum(e["type"] == "start_kyoku" for e in events)
    assert len(walls) >= n_rounds
    if with_scores:
        v = o.peek()
        return events, walls[:n_rounds], [int(v.players[i].score) for i in range(npl)]
    return events, walls[:n_rounds]


def to_mjsoul_rounds(events, walls, with_paishan=True, expectations=None):
    """MJAI events -> [[{"name", "data"}, ...], ...] (one list per round).  expectations: per hora event (in log order) a dict
    {count, fu, fans} for the Hule records (default zeros / empty)."""
    rounds, cur = [], None
    hora_no = 0
    st = {}
    for ev in events:
        ty = ev["type"]
        if ty == "start_kyoku":
            n = len(ev["scores"])
            st = dict(n=n, left=(55 if n == 3 else 70), doras=[mjai_to_mjsoul(ev["dora_marker"])], new_dora=False, after_kan=False,
                      reach=[False] * n, first=[True] * n, calls=False, oya=ev["oya"], first_draw=True, hules=None)
            d = dict(scores=list(ev["scores"]), dora_marker=st["doras"][0], doras=list(st["doras"]),
                     chang={"E": 0, "S": 1, "W": 2, "N": 3}[ev["bakaze"]], ju=ev["kyoku"] - 1, ben=ev["honba"], liqibang=ev["kyotaku"])
            for i in range(4):
                d[f"tiles{i}"] = [mjai_to_mjsoul(t) for t in ev["tehais"][i]] if i < n else []
            if with_paishan and n == 4:
                d["paishan"] = "".join(tid_to_mjsoul(t) for t in walls[len(rounds)])
            cur = [{"name": "NewRound", "data": d}]
            rounds.append(cur)
            continue
        if cur is None:
            continue
        if ty != "hora" and st["hules"] is not None:
            cur.append({"name": "Hule", "data": {"hules": st["hules"]}})
            st["hules"] = None
        a = ev.get("actor")
        if ty == "tsumo":
            st["left"] -= 1
            if st["first_draw"]:                       # the dealer's fourteenth tile is part of the deal
                st["first_draw"] = False
                cur[0]["data"][f"tiles{a}"].append(mjai_to_mjsoul(ev["pai"]))
                cur[0]["data"]["left_tile_count"] = st["left"]
                continue
            d = dict(seat=a, tile=mjai_to_mjsoul(ev["pai"]), left_tile_count=st["left"])
            if st["after_kan"]:
                d["doras"] = list(st["doras"])
                st["new_dora"] = False
            st["after_kan"] = False
            cur.append({"name": "DealTile", "data": d})
        elif ty == "dahai":
            liqi = st["reach"][a]
            d = dict(seat=a, tile=mjai_to_mjsoul(ev["pai"]), is_liqi=liqi, is_wliqi=liqi and st["first"][a] and not st["calls"],
                     moqie=bool(ev.get("tsumogiri")))
            if st["new_dora"]:
                d["doras"] = list(st["doras"])
                st["new_dora"] = False
            st["reach"][a] = False
            st["first"][a] = False
            cur.append({"name": "DiscardTile", "data": d})
        elif ty == "reach":
            st["reach"][a] = True
        elif ty in ("chi", "pon", "daiminkan"):
            st["calls"] = True
            cons = [mjai_to_mjsoul(t) for t in ev["consumed"]]
            cur.append({"name": "ChiPengGang", "data": dict(seat=a, type={"chi": 0, "pon": 1, "daiminkan": 2}[ty],
                                                            tiles=cons + [mjai_to_mjsoul(ev["pai"])], froms=[a] * len(cons) + [ev["target"]])})
            st["after_kan"] = ty == "daiminkan"
        elif ty == "ankan":
            st["calls"] = True
            t = mjai_to_mjsoul(ev["consumed"][0])
            cur.append({"name": "AnGangAddGang", "data": dict(seat=a, type=3, tiles=("5" + t[1]) if t[0] == "0" else t)})
            st["after_kan"] = True
        elif ty == "kakan":
            st["calls"] = True
            cur.append({"name": "AnGangAddGang", "data": dict(seat=a, type=2, tiles=mjai_to_mjsoul(ev["pai"]))})
            st["after_kan"] = True
        elif ty == "kita":
            cur.append({"name": "BaBei", "data": dict(seat=a, moqie=False)})
            st["after_kan"] = True
        elif ty == "dora":
            st["doras"].append(mjai_to_mjsoul(ev["dora_marker"]))
            st["new_dora"] = True
        elif ty == "hora":
            exp = (expectations or {}).get(hora_no, {})
            hora_no += 1
            zimo = a == ev["target"]
            last = next(x for x in reversed(cur) if x["name"] in ("DealTile", "DiscardTile", "AnGangAddGang", "BaBei", "NewRound"))
            if last["name"] == "NewRound":
                hu = last["data"][f"tiles{a}"][-1]
            elif last["name"] == "BaBei":
                hu = "4z"
            elif last["name"] == "AnGangAddGang":
                hu = last["data"]["tiles"]
            else:
                hu = last["data"]["tile"]
            h = dict(seat=a, hu_tile=hu, zimo=zimo, count=exp.get("count", 0), fu=exp.get("fu", 0),
                     fans=[{"id": y, "val": 1} for y in exp.get("fans", [])] + [{"id": 99, "val": 0}], hand=[], yiman=bool(exp.get("yiman", False)),
                     point_rong=exp.get("point_rong", 0), point_zimo_qin=exp.get("point_zimo_qin", 0),
                     point_zimo_xian=exp.get("point_zimo_xian", 0))
            if "paishan" not in cur[0]["data"] and ev.get("ura_markers"):
                h["li_doras"] = [mjai_to_mjsoul(t) for t in ev["ura_markers"]]
            st["hules"] = (st["hules"] or []) + [h]
        elif ty == "ryukyoku":
            if ev.get("reason") in (None, "exhaustive_draw", "nagashi_mangan"):
                cur.append({"name": "NoTile", "data": {}})
            else:
                cur.append({"name": "LiuJu", "data": {"type": 1, "seat": 0, "tiles": []}})
        elif ty in ("end_kyoku", "end_game"):
            cur = None
    return rounds
