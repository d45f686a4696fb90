"""CPU restatement of the reference's sequence (transformer) features -- TEST INFRASTRUCTURE, not a product path.

Follows riichienv-core/src/observation/sequence_features.rs function by function (cited below); pure Python over the
oracle's JSON event strings, for small cases only.  The features are functions of an Observation, whose `events`
field is the list of MJAI strings handed to it (observation/mod.rs:34, 71-82).  The device path defines them over the
seat's log of the CURRENT ROUND (see include/riichi_mi355x.h, rmj_encode_seq); `round_events` below extracts that list.
Pinned on the unit tests of sequence_features.rs:844-972 (tests/test_oracle_seq_features.py)."""
import json

from riichienv_amd import abi

SPARSE_PAD, MAX_SPARSE_LEN = 441, 25          # sequence_features.rs:18-21
PROG_PAD, CAND_PAD = (4, 276, 2, 2, 4), (279, 2, 2, 3)   # :28, :36
RED = (16, 52, 88)


def tile_id_to_kan37(t):                       # :46-72
    if t == 16:
        return 0
    if t == 52:
        return 10
    if t == 88:
        return 20
    tt = t // 4
    return tt + 1 if tt <= 8 else (tt + 2 if tt <= 17 else (tt + 3 if tt <= 33 else 0))


def _mjai_kan37(s):                            # :75-78
    try:
        return tile_id_to_kan37(abi.mjai_to_tid(s))
    except ValueError:
        return None


def encode_chi(consumed, called):              # :93-133
    tiles = sorted([called] + list(consumed))
    first = tiles[0] // 4
    suit = first // 9
    seq_start = first - suit * 9
    call_pos = called // 4 - suit * 9 - seq_start
    has_red = any(t in RED for t in tiles)
    involves_five = seq_start <= 4 <= seq_start + 2
    offset = sum(6 if s <= 4 <= s + 2 else 3 for s in range(seq_start))
    return suit * 30 + offset + ((3 + call_pos) if (involves_five and has_red) else call_pos)


def encode_pon(consumed, called):              # :144-184
    ct = called // 4
    suit = ct // 9
    if suit == 3:
        return 33 + (ct - 27)
    rank = ct - suit * 9
    if rank == 4:
        sub = 2 if called in RED else (1 if any(t in RED for t in consumed) else 0)
        return suit * 11 + 4 + sub
    return suit * 11 + (rank if rank < 4 else rank + 2)


def relative_from(actor, target):              # :188-190
    return (target - actor + 3) % 4 if target - actor + 3 >= 0 else -((-(target - actor + 3)) % 4)


def _consumed(ev):                             # :193-205 / :674-686
    out = []
    for s in ev.get("consumed") or []:
        try:
            out.append(abi.mjai_to_tid(s))
        except ValueError:
            pass
    return out


def progression(events, cap=512):              # :503-671 (the same entries as process_single_event_progression :213-314)
    prog, pending = [], None
    for s in events:
        try:
            v = json.loads(s)
        except ValueError:
            continue
        ty = v.get("type")
        if ty == "start_kyoku":
            prog.append((4, 0, 2, 2, 4))
        elif ty == "reach":
            if isinstance(v.get("actor"), int):
                pending = v["actor"]
        elif ty == "dahai":
            actor, pai = v.get("actor", 0), v.get("pai", "?")
            k = None if pai == "?" else _mjai_kan37(pai)
            if k is not None:
                liqi = 0
                if pending == actor:
                    pending, liqi = None, 1
                prog.append((actor, 1 + k, 1 if v.get("tsumogiri", False) else 0, liqi, 4))
        elif ty in ("chi", "pon"):
            actor, target, pai = v.get("actor", 0), v.get("target", 0), v.get("pai", "?")
            cons = _consumed(v)
            if pai != "?" and len(cons) >= 2:
                try:
                    called = abi.mjai_to_tid(pai)
                except ValueError:
                    called = None
                if called is not None:
                    enc = 38 + encode_chi(cons, called) if ty == "chi" else 128 + encode_pon(cons, called)
                    prog.append((actor, enc, 2, 2, relative_from(actor, target)))
        elif ty == "daiminkan":
            actor, target, pai = v.get("actor", 0), v.get("target", 0), v.get("pai", "?")
            k = None if pai == "?" else _mjai_kan37(pai)
            if k is not None:
                prog.append((actor, 168 + k, 2, 2, relative_from(actor, target)))
        elif ty == "ankan":
            cons = _consumed(v)
            if cons:
                prog.append((v.get("actor", 0), 205 + cons[0] // 4, 2, 2, 4))
        elif ty == "kakan":
            pai = v.get("pai", "?")
            k = None if pai == "?" else _mjai_kan37(pai)
            if k is not None:
                prog.append((v.get("actor", 0), 239 + k, 2, 2, 4))
        if len(prog) >= cap:
            break
    return prog


def get_drawn_tile(events, pid):               # :410-435
    for s in reversed(events):
        try:
            v = json.loads(s)
        except ValueError:
            continue
        ty = v.get("type", "")
        if ty == "tsumo" and v.get("actor") == pid and isinstance(v.get("pai"), str) and v["pai"] != "?":
            try:
                return abi.mjai_to_tid(v["pai"])
            except ValueError:
                return None
        if ty in ("dahai", "chi", "pon", "daiminkan"):
            break
    return None


def find_last_discard_actor(events):           # :825-835
    for s in reversed(events):
        try:
            v = json.loads(s)
        except ValueError:
            continue
        if v.get("type") in ("dahai", "kakan"):
            return v.get("actor")
    return None


def sparse(obs, events, game_style=1):         # :331-378, tiles remaining :381-407
    pid = obs["player_id"]
    tok = [min(game_style, 1), 2 + min(pid, 3), 6 + min(obs["round_wind"], 2), 9 + min(obs["oya"], 3)]
    used = len(obs["hand"]) + sum(len(d) for d in obs["discards"]) + sum(len(m) for ms in obs["melds"] for m in ms) + len(obs["dora"])
    tok.append(13 + min(max(136 - (14 + used), 0), 69))
    for i, t in enumerate(obs["dora"][:5]):
        tok.append(83 + i * 37 + tile_id_to_kan37(t))
    tok += [268 + t for t in obs["hand"] if t < 136]
    d = get_drawn_tile(events, pid)
    if d is not None:
        tok.append(404 + tile_id_to_kan37(d))
    return tok


def numeric(obs, events):                      # :447-491
    pid = obs["player_id"]
    out = [float(obs["honba"]), float(obs["riichi_sticks"])] + [float(obs["scores"][(pid + i) % 4]) for i in range(4)]
    start = (obs["honba"], obs["riichi_sticks"], list(obs["scores"]))
    for s in events:
        try:
            v = json.loads(s)
        except ValueError:
            continue
        if v.get("type") == "start_kyoku":
            sc = [0, 0, 0, 0]
            for i, x in enumerate((v.get("scores") or [])[:4]):
                sc[i] = int(x)
            start = (v.get("honba", 0), v.get("kyotaku", 0), sc)
            break
    return out + [float(start[0]), float(start[1])] + [float(start[2][(pid + i) % 4]) for i in range(4)]


def candidates(obs, events, legal):            # :697-813; legal = packed actions in list order
    pid = obs["player_id"]
    drawn = get_drawn_tile(events, pid)
    out = []
    for a in legal:
        ty, tile, cons = abi.unpack_action(a)
        if ty == abi.DISCARD:
            if tile is not None:
                out.append((tile_id_to_kan37(tile), 1 if drawn == tile else 0, 2, 3))
        elif ty == abi.ANKAN:
            if cons:
                out.append((37 + cons[0] // 4, 2, 2, 3))
        elif ty == abi.KAKAN:
            t = tile if tile is not None else (cons[0] if cons else None)
            if t is not None:
                out.append((71 + tile_id_to_kan37(t), 2, 2, 3))
        elif ty == abi.TSUMO:
            out.append((108, 2, 2, 3))
        elif ty == abi.KYUSHU:
            out.append((109, 2, 2, 3))
        elif ty == abi.PASS:
            out.append((110, 2, 2, 3))
        elif ty in (abi.CHI, abi.PON, abi.DAIMINKAN, abi.RON):
            target = find_last_discard_actor(events)
            if ty == abi.RON:
                if target is not None:
                    out.append((278, 2, 2, relative_from(pid, target)))
                continue
            if tile is None or target is None:
                continue
            if ty == abi.DAIMINKAN:
                out.append((241 + tile_id_to_kan37(tile), 2, 2, relative_from(pid, target)))
            elif len(cons) >= 2:
                enc = 111 + encode_chi(cons, tile) if ty == abi.CHI else 201 + encode_pon(cons, tile)
                out.append((enc, 2, 2, relative_from(pid, target)))
        # Riichi (:739-745) and Kita (:811) produce no candidate
    return out


def round_events(log):
    """the part of a seat's log that belongs to the current round: from its last start_kyoku on"""
    start = 0
    for i, s in enumerate(log):
        if '"type":"start_kyoku"' in s:
            start = i
    return log[start:]


def observation_of(game, pid):
    """the fields of get_observation (state/mod.rs:189-263) the sequence features read, from an oracle Game"""
    v = game.peek()
    ps = v.players
    return dict(player_id=pid, hand=list(ps[pid].hand[: ps[pid].hand_len]),
                melds=[[list(m.tiles[: m.n_tiles]) for m in p.melds[: p.n_melds]] for p in ps],
                discards=[list(p.discards[: p.n_discards]) for p in ps], dora=list(v.dora[: v.n_dora]),
                scores=[p.score for p in ps], honba=v.honba, riichi_sticks=v.riichi_sticks, round_wind=v.round_wind, oya=v.oya)
