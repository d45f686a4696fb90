"""Rebuild per-kyoku walls from a real MJAI log (tests/golden/126_204_0_mjai.jsonl, the reference's own
fixture tests/data/126_204_0_mjai.jsonl) and drive an environment through the logged actions.

The log is in the third-party MJAI format (tile strings, not 136-ids), so physical tile ids are assigned
consistently per kyoku: red fives are ids 16/52/88 (parser.rs:301-334), other copies are handed out in order.
"""
import json

from riichienv_amd import abi

HONORS = ["E", "S", "W", "N", "P", "F", "C"]


def tile_type(s):
    if s in HONORS:
        return 27 + HONORS.index(s)
    n = int(s[0])
    suit = "mps".index(s[1])
    return suit * 9 + n - 1


def is_red(s):
    return len(s) == 3 and s[2] == "r"


class TileBank:
    """Hands out unused physical copies of a tile string."""

    def __init__(self):
        self.used = set()

    def take(self, s):
        t = tile_type(s)
        if is_red(s):
            tid = t * 4
            assert tid in (16, 52, 88) and tid not in self.used
        else:
            start = 1 if t in (4, 13, 22) else 0
            tid = next(t * 4 + k for k in range(start, 4) if t * 4 + k not in self.used)
        self.used.add(tid)
        return tid

    def rest(self):
        return [t for t in range(136) if t not in self.used]


def split_kyoku(events):
    kyoku, cur = [], None
    for e in events:
        if e["type"] == "start_kyoku":
            cur = [e]
        elif cur is not None:
            cur.append(e)
            if e["type"] == "end_kyoku":
                kyoku.append(cur)
                cur = None
    return kyoku


def build_wall(kevents):
    """Wall in the reference's `reset(wall=...)` orientation (draw order: wall[0] is dealt first;
    state/wall.rs:69-80 reverses it, state/mod.rs:1750-1768 pops from the end)."""
    sk = kevents[0]
    oya = sk["oya"]
    bank = TileBank()
    wall = [None] * 136
    hands = [[bank.take(s) for s in sk["tehais"][p]] for p in range(4)]
    # deal slots: pop #n -> seat (state/mod.rs:1750-1765)
    slot_of = [[] for _ in range(4)]
    for n in range(48):
        idx = (n % 16) // 4
        slot_of[(idx + oya) % 4].append(n)
    for n in range(48, 52):
        slot_of[((n - 48) + oya) % 4].append(n)
    for p in range(4):
        for n, tid in zip(slot_of[p], hands[p]):
            wall[n] = tid
    wall[131] = bank.take(sk["dora_marker"])  # W[4] after reverse
    live = 52
    rinshan = 0
    n_dora = 1
    after_kan = False
    for e in kevents[1:]:
        t = e["type"]
        if t == "tsumo":
            if after_kan:
                wall[135 - rinshan] = bank.take(e["pai"])
                rinshan += 1
                after_kan = False
            else:
                wall[live] = bank.take(e["pai"])
                live += 1
        elif t in ("ankan", "kakan", "daiminkan"):
            after_kan = True
        elif t == "dora":
            wall[131 - 2 * n_dora] = bank.take(e["dora_marker"])  # W[4 + 2k]
            n_dora += 1
        elif t == "hora":
            for k, s in enumerate(e.get("ura_markers", [])):
                pos = 130 - 2 * k  # W[5 + 2k]
                if wall[pos] is None:
                    wall[pos] = bank.take(s)
    rest = bank.rest()
    for i in range(136):
        if wall[i] is None:
            wall[i] = rest.pop(0)
    assert sorted(wall) == list(range(136))
    return wall


def load_log(path):
    with open(path) as f:
        return [json.loads(l) for l in f if l.strip()]


def pick_from_hand(hand, s, prefer=None, exclude=()):
    """136-id in `hand` whose MJAI string is s."""
    t = tile_type(s)
    cands = [x for x in hand if x // 4 == t and (x in (16, 52, 88)) == is_red(s) and x not in exclude]
    assert cands, (s, hand)
    if prefer is not None and prefer in cands:
        return prefer
    return cands[0]


class ReplayDriver:
    """Feeds the logged player decisions of one kyoku to an env adapter.

    The adapter exposes: status() -> (active_mask, phase, done); peek() -> StateView; legal(seat) -> packed list;
    step(dict seat->packed action).
    """

    def __init__(self, env):
        self.env = env

    def _hand(self, seat):
        v = self.env.peek()
        p = v.players[seat]
        return list(p.hand[: p.hand_len]), v

    def _submit(self, acts):
        for seat, a in acts.items():
            assert a in self.env.legal(seat), (seat, abi.unpack_action(a), [abi.unpack_action(x) for x in self.env.legal(seat)])
        self.env.step(acts)

    def _pass_all(self):
        act, ph, dn = self.env.status()
        if not dn and ph == abi.WAIT_RESPONSE:
            self._submit({s: abi.pack_action(abi.PASS) for s in range(4) if (act >> s) & 1})

    def run_kyoku(self, kevents):
        i = 1
        n = len(kevents)
        while i < n:
            e = kevents[i]
            t = e["type"]
            act, ph, dn = self.env.status()
            if t in ("tsumo", "dora", "reach_accepted", "end_kyoku"):
                if t == "tsumo" and ph == abi.WAIT_RESPONSE:
                    self._pass_all()  # nobody claimed the previous discard
                i += 1
                continue
            if t == "dahai":
                hand, v = self._hand(e["actor"])
                drawn = v.drawn_tile if v.drawn_tile >= 0 else None
                tid = pick_from_hand(hand, e["pai"], prefer=drawn if e["tsumogiri"] else None,
                                     exclude=() if e["tsumogiri"] or drawn is None else (drawn,))
                self._submit({e["actor"]: abi.pack_action(abi.DISCARD, tid)})
            elif t == "reach":
                self._submit({e["actor"]: abi.pack_action(abi.RIICHI)})
            elif t in ("pon", "chi", "daiminkan"):
                hand, v = self._hand(e["actor"])
                cons, used = [], []
                for s in e["consumed"]:
                    x = pick_from_hand(hand, s, exclude=used)
                    used.append(x)
                    cons.append(x)
                kind = {"pon": abi.PON, "chi": abi.CHI, "daiminkan": abi.DAIMINKAN}[t]
                acts = {s: abi.pack_action(abi.PASS) for s in range(4) if (act >> s) & 1}
                acts[e["actor"]] = abi.pack_action(kind, v.last_discard_tile, cons)
                self._submit(acts)
            elif t == "ankan":
                hand, v = self._hand(e["actor"])
                ty = tile_type(e["consumed"][0])
                cons = [ty * 4 + k for k in range(4)]
                self._submit({e["actor"]: abi.pack_action(abi.ANKAN, ty * 4, cons)})
            elif t == "kakan":
                hand, v = self._hand(e["actor"])
                tid = pick_from_hand(hand, e["pai"])
                a = next(x for x in self.env.legal(e["actor"]) if abi.unpack_action(x)[0] == abi.KAKAN and abi.unpack_action(x)[1] == tid)
                self._submit({e["actor"]: a})
            elif t == "hora":
                winners = []
                while i < n and kevents[i]["type"] == "hora":
                    winners.append(kevents[i])
                    i += 1
                i -= 1
                if winners[0]["actor"] == winners[0]["target"]:
                    self._submit({winners[0]["actor"]: next(x for x in self.env.legal(winners[0]["actor"]) if abi.unpack_action(x)[0] == abi.TSUMO)})
                else:
                    acts = {s: abi.pack_action(abi.PASS) for s in range(4) if (act >> s) & 1}
                    for w in winners:
                        acts[w["actor"]] = next(x for x in self.env.legal(w["actor"]) if abi.unpack_action(x)[0] == abi.RON)
                    self._submit(acts)
            elif t == "ryukyoku":
                if ph == abi.WAIT_RESPONSE:
                    self._pass_all()  # last discard unclaimed -> exhaustive draw fires inside that step
            i += 1


def comparable(ev):
    """Keys shared by the third-party log and the reference's own emission (Appendix A of SURVEY.md)."""
    t = ev["type"]
    keep = {"type": t}
    for k in ("actor", "target", "pai", "tsumogiri", "deltas", "ura_markers", "dora_marker"):
        if k in ev:
            keep[k] = ev[k]
    if t == "ankan":
        keep.pop("pai", None)  # the reference adds a non-standard "pai" key to ankan
    if "consumed" in ev:
        keep["consumed"] = sorted(ev["consumed"])
    if t == "start_kyoku":
        for k in ("bakaze", "honba", "kyoku", "kyotaku", "oya", "scores"):
            keep[k] = ev[k]
        keep["tehais"] = [sorted(h) for h in ev["tehais"]]
    return keep
