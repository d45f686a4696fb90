"""MJAI logs -> (observation, action) samples in lock-step on the GPU (MJAI part).

The reference reads logs with MjaiReplay.from_jsonl and walks them with KyokuStepIterator on the host, one game at a time
(riichienv-core/src/replay/mjai_replay.rs:65-156, replay/mod.rs).  Here many logs advance together: event k of every
log is applied with rmj_apply_events (= RiichiEnv.apply_event), and whenever the NEXT event of a log is a player's
decision the acting seat's legal list, mask, feature tensor and the action it took (select_action_from_mjai,
observation/mjai_select.rs:88-194, encoded as the 82- / 60-way id) are emitted - the (obs, action) pairs of
behaviour-cloning / offline-RL datasets.  This is the event-stream semantics of apply_event (tile names map to one id
per name, no wall); it is not a restatement of the reference's MjSoul-style replay pipeline (LogAction / grp_features).
"""
from __future__ import annotations

import gzip
import json

import numpy as np

from . import abi, mjai, vecenv

# events that are a decision of their actor, and the seats' implicit "none" (pass) after a discard / kita
_ACTOR_DECISIONS = ("dahai", "chi", "pon", "daiminkan", "kan", "ankan", "kakan", "reach", "hora", "kita")


def load_mjai_jsonl(path):
    """MjaiReplay.from_jsonl (replay/mjai_replay.rs:275-300): one JSON event per line; gzip is detected by its magic
    bytes, not by the file name."""
    with open(path, "rb") as f:
        magic = f.read(2)
    opener = gzip.open if magic == b"\x1f\x8b" else open
    with opener(path, "rt") as f:
        return [json.loads(line) for line in f if line.strip()]


class Kyoku:
    """One round of an MJAI log: LogKyoku as MjaiReplay builds it (replay/mjai_replay.rs:158-268, 386-625) - the round
    header, the action list in the reference's action vocabulary (replay/mod.rs:35-80) and the GRP features."""

    _BAKAZE = {"S": 1, "W": 2, "N": 3}

    def __init__(self, ev):
        self.mjai_events = [ev]                      # the raw MJAI events of the round (start_kyoku first)
        self.scores = list(ev["scores"])
        self.end_scores = list(ev["scores"])
        self.chang = self._BAKAZE.get(ev.get("bakaze", "E"), 0)
        self.ju = int(ev["kyoku"]) - 1
        self.ben = int(ev.get("honba", 0))
        self.liqibang = int(ev.get("kyoutaku", ev.get("kyotaku", 0)))   # serde alias, mjai_replay.rs:77-78
        self.doras = [ev["dora_marker"]]
        self.ura_doras = []
        self.hands = [list(h) for h in ev["tehais"]][: len(self.scores)]
        n = len(self.scores)
        self.left_tile_count = 55 if n == 3 else 70
        self.wliqi = [False] * n
        self.actions = []
        self._liqi, self._reached, self._accepted, self._first = [False] * n, [False] * n, [False] * n, [True] * n
        self._has_calls = False
        self._pending_hule = []

    def _flush(self):
        if self._pending_hule:
            self.actions.append({"name": "Hule", "hules": self._pending_hule})
            self._pending_hule = []

    def _feed(self, ev):  # MjaiReplay::process_event, mjai_replay.rs:386-625
        self.mjai_events.append(ev)
        ty = ev.get("type")
        if ty != "hora":
            self._flush()
        a = ev.get("actor")
        if ty == "tsumo":
            self.actions.append({"name": "DealTile", "seat": a, "tile": ev["pai"]})
            self.left_tile_count = max(self.left_tile_count - 1, 0)
        elif ty == "dahai":
            is_liqi = self._liqi[a]
            is_wliqi = is_liqi and self._first[a] and not self._has_calls
            if is_wliqi:
                self.wliqi[a] = True
            self.actions.append({"name": "DiscardTile", "seat": a, "tile": ev["pai"], "is_liqi": is_liqi, "is_wliqi": is_wliqi})
            self._first[a] = False
            if is_liqi:
                self._liqi[a] = False
        elif ty == "reach":
            self._liqi[a] = self._reached[a] = True
        elif ty == "reach_accepted":
            self._accepted[a] = True
        elif ty in ("chi", "pon", "kan", "daiminkan"):
            self._has_calls = True
            kind = {"chi": "Chi", "pon": "Pon"}.get(ty, "Daiminkan")
            self.actions.append({"name": "ChiPengGang", "seat": a, "meld_type": kind, "tiles": [ev["pai"]] + list(ev["consumed"]),
                                 "froms": [ev["target"]] + [a] * len(ev["consumed"])})
        elif ty == "ankan":
            self._has_calls = True
            self.actions.append({"name": "AnGangAddGang", "seat": a, "meld_type": "Ankan", "tiles": list(ev["consumed"])})
        elif ty == "kakan":
            self._has_calls = True
            self.actions.append({"name": "AnGangAddGang", "seat": a, "meld_type": "Kakan", "tiles": [ev["pai"]]})
        elif ty == "dora":
            self.doras.append(ev["dora_marker"])
            self.actions.append({"name": "Dora", "dora_marker": ev["dora_marker"]})
        elif ty == "hora":
            if ev.get("uradora_markers") or ev.get("ura_markers"):
                self.ura_doras = list(ev.get("uradora_markers") or ev.get("ura_markers"))
            if ev.get("scores") is not None:
                self.end_scores = list(ev["scores"])
            elif ev.get("deltas", ev.get("delta")) is not None:
                first = not self._pending_hule
                for i, d in enumerate(ev.get("deltas", ev.get("delta"))[: len(self.end_scores)]):
                    # the first hora of a batch starts from the round's scores minus the ACCEPTED riichi deposits; further
                    # hora events (double / triple ron) add their deltas (mjai_replay.rs:581-599)
                    self.end_scores[i] = (self.scores[i] + d - (1000 if self._accepted[i] else 0)) if first else self.end_scores[i] + d
            # the winning tile: the event's `pai`, else the tile of the last action (mjai_replay.rs:541-559)
            if ev.get("pai") is not None:
                hu = abi.mjai_to_tid(ev["pai"], True)
            else:
                last = self.actions[-1] if self.actions else None
                hu = 0
                if last is not None and last["name"] in ("DealTile", "DiscardTile"):
                    hu = abi.mjai_to_tid(last["tile"], True)
                elif last is not None and last["name"] == "AnGangAddGang":
                    hu = abi.mjai_to_tid(last["tiles"][0], True)
            ur = ev.get("uradora_markers") if ev.get("uradora_markers") is not None else ev.get("ura_markers")
            self._pending_hule.append({"seat": a, "zimo": a == ev.get("target"), "hu_tile": hu, "count": ev.get("han") or 0,
                                       "fu": ev.get("fu") or 0,
                                       "li_doras": None if ur is None else [abi.mjai_to_tid(t, True) for t in ur]})
        elif ty == "kita":
            self.actions.append({"name": "BaBei", "seat": a})
        elif ty == "ryukyoku":
            if ev.get("scores") is not None:
                self.end_scores = list(ev["scores"])
            elif ev.get("deltas", ev.get("delta")) is not None:
                for i, d in enumerate(ev.get("deltas", ev.get("delta"))[: len(self.end_scores)]):
                    self.end_scores[i] = self.scores[i] + d - (1000 if self._reached[i] else 0)   # :614-621 uses `reached` here
            self.actions.append({"name": "NoTile"})

    def events(self):
        """LogKyoku.events (replay/mod.rs:1294-1500): NewRound followed by one entry per action"""
        head = {"name": "NewRound", "data": dict(scores=list(self.scores), doras=list(self.doras), dora_marker=self.doras[0],
                                                  chang=self.chang, ju=self.ju, ben=self.ben, liqibang=self.liqibang,
                                                  left_tile_count=self.left_tile_count,
                                                  **{f"tiles{i}": list(h) for i, h in enumerate(self.hands)})}
        return [head] + [{"name": a["name"], "data": {k: v for k, v in a.items() if k != "name"}} for a in self.actions]

    def take_win_result_contexts(self, ankan_from_consumed=True):
        """LogKyoku.take_win_result_contexts (replay/mod.rs:1089-1091)"""
        return WinResultContextIterator(self, ankan_from_consumed)

    def grp_features(self):
        """LogKyoku.grp_features (replay/mod.rs:1502-1522)"""
        return dict(chang=self.chang, ju=self.ju, ben=self.ben, liqibang=self.liqibang, scores=list(self.scores),
                    end_scores=list(self.end_scores), wliqi=list(self.wliqi),
                    delta_scores=[e - s for s, e in zip(self.scores, self.end_scores)] if len(self.scores) == len(self.end_scores) else [])


class WinResultContext:
    """WinResultContext (replay/mod.rs:2096-2180): the evaluator inputs of one win as the replay reconstructs them, the
    log's own expectation (MJAI logs carry none: han / fu 0, no yaku) and `actual` = HandEvaluator.calc of those inputs
    (filled in batch by evaluate_win_contexts)."""

    __slots__ = ("seat", "tiles", "melds", "agari_tile", "dora_indicators", "ura_indicators", "conditions", "expected_yaku",
                 "expected_han", "expected_fu", "actual", "sanma")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    def hand_case(self) -> abi.HandCase:
        hc = abi.HandCase()
        hc.n_tiles = len(self.tiles)
        for i, t in enumerate(self.tiles[:14]):
            hc.tiles[i] = t
        hc.n_melds = len(self.melds)
        for i, m in enumerate(self.melds[:4]):
            mv = hc.melds[i]
            mv.meld_type = m["meld_type"]
            mv.n_tiles = len(m["tiles"])
            for j, t in enumerate(m["tiles"][:4]):
                mv.tiles[j] = t
            mv.opened = 1 if m["opened"] else 0
            mv.from_who = m["from_who"]
            mv.called_tile = -1 if m["called_tile"] is None else m["called_tile"]
        hc.win_tile = self.agari_tile
        hc.n_dora = min(len(self.dora_indicators), 5)
        for i, t in enumerate(self.dora_indicators[:5]):
            hc.dora[i] = t
        hc.n_ura = min(len(self.ura_indicators), 5)
        for i, t in enumerate(self.ura_indicators[:5]):
            hc.ura[i] = t
        c = self.conditions
        for k in ("tsumo", "riichi", "double_riichi", "ippatsu", "haitei", "houtei", "rinshan", "chankan", "tsumo_first_turn"):
            setattr(hc, k, 1 if c[k] else 0)
        hc.player_wind, hc.round_wind, hc.honba = c["player_wind"], c["round_wind"], c["honba"]
        hc.kita_count = c["kita_count"]
        hc.is_sanma = 1 if self.sanma else 0
        return hc


_MELD_ID = {"Chi": abi.MELD_CHI, "Pon": abi.MELD_PON, "Daiminkan": abi.MELD_DAIMINKAN, "Ankan": abi.MELD_ANKAN,
            "Kakan": abi.MELD_KAKAN}


def _match_and_remove(hand, target):
    """TileConverter::match_and_remove_u8 (replay/mod.rs:2243-2255): the exact id, else any copy of the type."""
    if target in hand:
        hand.remove(target)
        return True
    for k, x in enumerate(hand):
        if x // 4 == target // 4:
            del hand[k]
            return True
    return False


class WinResultContextIterator:
    """WinResultContextIterator (replay/mod.rs:1593-2094) over one Kyoku built from an MJAI log: walks the round's actions,
    keeps the seats' hands / melds / riichi, ippatsu, rinshan and first-turn flags, the dora indicators and the tile
    count, and yields one WinResultContext per `hora`.  An MJAI log carries no wall (`paishan`), so the wall-dependent
    branches (_recalc_doras, _get_ura_indicators) are the reference's no-wall paths: indicators come from the `dora`
    events, ura indicators from the hora event.  `actual` is left None; evaluate_win_contexts fills it for many
    contexts at once on the GPU.
    Reference quirk: MjaiReplay builds every Ankan with tile_raw_id = 0 (mjai_replay.rs:507-514), and the iterator
    removes "four tiles of type tile_raw_id" (replay/mod.rs:1912-1935), i.e. it treats every MJAI ankan as 1m.
    `ankan_from_consumed=False` reproduces that; the default uses the type of the consumed tiles."""

    def __init__(self, kyoku: Kyoku, ankan_from_consumed=True):
        n = len(kyoku.scores)
        self.k = kyoku
        self.n = n
        self.idx = 0
        self.pending = []
        self.melds = [[] for _ in range(4)]
        self.hands = [[abi.mjai_to_tid(t, True) for t in h] for h in kyoku.hands] + [[] for _ in range(4 - n)]
        self.liqi, self.wliqi, self.ippatsu, self.rinshan = [False] * 4, [False] * 4, [False] * 4, [False] * 4
        self.first = [True] * 4
        self.was_kakan, self.kakan_tile, self.was_babei = False, None, False
        self.ippatsu_before_babei = [False] * 4
        self.doras = [abi.mjai_to_tid(kyoku.doras[0], True)]     # kyoku.doras grows with the dora events: start from the header's
        self.left = 55 if n == 3 else 70
        self.kita = [0] * 4
        self.ankan_from_consumed = ankan_from_consumed

    def __iter__(self):
        return self

    def _after_kakan_reset(self):
        if self.was_kakan:
            self.ippatsu = [False] * 4
            self.first = [False] * 4
            self.was_kakan = self.was_babei = False
            self.kakan_tile = None

    def __next__(self):
        if self.pending:
            return self.pending.pop(0)
        acts = self.k.actions
        while self.idx < len(acts):
            a = acts[self.idx]
            self.idx += 1
            name = a["name"]
            if name != "Hule":
                self.rinshan = [False] * 4
                if name != "BaBei":
                    self.was_babei = False
            if name == "DiscardTile":
                seat = a["seat"]
                self._after_kakan_reset()
                if a["is_wliqi"]:
                    self.wliqi[seat] = self.ippatsu[seat] = True
                if a["is_liqi"]:
                    self.liqi[seat] = self.ippatsu[seat] = True
                else:
                    self.ippatsu[seat] = False
                self.first[seat] = False
                _match_and_remove(self.hands[seat], abi.mjai_to_tid(a["tile"], True))
            elif name == "DealTile":
                seat = a["seat"]
                self._after_kakan_reset()
                self.hands[seat].append(abi.mjai_to_tid(a["tile"], True))
                if self.left > 0:
                    self.left -= 1
            elif name == "ChiPengGang":
                seat = a["seat"]
                self.rinshan = [False] * 4
                self.ippatsu = [False] * 4
                self.first = [False] * 4
                self.was_kakan = self.was_babei = False
                self.kakan_tile = None
                tl = [abi.mjai_to_tid(t, True) for t in a["tiles"]]
                for t, f in zip(tl, a["froms"]):
                    if f == seat:
                        _match_and_remove(self.hands[seat], t)
                frm = next((f for f in a["froms"] if f != seat), -1)
                called = next((t for t, f in zip(tl, a["froms"]) if f != seat), None)
                self.melds[seat].append(dict(meld_type=_MELD_ID[a["meld_type"]], tiles=tl, opened=True, from_who=frm,
                                             called_tile=called))
                if a["meld_type"] == "Daiminkan":
                    self.rinshan[seat] = True
            elif name == "Dora":
                self.doras.append(abi.mjai_to_tid(a["dora_marker"], True))     # no wall: replay/mod.rs:1869-1871
            elif name == "AnGangAddGang":
                seat = a["seat"]
                self.rinshan = [False] * 4
                tl = [abi.mjai_to_tid(t, True) for t in a["tiles"]]
                if a["meld_type"] == "Ankan":
                    self.ippatsu = [False] * 4
                    self.first = [False] * 4
                    self.was_kakan = self.was_babei = False
                    self.kakan_tile = None
                    t34 = tl[0] // 4 if (self.ankan_from_consumed and tl) else 0
                    for _ in range(4):
                        for k, x in enumerate(self.hands[seat]):
                            if x // 4 == t34:
                                del self.hands[seat][k]
                                break
                    self.melds[seat].append(dict(meld_type=abi.MELD_ANKAN, tiles=[t34 * 4 + i for i in range(4)], opened=False,
                                                 from_who=-1, called_tile=None))
                    self.rinshan[seat] = True
                else:
                    self.was_kakan, self.kakan_tile = True, tl[0]
                    self.rinshan[seat] = True
                    for m in self.melds[seat]:
                        if m["meld_type"] == abi.MELD_PON and m["tiles"][0] // 4 == tl[0] // 4:
                            m["meld_type"] = abi.MELD_KAKAN
                            m["tiles"] = m["tiles"] + [tl[0]]
                            break
                    else:
                        self.melds[seat].append(dict(meld_type=abi.MELD_KAKAN, tiles=tl, opened=True, from_who=-1, called_tile=None))
                    _match_and_remove(self.hands[seat], tl[0])
            elif name == "BaBei":
                seat = a["seat"]
                self.ippatsu_before_babei = list(self.ippatsu)
                self.ippatsu = [False] * 4
                self.first = [False] * 4
                self.was_babei = True
                for k, x in enumerate(self.hands[seat]):
                    if x // 4 == 30:
                        del self.hands[seat][k]
                        break
                self.kita[seat] += 1
                self.rinshan[seat] = True
            elif name == "Hule":
                for h in a["hules"]:
                    seat, zimo = h["seat"], h["zimo"]
                    win = h["hu_tile"]
                    chankan = (not zimo) and self.was_kakan and self.kakan_tile is not None and self.kakan_tile // 4 == win // 4
                    ipp = self.ippatsu_before_babei[seat] if (not zimo and self.was_babei) else self.ippatsu[seat]
                    hand = list(self.hands[seat])
                    cond = dict(tsumo=zimo, riichi=self.liqi[seat], double_riichi=self.wliqi[seat], ippatsu=ipp,
                                haitei=self.left == 0 and zimo and not self.rinshan[seat],
                                houtei=self.left == 0 and not zimo and not self.rinshan[seat], rinshan=self.rinshan[seat],
                                chankan=chankan, tsumo_first_turn=self.first[seat] and zimo,
                                player_wind=(seat + self.n - self.k.ju) % self.n, round_wind=self.k.chang, honba=0,
                                kita_count=self.kita[seat])
                    if not zimo:
                        hand.append(win)
                    ura = list(h["li_doras"]) if (self.liqi[seat] and h.get("li_doras") is not None) else \
                        ([abi.mjai_to_tid(t, True) for t in self.k.ura_doras] if self.liqi[seat] else [])
                    self.pending.append(WinResultContext(seat=seat, tiles=hand, melds=[dict(m) for m in self.melds[seat]],
                                                         agari_tile=win, dora_indicators=list(self.doras), ura_indicators=ura,
                                                         conditions=cond, expected_yaku=[], expected_han=h.get("count", 0),
                                                         expected_fu=h.get("fu", 0), actual=None, sanma=self.n == 3))
                if self.pending:
                    return self.pending.pop(0)
        raise StopIteration


def evaluate_win_contexts(contexts, device=0):
    """`actual` of many WinResultContexts in ONE batch on the GPU: rmj_eval_hands = HandEvaluator(tiles, melds).calc(agari
    tile, dora, ura, conditions) (replay/mod.rs:2060-2068).  Returns the contexts."""
    contexts = list(contexts)
    if contexts:
        for c, r in zip(contexts, vecenv.eval_hands([c.hand_case() for c in contexts], device=device)):
            c.actual = r
    return contexts


class MjaiReplay:
    """MjaiReplay (replay/mjai_replay.rs:270-384): an MJAI log split into rounds.  `steps` of the reference's kyoku objects
    (observation, action) is ReplayBatch.samples() here (many logs in lock-step on the GPU)."""

    def __init__(self, rounds, events):
        self.rounds = rounds
        self.events = events

    @classmethod
    def from_jsonl(cls, path, rule=None):
        if rule not in (None, "tenhou", "mjsoul"):
            raise ValueError(f"Unknown rule: '{rule}'. Expected 'tenhou' or 'mjsoul'")   # mjai_replay.rs:282-287
        try:
            events = load_mjai_jsonl(path)
        except OSError as e:
            raise ValueError(f"Failed to open file: {e}")
        rounds, cur = [], None
        for ev in events:
            ty = ev.get("type")
            if ty == "start_kyoku":
                if cur is not None:
                    cur._flush()
                    rounds.append(cur)
                cur = Kyoku(ev)
            elif ty in ("end_kyoku", "end_game"):
                if cur is not None:
                    cur._flush()
                    rounds.append(cur)
                    cur = None
            elif cur is not None:
                cur._feed(ev)
        if cur is not None:
            cur._flush()
            rounds.append(cur)
        for i in range(len(rounds) - 1):          # the next round's start scores are authoritative (mjai_replay.rs:363-367)
            rounds[i].end_scores = list(rounds[i + 1].scores)
        return cls(rounds, events)

    def num_rounds(self):
        return len(self.rounds)

    def take_kyokus(self):
        return iter(self.rounds)


class ReplayBatch:
    """Lock-step replay of `logs` (lists of MJAI event dicts) on one GPU."""

    def __init__(self, logs, game_mode=2, device=0, extended=False, include_pass=True, masked_ok=False, env=None):
        self.logs = [list(l) for l in logs]
        self.n = len(self.logs)
        self.env = env or vecenv.VecRiichiEnv(self.n, game_mode=game_mode, seed=0, device=device, skip_mjai_logging=True)
        self.sanma = self.env.game_mode >= 3
        self.extended = extended
        self.include_pass = include_pass
        self.masked_ok = masked_ok

    def _encode_id(self, packed):
        t, tile, cons = abi.unpack_action(packed)
        from .compat import Action, ActionType

        a = Action(ActionType(t), tile, cons)
        return a.encode_3p() if self.sanma else a.encode()

    def _decisions_before(self, k, legal, cnt, active, drawn):
        """(game, seat, packed action) of every decision taken by event k of each log, given the state before it."""
        out = []
        for g, log in enumerate(self.logs):
            if k >= len(log):
                continue
            ev = log[k]
            ty = ev.get("type")
            act_mask = int(active[g])
            if ty in _ACTOR_DECISIONS or (ty == "ryukyoku" and act_mask and cnt[g].sum() > 0):
                seats = [int(ev["actor"])] if "actor" in ev else [s for s in range(4) if (act_mask >> s) & 1]
                for s in seats:
                    if not (act_mask >> s) & 1 or cnt[g, s] == 0:
                        continue
                    sel = mjai.select_action_from_mjai(legal[g, s, : cnt[g, s]], ev, drawn[g], self.sanma)
                    if sel is not None:
                        out.append((g, s, sel))
                if self.include_pass and ty in ("chi", "pon", "daiminkan", "kan", "hora"):
                    for s in range(4):      # the other seats that were offered a claim and let it go
                        if s != int(ev.get("actor", -1)) and (act_mask >> s) & 1 and cnt[g, s] > 0:
                            sel = mjai.select_action_from_mjai(legal[g, s, : cnt[g, s]], {"type": "none"}, None, self.sanma)
                            if sel is not None:
                                out.append((g, s, sel))
            elif self.include_pass and ty == "tsumo" and act_mask and self._phase[g] == abi.WAIT_RESPONSE:
                for s in range(4):          # everybody passed on the previous discard
                    if (act_mask >> s) & 1 and cnt[g, s] > 0:
                        sel = mjai.select_action_from_mjai(legal[g, s, : cnt[g, s]], {"type": "none"}, None, self.sanma)
                        if sel is not None:
                            out.append((g, s, sel))
        return out

    def samples(self):
        """Generator over event indices: yields dicts with `game`, `seat`, `action_id`, `action` (packed), `mask`
        ([k, 82] / [k, 60]) and `obs` ([k, C, W] f32) for the k decisions taken at that index."""
        steps = max((len(l) for l in self.logs), default=0)
        nmask = 60 if self.sanma else 82
        for k in range(steps):
            act, ph, done = self.env.status()
            self._phase = ph
            legal, cnt = self.env.legal()
            active = np.where(done.astype(bool), 0, act)
            drawn = [None] * self.n
            pending = [g for g in range(self.n) if k < len(self.logs[g]) and self.logs[g][k].get("type") == "dahai"]
            for g in pending:
                v = self.env.peek(g)
                drawn[g] = None if v.drawn_tile < 0 else int(v.drawn_tile)
            dec = self._decisions_before(k, legal, cnt, active, drawn)
            if dec:
                enc = self.env.encode_extended() if self.extended else self.env.encode()
                mask = self.env.mask()
                gs = np.array([d[0] for d in dec])
                ss = np.array([d[1] for d in dec])
                yield {"index": k, "game": gs, "seat": ss, "action": np.array([d[2] for d in dec], dtype=np.uint64),
                       "action_id": np.array([self._encode_id(d[2]) for d in dec], dtype=np.int64),
                       "mask": mask[gs, ss][:, :nmask].copy(), "obs": enc[gs, ss].copy()}
            self.env.apply_events([l[k] if k < len(l) else None for l in self.logs], masked_ok=self.masked_ok)
