"""MJAI logs -> (observation, action) samples in lock-step on the GPU (MJAI part).

The reference reads logs with MjaiReplay.from_jsonl and walks them with KyokuStepIterator on the host, one game at a time
(riichienv-core/src/replay/mjai_replay.rs:65-156, replay/mod.rs).  Here many logs advance together: event k of every
log is applied with rmj_apply_events (= RiichiEnv.apply_event), and whenever the NEXT event of a log is a player's
decision the acting seat's legal list, mask, feature tensor and the action it took (select_action_from_mjai,
observation/mjai_select.rs:88-194, encoded as the 82- / 60-way id) are emitted - the (obs, action) pairs of
behaviour-cloning / offline-RL datasets.  This is the event-stream semantics of apply_event (tile names map to one id
per name, no wall).  The host-side kyoku API of both log formats lives here as well: MjaiReplay / MjSoulReplay -> Kyoku
(events, grp_features) -> WinResultContextIterator -> evaluate_win_contexts (all wins in one rmj_eval_hands launch).
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
        head = {"name": "NewRound", "data": dict(scores=list(self.scores), doras=list(self.doras), dora_marker=self.doras[0] if self.doras else None,
                                                  chang=self.chang, ju=self.ju, ben=self.ben, liqibang=self.liqibang,
                                                  left_tile_count=self.left_tile_count,
                                                  **{f"tiles{i}": list(h) for i, h in enumerate(self.hands)})}
        return [head] + [{"name": a["name"], "data": {k: v for k, v in a.items() if k != "name"}} for a in self.actions]

    def to_mjai_events(self, reveal_listed_doras=True):
        """The round as MJAI events (start_kyoku ... end_kyoku), whatever its source: what ReplayBatch / rmj_apply_events
        consume.  For a Mahjong Soul round this is where the reference's LogKyoku.steps sets its state up (replay/mod.rs:
        1094-1290): a 14-tile hand marks the dealer, whose drawn tile is the tile of the first action if that is the dealer's
        own discard / tsumo / kan, else the last tile of the list; the deposit of a riichi is confirmed (reach_accepted) by the
        next draw / call / draw-end and dropped by a Ron (apply_log_action, state/event_handler.rs:396-423, :683-688); new
        indicators listed by a draw are revealed before it, those listed by a discard after it; tsumogiri = the discard is the
        drawn tile (event_handler.rs:343-347).  reveal_listed_doras=False keeps the reference's step iterator as it is:
        apply_log_action ignores the `doras` lists of DealTile / DiscardTile, so its observations of a Mahjong Soul record
        never show a kan dora (only explicit `dora` actions do); the default reveals them."""
        if getattr(self, "source", None) != "mjsoul":
            return list(self.mjai_events)
        from .mjai import tid_to_mjai

        n = len(self.scores)
        kind = lambda t: t[:2] if t[0].isdigit() else t      # noqa: E731  ("5mr" and "5m" are one type)
        hands = [list(h) for h in self.hands[:n]]
        oya = next((i for i, h in enumerate(hands) if len(h) == 14), self.ju % max(n, 1))
        drawn = [None] * n
        ev = []
        first_draw = None
        if len(hands[oya]) == 14:
            first_draw = hands[oya][-1]
            a0 = self.actions[0] if self.actions else {"name": "Other"}
            if a0["name"] == "Hule":
                first_draw = next((tid_to_mjai(h["hu_tile"]) for h in a0["hules"] if h["seat"] == oya and h["zimo"]), first_draw)
            elif a0["name"] == "DiscardTile" and a0["seat"] == oya:
                first_draw = a0["tile"]
            elif a0["name"] == "AnGangAddGang" and a0["seat"] == oya:
                first_draw = a0["tiles"][0]
            if first_draw not in hands[oya]:
                first_draw = next((t for t in hands[oya] if kind(t) == kind(first_draw)), hands[oya][-1])
            hands[oya].remove(first_draw)
        ev.append({"type": "start_kyoku", "bakaze": "ESWN"[self.chang] if 0 <= self.chang < 4 else "E", "kyoku": self.ju + 1, "honba": self.ben,
                   "kyotaku": self.liqibang, "oya": oya, "scores": list(self.scores), "dora_marker": self.doras[0] if self.doras else "?",
                   "tehais": [list(h) for h in hands]})
        if first_draw is not None:
            ev.append({"type": "tsumo", "actor": oya, "pai": first_draw})
            hands[oya].append(first_draw)
            drawn[oya] = first_draw
        known = max(len(self.doras), 1)
        melds = [[] for _ in range(n)]
        pending_reach = None
        last_discard = None

        def accept():
            nonlocal pending_reach
            if pending_reach is not None:
                ev.append({"type": "reach_accepted", "actor": pending_reach})
                pending_reach = None

        def reveal(listed):
            nonlocal known
            for t in ((listed or []) if reveal_listed_doras else [])[known:]:
                ev.append({"type": "dora", "dora_marker": t})
                known += 1

        def take(seat, t):
            h = hands[seat]
            if t in h:
                h.remove(t)
            else:
                k = next((x for x in h if kind(x) == kind(t)), None)
                if k is not None:
                    h.remove(k)

        for a in self.actions:
            name = a["name"]
            if name == "DealTile":
                accept()
                reveal(a.get("doras"))
                s = a["seat"]
                ev.append({"type": "tsumo", "actor": s, "pai": a["tile"]})
                hands[s].append(a["tile"])
                drawn[s] = a["tile"]
            elif name == "DiscardTile":
                s = a["seat"]
                riichi = a["is_liqi"] or a["is_wliqi"]
                if riichi:
                    ev.append({"type": "reach", "actor": s})
                ev.append({"type": "dahai", "actor": s, "pai": a["tile"], "tsumogiri": drawn[s] == a["tile"]})
                take(s, a["tile"])
                drawn[s] = None
                last_discard = s
                if riichi:
                    pending_reach = s
                reveal(a.get("doras"))
            elif name == "ChiPengGang":
                accept()
                s = a["seat"]
                own = [t for t, f in zip(a["tiles"], a["froms"]) if f == s]
                other = [(t, f) for t, f in zip(a["tiles"], a["froms"]) if f != s]
                pai, target = other[0] if other else (a["tiles"][0], last_discard if last_discard is not None else 0)
                ty = {"Chi": "chi", "Pon": "pon", "Daiminkan": "daiminkan"}.get(a["meld_type"], "chi")
                ev.append({"type": ty, "actor": s, "target": target, "pai": pai, "consumed": own})
                for t in own:
                    take(s, t)
                melds[s].append((ty, pai, own))
                drawn[s] = None
            elif name == "AnGangAddGang":
                s, t = a["seat"], a["tiles"][0]
                if a["meld_type"] == "Ankan":
                    own = [x for x in hands[s] if kind(x) == kind(t)][:4]
                    own += [kind(t)] * (4 - len(own))
                    ev.append({"type": "ankan", "actor": s, "consumed": own})
                    for x in own:
                        take(s, x)
                else:
                    pon = next((m for m in melds[s] if m[0] == "pon" and kind(m[1]) == kind(t)), None)
                    ev.append({"type": "kakan", "actor": s, "pai": t, "consumed": ([pon[1]] + list(pon[2])) if pon else [kind(t)] * 3})
                    take(s, t)
                last_discard = s
                drawn[s] = None
            elif name == "BaBei":
                accept()
                s = a["seat"]
                ev.append({"type": "kita", "actor": s, "pai": "N"})
                take(s, "N")
                last_discard = s
                drawn[s] = None
            elif name == "Dora":
                ev.append({"type": "dora", "dora_marker": a["dora_marker"]})
                known += 1
            elif name == "Hule":
                if a["hules"] and not a["hules"][0]["zimo"]:
                    pending_reach = None
                for h in a["hules"]:
                    ev.append({"type": "hora", "actor": h["seat"], "target": h["seat"] if h["zimo"] else (last_discard if last_discard is not None else 0),
                               "pai": tid_to_mjai(h["hu_tile"])})
            elif name == "NoTile":
                accept()
                ev.append({"type": "ryukyoku"})
            elif name == "LiuJu":
                if n == 4:
                    accept()
                ev.append({"type": "ryukyoku", "reason": "abortive"})
        ev.append({"type": "end_kyoku"})
        return ev

    rule = "tenhou"

    def steps(self, seat=None, rule=None, skip_single_action=True, device=0, extended=False):
        """LogKyoku.steps (replay/mod.rs:1094-1290, KyokuStepIterator :199-560): the decisions of this round, one by one, as
        `(obs, action)` for one seat or `(seat, obs, action)` for all - obs a StepObservation (feature tensor, mask, legal
        actions of the deciding seat), action the compat.Action the log took.  A riichi is two decisions (Riichi, then the
        discard with Riichi no longer offered); seats that let a claim go yield Pass decisions, delivered before the decision of
        the seat that did claim, highest seat first (the iterator pops its queue from the back, :211); skip_single_action drops
        decisions with at most one legal action.  The round runs as a one-game ReplayBatch on the GPU - datasets over many
        logs should use ReplayBatch directly, which replays them in lock-step."""
        n = len(self.scores)
        bits = abi.RULE_MJSOUL if (rule or self.rule) == "mjsoul" else abi.RULE_TENHOU
        key = (5 if n == 3 else 2, bits, device)
        pool = _STEP_ENVS.setdefault(key, [])       # one-game environments are reused: a start_kyoku event rewrites the round
        env = pool.pop() if pool else vecenv.VecRiichiEnv(1, game_mode=key[0], seed=0, rule_bits=bits, device=device, skip_mjai_logging=True)
        try:
            yield from self._steps_on(env, seat, skip_single_action, extended)
        finally:
            pool.append(env)

    def _steps_on(self, env, seat, skip_single_action, extended):
        from .compat import Action

        n = len(self.scores)
        rb = ReplayBatch([[{"type": "start_game"}] + self.to_mjai_events()], extended=extended, include_pass=True, env=env)
        for smp in rb.samples():
            order = list(range(len(smp["seat"])))
            is_pass = [abi.unpack_action(int(a))[0] == abi.PASS for a in smp["action"]]
            order = sorted((j for j in order if is_pass[j]), key=lambda j: -int(smp["seat"][j])) + [j for j in order if not is_pass[j]]
            for j in order:
                s = int(smp["seat"][j])
                legal = [int(a) for a in smp["legal"][j]]
                if seat is not None and s != seat:
                    continue
                if skip_single_action and len(legal) <= 1:
                    continue
                obs = StepObservation(s, n, smp["obs"][j], smp["mask"][j], legal)
                act = Action._from_packed(int(smp["action"][j]), actor=s if is_pass[j] else None)
                yield (obs, act) if seat is not None else (s, obs, act)

    def take_win_result_contexts(self, ankan_from_consumed=True):
        """LogKyoku.take_win_result_contexts (replay/mod.rs:1089-1091)"""
        return WinResultContextIterator(self, ankan_from_consumed)

    game_end_scores = None        # set by MjSoulReplay.from_dict (None for MJAI logs, mjai_replay.rs:266)

    @staticmethod
    def _ranks(scores):
        """get_ranks (replay/mod.rs:1542-1552): 0 = top; equal scores rank by seat index"""
        order = sorted(range(len(scores)), key=lambda i: (-scores[i], i))
        ranks = [0] * len(scores)
        for r, seat in enumerate(order):
            ranks[seat] = r
        return ranks

    def take_grp_features(self):
        """LogKyoku.take_grp_features (replay/mod.rs:1524-1589): the round's scores / ranks before and after, the final ranks of
        the game (from game_end_scores when the reader computed them, else the round's own end ranks) and the dealt hands as ids"""
        ini = list(self.scores)
        end = list(self.end_scores) if self.end_scores else ini
        ri, re_ = self._ranks(ini), self._ranks(end)
        out = dict(chang=self.chang, ju=self.ju, ben=self.ben, liqibang=self.liqibang,
                   round_initial_scores=ini, round_end_scores=end, round_delta_scores=[e - s_ for s_, e in zip(ini, end)],
                   round_initial_ranks=ri, round_end_ranks=re_, round_delta_ranks=[e - s_ for s_, e in zip(ri, re_)],
                   final_ranks=self._ranks(self.game_end_scores) if self.game_end_scores is not None else list(re_))
        for i, h in enumerate(self.hands):
            out[f"player{i}_initial_hand_tids"] = [abi.mjai_to_tid(t, True) for t in h]
        return out

    def grp_features(self):
        """LogKyoku.grp_features (replay/mod.rs:1502-1522)"""
        return dict(chang=self.chang, ju=self.ju, ben=self.ben, liqibang=self.liqibang, scores=list(self.scores),
                    end_scores=list(self.end_scores), wliqi=list(self.wliqi),
                    delta_scores=[e - s for s, e in zip(self.scores, self.end_scores)] if len(self.scores) == len(self.end_scores) else [])


_STEP_ENVS = {}   # (game mode, rule bits, device) -> idle one-game environments of Kyoku.steps


class StepObservation:
    """What a decision of Kyoku.steps carries of the reference's Observation: the seat, its feature tensor (encode() = the
    74 x 34 / 74 x 27 block, or the 215-channel one with extended=True), the action mask and the legal actions."""

    __slots__ = ("player_id", "num_players", "tensor", "_mask", "_legal")

    def __init__(self, player_id, num_players, tensor, mask, legal):
        self.player_id, self.num_players, self.tensor, self._mask, self._legal = player_id, num_players, tensor, mask, legal

    @property
    def action_space_size(self):   # (a getter, like the reference's Observation: observation/python.rs:113-116)
        return 60 if self.num_players == 3 else 82

    def legal_actions(self):
        from .compat import Action

        return [Action._from_packed(a) for a in self._legal]

    def mask(self) -> bytes:
        return bytes(bytearray(int(x) for x in self._mask))

    def encode(self) -> bytes:
        return self.tensor.tobytes()


class WinResultContext:
    """WinResultContext (replay/mod.rs:2096-2180): the evaluator inputs of one win as the replay reconstructs them, the
    log's own expectation (MJAI logs carry none: han / fu 0, no yaku) and `actual` = HandEvaluator.calc of those inputs
    (filled in batch by evaluate_win_contexts)."""

    __slots__ = ("seat", "tiles", "melds", "agari_tile", "dora_indicators", "ura_indicators", "conditions", "expected_yaku",
                 "expected_han", "expected_fu", "actual", "sanma")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    def create_calculator(self):
        """replay/mod.rs:2160-2163: a HandEvaluator (3P: HandEvaluator3P) over this context's tiles and melds"""
        from .compat import Meld
        from .hand import HandEvaluator, HandEvaluator3P

        melds = [Meld(m["meld_type"], list(m["tiles"]), m["opened"], m["from_who"], m["called_tile"]) for m in self.melds]
        return (HandEvaluator3P if self.sanma else HandEvaluator)(list(self.tiles), melds)

    def calculate(self, calculator, conditions=None):
        """replay/mod.rs:2165-2179: calculator.calc(agari tile, indicators, ura indicators, conditions or the context's own)"""
        from .hand import Conditions

        cond = conditions
        if cond is None:
            c = self.conditions
            cond = Conditions(**{k: c[k] for k in ("tsumo", "riichi", "double_riichi", "ippatsu", "haitei", "houtei", "rinshan", "chankan",
                                                    "tsumo_first_turn", "player_wind", "round_wind", "honba", "kita_count")})
        return calculator.calc(self.agari_tile, list(self.dora_indicators), cond, list(self.ura_indicators))

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


_MS_HONORS = "ESWNPFC"


def mjsoul_tile_to_mjai(t):
    """MjSoul tile string -> MJAI name, by TileConverter::parse_tile (replay/mod.rs:2184-2202): digit + suit, 0 = red five,
    z1..z7 = E S W N P F C; malformed strings fall back to the reference's defaults (number 0 -> red five, suit -> 1m)."""
    if not t:
        return "1m"
    d, su = t[0], t[1:]
    num = int(d) if d.isdigit() else 0
    red = num == 0
    num = 5 if red else num
    if su == "z":
        return _MS_HONORS[num - 1] if 1 <= num <= 7 else "1m"
    if su not in ("m", "p", "s"):
        return "1m"
    return f"{num}{su}" + ("r" if red else "")


def _parse_paishan(s):
    """parse_paishan (replay/mod.rs:1627-1637): two characters per tile, draw order"""
    return [abi.mjai_to_tid(mjsoul_tile_to_mjai(s[i:i + 2]), True) for i in range(0, len(s) - 1, 2)]


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
        if getattr(kyoku, "source", "mjai") == "mjsoul":
            self.doras = [abi.mjai_to_tid(t, True) for t in kyoku.doras]          # current_doras = kyoku.doras (replay/mod.rs:1663)
            self.left = kyoku.left_tile_count                                       # the header's count (default 70, :443)
        else:
            self.doras = [abi.mjai_to_tid(kyoku.doras[0], True)]  # an MJAI kyoku's doras grow with its dora events: start from the header's
            self.left = 55 if n == 3 else 70
        self.kita = [0] * 4
        self.ankan_from_consumed = ankan_from_consumed
        # wall-dependent bookkeeping (replay/mod.rs:1634-1724): MjSoul records may carry the wall (`paishan`)
        self.wall = _parse_paishan(kyoku.paishan) if getattr(kyoku, "paishan", None) else []
        self.dora_count = 1
        self.pending_minkan_doras = 0

    def __iter__(self):
        return self

    def _recalc_doras(self):      # replay/mod.rs:1673-1690: indicator i of the wall sits at len - 5 - 2i
        if not self.wall:
            return
        n = len(self.wall)
        self.doras = [self.wall[n - 5 - 2 * i] for i in range(self.dora_count) if n >= 5 + 2 * i]

    def _sync_doras_with_wall(self):   # replay/mod.rs:1692-1708
        if not self.wall:
            return
        if len(self.doras) > self.dora_count:      # the log shows more indicators: trust it
            self.dora_count = len(self.doras)
            self.pending_minkan_doras = 0
        elif self.dora_count > len(self.doras):
            self._recalc_doras()

    def _ura_from_wall(self):          # replay/mod.rs:1710-1724
        n = len(self.wall)
        return [self.wall[n - 6 - 2 * i] for i in range(self.dora_count) if n >= 6 + 2 * i]

    def _flush_pending_doras(self):
        if self.pending_minkan_doras > 0:
            self.dora_count += self.pending_minkan_doras
            self.pending_minkan_doras = 0

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
                if a.get("doras") is not None:
                    self.doras = [abi.mjai_to_tid(t, True) for t in a["doras"]]
                self._flush_pending_doras()          # a discard reveals the pending kan indicators
                self._sync_doras_with_wall()
            elif name == "DealTile":
                seat = a["seat"]
                self._after_kakan_reset()
                self.hands[seat].append(abi.mjai_to_tid(a["tile"], True))
                if a.get("left_tile_count") is not None:
                    self.left = a["left_tile_count"]
                elif self.left > 0:
                    self.left -= 1
                if a.get("doras") is not None:       # a draw that shows indicators is a replacement draw
                    self.doras = [abi.mjai_to_tid(t, True) for t in a["doras"]]
                    self.rinshan[seat] = True
                self._sync_doras_with_wall()
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
                    self._flush_pending_doras()
                    self.pending_minkan_doras += 1
            elif name == "Dora":
                if not self.wall:
                    self.doras.append(abi.mjai_to_tid(a["dora_marker"], True))     # no wall: replay/mod.rs:1869-1871
                else:
                    self.dora_count += 1
                    if self.pending_minkan_doras > 0:
                        self.pending_minkan_doras -= 1
                    self._sync_doras_with_wall()
            elif name == "AnGangAddGang":
                seat = a["seat"]
                self.rinshan = [False] * 4
                tl = [abi.mjai_to_tid(t, True) for t in a["tiles"]]
                self._flush_pending_doras()          # a new kan flushes the indicators pending from an earlier open kan
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
                    if self.wall:
                        self.dora_count += 1         # a concealed kan shows its indicator at once
                else:
                    self.pending_minkan_doras += 1   # an added kan shows it after the discard
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
                self._sync_doras_with_wall()
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
                    if not self.liqi[seat]:
                        ura = []
                    elif h.get("li_doras") is not None:
                        ura = list(h["li_doras"])
                    elif self.wall:
                        ura = self._ura_from_wall()
                    else:
                        ura = [abi.mjai_to_tid(t, True) for t in self.k.ura_doras]
                    self.pending.append(WinResultContext(seat=seat, tiles=hand, melds=[dict(m) for m in self.melds[seat]],
                                                         agari_tile=win, dora_indicators=list(self.doras), ura_indicators=ura,
                                                         conditions=cond, expected_yaku=list(h.get("fans", [])), expected_han=h.get("count", 0),
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


def _gpu_tenpai(cases, device=0):
    """HandEvaluator(hand, melds).is_tenpai() of many hands in one rmj_eval_hands launch"""
    from .vecenv import eval_hands

    return [bool(r.is_tenpai) for r in eval_hands(cases, device=device)] if cases else []


class LogRoundWalker:
    """GameState / GameState3P::apply_log_action (state/event_handler.rs:332-891, state_3p/event_handler.rs:365-840) reduced to
    what decides the scores: the record's actions move tiles between hands, melds and rivers, riichi deposits are taken when the
    next draw / call / draw-end confirms them (a Ron on the riichi discard voids the deposit), a Hule pays the points the RECORD
    carries (point_rong / point_zimo_qin / point_zimo_xian, honba added here, pao split from the melds the walker saw), an
    exhaustive draw pays nagashi mangan or the tenpai / noten payments.  The one piece of hand math - is_tenpai of the hands at
    NoTile - is left to the caller (`pending_cases` -> `finish`), so many records resolve it in one GPU launch."""

    _NO_COND = dict(tsumo=False, riichi=False, double_riichi=False, ippatsu=False, haitei=False, houtei=False, rinshan=False,
                    chankan=False, tsumo_first_turn=False, player_wind=0, round_wind=0, honba=0, kita_count=0)

    def __init__(self, k: Kyoku):
        n = self.n = len(k.scores)
        if n not in (3, 4):              # try_into().unwrap_or(...) of mjsoul_replay.rs:266, :299
            n = self.n = 4
            self.scores = [25000] * 4
        else:
            self.scores = list(k.scores)
        self.oya = k.ju % n
        self.honba, self.sticks = k.ben, k.liqibang
        self.hands = [sorted(abi.mjai_to_tid(t, True) for t in (k.hands[i] if i < len(k.hands) else [])) for i in range(n)]
        self.melds = [[] for _ in range(n)]
        self.pao = [dict() for _ in range(n)]
        self.nagashi = [True] * n
        self.declared = [False] * n
        self.pending_riichi = None
        self.last_discard = None
        self.current = self.oya
        self.actions = list(k.actions)
        self.pos = 0
        self.pending_cases = None

    def _take_deposit(self):
        if self.pending_riichi is not None:
            self.scores[self.pending_riichi] -= 1000
            self.sticks += 1
            self.pending_riichi = None

    @staticmethod
    def _remove(hand, t):
        if t in hand:
            hand.remove(t)

    def run(self):
        """Applies actions until the end, or until a NoTile needs the tenpai flags (then pending_cases holds one HandCase per
        seat and finish() resumes).  Returns True when the round is complete."""
        while self.pos < len(self.actions):
            a = self.actions[self.pos]
            self.pos += 1
            name = a["name"]
            n = self.n
            if name == "DiscardTile":
                s, t = a["seat"], abi.mjai_to_tid(a["tile"], True)
                self._remove(self.hands[s], t)
                self.nagashi[s] = self.nagashi[s] and (t // 4 >= 27 or (t // 4) % 9 in (0, 8))
                if a["is_liqi"] or a["is_wliqi"]:
                    if n == 3 or not self.declared[s]:
                        self.pending_riichi = s
                    self.declared[s] = True
                self.last_discard = (s, t)
                self.current = (s + 1) % n
            elif name == "DealTile":
                self._take_deposit()
                self.hands[a["seat"]].append(abi.mjai_to_tid(a["tile"], True))
                self.current = a["seat"]
            elif name == "ChiPengGang":
                self._take_deposit()
                s = a["seat"]
                if self.last_discard is not None:
                    self.nagashi[self.last_discard[0]] = False
                tiles = [abi.mjai_to_tid(t, True) for t in a["tiles"]]
                froms = list(a["froms"])
                for i, t in enumerate(tiles):
                    if i < len(froms) and froms[i] == s:
                        self._remove(self.hands[s], t)
                other = [(t, f) for t, f in zip(tiles, froms) if f != s]
                from_who = other[0][1] if other else -1
                ct = other[0][0] if other else None
                kind = a["meld_type"]
                self.melds[s].append(dict(meld_type=_MELD_ID[kind], tiles=tiles, opened=True, from_who=from_who, called_tile=ct))
                if kind in ("Pon", "Daiminkan") and ct is not None:
                    groups = [m["tiles"][0] // 4 for m in self.melds[s] if m["meld_type"] != abi.MELD_CHI]
                    if 31 <= ct // 4 <= 33 and sum(31 <= g <= 33 for g in groups) == 3:
                        self.pao[s][37] = max(from_who, 0)
                    elif 27 <= ct // 4 <= 30 and sum(27 <= g <= 30 for g in groups) == 4:
                        self.pao[s][50] = max(from_who, 0)
                self.current = s
            elif name == "AnGangAddGang":
                s, t = a["seat"], abi.mjai_to_tid(a["tiles"][0], True)
                if a["meld_type"] == "Ankan":
                    for _ in range(4):
                        for i, x in enumerate(self.hands[s]):
                            if x // 4 == t // 4:
                                del self.hands[s][i]
                                break
                    b = t // 4 * 4
                    self.melds[s].append(dict(meld_type=abi.MELD_ANKAN, tiles=[b, b + 1, b + 2, b + 3], opened=False, from_who=-1,
                                              called_tile=None))
                else:
                    self._remove(self.hands[s], t)
                    for m in self.melds[s]:
                        if m["meld_type"] == abi.MELD_PON and m["tiles"][0] // 4 == t // 4:
                            m["meld_type"] = abi.MELD_KAKAN
                            m["tiles"] = sorted(m["tiles"] + [t])
                            break
                self.last_discard = (s, t)
                self.current = s
            elif name == "BaBei" and n == 3:
                self._take_deposit()
                s = a["seat"]
                for i, x in enumerate(self.hands[s]):
                    if x // 4 == 30:
                        self.last_discard = (s, x)
                        del self.hands[s][i]
                        break
                self.current = s
            elif name == "Hule":
                self._hule(a["hules"])
            elif name == "NoTile":
                self._take_deposit()
                winners = [i for i in range(n) if self.nagashi[i]]
                if winners:
                    for w in winners:          # mangan tsumo without honba: calculate_score(5, 30, is_oya, true, 0, np)
                        for i in range(n):
                            if i != w:
                                pay = 4000 if (w == self.oya or i == self.oya) else 2000
                                self.scores[i] -= pay
                                self.scores[w] += pay
                else:
                    self.pending_cases = [WinResultContext(seat=i, tiles=list(self.hands[i]), melds=[dict(m) for m in self.melds[i]],
                                                           agari_tile=0, dora_indicators=[], ura_indicators=[],
                                                           conditions=self._NO_COND, sanma=n == 3).hand_case() for i in range(n)]
                    return False
            elif name == "LiuJu":
                if n == 4:                     # (the 3P handler leaves a pending deposit alone, state_3p/event_handler.rs:834-837)
                    self._take_deposit()
        return True

    def finish(self, tenpai):
        """the tenpai flags of pending_cases -> the noten payments (3000 shared among four, 2000 among three), then the rest"""
        n = self.n
        self.pending_cases = None
        num = sum(bool(t) for t in tenpai)
        if 0 < num < n:
            pot = 3000 if n == 4 else 2000
            for i, tp in enumerate(tenpai):
                self.scores[i] += pot // num if tp else -(pot // (n - num))
        return self.run()

    def _hule(self, hules):
        n, oya = self.n, self.oya
        tsumo = lambda h: h["zimo"] and (n == 4 or h["seat"] == self.current)   # noqa: E731  (3P: a zimo flag on another seat's turn is a Ron)
        if hules and not tsumo(hules[0]):
            self.pending_riichi = None
        honba, sticks, honba_taken = self.honba, self.sticks, False
        for h in hules:
            w = h["seat"]
            total = pao_val = 0
            payer = None
            if h.get("yiman"):
                for y in h.get("fans", []):
                    val = 2 if y in (47, 48, 49, 50) else 1
                    total += val
                    if y in self.pao[w]:
                        pao_val += val
                        payer = self.pao[w][y]
            qin, xian, rong = int(h.get("point_zimo_qin", 0)), int(h.get("point_zimo_xian", 0)), int(h.get("point_rong", 0))

            def move(src, amount):
                self.scores[src] -= amount
                self.scores[w] += amount

            if tsumo(h):
                is_oya = w == oya
                if pao_val > 0 and n == 4:
                    unit = 48000 if is_oya else 32000
                    move(payer, pao_val * unit)
                    rest = (total - pao_val) * unit
                    if rest > 0:
                        for i in range(4):
                            if i != w:
                                move(i, rest // 3 if is_oya else (rest // 2 if i == oya else rest // 4))
                    move(payer, honba * 300)
                elif pao_val > 0:
                    whole = xian * 2 if is_oya else qin + xian
                    pao_amt = whole * pao_val // total
                    rest = whole - pao_amt
                    move(payer, pao_amt)
                    if rest > 0:
                        for i in range(3):
                            if i != w:
                                move(i, rest // 2 if is_oya else ((qin if i == oya else xian) * rest // whole))
                    move(payer, honba * 200)
                else:
                    for i in range(n):
                        if i != w:
                            move(i, (xian if (is_oya or i != oya) else qin) + honba * 100)
            elif self.last_discard is not None:
                target = self.last_discard[0]
                hb = 0 if honba_taken else honba * 100 * (n - 1)
                honba_taken = True
                if n == 4:
                    # (the 4P handler takes the first pao entry among the fans, whatever the yakuman count)
                    p4 = next((self.pao[w][y] for y in h.get("fans", []) if h.get("yiman") and y in self.pao[w]), None)
                    if p4 is not None:           # the liable seat and the discarder pay half each; the winner receives point_rong
                        self.scores[p4] -= rong // 2 + hb
                        self.scores[target] -= rong // 2
                        self.scores[w] += rong + hb
                    else:
                        move(target, rong + hb)
                elif pao_val > 0:
                    pao_amt = rong * pao_val // total
                    move(payer if payer is not None else target, pao_amt // 2 + hb)
                    move(target, rong - pao_amt // 2)
                else:
                    move(target, rong + hb)
        if hules:
            self.scores[hules[0]["seat"]] += sticks * 1000
            self.sticks = 0


def game_end_scores(kyokus, tenpai=None, device=0):
    """The end scores of many rounds (the last round of each record: mjsoul_replay.rs:259-339) - every walker runs on the host,
    the is_tenpai checks of all exhaustive draws go out as ONE rmj_eval_hands batch.  `tenpai` maps a list of HandCases to
    flags (default: the GPU)."""
    walkers = [LogRoundWalker(k) for k in kyokus]
    waiting = [w for w in walkers if not w.run()]
    while waiting:
        cases = [c for w in waiting for c in w.pending_cases]
        flags = (tenpai or (lambda cs: _gpu_tenpai(cs, device)))(cases)
        nxt, at = [], 0
        for w in waiting:
            k = len(w.pending_cases)
            if not w.finish(flags[at: at + k]):
                nxt.append(w)
            at += k
        waiting = nxt
    return [list(w.scores) for w in walkers]


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
        for k in rounds:
            k.rule = rule or "tenhou"
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

    def _robbed_kan_tile(self, log, k):
        """The tile a `hora` at index k robs: the log's previous action is a kakan (chankan) or an ankan (kokushi) by the hora's
        target - `dora` events in between do not count - else None.  The event-stream state machine sets up no claims behind a kan
        (event_handler.rs:290-305), so this decision is not in the published lists; the reference's walker still yields it:
        apply_log_action leaves the kan tile in last_discard (state/event_handler.rs:649-661) and the Hule arm builds a Ron on it
        (replay/mod.rs:483-527)."""
        ev = log[k]
        if ev.get("actor") is None or ev.get("target") is None or int(ev["actor"]) == int(ev["target"]):
            return None
        j = k - 1
        while j >= 0 and log[j].get("type") == "dora":
            j -= 1
        if j < 0 or int(log[j].get("actor", -1)) != int(ev["target"]):
            return None
        if log[j].get("type") == "kakan":
            return abi.mjai_to_tid(log[j]["pai"], self.masked_ok)
        if log[j].get("type") == "ankan" and log[j].get("consumed"):
            return abi.mjai_to_tid(log[j]["consumed"][0], self.masked_ok)
        return None

    def _decisions_before(self, k, legal, cnt, active, drawn):
        """(game, seat, packed action) of every decision taken by event k of each log, given the state before it.  A fourth member
        marks a decision the published lists do not hold (the Ron on a robbed kan): the walker's own list, see samples()."""
        out = []
        for g, log in enumerate(self.logs):
            if k >= len(log):
                continue
            ev = log[k]
            ty = ev.get("type")
            act_mask = int(active[g])
            if ty == "hora" and not self._done[g] and not (act_mask >> int(ev.get("actor", 0))) & 1:
                tile = self._robbed_kan_tile(log, k)
                if tile is not None and not (k > 0 and log[k - 1].get("type") == "hora"):   # (the first winner only, replay/mod.rs:483-485)
                    ron, pas = abi.pack_action(abi.RON, tile), abi.pack_action(abi.PASS)
                    out.append((g, int(ev["actor"]), ron, [ron, pas]))
                    continue
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
            self._done = done.astype(bool)
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
                m = mask[gs, ss][:, :nmask].copy()
                lg = [legal[g, s, : cnt[g, s]].copy() for g, s in zip(gs, ss)]
                for j, d in enumerate(dec):
                    if len(d) > 3:   # the Ron on a robbed kan: what get_observation_for_replay builds (state/mod.rs:265-325) - the walker's
                        # current_claims are empty outside its pass look-ahead, so the seat's list is the pushed Ron and Pass
                        lg[j] = np.array(d[3], dtype=np.uint64)
                        m[j] = 0
                        for a in d[3]:
                            m[j, self._encode_id(int(a))] = 1
                yield {"index": k, "game": gs, "seat": ss, "action": np.array([d[2] for d in dec], dtype=np.uint64),
                       "action_id": np.array([self._encode_id(d[2]) for d in dec], dtype=np.int64),
                       "mask": m, "obs": enc[gs, ss].copy(), "legal": lg}
            self.env.apply_events([l[k] if k < len(l) else None for l in self.logs], masked_ok=self.masked_ok, replay=True)


class MjSoulReplay:
    """MjSoulReplay (replay/mjsoul_replay.rs:20-688): Mahjong Soul records - per round a list of `{"name": ..., "data": ...}`
    actions starting with NewRound - as the same Kyoku objects MjaiReplay yields (tile names are translated to MJAI names, the
    action vocabulary is the reference's own, replay/mod.rs:35-80).  The reference ships no MjSoul record; the reader follows
    the serde schema of mjsoul_replay.rs (field aliases, defaults, `#[serde(other)]`) and is tested on records converted from
    games the oracle played (tests/test_mjsoul_replay.py).  from_dict also yields game_end_scores (the last round replayed by
    LogRoundWalker, the reference's GameState.apply_log_action; from_json does not, as in the reference)."""

    _MELD = {0: "Chi", 1: "Pon", 2: "Daiminkan", 3: "Ankan"}

    def __init__(self, rounds):
        self.rounds = rounds

    @classmethod
    def from_json(cls, path):
        """mjsoul_replay.rs:168-194: a gzip file holding {"rounds": [[action, ...], ...]} (plain JSON is accepted as well)"""
        try:
            with open(path, "rb") as f:
                raw = f.read()
        except OSError as e:
            raise ValueError(f"Failed to open file: {e}")
        if raw[:2] == b"\x1f\x8b":
            raw = gzip.decompress(raw)
        try:
            log = json.loads(raw)
        except ValueError as e:
            raise ValueError(f"Failed to parse JSON: {e}")
        return cls._from_rounds(log["rounds"])

    @classmethod
    def from_dict(cls, paifu, tenpai=None, device=0):
        """mjsoul_replay.rs:196-342: a paifu dict {"data": rounds, ...} or the list of rounds itself.  Like the reference this
        entry also replays the last round through the log walker (LogRoundWalker = apply_log_action): its result is the last
        round's end_scores and every round's game_end_scores (take_grp_features' final_ranks).  `tenpai`: see game_end_scores."""
        return cls.from_dicts([paifu], tenpai=tenpai, device=device)[0]

    @classmethod
    def from_dicts(cls, paifus, tenpai=None, device=0):
        """from_dict of many records; the tenpai checks of all their final exhaustive draws share one GPU launch"""
        out = []
        for paifu in paifus:
            if isinstance(paifu, dict):
                if "data" not in paifu:
                    raise ValueError("Invalid dict format: missing 'data'")
                out.append(cls._from_rounds(paifu["data"]))
            elif isinstance(paifu, list):
                out.append(cls._from_rounds(paifu))
            else:
                raise ValueError("Invalid input format: expected dict or list")
        with_rounds = [r for r in out if r.rounds]
        for r, ges in zip(with_rounds, game_end_scores([r.rounds[-1] for r in with_rounds], tenpai=tenpai, device=device)):
            r.rounds[-1].end_scores = list(ges)
            for k in r.rounds:
                k.game_end_scores = list(ges)
        return out

    @classmethod
    def _from_rounds(cls, rounds_raw):
        rounds = [cls._kyoku_from_raw_actions(r) for r in rounds_raw]
        for i in range(len(rounds) - 1):           # the next round's start scores (mjsoul_replay.rs:189-191)
            rounds[i].end_scores = list(rounds[i + 1].scores)
        return cls(rounds)

    @classmethod
    def _kyoku_from_raw_actions(cls, raw):        # mjsoul_replay.rs:440-561
        k = Kyoku.__new__(Kyoku)
        k.source = "mjsoul"
        k.rule = "mjsoul"                         # mjsoul_replay.rs:541
        k.mjai_events = []
        k.scores, k.doras, k.ura_doras, k.hands = [], [], [], [[] for _ in range(4)]
        k.chang = k.ju = k.ben = k.liqibang = 0
        k.left_tile_count, k.paishan = 70, None
        head = raw[0] if raw else None
        if head is not None and head.get("name") == "NewRound":
            d = head.get("data", {})
            k.scores = list(d["scores"])
            da = d.get("dora_indicators") if d.get("dora_indicators") is not None else d.get("doras")
            if da is not None:
                k.doras = [mjsoul_tile_to_mjai(t) for t in da]
            elif d.get("dora_marker") is not None:
                k.doras = [mjsoul_tile_to_mjai(d["dora_marker"])]
            k.hands = [[mjsoul_tile_to_mjai(t) for t in d[f"tiles{i}"]] for i in range(4)]
            k.chang, k.ju = int(d["chang"]), int(d["ju"])
            k.ben = int(d["ben"] if d.get("ben") is not None else (d.get("honba") or 0))
            k.liqibang = int(d["liqibang"])
            k.left_tile_count = int(d["left_tile_count"]) if d.get("left_tile_count") is not None else 70
            if d.get("ura_doras") is not None:
                k.ura_doras = [mjsoul_tile_to_mjai(t) for t in d["ura_doras"]]
            k.paishan = d.get("paishan")
        # (the reference keeps four hand vectors whatever the player count; Kyoku keeps one per score like MjaiReplay)
        k.hands = k.hands[: len(k.scores)] if k.scores else k.hands
        k.end_scores = list(k.scores)
        k.actions = [a for a in (cls._parse_raw_action(x) for x in raw) if a is not None]
        k.wliqi = [False] * max(len(k.scores), 0)
        for a in k.actions:
            if a["name"] == "DiscardTile" and a["is_wliqi"] and a["seat"] < len(k.wliqi):
                k.wliqi[a["seat"]] = True
        k._pending_hule = []
        return k

    @classmethod
    def _parse_raw_action(cls, x):                # mjsoul_replay.rs:563-687
        name, d = x.get("name"), x.get("data", {}) or {}
        tiles = lambda v: [mjsoul_tile_to_mjai(t) for t in v]   # noqa: E731
        if name == "NewRound":
            return {"name": "Other"}
        if name == "DiscardTile":
            return {"name": "DiscardTile", "seat": d["seat"], "tile": mjsoul_tile_to_mjai(d["tile"]), "is_liqi": bool(d.get("is_liqi", False)),
                    "is_wliqi": bool(d.get("is_wliqi", False)), "doras": tiles(d["doras"]) if d.get("doras") else None}
        if name == "DealTile":
            doras = tiles(d["doras"]) if d.get("doras") else None
            if doras is None and d.get("dora_marker") is not None:
                doras = [mjsoul_tile_to_mjai(d["dora_marker"])]
            return {"name": "DealTile", "seat": d["seat"], "tile": mjsoul_tile_to_mjai(d["tile"]), "doras": doras,
                    "left_tile_count": d.get("left_tile_count")}
        if name == "ChiPengGang":
            return {"name": "ChiPengGang", "seat": d["seat"], "meld_type": cls._MELD.get(d["type"], "Chi"), "tiles": tiles(d["tiles"]),
                    "froms": list(d["froms"])}
        if name == "AnGangAddGang":
            return {"name": "AnGangAddGang", "seat": d["seat"], "meld_type": "Ankan" if d["type"] == 3 else "Kakan",
                    "tiles": [mjsoul_tile_to_mjai(d["tiles"])]}
        if name == "Hule":
            hs = []
            for h in d["hules"]:
                li = h.get("ura_dora_indicators") if h.get("ura_dora_indicators") is not None else h.get("li_doras")
                hs.append({"seat": h["seat"], "hu_tile": abi.mjai_to_tid(mjsoul_tile_to_mjai(h["hu_tile"]), True), "zimo": bool(h["zimo"]),
                           "count": h["count"], "fu": h["fu"], "fans": [f["id"] for f in h["fans"] if f.get("val", 0) > 0],
                           "li_doras": None if li is None else [abi.mjai_to_tid(mjsoul_tile_to_mjai(t), True) for t in li],
                           "yiman": bool(h["yiman"]), "point_rong": h["point_rong"], "point_zimo_qin": h["point_zimo_qin"],
                           "point_zimo_xian": h["point_zimo_xian"]})
            return {"name": "Hule", "hules": hs}
        if name == "dora":
            return {"name": "Dora", "dora_marker": mjsoul_tile_to_mjai(d["dora_marker"])}
        if name == "NoTile":
            return {"name": "NoTile"}
        if name == "BaBei":
            return {"name": "BaBei", "seat": d["seat"], "moqie": bool(d.get("moqie", False))}
        if name == "LiuJu":
            return {"name": "LiuJu", "lj_type": d.get("type", 0), "seat": d.get("seat", 0), "tiles": tiles(d.get("tiles", []))}
        return {"name": "Other"}

    def num_rounds(self):
        return len(self.rounds)

    def take_kyokus(self):
        return iter(self.rounds)

    def to_mjai(self, reveal_listed_doras=True):
        """The record as one MJAI event list (start_game, the rounds' Kyoku.to_mjai_events, end_game): the input of ReplayBatch,
        which turns it into (observation, mask, action) samples on the GPU - this framework's form of LogKyoku.steps for
        Mahjong Soul records (replay/mod.rs:1094-1290 + KyokuStepIterator)."""
        out = [{"type": "start_game"}]
        for k in self.rounds:
            out += k.to_mjai_events(reveal_listed_doras)
        return out + [{"type": "end_game"}]

    def verify(self, evaluate=None):
        """mjsoul_replay.rs:255-436: evaluate every win of every round and compare with what the record expects - the yaku
        sets without the dora kinds (31, 32, 33), han (yakuman counts normalised) and fu.  `evaluate` maps a list of
        WinResultContexts to evaluated contexts (default: one rmj_eval_hands launch).  Returns (wins, mismatches)."""
        ctxs = [c for k in self.rounds for c in k.take_win_result_contexts()]
        ctxs = (evaluate or evaluate_win_contexts)(ctxs)
        ignored, yakuman_ids = {31, 32, 33}, set(range(35, 51))
        bad = 0
        for c in ctxs:
            sim = list(c.actual.yaku[: c.actual.n_yaku])
            exp = list(c.expected_yaku)
            exp_han = c.expected_han * 13 if (any(y in yakuman_ids for y in exp) and c.expected_han < 13) else c.expected_han
            mismatch = sorted(y for y in sim if y not in ignored) != sorted(y for y in exp if y not in ignored)
            if not mismatch:
                sim_ign, exp_ign = sum(y in ignored for y in sim), sum(y in ignored for y in exp)
                if exp_han < 13 and c.actual.han != exp_han - exp_ign + sim_ign:
                    mismatch = c.actual.han != exp_han
                elif (c.actual.han >= 13) != (exp_han >= 13):
                    mismatch = True
                if not mismatch and exp_han < 13 and c.actual.fu != c.expected_fu:
                    mismatch = True
            bad += mismatch
        return len(ctxs), bad
