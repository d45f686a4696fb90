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
            self._pending_hule.append({"seat": a, "zimo": a == ev.get("target"), "hu_tile": ev.get("pai")})
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

    def grp_features(self):
        """LogKyoku.grp_features (replay/mod.rs:1502-1522)"""
        return dict(chang=self.chang, ju=self.ju, ben=self.ben, liqibang=self.liqibang, scores=list(self.scores),
                    end_scores=list(self.end_scores), wliqi=list(self.wliqi),
                    delta_scores=[e - s for s, e in zip(self.scores, self.end_scores)] if len(self.scores) == len(self.end_scores) else [])


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
