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
    """MjaiReplay.from_jsonl (replay/mjai_replay.rs): one JSON event per line, optionally gzip-compressed."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as f:
        return [json.loads(line) for line in f if line.strip()]


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
