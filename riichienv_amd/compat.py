"""Drop-in, reference-named scalar API over the batched HIP path (n = 1 `VecRiichiEnv`).

Mirrors the user-facing surface of smly/RiichiEnv for the step hot path so that the README loop and the
reference's own tests read the same (names, argument meaning, error behaviour):

    RiichiEnv        riichienv-python/src/env.rs:74-872
    Action / ActionType / Phase     riichienv-core/src/action.rs:29-149, 350-545
    Observation      riichienv-core/src/observation/python.rs:76-133, 457-806
    GameRule         riichienv-core/src/rule.rs:10-57
    Meld / MeldType  riichienv-core/src/types.rs:54-190
    RandomAgent      src/riichienv/agents/random_agent.py:6-15

Everything computes through the C-ABI library (HIP); there is no CPU fallback.
"""
from __future__ import annotations

import enum
import json
import random

import numpy as np

from . import abi, mjai, vecenv


class ActionType(enum.IntEnum):  # action.rs:55-68 (pyclass rename_all = SCREAMING_SNAKE_CASE)
    DISCARD = 0
    CHI = 1
    PON = 2
    DAIMINKAN = 3
    RON = 4
    RIICHI = 5
    TSUMO = 6
    PASS = 7
    ANKAN = 8
    KAKAN = 9
    KYUSHU_KYUHAI = 10
    KITA = 11


# PascalCase aliases of the reference's Python layer (src/riichienv/action.py:5-17, deprecated there, still exported)
for _n in list(ActionType):
    setattr(ActionType, "".join(w.capitalize() for w in _n.name.split("_")), _n)


class GameType(enum.IntEnum):  # src/riichienv/game_mode.py
    YON_IKKYOKU = 0
    YON_TONPUSEN = 1
    YON_HANCHAN = 2
    SAN_IKKYOKU = 3
    SAN_TONPUSEN = 4
    SAN_HANCHAN = 5


class Phase(enum.IntEnum):  # action.rs:29-33
    WaitAct = 0
    WaitResponse = 1


class MeldType(enum.IntEnum):  # types.rs:54-62
    Chi = 0
    Pon = 1
    Daiminkan = 2
    Ankan = 3
    Kakan = 4


_HONORS = ["E", "S", "W", "N", "P", "F", "C"]


def tid_to_mjai(tid: int) -> str:  # parser.rs:301-334
    if tid == 16:
        return "5mr"
    if tid == 52:
        return "5pr"
    if tid == 88:
        return "5sr"
    if tid < 108:
        return f"{(tid % 36) // 4 + 1}{'mps'[tid // 36]}"
    return _HONORS[(tid - 108) // 4]


class Action:
    """action.rs:76-105: consume_tiles are sorted on construction."""

    __slots__ = ("action_type", "tile", "consume_tiles", "actor")

    def __init__(self, type=ActionType.PASS, tile=None, consume_tiles=(), actor=None):
        self.action_type = ActionType(int(type))
        self.tile = None if tile is None else int(tile)
        self.consume_tiles = sorted(int(t) for t in consume_tiles)
        self.actor = actor

    @staticmethod
    def _from_packed(v: int, actor=None) -> "Action":
        t, tile, cons = abi.unpack_action(int(v))
        return Action(ActionType(t), tile, cons, actor)

    def _pack(self) -> int:
        return abi.pack_action(int(self.action_type), self.tile, self.consume_tiles)

    def encode(self) -> int:  # action.rs:158-227
        t = self.action_type
        if t == ActionType.DISCARD:
            if self.tile is None:
                raise ValueError("Discard action requires a tile")
            return self.tile // 4
        if t == ActionType.RIICHI:
            return 37
        if t == ActionType.CHI:
            if self.tile is None:
                raise ValueError("Chi action requires a target tile")
            target = self.tile // 4
            ts = sorted({c // 4 for c in self.consume_tiles} | {target})
            if len(ts) != 3:
                raise ValueError(f"Invalid Chi tiles: target={self.tile}, consumed={self.consume_tiles}")
            return 38 if target == ts[0] else (39 if target == ts[1] else 40)
        if t == ActionType.PON:
            return 41
        if t == ActionType.DAIMINKAN:
            if self.tile is None:
                raise ValueError("Daiminkan action requires a tile")
            return 42 + self.tile // 4
        if t in (ActionType.ANKAN, ActionType.KAKAN):
            if not self.consume_tiles:
                raise ValueError("Ankan/Kakan action requires consumed tiles")
            return 42 + self.consume_tiles[0] // 4
        if t in (ActionType.RON, ActionType.TSUMO):
            return 79
        if t == ActionType.KYUSHU_KYUHAI:
            return 80
        if t == ActionType.PASS:
            return 81
        raise ValueError("Kita action is not valid in 4-player mode")

    def encode_3p(self) -> int:  # action.rs:262-346 (60-way space, compact tile index action.rs:14-22)
        def compact(tile):
            t34 = tile // 4
            c = 0 if t34 == 0 else (t34 - 7 if 8 <= t34 < 34 else None)
            if c is None:
                raise ValueError(f"Tile type {t34} (manzu 2-8) is not valid in 3P mode")
            return c

        t = self.action_type
        if t == ActionType.DISCARD:
            if self.tile is None:
                raise ValueError("Discard action requires a tile")
            return compact(self.tile)
        if t == ActionType.RIICHI:
            return 27
        if t == ActionType.CHI:
            raise ValueError("Chi is not allowed in 3P mode")
        if t == ActionType.PON:
            return 28
        if t == ActionType.DAIMINKAN:
            if self.tile is None:
                raise ValueError("Daiminkan action requires a tile")
            return 29 + compact(self.tile)
        if t in (ActionType.ANKAN, ActionType.KAKAN):
            if not self.consume_tiles:
                raise ValueError("Ankan/Kakan action requires consumed tiles")
            return 29 + compact(self.consume_tiles[0])
        if t in (ActionType.RON, ActionType.TSUMO):
            return 56
        if t == ActionType.KYUSHU_KYUHAI:
            return 57
        if t == ActionType.PASS:
            return 58
        return 59  # Kita

    def to_mjai(self) -> str:  # action.rs:107-149 (serde_json BTreeMap -> alphabetical keys)
        names = {ActionType.DISCARD: "dahai", ActionType.CHI: "chi", ActionType.PON: "pon", ActionType.DAIMINKAN: "daiminkan",
                 ActionType.ANKAN: "ankan", ActionType.KAKAN: "kakan", ActionType.RIICHI: "reach", ActionType.TSUMO: "hora",
                 ActionType.RON: "hora", ActionType.KYUSHU_KYUHAI: "ryukyoku", ActionType.KITA: "kita", ActionType.PASS: "none"}
        d = {"type": names[self.action_type]}
        if self.actor is not None:
            d["actor"] = self.actor
        if self.tile is not None and self.action_type not in (ActionType.TSUMO, ActionType.RON, ActionType.RIICHI):
            d["pai"] = tid_to_mjai(self.tile)
        if self.consume_tiles:
            d["consumed"] = [tid_to_mjai(t) for t in self.consume_tiles]
        return json.dumps(dict(sorted(d.items())), separators=(",", ":"))

    def __eq__(self, o):
        return (isinstance(o, Action) and self.action_type == o.action_type and self.tile == o.tile
                and self.consume_tiles == o.consume_tiles and self.actor == o.actor)

    def __repr__(self):
        return (f"Action(action_type={self.action_type.name}, tile={self.tile}, consume_tiles={self.consume_tiles}, "
                f"actor={self.actor})")


class Meld:  # types.rs:98-190
    def __init__(self, meld_type, tiles, opened, from_who=-1, called_tile=None):
        self.meld_type = MeldType(int(meld_type))
        self.tiles = list(tiles)
        self.opened = bool(opened)
        self.from_who = from_who
        self.called_tile = called_tile

    def __repr__(self):
        return f"Meld({self.meld_type.name}, {self.tiles}, opened={self.opened}, from_who={self.from_who})"


class GameRule:  # rule.rs:10-57
    FIELDS = ["allows_ron_on_ankan_for_kokushi_musou", "is_kokushi_musou_13machi_double", "is_suuankou_tanki_double",
              "is_junsei_chuurenpoutou_double", "is_daisuushii_double", "yakuman_pao_is_liability_only", "sanchaho_is_draw",
              "kuikae_forbidden"]

    def __init__(self, allows_ron_on_ankan_for_kokushi_musou=False, is_kokushi_musou_13machi_double=False,
                 is_suuankou_tanki_double=False, is_junsei_chuurenpoutou_double=False, is_daisuushii_double=False,
                 yakuman_pao_is_liability_only=False, sanchaho_is_draw=False, kuikae_forbidden=True):
        for k, v in zip(self.FIELDS, (allows_ron_on_ankan_for_kokushi_musou, is_kokushi_musou_13machi_double,
                                      is_suuankou_tanki_double, is_junsei_chuurenpoutou_double, is_daisuushii_double,
                                      yakuman_pao_is_liability_only, sanchaho_is_draw, kuikae_forbidden)):
            setattr(self, k, bool(v))

    @staticmethod
    def default_tenhou():
        return GameRule(sanchaho_is_draw=True, kuikae_forbidden=True)

    @staticmethod
    def default_mjsoul():
        return GameRule(True, True, True, True, True, True, False, True)

    def bits(self) -> int:
        return sum((1 << i) for i, k in enumerate(self.FIELDS) if getattr(self, k))


class Observation:
    """Per-seat snapshot (observation/mod.rs:24-56; state/mod.rs:189-263)."""

    def __init__(self, player_id, view, legal, mask, waits, new_events, events, encoder, num_players=4, ext_encoder=None,
                 aux_encoder=None, seq_encoder=None):
        pid = player_id
        self.player_id = pid
        self.num_players = num_players            # 3: Observation3P (observation_3p/mod.rs)
        players = list(view.players)[:num_players]
        self.hands = [list(p.hand[: p.hand_len]) if i == pid else [] for i, p in enumerate(players)]
        self.hand = self.hands[pid]
        self.melds = [[Meld(m.meld_type, list(m.tiles[: m.n_tiles]), bool(m.opened), m.from_who,
                            None if m.called_tile < 0 else m.called_tile) for m in p.melds[: p.n_melds]] for p in players]
        self.discards = [list(p.discards[: p.n_discards]) for p in players]
        self.dora_indicators = list(view.dora[: view.n_dora])
        self.scores = [p.score for p in players]
        self.riichi_declared = [bool(p.riichi_declared) for p in players]
        self.honba = view.honba
        self.riichi_sticks = view.riichi_sticks
        self.round_wind = view.round_wind
        self.oya = view.oya
        self.kyoku_index = view.kyoku_idx
        self.waits = [t for t in range(34) if (waits >> t) & 1]
        self.is_tenpai = bool(self.waits)
        self.riichi_sutehais = [None if p.riichi_sutehai < 0 else p.riichi_sutehai for p in players]
        self.last_tedashis = [None if p.last_tedashi < 0 else p.last_tedashi for p in players]
        # the reference hands the DISCARDER'S SEAT over here (state/mod.rs:252 destructures (pid, tile) as (tile, _pid))
        self.last_discard = None if view.last_discard_pid < 0 else view.last_discard_pid
        self.drawn_tile = None if view.drawn_tile < 0 else view.drawn_tile
        self._legal_actions = legal
        self._mask = mask
        self._new_events = new_events
        self._log = events  # the seat's whole log (not a reference field)
        self._encoder = encoder
        self._ext_encoder = ext_encoder
        self._aux_encoder = aux_encoder
        self._seq_encoder = seq_encoder

    @property
    def action_space_size(self):  # observation/python.rs:113-116: a #[getter] (an attribute, not a method: found by transcribing tests/env/test_sanma.py:222-226 in round 6)
        return 60 if self.num_players == 3 else 82

    def select_action_from_mjai(self, mjai_data):  # observation/mjai_select.rs:88-194
        packed = mjai.select_action_from_mjai([a._pack() for a in self._legal_actions], mjai_data, self.drawn_tile,
                                              self.num_players == 3)
        return None if packed is None else Action._from_packed(packed, self.player_id)

    def encode_extended(self) -> bytes:  # observation/python.rs:1271-1296 -> 215 x 34 (3P: 215 x 27) f32
        return self._ext_encoder(self.player_id).tobytes()

    # ---- the separately exposed feature blocks of the reference = slices of the extended tensor
    #      (observation/python.rs:183-236, 808-875, 927-1268 restate the blocks of observation/encode.rs:293-585)
    def _ext(self):
        w = 27 if self.num_players == 3 else 34
        return np.frombuffer(self.encode_extended(), dtype=np.float32).reshape(215, w)

    def encode_discard_history_decay(self, decay_rate=None) -> bytes:  # (np, W)
        if decay_rate is not None and abs(decay_rate - 0.2) > 1e-12:
            raise NotImplementedError("the device table holds exp(-0.2 * age); other decay rates are not built")
        return self._ext()[74:74 + self.num_players].tobytes()

    def encode_shanten_efficiency(self) -> bytes:  # (np, 4): shanten, effective tiles, best ukeire, turn
        e = self._ext()
        return np.ascontiguousarray(e[78:78 + 4 * self.num_players, 0].reshape(self.num_players, 4)).tobytes()

    def encode_ankan_overview(self) -> bytes:  # (np, W)
        return self._ext()[94:94 + self.num_players].tobytes()

    def encode_fuuro_overview(self) -> bytes:  # (np, 4, 5, W)
        return self._ext()[98:98 + 20 * self.num_players].tobytes()

    def encode_action_availability(self) -> bytes:  # (11,)
        return np.ascontiguousarray(self._ext()[178:189, 0]).tobytes()

    def encode_discard_candidates(self) -> bytes:  # (5,)
        return np.ascontiguousarray(self._ext()[189:194, 0]).tobytes()

    def encode_pass_context(self) -> bytes:  # (3,)
        return np.ascontiguousarray(self._ext()[194:197, 0]).tobytes()

    def encode_last_tedashis(self) -> bytes:  # (np - 1, 3)
        return np.ascontiguousarray(self._ext()[197:197 + 3 * (self.num_players - 1), 0]).tobytes()

    def encode_riichi_sutehais(self) -> bytes:  # (np - 1, 3)
        return np.ascontiguousarray(self._ext()[206:206 + 3 * (self.num_players - 1), 0]).tobytes()

    # ---- blocks outside the extended tensor: one device launch per block (rmj_encode_aux)
    def encode_kawa_overview(self) -> bytes:  # observation/python.rs:881-925 -> (np, 7, W)
        return self._aux_encoder("encode_kawa_overview").tobytes()

    def encode_yaku_possibility(self) -> bytes:  # observation/python.rs:327-455 -> (np, 21, 2)
        return self._aux_encoder("encode_yaku_possibility").tobytes()

    def encode_furiten_ron_possibility(self) -> bytes:  # observation/python.rs:251-293 -> (np, 21)
        return self._aux_encoder("encode_furiten_ron_possibility").tobytes()

    # ---- sequence (transformer) features, observation/sequence_features.rs: variable-length arrays like the reference's,
    #      computed by rmj_encode_seq over the events of the current round (header: the reference uses the Observation's
    #      own `events`, i.e. the log since the seat's previous observation)
    def _seq(self, game_style=1):
        if self.num_players == 3:
            raise AttributeError("sequence features exist for 4-player observations only (observation/sequence_features.rs)")
        return self._seq_encoder(int(game_style))

    def encode_seq_sparse(self, game_style=1) -> bytes:  # python.rs:1302-1317 -> u16[5..25]
        o = self._seq(game_style)
        return o["sparse"][0, self.player_id, : o["n_sparse"][0, self.player_id]].tobytes()

    def encode_seq_numeric(self) -> bytes:  # python.rs:1319-1333 -> f32[12]
        return self._seq()["numeric"][0, self.player_id].tobytes()

    def encode_seq_progression(self) -> bytes:  # python.rs:1335-1350 -> u16[n][5]
        o = self._seq()
        n = int(o["n_progression"][0])
        if n == 0xFFFF:
            raise RuntimeError("the event ring no longer holds the round's start_kyoku: create the env with a larger event ring")
        return o["progression"][0, :n].tobytes()

    def encode_seq_candidates(self) -> bytes:  # python.rs:1352-1362 -> u16[n][4]
        o = self._seq()
        return o["candidates"][0, self.player_id, : o["n_candidates"][0, self.player_id]].tobytes()

    def legal_actions(self):  # observation/python.rs:93-96
        return list(self._legal_actions)

    def mask(self) -> bytes:  # observation/python.rs:98-111
        return bytes(self._mask)

    def new_events(self):
        return list(self._new_events)

    @property
    def events(self):  # observation/python.rs:81-91: the Observation's events ARE the new events (observation/mod.rs:138-140), as dicts
        return [json.loads(s) for s in self._new_events]

    def find_action(self, action_id):  # observation/mod.rs:117-129
        for a in self._legal_actions:
            try:
                if (a.encode_3p() if self.num_players == 3 else a.encode()) == action_id:
                    return a
            except ValueError:
                pass
        return None

    def encode(self) -> bytes:  # observation/python.rs:457-806 -> 74*34 f32, channel-major
        return self._encoder(self.player_id).tobytes()


class RiichiEnv:
    """Scalar environment with the reference's method names, backed by one game on the GPU."""

    def __init__(self, game_mode=None, skip_mjai_logging=False, seed=None, round_wind=None, rule=None, device=0, reference_rng=True):
        """reference_rng (not a parameter of the reference; default on since round 6): RiichiEnv(seed=s) deals the walls the REFERENCE deals for s -
        StdRng::seed_from_u64(splitmix64(s + hand_index)), rand's shuffle, salt, SHA-256 digest (state/wall.rs:36-67) - and `salt` / `wall_digest`
        read like WallState's fields.  (The chain is restated from the published algorithms of rand 0.9 / rand_core 0.9 / chacha20 / sha2; its
        primitives are pinned on published vectors, the seed expansion and the index draws on nothing outside this repository until
        tests/golden/ref_rng_vectors.json exists - INTEGRATION.md has the 30-line Rust program that writes it.)  False: the build's own shuffle."""
        self._mode = vecenv._mode_id(game_mode)
        self._rule = rule or GameRule.default_tenhou()
        s = random.getrandbits(63) if seed is None else int(seed)
        self._v = vecenv.VecRiichiEnv(1, game_mode=self._mode, seeds=np.array([s], np.uint64), rule_bits=self._rule.bits(),
                                      skip_mjai_logging=skip_mjai_logging, round_wind=round_wind or 0, device=device,
                                      event_ring=8192, reference_rng=bool(reference_rng))
        self._seed, self._skip_log = (None if seed is None else int(seed)), bool(skip_mjai_logging)
        self._cursor = [0, 0, 0, 0]  # player_event_counts (state/mod.rs:211-218)
        self._applied = None         # host-side logs of apply_event / observe_event (see apply_event)
        self._np = 3 if self._mode >= 3 else 4

    @property
    def num_players(self):
        return self._np

    @property
    def salt(self):
        """WallState.salt (state/wall.rs:15, 48-50): 16 hex digits of the current wall's shuffle; "" before a seeded shuffle, after a start_kyoku
        event, or with reference_rng=False"""
        return self._v.wall_digest(0)[0]

    @property
    def wall_digest(self):
        """WallState.wall_digest (state/wall.rs:16, 51-55): SHA-256(salt || wall) as 64 hex digits; stale after a load_wall like the reference's"""
        return self._v.wall_digest(0)[1]

    def clone(self):
        """RiichiEnv.clone / __copy__ / __deepcopy__ (env.rs:358-372): an independent environment in the same state"""
        import copy

        out = object.__new__(RiichiEnv)
        out.__dict__.update({k: v for k, v in self.__dict__.items() if k != "_v"})
        out._v = self._v.clone()
        out._cursor = list(self._cursor)
        out._applied = copy.deepcopy(self._applied)
        if hasattr(self, "_applied_seat"):
            out._applied_seat = copy.deepcopy(self._applied_seat)
        return out

    def __copy__(self):
        return self.clone()

    def __deepcopy__(self, memo):
        return self.clone()

    # ---- MJAI event ingestion (env.rs:880-948; full-information streams, see rmj_apply_events) -------------
    def apply_event(self, event):
        self._v.apply_events([event], masked_ok=True)  # "?" -> tile 0 like parse_mjai_tile (event_handler.rs:8-10)
        # apply_and_log (env.rs:52-72): the caller's event is pushed into mjai_log and, masked per seat like
        # _push_mjai_event (state/mod.rs:2094-2148), into the seats' logs; start_game restarts them.  The device does not
        # append applied events to its ring, so the binding keeps these logs on the host from the first apply_event on.
        if event.get("type") == "start_game" or self._applied is None:
            fresh = event.get("type") == "start_game"
            self._applied = [] if fresh else list(self._v.mjai_log(0))
            self._applied_seat = [[] if fresh else list(self._v.mjai_log(0, p)) for p in range(self._np)]
            if fresh:
                self._cursor = [0, 0, 0, 0]
        dumps = lambda e: json.dumps(e, sort_keys=True, separators=(",", ":"), ensure_ascii=False)  # serde_json::Value order
        self._applied.append(dumps(event))
        for p in range(self._np):
            ev = event
            if event.get("type") == "start_kyoku" and isinstance(event.get("tehais"), list):
                ev = dict(event, tehais=[h if i == p else ["?"] * len(h) for i, h in enumerate(event["tehais"])])
            elif event.get("type") == "tsumo" and event.get("actor") != p:
                ev = dict(event, pai="?")
            self._applied_seat[p].append(dumps(ev))

    def observe_event(self, event, player_id):
        self.apply_event(event)
        if event.get("type") in ("start_game", "start_kyoku", "reach_accepted", "dora", "hora", "ryukyoku", "end_kyoku", "end_game"):
            return None
        obs = self.get_observation(player_id)
        return obs if obs.legal_actions() else None

    # ---- core loop -----------------------------------------------------------------------------------
    def reset(self, oya=None, wall=None, round_wind=None, scores=None, honba=None, kyotaku=None, seed=None):
        if scores is not None and len(scores) != self._np:
            raise ValueError(f"scores length {len(scores)} does not match number of players {self._np}")  # env.rs:815-823
        self._v.reset(walls=None if wall is None else np.array(wall, np.uint8)[None], oya=None if oya is None else [oya],
                      round_wind=None if round_wind is None else [round_wind],
                      scores=None if scores is None else np.array(scores, np.int32)[None],
                      honba=None if honba is None else [honba], kyotaku=None if kyotaku is None else [kyotaku])
        self._cursor = [0, 0, 0, 0]
        self._applied = None
        return self.get_observations(self.active_players)

    def step(self, actions):
        a = np.full((1, 4), abi.NO_ACTION, np.uint64)
        for pid, act in dict(actions).items():
            a[0, int(pid)] = act._pack()
        self._v.step(a)
        if self._v.peek(0).last_error_pid >= 0:  # env.rs:865-869 (quirk Q9)
            return {}
        return self.get_observations(self.active_players)

    def get_observations(self, players=None):
        pids = list(range(self._np)) if players is None else list(players)
        view = self._v.peek(0)
        legal, cnt = self._v.legal()
        mask = self._v.mask()
        waits = self._v.waits()
        act = view.active_mask
        enc = {}

        def encoder(pid):
            if "a" not in enc:
                enc["a"] = self._v.encode()
            return enc["a"][0, pid]

        def ext_encoder(pid):
            if "x" not in enc:
                enc["x"] = self._v.encode_extended()
            return enc["x"][0, pid]

        def aux_encoder(name):
            if name not in enc:
                enc[name] = getattr(self._v, name)()
            return enc[name][0]

        def seq_encoder(game_style):
            if ("seq", game_style) not in enc:
                enc[("seq", game_style)] = self._v.encode_seq(game_style)
            return enc[("seq", game_style)]

        nmask = 60 if self._np == 3 else 82
        out = {}
        for pid in pids:
            active = bool((act >> pid) & 1) and not view.is_done and (
                (view.phase == Phase.WaitAct and view.current_player == pid) or view.phase == Phase.WaitResponse)
            la = [Action._from_packed(x, pid) for x in legal[0, pid, : cnt[0, pid]]] if active else []
            w = int(waits[0, pid]) if active else self._waits_of(view, pid)
            log = self._v.mjai_log(0, pid) if self._applied is None else self._applied_seat[pid]
            new = log[self._cursor[pid]:]
            self._cursor[pid] = len(log)
            out[pid] = Observation(pid, view, la, mask[0, pid][:nmask] if active else np.zeros(nmask, np.uint8), w, new, log,
                                   encoder, self._np, ext_encoder, aux_encoder, seq_encoder)
        return out

    def get_observation(self, player_id):
        return self.get_observations([player_id])[player_id]

    def _waits_of(self, view, pid):
        p = view.players[pid]
        if p.hand_len + 3 * p.n_melds != 13:
            return 0
        hc = abi.HandCase()
        hc.n_tiles = p.hand_len
        for i in range(p.hand_len):
            hc.tiles[i] = p.hand[i]
        hc.n_melds = p.n_melds
        for i in range(p.n_melds):
            hc.melds[i] = p.melds[i]
        return vecenv.eval_hands([hc])[0].waits

    def _get_legal_actions(self, pid):
        legal, cnt = self._v.legal()
        return [Action._from_packed(x, pid) for x in legal[0, pid, : cnt[0, pid]]]

    def done(self):
        return bool(self._v.done()[0])

    def scores(self):
        return [int(x) for x in self._v.scores()[0]][: self._np]

    def ranks(self):  # env.rs:673-689
        return [int(x) for x in self._v.ranks()[0]][: self._np]

    def points(self, rule_name="basic"):  # env.rs:691-727
        if self._np == 3:
            presets = {"basic": (1.0, 35000.0, [40.0, 0.0, -40.0])}
        else:
            presets = {"basic": (1.0, 25000.0, [50.0, 10.0, -10.0, -50.0]), "ouza-tyoujyo": (0.0, 25000.0, [100.0, 40.0, -40.0, -100.0]),
                       "ouza-normal": (0.0, 25000.0, [50.0, 20.0, -20.0, -50.0])}
        if rule_name not in presets:
            raise ValueError(f"Unknown preset rule{' for 3P' if self._np == 3 else ''}: {rule_name}")
        w, base, uma = presets[rule_name]
        return [(s - base) / 1000.0 * w + uma[r - 1] for s, r in zip(self.scores(), self.ranks())]

    @property
    def mjai_log(self):  # env.rs:729-739
        return [json.loads(s) for s in (self._v.mjai_log(0) if self._applied is None else self._applied)]

    # ---- state getters (env.rs:134-622) ----------------------------------------------------------------
    def _view(self):
        return self._v.peek(0)

    def _poke(self, fn):
        v = self._v.peek(0)
        fn(v)
        self._v.poke(0, v)

    @property
    def phase(self):
        return Phase(self._view().phase)

    @phase.setter
    def phase(self, ph):
        self._poke(lambda v: setattr(v, "phase", int(ph)))

    @property
    def current_player(self):
        return self._view().current_player

    @current_player.setter
    def current_player(self, p):
        self._poke(lambda v: setattr(v, "current_player", int(p)))

    @property
    def active_players(self):
        m = self._view().active_mask
        return [p for p in range(4) if (m >> p) & 1]

    @active_players.setter
    def active_players(self, ps):
        self._poke(lambda v: setattr(v, "active_mask", sum(1 << int(p) for p in ps)))

    @property
    def hands(self):
        v = self._view()
        return [list(p.hand[: p.hand_len]) for p in v.players]

    @hands.setter
    def hands(self, hs):
        def f(v):
            for p, h in enumerate(hs):
                v.players[p].hand_len = len(h)
                for i, t in enumerate(h):
                    v.players[p].hand[i] = t
        self._poke(f)

    @property
    def melds(self):
        v = self._view()
        return [[Meld(m.meld_type, list(m.tiles[: m.n_tiles]), bool(m.opened), m.from_who,
                      None if m.called_tile < 0 else m.called_tile) for m in p.melds[: p.n_melds]] for p in v.players]

    @melds.setter
    def melds(self, ms):
        def f(v):
            for p, lst in enumerate(ms):
                v.players[p].n_melds = len(lst)
                for i, m in enumerate(lst):
                    mv = v.players[p].melds[i]
                    mv.meld_type = int(m.meld_type)
                    mv.n_tiles = len(m.tiles)
                    for k, t in enumerate(sorted(m.tiles)):
                        mv.tiles[k] = t
                    mv.opened = int(m.opened)
                    mv.from_who = m.from_who
                    mv.called_tile = -1 if m.called_tile is None else m.called_tile
        self._poke(f)

    @property
    def discards(self):
        v = self._view()
        return [list(p.discards[: p.n_discards]) for p in v.players]

    @discards.setter
    def discards(self, ds):
        def f(v):
            for p, d in enumerate(ds):
                v.players[p].n_discards = len(d)
                for i, t in enumerate(d):
                    v.players[p].discards[i] = t
        self._poke(f)

    @property
    def drawn_tile(self):
        d = self._view().drawn_tile
        return None if d < 0 else d

    @drawn_tile.setter
    def drawn_tile(self, t):
        self._poke(lambda v: setattr(v, "drawn_tile", -1 if t is None else int(t)))

    @property
    def needs_tsumo(self):
        return bool(self._view().needs_tsumo)

    @needs_tsumo.setter
    def needs_tsumo(self, b):
        self._poke(lambda v: setattr(v, "needs_tsumo", int(bool(b))))

    @property
    def is_first_turn(self):
        return bool(self._view().is_first_turn)

    @is_first_turn.setter
    def is_first_turn(self, b):
        self._poke(lambda v: setattr(v, "is_first_turn", int(bool(b))))

    @property
    def riichi_declared(self):
        return [bool(p.riichi_declared) for p in self._view().players]

    @riichi_declared.setter
    def riichi_declared(self, bs):
        def f(v):
            for p, b in enumerate(bs):
                v.players[p].riichi_declared = int(bool(b))
        self._poke(f)

    @property
    def dora_indicators(self):
        v = self._view()
        return list(v.dora[: v.n_dora])

    @property
    def wall(self):
        v = self._view()
        return list(v.wall[: v.wall_len])

    @property
    def pao(self):
        out = []
        for p in self._view().players:
            d = {}
            if p.pao_daisangen >= 0:
                d[37] = p.pao_daisangen
            if p.pao_daisuushi >= 0:
                d[50] = p.pao_daisuushi
            out.append(d)
        return out

    # per-seat flags (env.rs:300-470): list getters, list setters
    def _seat_flag(name, cast=bool):   # noqa: N805
        def get(self):
            return [cast(getattr(p, name)) for p in list(self._view().players)[: self._np]]

        def put(self, values):
            def f(v):
                for p, x in enumerate(values):
                    setattr(v.players[p], name, int(x))
            self._poke(f)
        return property(get, put)

    riichi_stage = _seat_flag("riichi_stage")
    double_riichi_declared = _seat_flag("double_riichi_declared")
    missed_agari_riichi = _seat_flag("missed_agari_riichi")
    missed_agari_doujun = _seat_flag("missed_agari_doujun")
    nagashi_eligible = _seat_flag("nagashi_eligible")
    ippatsu_cycle = _seat_flag("ippatsu_cycle")
    score_deltas = _seat_flag("score_delta", int)
    del _seat_flag

    @property
    def forbidden_discards(self):
        return [list(p.forbidden[: p.n_forbidden]) for p in list(self._view().players)[: self._np]]

    @forbidden_discards.setter
    def forbidden_discards(self, lists):
        def f(v):
            for p, lst in enumerate(lists):
                v.players[p].n_forbidden = min(len(lst), 2)
                for i, t in enumerate(list(lst)[:2]):
                    v.players[p].forbidden[i] = t
        self._poke(f)

    @property
    def is_done(self):
        return bool(self._view().is_done)

    @property
    def is_rinshan_flag(self):
        return bool(self._view().is_rinshan_flag)

    @is_rinshan_flag.setter
    def is_rinshan_flag(self, b):
        self._poke(lambda v: setattr(v, "is_rinshan_flag", int(bool(b))))

    @property
    def riichi_pending_acceptance(self):
        x = self._view().riichi_pending_acceptance
        return None if x < 0 else int(x)

    @riichi_pending_acceptance.setter
    def riichi_pending_acceptance(self, x):
        self._poke(lambda v: setattr(v, "riichi_pending_acceptance", -1 if x is None else int(x)))

    @property
    def pending_kan_dora_count(self):
        return int(self._view().pending_kan_dora_count)

    @pending_kan_dora_count.setter
    def pending_kan_dora_count(self, n):
        self._poke(lambda v: setattr(v, "pending_kan_dora_count", int(n)))

    @property
    def pending_kan(self):
        """(seat, Action) of a kan waiting for the chankan answers, else None (state/mod.rs:61)"""
        v = self._view()
        return None if v.pending_kan_pid < 0 else (int(v.pending_kan_pid), Action._from_packed(v.pending_kan_action, int(v.pending_kan_pid)))

    @property
    def last_discard(self):
        v = self._view()
        return None if v.last_discard_pid < 0 else (int(v.last_discard_pid), int(v.last_discard_tile))

    @last_discard.setter
    def last_discard(self, x):
        def f(v):
            v.last_discard_pid, v.last_discard_tile = (-1, -1) if x is None else (int(x[0]), int(x[1]))
        self._poke(f)

    @property
    def current_claims(self):
        """{seat: legal claim actions} while the round waits for answers (state/mod.rs:60)"""
        v = self._view()
        if v.phase != Phase.WaitResponse:
            return {}
        return {p: self._get_legal_actions(p) for p in range(self._np) if (v.active_mask >> p) & 1}

    @property
    def agari_results(self):
        """{seat: WinResult fields} of the round that ended the game (env.rs:606-607)"""
        return self._v.win_results(0)

    last_agari_results = agari_results

    def game_mode(self):
        return self._mode

    game_type = property(lambda self: GameType(self._mode))
    seed = property(lambda self: self._seed)
    skip_mjai_logging = property(lambda self: self._skip_log)
    player_event_counts = property(lambda self: list(self._cursor[: self._np]))

    oya = property(lambda self: self._view().oya)
    honba = property(lambda self: self._view().honba)
    kyoku_idx = property(lambda self: self._view().kyoku_idx)
    round_wind = property(lambda self: self._view().round_wind)
    riichi_sticks = property(lambda self: self._view().riichi_sticks)
    turn_count = property(lambda self: self._view().turn_count)
    drawable_count = property(lambda self: self._view().drawable_count)
    rinshan_draw_count = property(lambda self: self._view().rinshan_draw_count)

    def set_scores(self, scores):  # env.rs:636-671
        def f(v):
            for p, s in enumerate(scores):
                v.players[p].score = int(s)
        self._poke(f)

    def set_state(self, oya=None, round_wind=None):
        def f(v):
            if oya is not None:
                v.oya = oya
                v.kyoku_idx = oya
            if round_wind is not None:
                v.round_wind = round_wind
        self._poke(f)


class RandomAgent:  # src/riichienv/agents/random_agent.py:6-15
    def __init__(self, seed=None):
        self._rng = random.Random(seed)

    def act(self, obs: Observation) -> Action:
        return self._rng.choice(obs.legal_actions())
