"""ctypes mirror of include/riichi_mi355x.h (POD structs + packed-action helpers).

Pure data-layout definitions: no compute.  Shared by the product binding
(riichienv_amd.vecenv) and by the test-only oracle binding (oracle/oracle.py).
"""
from __future__ import annotations

import ctypes as C

NP = 4
MAX_LEGAL = 64
ACTION_SPACE_4P = 82
ACTION_SPACE_3P = 60
MAX_DISCARDS = 32
NO_ACTION = 0xFFFFFFFFFFFFFFFF
TILE_NONE = 0xFF

# ActionType (reference: riichienv-core/src/action.rs:55-68)
DISCARD, CHI, PON, DAIMINKAN, RON, RIICHI, TSUMO, PASS, ANKAN, KAKAN, KYUSHU, KITA = range(12)
ACTION_NAMES = ["DISCARD", "CHI", "PON", "DAIMINKAN", "RON", "RIICHI", "TSUMO", "PASS", "ANKAN", "KAKAN",
                "KYUSHU_KYUHAI", "KITA"]
WAIT_ACT, WAIT_RESPONSE = 0, 1
MELD_CHI, MELD_PON, MELD_DAIMINKAN, MELD_ANKAN, MELD_KAKAN = range(5)
MELD_NAMES = {"chi": 0, "pon": 1, "daiminkan": 2, "ankan": 3, "kakan": 4}

RULE_TENHOU = 64 | 128
RULE_MJSOUL = 1 | 2 | 4 | 8 | 16 | 32 | 128
RULE_REFERENCE_RNG = 256  # not a GameRule field: seed -> wall through the reference's StdRng / shuffle / salt / digest (include/riichi_mi355x.h)


def pack_action(atype: int, tile: int | None = None, consume=()) -> int:
    cons = sorted(consume)
    v = atype & 0xFF
    v |= (TILE_NONE if tile is None else tile) << 8
    v |= len(cons) << 16
    for i, c in enumerate(cons[:4]):
        v |= c << (24 + 8 * i)
    return v


def unpack_action(v: int):
    atype = v & 0xFF
    tile = (v >> 8) & 0xFF
    n = (v >> 16) & 0xFF
    cons = [(v >> (24 + 8 * i)) & 0xFF for i in range(min(n, 4))]
    return atype, (None if tile == TILE_NONE else tile), cons


class MeldView(C.Structure):
    _fields_ = [("meld_type", C.c_uint8), ("n_tiles", C.c_uint8), ("tiles", C.c_uint8 * 4), ("opened", C.c_uint8),
                ("from_who", C.c_int8), ("called_tile", C.c_int16)]


class PlayerView(C.Structure):
    _fields_ = [("hand_len", C.c_uint8), ("hand", C.c_uint8 * 14), ("n_melds", C.c_uint8), ("melds", MeldView * 4),
                ("n_discards", C.c_uint8), ("discards", C.c_uint8 * MAX_DISCARDS),
                ("discard_from_hand_bits", C.c_uint32), ("discard_is_riichi_bits", C.c_uint32),
                ("riichi_declaration_index", C.c_int8), ("score", C.c_int32), ("score_delta", C.c_int32),
                ("riichi_declared", C.c_uint8), ("riichi_stage", C.c_uint8), ("double_riichi_declared", C.c_uint8),
                ("missed_agari_riichi", C.c_uint8), ("missed_agari_doujun", C.c_uint8),
                ("nagashi_eligible", C.c_uint8), ("ippatsu_cycle", C.c_uint8),
                ("pao_daisangen", C.c_int8), ("pao_daisuushi", C.c_int8),
                ("n_forbidden", C.c_uint8), ("forbidden", C.c_uint8 * 2),
                ("riichi_sutehai", C.c_int16), ("last_tedashi", C.c_int16),
                ("n_kita", C.c_uint8), ("kita", C.c_uint8 * 4)]


class StateView(C.Structure):
    _fields_ = [("wall_len", C.c_uint8), ("wall", C.c_uint8 * 136), ("n_dora", C.c_uint8), ("dora", C.c_uint8 * 5),
                ("rinshan_draw_count", C.c_uint8), ("pending_kan_dora_count", C.c_uint8),
                ("drawable_count", C.c_uint8), ("wall_seed", C.c_uint64), ("hand_index", C.c_uint64),
                ("players", PlayerView * NP),
                ("current_player", C.c_uint8), ("is_done", C.c_uint8), ("needs_tsumo", C.c_uint8),
                ("phase", C.c_uint8), ("active_mask", C.c_uint8),
                ("turn_count", C.c_uint32), ("riichi_sticks", C.c_uint32),
                ("last_discard_pid", C.c_int16), ("last_discard_tile", C.c_int16), ("pending_kan_pid", C.c_int16),
                ("pending_kan_action", C.c_uint64),
                ("oya", C.c_uint8), ("honba", C.c_uint8), ("kyoku_idx", C.c_uint8), ("round_wind", C.c_uint8),
                ("is_rinshan_flag", C.c_uint8), ("is_first_turn", C.c_uint8),
                ("riichi_pending_acceptance", C.c_int16), ("drawn_tile", C.c_int16), ("last_error_pid", C.c_int16)]


class Event(C.Structure):
    _fields_ = [("type", C.c_uint8), ("actor", C.c_uint8), ("target", C.c_uint8), ("tile", C.c_uint8),
                ("consumed", C.c_uint8 * 4), ("deltas", C.c_int32 * 4), ("flags", C.c_uint8), ("n_ura", C.c_uint8),
                ("ura", C.c_uint8 * 5), ("pad", C.c_uint8)]


# RMJ_EV_* (include/riichi_mi355x.h)
EV_NONE, EV_START_GAME, EV_START_KYOKU, EV_TSUMO, EV_DAHAI, EV_REACH, EV_REACH_ACCEPTED, EV_CHI, EV_PON, EV_DAIMINKAN, \
    EV_ANKAN, EV_KAKAN, EV_DORA, EV_HORA, EV_RYUKYOKU, EV_END_KYOKU, EV_END_GAME, EV_KITA, EV_TEHAI = range(19)
EVENT_SLOTS = 3  # records per game and call of rmj_apply_events (start_kyoku = START_KYOKU + 2 x TEHAI)
_HONORS = ["E", "S", "W", "N", "P", "F", "C"]


def mjai_to_tid(s: str, masked_ok: bool = False) -> int:
    """parser.rs:336-385 mjai_to_tid: one id per tile name (copy 0; plain 5 = copy 1, red 5 = copy 0).  The reference
    maps an unparsable string (e.g. the masked "?") to tile 0 (parse_mjai_tile, event_handler.rs:8-10); with
    masked_ok the same happens here (bot-side streams: the state of the masked seats is then garbage, as in the
    reference, and only the observing seat's outputs are meaningful), otherwise it raises."""
    if s in _HONORS:
        return 108 + _HONORS.index(s) * 4
    if s in ("5mr", "5pr", "5sr"):
        return {"5mr": 16, "5pr": 52, "5sr": 88}[s]
    if len(s) >= 2 and s[0].isdigit() and s[1] in "mpsz":
        num, suit = int(s[0]), s[1]
        if suit == "z":
            if 1 <= num <= 7:
                return 108 + (num - 1) * 4
        else:
            si = "mps".index(suit)
            if num == 0:
                return si * 36 + 16
            if 1 <= num <= 9:
                base = si * 36 + (num - 1) * 4
                return base + 1 if num == 5 else base
    if masked_ok:
        return 0
    raise ValueError(f"cannot map MJAI tile {s!r} (pass masked_ok=True to ingest masked streams like the reference)")


def event_records_from_mjai(ev: dict, num_players: int = 4, masked_ok: bool = False):
    """MJAI event dict (replay/mjai_replay.rs MjaiEvent) -> up to EVENT_SLOTS binary records for rmj_apply_events /
    the oracle.  Unknown event types map to a NONE record (MjaiEvent::Other: no state change)."""
    recs = (Event * EVENT_SLOTS)()
    ty = ev.get("type")
    e = recs[0]
    actor = int(ev.get("actor", 0) or 0)
    simple = {"start_game": EV_START_GAME, "reach": EV_REACH, "reach_accepted": EV_REACH_ACCEPTED, "hora": EV_HORA,
              "ryukyoku": EV_RYUKYOKU, "end_kyoku": EV_END_KYOKU, "end_game": EV_END_GAME, "kita": EV_KITA}
    if ty in simple:
        e.type, e.actor = simple[ty], actor
    elif ty == "start_kyoku":
        e.type = EV_START_KYOKU
        e.actor = int(ev["oya"])
        e.target = int(ev["kyoku"])
        e.tile = mjai_to_tid(ev["dora_marker"], masked_ok)
        kyotaku = int(ev.get("kyoutaku", ev.get("kyotaku", 0)))
        e.consumed[0] = "ESWN".index(ev["bakaze"]) if ev["bakaze"] in "ESWN" else 0
        e.consumed[1] = int(ev["honba"])
        e.consumed[2], e.consumed[3] = kyotaku & 0xFF, (kyotaku >> 8) & 0xFF
        for i, sc in enumerate(ev["scores"][:4]):
            e.deltas[i] = int(sc)
        tehais = ev["tehais"]
        for half in range(2):
            t = recs[1 + half]
            t.type, t.actor = EV_TEHAI, half
            payload = []
            for q in range(2):
                seat = 2 * half + q
                hand = [mjai_to_tid(x, masked_ok) for x in tehais[seat]] if seat < min(num_players, len(tehais)) else [0] * 13
                if len(hand) != 13:
                    raise ValueError("start_kyoku: every tehai must hold 13 tiles")
                payload += hand
            C.memmove(C.addressof(t) + 4, bytes(payload), 26)
    elif ty in ("tsumo", "dahai", "kakan"):
        e.type = {"tsumo": EV_TSUMO, "dahai": EV_DAHAI, "kakan": EV_KAKAN}[ty]
        e.actor, e.tile = actor, mjai_to_tid(ev["pai"], masked_ok)
        if ty == "dahai":
            e.flags = 1 if ev.get("tsumogiri") else 0
    elif ty in ("pon", "chi", "daiminkan", "kan", "ankan"):
        e.type = {"pon": EV_PON, "chi": EV_CHI, "daiminkan": EV_DAIMINKAN, "kan": EV_DAIMINKAN, "ankan": EV_ANKAN}[ty]
        e.actor = actor
        e.target = int(ev.get("target", 0) or 0)
        if ty != "ankan":
            e.tile = mjai_to_tid(ev["pai"], masked_ok)
        cons = [mjai_to_tid(x, masked_ok) for x in ev["consumed"]][:4]
        for i, c in enumerate(cons):
            e.consumed[i] = c
        e.flags = (len(cons) << 4) & 0xFF
    elif ty == "dora":
        e.type, e.tile = EV_DORA, mjai_to_tid(ev["dora_marker"], masked_ok)
    else:
        e.type = EV_NONE
    return recs


class HandCase(C.Structure):
    _fields_ = [("n_tiles", C.c_uint8), ("tiles", C.c_uint8 * 14), ("n_melds", C.c_uint8), ("melds", MeldView * 4),
                ("win_tile", C.c_uint8), ("n_dora", C.c_uint8), ("dora", C.c_uint8 * 5), ("n_ura", C.c_uint8),
                ("ura", C.c_uint8 * 5),
                ("tsumo", C.c_uint8), ("riichi", C.c_uint8), ("double_riichi", C.c_uint8), ("ippatsu", C.c_uint8),
                ("haitei", C.c_uint8), ("houtei", C.c_uint8), ("rinshan", C.c_uint8), ("chankan", C.c_uint8),
                ("tsumo_first_turn", C.c_uint8), ("player_wind", C.c_uint8), ("round_wind", C.c_uint8),
                ("kita_count", C.c_uint8), ("is_sanma", C.c_uint8), ("honba", C.c_uint32)]


class HandResult(C.Structure):
    _fields_ = [("is_win", C.c_uint8), ("yakuman", C.c_uint8), ("has_win_shape", C.c_uint8), ("n_yaku", C.c_uint8),
                ("yaku", C.c_uint8 * 20), ("han", C.c_uint32), ("fu", C.c_uint32), ("ron_agari", C.c_uint32),
                ("tsumo_agari_oya", C.c_uint32), ("tsumo_agari_ko", C.c_uint32), ("waits", C.c_uint64),
                ("is_tenpai", C.c_uint8), ("is_agari", C.c_uint8), ("pad", C.c_uint8 * 6)]


class WinResult(C.Structure):
    _fields_ = [("is_win", C.c_uint8), ("yakuman", C.c_uint8), ("has_win_shape", C.c_uint8), ("n_yaku", C.c_uint8),
                ("yaku", C.c_uint8 * 20), ("han", C.c_uint32), ("fu", C.c_uint32), ("ron_agari", C.c_uint32),
                ("tsumo_agari_oya", C.c_uint32), ("tsumo_agari_ko", C.c_uint32), ("pao_payer", C.c_int8), ("pad", C.c_uint8 * 3)]


class EventViews(C.Structure):   # RmjEventViews
    _fields_ = [("n_games", C.c_uint32), ("ring", C.c_uint32), ("events", C.c_void_p), ("ev_count", C.c_void_p),
                ("ev_count_stride", C.c_uint32), ("reserved", C.c_uint32), ("lost", C.c_void_p), ("ev_base", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("n_games", C.c_uint32), ("game_mode", C.c_uint8), ("skip_mjai_logging", C.c_uint8),
                ("round_wind", C.c_uint8), ("reserved0", C.c_uint8), ("rule_bits", C.c_uint32),
                ("device", C.c_int32), ("base_seed", C.c_uint64), ("game_offset", C.c_uint64),
                ("seeds", C.POINTER(C.c_uint64)), ("event_ring", C.c_uint32), ("reserved1", C.c_uint32)]


class SeqBuffers(C.Structure):
    _fields_ = [("sparse", C.c_void_p), ("n_sparse", C.c_void_p), ("numeric", C.c_void_p), ("progression", C.c_void_p),
                ("n_progression", C.c_void_p), ("candidates", C.c_void_p), ("n_candidates", C.c_void_p)]


SEQ_SPARSE, SEQ_PROG, SEQ_CAND, SEQ_DELTA_PROG = 25, 256, 64, 64


class BenchResult(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("step_kernel_ms", C.c_double), ("env_steps", C.c_uint64),
                ("launches", C.c_uint32), ("launches_in_flight", C.c_uint32), ("full_path_steps", C.c_uint64),
                ("queued", C.c_uint32), ("reserved", C.c_uint32)]


def hand_case_from_fixture(case: dict) -> HandCase:
    """Build a HandCase from one entry of the reference's agari_*.json fixtures
    (riichienv-core/tests/agari_correctness.rs:29-84)."""
    hc = HandCase()
    tiles = case["tiles_136"]
    hc.n_tiles = len(tiles)
    for i, t in enumerate(tiles):
        hc.tiles[i] = t
    hc.n_melds = len(case["melds"])
    for i, m in enumerate(case["melds"]):
        mv = hc.melds[i]
        mv.meld_type = MELD_NAMES[m["meld_type"]]
        mv.n_tiles = len(m["tiles"])
        for j, t in enumerate(m["tiles"]):
            mv.tiles[j] = t
        mv.opened = 1 if m["opened"] else 0
        mv.from_who = m["from_who"]
        mv.called_tile = -1
    hc.win_tile = case["win_tile_136"]
    hc.n_dora = len(case["dora_indicators"])
    for i, t in enumerate(case["dora_indicators"]):
        hc.dora[i] = t
    hc.n_ura = len(case["ura_indicators"])
    for i, t in enumerate(case["ura_indicators"]):
        hc.ura[i] = t
    c = case["conditions"]
    for k in ("tsumo", "riichi", "double_riichi", "ippatsu", "haitei", "houtei", "rinshan", "chankan",
              "tsumo_first_turn"):
        setattr(hc, k, 1 if c[k] else 0)
    hc.player_wind = c["player_wind"]
    hc.round_wind = c["round_wind"]
    hc.honba = c["honba"]
    hc.kita_count = c.get("kita_count", 0)
    hc.is_sanma = 1 if c.get("is_sanma", False) else 0
    return hc
