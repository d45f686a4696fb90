"""Host-side MJAI helpers of the reference's binding layer (pure protocol logic, no device work):
tid_to_mjai (parser.rs:301-334) and Observation.select_action_from_mjai (observation/mjai_select.rs:88-194), which maps
a bot's MJAI reply onto one of the legal actions published by the step / apply_events kernels."""
from __future__ import annotations

import json

from . import abi

_HONORS = ["E", "S", "W", "N", "P", "F", "C"]


def tid_to_mjai(tid: int) -> str:
    """parser.rs:301-334"""
    if tid in (16, 52, 88):
        return {16: "5mr", 52: "5pr", 88: "5sr"}[tid]
    kind = tid // 36
    if kind < 3:
        return f"{(tid % 36) // 4 + 1}{'mps'[kind]}"
    num = (tid - 108) // 4 + 1
    return _HONORS[num - 1] if 1 <= num <= 7 else f"{num}z"


def _parse(msg):
    """parse_mjai_message (mjai_select.rs:19-70): JSON string or dict -> (type, pai, tsumogiri, consumed)."""
    if isinstance(msg, str):
        try:
            v = json.loads(msg)
        except ValueError:
            return None
        if not isinstance(v, dict) or not isinstance(v.get("type"), str):
            return None
        pai = v.get("pai") if isinstance(v.get("pai"), str) else ""
        tg = v.get("tsumogiri") if isinstance(v.get("tsumogiri"), bool) else None
        cons = [x for x in v["consumed"] if isinstance(x, str)] if isinstance(v.get("consumed"), list) else None
        return v["type"], pai, tg, cons
    if isinstance(msg, dict):
        ty = msg.get("type") if isinstance(msg.get("type"), str) else ""
        pai = msg.get("pai", msg.get("tile"))
        pai = pai if isinstance(pai, str) else ""
        tg = msg.get("tsumogiri") if isinstance(msg.get("tsumogiri"), bool) else None
        cons = msg.get("consumed")
        cons = list(cons) if isinstance(cons, (list, tuple)) and all(isinstance(x, str) for x in cons) else None
        return ty, pai, tg, cons
    return None


def select_action_from_mjai(legal_actions, mjai_data, drawn_tile=None, three_player=False):
    """mjai_select.rs:88-194 on packed actions (abi.pack_action): the first legal action matching the MJAI message, or
    None.  `legal_actions` is a seat's ordered list as returned by VecRiichiEnv.legal()."""
    parsed = _parse(mjai_data)
    if parsed is None:
        return None
    atype, tile_str, tsumogiri, consumed = parsed
    acts = [(int(a),) + abi.unpack_action(int(a)) for a in legal_actions]   # (packed, type, tile|None, consume)
    if atype == "hora":
        return next((a[0] for a in acts if a[1] in (abi.TSUMO, abi.RON)), None)
    if atype == "none":
        return next((a[0] for a in acts if a[1] == abi.PASS), None)
    table = {"dahai": abi.DISCARD, "pon": abi.PON, "kakan": abi.KAKAN, "daiminkan": abi.DAIMINKAN, "ankan": abi.ANKAN,
             "reach": abi.RIICHI, "ryukyoku": abi.KYUSHU}
    if not three_player:
        table["chi"] = abi.CHI
    else:
        table["kita"] = abi.KITA
    tt = table.get(atype)
    if tt is None:
        return None
    if tt == abi.DISCARD:
        cands = [a for a in acts if a[1] == abi.DISCARD and (tile_str == "" or (a[2] is not None and tid_to_mjai(a[2]) == tile_str))]
        if not cands:
            return None
        if tsumogiri is not None and drawn_tile is not None:
            for a in cands:
                if (a[2] == drawn_tile) == tsumogiri:
                    return a[0]
        return cands[0][0]
    for a in acts:
        if a[1] != tt:
            continue
        if consumed is not None:
            if sorted(tid_to_mjai(t) for t in a[3]) != sorted(consumed) or len(a[3]) != len(consumed):
                continue
            if tile_str and tt in (abi.CHI, abi.PON, abi.DAIMINKAN, abi.KAKAN):
                if a[2] is None or tid_to_mjai(a[2]) != tile_str:
                    continue
            return a[0]
        if tile_str:
            if a[2] is not None and tid_to_mjai(a[2]) == tile_str:
                return a[0]
            continue
        return a[0]
    return None
