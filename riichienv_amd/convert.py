"""Tile-name helpers under the reference's names (src/riichienv/convert.py): 136-ids <-> "mpsz" names ("1z", "5p", "0p" = red
five) <-> MJAI names ("E", "5p", "5pr"), list versions that hand out distinct copies, and paishan_to_wall (a Mahjong Soul wall
string -> 136 ids).  Host-side string code, no device work.  Copy 0 of a five is the red one (16 / 52 / 88), so the canonical id of
a plain five is copy 1; every other name maps to copy 0 of its type.  Checked against outputs of the reference module
(tests/golden/convert_vectors.json, scripts/gen_convert_vectors.py) and its tests (tests/test_convert.py, tests/env/test_paishan.py)."""
from __future__ import annotations

_RED = {16: "m", 52: "p", 88: "s"}
_HONORS = "ESWNPFC"


def _check_tid(tid):
    if not (0 <= tid < 136):
        raise ValueError(f"Invalid TID: {tid}")


def tid_to_mpsz(tid: int) -> str:
    _check_tid(tid)
    if tid in _RED:
        return "0" + _RED[tid]
    t34 = tid // 4
    return f"{t34 % 9 + 1}{'mpsz'[t34 // 9]}"


def tid_to_mjai(tid: int) -> str:
    _check_tid(tid)
    if tid in _RED:
        return "5" + _RED[tid] + "r"
    t34 = tid // 4
    return _HONORS[t34 - 27] if t34 >= 27 else f"{t34 % 9 + 1}{'mps'[t34 // 9]}"


def mpsz_to_tid(mpsz_str: str) -> int:
    """canonical id: "1z" -> 108, "0p" -> 52, "5p" -> 53"""
    if not mpsz_str:
        raise ValueError("Empty string")
    suit, num_str = mpsz_str[-1], mpsz_str[:-1]
    if suit not in "mpsz" or len(suit) != 1:
        raise ValueError(f"Invalid suit: {suit}")
    try:
        num = int(num_str)
    except ValueError as e:
        raise ValueError(f"Invalid number: {num_str}") from e
    if suit == "z":
        if not (1 <= num <= 7):
            raise ValueError(f"Invalid honor number: {num}")
        return 108 + (num - 1) * 4
    base = 36 * "mps".index(suit)
    if num == 0:
        return base + 16
    if not (1 <= num <= 9):
        raise ValueError(f"Invalid number: {num}")
    return base + 17 if num == 5 else base + (num - 1) * 4


def mjai_to_tid(mjai_str: str) -> int:
    if mjai_str in _HONORS and len(mjai_str) == 1:
        return 108 + _HONORS.index(mjai_str) * 4
    if mjai_str.endswith("r"):
        core = mjai_str[:-1]
        if core not in ("5m", "5p", "5s"):
            raise ValueError(f"Invalid red spec: {mjai_str}")
        return mpsz_to_tid("0" + core[1:])
    return mpsz_to_tid(mjai_str)


def mpsz_to_mjai(mpsz_str: str) -> str:
    return tid_to_mjai(mpsz_to_tid(mpsz_str))


def mjai_to_mpsz(mjai_str: str) -> str:
    return tid_to_mpsz(mjai_to_tid(mjai_str))


def tid_to_mpsz_list(tid_list):
    return [tid_to_mpsz(t) for t in tid_list]


def tid_to_mjai_list(tid_list):
    return [tid_to_mjai(t) for t in tid_list]


def _distinct(bases):
    """the k-th occurrence of a canonical id becomes id + k (["1m", "1m"] -> [0, 1]; "0m" and "5m" count separately)"""
    seen, out = {}, []
    for b in bases:
        k = seen.get(b, 0)
        out.append(b + k)
        seen[b] = k + 1
    return out


def mpsz_to_tid_list(mpsz_list):
    return _distinct(mpsz_to_tid(s) for s in mpsz_list)


def mjai_to_tid_list(mjai_list):
    return _distinct(mjai_to_tid(s) for s in mjai_list)


def mpsz_to_mjai_list(mpsz_list):
    return [mpsz_to_mjai(s) for s in mpsz_list]


def mjai_to_mpsz_list(mjai_list):
    return [mjai_to_mpsz(s) for s in mjai_list]


def paishan_to_wall(paishan_str: str):
    """"1m2m3p..." (two characters per tile) -> distinct 136-ids in the order of the string"""
    if len(paishan_str) % 2:
        raise ValueError(f"Invalid paishan string length: {len(paishan_str)}")
    return _distinct(mpsz_to_tid(paishan_str[i: i + 2]) for i in range(0, len(paishan_str), 2))
