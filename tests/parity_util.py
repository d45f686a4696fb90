"""Helpers shared by the GPU parity tests: compare the HIP path with the oracle."""
import ctypes as C

import numpy as np

from riichienv_amd import abi


def view_bytes(v: abi.StateView) -> bytes:
    return bytes(C.string_at(C.byref(v), C.sizeof(v)))


def _struct_to_dict(s):
    out = {}
    for name, _ in s._fields_:
        val = getattr(s, name)
        if isinstance(val, C.Array):
            if len(val) and isinstance(val[0], C.Structure):
                val = [_struct_to_dict(x) for x in val]
            else:
                val = list(val)
        elif isinstance(val, C.Structure):
            val = _struct_to_dict(val)
        out[name] = val
    return out


def normalize_view(v: abi.StateView) -> dict:
    """Canonical dict of the semantically meaningful part of a state view (unused tail slots zeroed)."""
    d = _struct_to_dict(v)
    d["wall"] = d["wall"][: d["wall_len"]]
    d["dora"] = d["dora"][: d["n_dora"]]
    for p in d["players"]:
        p["hand"] = p["hand"][: p["hand_len"]]
        p["melds"] = p["melds"][: p["n_melds"]]
        for m in p["melds"]:
            m["tiles"] = m["tiles"][: m["n_tiles"]]
        p["discards"] = p["discards"][: p["n_discards"]]
        p["forbidden"] = p["forbidden"][: p["n_forbidden"]]
    if d["pending_kan_pid"] < 0:
        d["pending_kan_action"] = 0
    if d["last_discard_pid"] < 0:
        d["last_discard_tile"] = -1
    return d


def diff_dict(a, b, path=""):
    out = []
    if isinstance(a, dict):
        for k in a:
            out += diff_dict(a[k], b[k], path + "." + k)
    elif isinstance(a, list):
        if len(a) != len(b):
            out.append(f"{path}: len {len(a)} != {len(b)} ({a} vs {b})")
        else:
            for i, (x, y) in enumerate(zip(a, b)):
                out += diff_dict(x, y, f"{path}[{i}]")
    elif a != b:
        out.append(f"{path}: {a} != {b}")
    return out


def fmt_action(a):
    t, tile, cons = abi.unpack_action(int(a))
    return f"{abi.ACTION_NAMES[t] if t < 12 else t}({tile},{cons})"
