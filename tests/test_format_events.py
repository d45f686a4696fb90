"""The C batch formatter of binary MJAI records (rmj_format_events, riichienv_amd/csrc/rmj_host.h) against the single-event
formatter and against the reference's real hanchan log (tests/golden/126_204_0_mjai.jsonl = the reference's
tests/data file): tsumo / dahai / calls / dora / reach events must come back as the log's own JSON objects."""
import ctypes as C
import json
import os

import numpy as np

from riichienv_amd import abi, vecenv


def _records_of_log(golden_dir):
    evs = [json.loads(x) for x in open(os.path.join(golden_dir, "126_204_0_mjai.jsonl")) if x.strip()]
    recs, src = [], []
    for ev in evs:
        r = abi.event_records_from_mjai(ev)
        k = 3 if ev["type"] == "start_kyoku" else 1
        if r[0].type == abi.EV_NONE:
            continue
        for i in range(k):
            recs.append(bytes(r[i]))
        src.append(ev)
    return np.frombuffer(b"".join(recs), np.uint8).reshape(-1, 32).copy(), src


def _single(L, ev, seat):
    out, i, n = [], 0, len(ev)
    buf = C.create_string_buffer(4096)
    while i < n:
        used = L.rmj_format_event(C.cast(ev[i:].ctypes.data, C.POINTER(abi.Event)), n - i, seat, buf, 4096)
        assert used > 0
        out.append(buf.value.decode())
        i += used
    return out


def _batch(L, ev, offsets, seat):
    n = len(offsets) - 1
    toffs = np.zeros(n + 1, np.uint64)
    need = C.c_uint64()
    assert L.rmj_format_events(ev.ctypes.data, offsets.ctypes.data, n, seat, None, 0, toffs.ctypes.data, C.byref(need)) != 0   # size pass
    buf = np.zeros(int(need.value), np.uint8)
    assert L.rmj_format_events(ev.ctypes.data, offsets.ctypes.data, n, seat, buf.ctypes.data, int(need.value), toffs.ctypes.data, C.byref(need)) == 0
    raw = buf.tobytes()
    return [raw[int(toffs[g]): int(toffs[g + 1])].decode().split("\n")[:-1] for g in range(n)]


def test_batch_formatter_equals_single_formatter_and_the_log(golden_dir):
    L = vecenv.load_lib()
    ev, src = _records_of_log(golden_dir)
    # cut the stream into "games" at the start_kyoku records (a triple is never split), plus an empty game
    cuts = [0] + [i for i in range(1, len(ev)) if ev[i, 0] == abi.EV_START_KYOKU] + [len(ev), len(ev)]
    offsets = np.array(cuts, np.uint32)
    for seat in (-1, 0, 3):
        one = _single(L, ev, seat)
        many = _batch(L, ev, offsets, seat)
        assert [s for g in many for s in g] == one
        assert many[-1] == []
    full = _single(L, ev, -1)
    assert len(full) == len(src)
    for s, want in zip(full, src):
        got = json.loads(s)
        if want["type"] in ("tsumo", "dahai", "pon", "chi", "daiminkan", "kakan", "dora", "reach", "reach_accepted", "end_kyoku", "end_game"):
            assert got == {k: v for k, v in want.items() if k in got}, (got, want)
            assert s == json.dumps(got, sort_keys=True, separators=(",", ":"))     # alphabetical keys, no spaces (state/mod.rs:2094-2148)
        if want["type"] == "start_kyoku":
            assert got["tehais"] == want["tehais"] and got["scores"] == want["scores"] and got["oya"] == want["oya"]


def test_a_window_that_starts_inside_a_start_kyoku_triple():
    """a ring that was lapped can start with the tehai records of a lost start_kyoku: they are skipped, the rest is formatted"""
    L = vecenv.load_lib()
    r = abi.event_records_from_mjai({"type": "start_kyoku", "bakaze": "E", "dora_marker": "1m", "kyoku": 1, "honba": 0, "kyotaku": 0, "oya": 0,
                                     "scores": [25000] * 4, "tehais": [["1m"] * 13] * 4})
    t = abi.event_records_from_mjai({"type": "tsumo", "actor": 2, "pai": "5pr"})
    ev = np.frombuffer(bytes(r[1]) + bytes(r[2]) + bytes(t[0]) + bytes(r[0]), np.uint8).reshape(-1, 32).copy()   # ... and ends with a cut triple
    got = _batch(L, ev, np.array([0, 4], np.uint32), -1)
    assert got == [['{"actor":2,"pai":"5pr","type":"tsumo"}']]
