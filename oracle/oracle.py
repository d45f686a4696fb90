"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (the CPU restatement of riichienv-core).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from riichienv_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "liboracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, os.environ.get("RMJ_ORACLE_LIB", "liboracle.so"))   # (liboracle_asan.so: scripts/run_sanitizers.sh)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_game_new.restype = C.c_void_p
        L.orc_game_new.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_uint32]
        L.orc_game_free.argtypes = [C.c_void_p]
        L.orc_game_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_game_step.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_game_legal.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_game_mask.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_game_waits.restype = C.c_uint64
        L.orc_game_waits.argtypes = [C.c_void_p, C.c_int]
        L.orc_game_status.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_game_step_count.restype = C.c_uint64
        L.orc_game_step_count.argtypes = [C.c_void_p]
        L.orc_game_peek.argtypes = [C.c_void_p, C.POINTER(abi.StateView)]
        L.orc_game_poke.argtypes = [C.c_void_p, C.POINTER(abi.StateView)]
        L.orc_game_log_len.restype = C.c_uint32
        L.orc_game_log_len.argtypes = [C.c_void_p, C.c_int]
        L.orc_game_log_get.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_char_p, C.c_uint32]
        L.orc_game_random_actions.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_game_greedy_actions.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
        L.orc_eval_hands.argtypes = [C.POINTER(abi.HandCase), C.c_uint32, C.POINTER(abi.HandResult)]
        L.orc_agari_counts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_is_tenpai_free.argtypes = [C.c_void_p]
        L.orc_calculate_score.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_void_p]
        L.orc_find_divisions.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_action_encode.argtypes = [C.c_uint64]
        L.orc_action_encode_3p.argtypes = [C.c_uint64]
        L.orc_game_encode.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_game_encode_extended.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for f in (L.orc_game_encode_kawa_overview, L.orc_game_encode_yaku_possibility, L.orc_game_encode_furiten_ron):
            f.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_shanten.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_effective_tiles.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_best_ukeire.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_game_apply_event.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_tid_to_mjai.argtypes = [C.c_uint8, C.c_char_p]
        L.orc_game_win_results.argtypes = [C.c_void_p, C.POINTER(abi.WinResult)]
        L.orc_bench_rollout.restype = C.c_uint64
        L.orc_bench_rollout.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32,
                                        C.c_int, C.POINTER(C.c_double)]
        L.orc_game_wall_meta.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_chacha_block.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_seed_from_u64.argtypes = [C.c_uint64, C.c_void_p]
        L.orc_stdrng_words.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_sha256.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p]
        L.orc_random_range_u32.restype = C.c_uint32
        L.orc_random_range_u32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_reference_wall.restype = C.c_uint32
        L.orc_reference_wall.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_char_p, C.c_char_p]
        _LIB = L
    return _LIB


# ---- ref_rng.hpp (the restated rand / chacha20 / sha2 algorithms behind state/wall.rs:36-56)
def chacha_block(key_words, counter=0, stream=0, rounds=12):
    k = np.ascontiguousarray(key_words, dtype=np.uint32)
    out = np.zeros(16, np.uint32)
    lib().orc_chacha_block(k.ctypes.data, counter, stream, rounds, out.ctypes.data)
    return out


def seed_from_u64(s: int) -> bytes:
    out = np.zeros(32, np.uint8)
    lib().orc_seed_from_u64(s, out.ctypes.data)
    return out.tobytes()


def stdrng_words(seed32: bytes, n: int):
    sd = np.frombuffer(seed32, np.uint8).copy()
    out = np.zeros(n, np.uint32)
    lib().orc_stdrng_words(sd.ctypes.data, n, out.ctypes.data)
    return out


def sha256(msg: bytes) -> bytes:
    out = np.zeros(32, np.uint8)
    lib().orc_sha256(msg, len(msg), out.ctypes.data)
    return out.tobytes()


def random_range_u32(seed32: bytes, skip_words: int, bound: int):
    sd = np.frombuffer(seed32, np.uint8).copy()
    used = C.c_uint32()
    v = lib().orc_random_range_u32(sd.ctypes.data, skip_words, bound, C.byref(used))
    return v, used.value


def reference_wall(hand_seed: int, sanma=False):
    """(w before the reversal, salt, digest, u32 words drawn) of state/wall.rs:36-56 for StdRng::seed_from_u64(hand_seed)"""
    w = np.zeros(108 if sanma else 136, np.uint8)
    salt, dg = C.create_string_buffer(17), C.create_string_buffer(65)
    n = lib().orc_reference_wall(hand_seed, int(sanma), w.ctypes.data, salt, dg)
    return w, salt.value.decode(), dg.value.decode(), n


def eval_hands(cases):
    arr = (abi.HandCase * len(cases))(*cases)
    out = (abi.HandResult * len(cases))()
    lib().orc_eval_hands(arr, len(cases), out)
    return list(out)


def agari_counts(counts: np.ndarray):
    counts = np.ascontiguousarray(counts, dtype=np.uint8)
    n = counts.shape[0]
    ag = np.zeros(n, np.uint8)
    tp = np.zeros(n, np.uint8)
    w = np.zeros(n, np.uint64)
    lib().orc_agari_counts(counts.ctypes.data, n, ag.ctypes.data, tp.ctypes.data, w.ctypes.data)
    return ag, tp, w


def calculate_score(han, fu, is_oya, is_tsumo, honba, np_):
    a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (han, fu, is_oya, is_tsumo)]
    hb = np.ascontiguousarray(honba, dtype=np.uint32)
    npl = np.ascontiguousarray(np_, dtype=np.uint8)
    n = len(a[0])
    out = np.zeros((n, 4), np.uint32)
    lib().orc_calculate_score(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, hb.ctypes.data,
                              npl.ctypes.data, n, out.ctypes.data)
    return out


def shanten(counts, sanma=False):
    counts = np.ascontiguousarray(counts, dtype=np.uint8)
    out = np.zeros(counts.shape[0], np.int8)
    lib().orc_shanten(counts.ctypes.data, counts.shape[0], int(sanma), out.ctypes.data)
    return out


def effective_tiles(counts, sanma=False):
    """calculate_effective_tiles(_3p)_with_discard on [n][34] type histograms (0xFFFFFFFF for a 3n hand)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint8)
    out = np.zeros(counts.shape[0], np.uint32)
    lib().orc_effective_tiles(counts.ctypes.data, counts.shape[0], int(sanma), out.ctypes.data)
    return out


def best_ukeire(counts, visible, sanma=False):
    """calculate_best_ukeire(_3p) on [n][34] hand / visible type histograms."""
    counts = np.ascontiguousarray(counts, dtype=np.uint8)
    visible = np.ascontiguousarray(visible, dtype=np.uint8)
    out = np.zeros(counts.shape[0], np.uint32)
    lib().orc_best_ukeire(counts.ctypes.data, visible.ctypes.data, counts.shape[0], int(sanma), out.ctypes.data)
    return out


def tid_to_mjai(t: int) -> str:
    b = C.create_string_buffer(8)
    lib().orc_tid_to_mjai(t, b)
    return b.value.decode()


class Game:
    """One oracle game (reference: GameState + the env.rs reset/step binding)."""

    def __init__(self, game_mode=0, seed=None, rule_bits=abi.RULE_TENHOU, skip_log=False, round_wind=0):
        self.L = lib()
        self.sanma = game_mode >= 3
        self.h = self.L.orc_game_new(game_mode, int(skip_log), seed or 0, 0 if seed is None else 1, round_wind, rule_bits)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_game_free(self.h)
            self.h = None

    def reset(self, wall=None, oya=-1, round_wind=-1, scores=None, honba=-1, kyotaku=-1):
        w = None
        if wall is not None:
            w = (C.c_uint8 * 136)(*wall)
        s = None
        if scores is not None:
            s = (C.c_int32 * 4)(*scores)
        self.L.orc_game_reset(self.h, w, oya, round_wind, s, honba, kyotaku)

    def step(self, actions):
        """actions: dict seat -> packed action, or a length-4 sequence."""
        arr = (C.c_uint64 * 4)(*([abi.NO_ACTION] * 4))
        if isinstance(actions, dict):
            for k, v in actions.items():
                arr[k] = v
        else:
            for k in range(4):
                arr[k] = int(actions[k])
        self.L.orc_game_step(self.h, arr)

    def legal(self, pid):
        out = (C.c_uint64 * abi.MAX_LEGAL)()
        n = self.L.orc_game_legal(self.h, pid, out)
        return [out[i] for i in range(min(n, abi.MAX_LEGAL))]

    def mask(self, pid):
        m = np.zeros(82, np.uint8)
        self.L.orc_game_mask(self.h, pid, m.ctypes.data)
        return m

    def waits(self, pid):
        return self.L.orc_game_waits(self.h, pid)

    def wall_meta(self):
        """(salt, wall_digest) of the current wall: state/wall.rs:15-16"""
        salt, dg = C.create_string_buffer(17), C.create_string_buffer(65)
        self.L.orc_game_wall_meta(self.h, salt, dg)
        return salt.value.decode(), dg.value.decode()

    def status(self):
        a, p, d = C.c_uint8(), C.c_uint8(), C.c_uint8()
        self.L.orc_game_status(self.h, C.byref(a), C.byref(p), C.byref(d))
        return a.value, p.value, d.value

    @property
    def step_count(self):
        return self.L.orc_game_step_count(self.h)

    def peek(self) -> abi.StateView:
        v = abi.StateView()
        self.L.orc_game_peek(self.h, C.byref(v))
        return v

    def poke(self, v: abi.StateView):
        self.L.orc_game_poke(self.h, C.byref(v))

    def log(self, seat=-1):
        n = self.L.orc_game_log_len(self.h, seat)
        buf = C.create_string_buffer(4096)
        out = []
        for i in range(n):
            self.L.orc_game_log_get(self.h, seat, i, buf, 4096)
            out.append(buf.value.decode())
        return out

    def encode(self, pid, sanma=False):
        out = np.zeros((74, 27 if sanma else 34), np.float32)
        self.L.orc_game_encode(self.h, pid, out.ctypes.data)
        return out

    def encode_extended(self, pid):
        out = np.zeros((215, 27 if self.sanma else 34), np.float32)
        self.L.orc_game_encode_extended(self.h, pid, out.ctypes.data)
        return out

    def encode_kawa_overview(self):
        np_ = 3 if self.sanma else 4
        out = np.zeros((np_, 7, 27 if self.sanma else 34), np.float32)
        self.L.orc_game_encode_kawa_overview(self.h, out.ctypes.data)
        return out

    def encode_yaku_possibility(self):
        out = np.zeros((3 if self.sanma else 4, 21, 2), np.float32)
        self.L.orc_game_encode_yaku_possibility(self.h, out.ctypes.data)
        return out

    def encode_furiten_ron_possibility(self):
        out = np.zeros((3 if self.sanma else 4, 21), np.float32)
        self.L.orc_game_encode_furiten_ron(self.h, out.ctypes.data)
        return out

    def win_results(self):
        arr = (abi.WinResult * 4)()
        m = self.L.orc_game_win_results(self.h, arr)
        out = {}
        for p in range(4):
            if (m >> p) & 1:
                w = arr[p]
                out[p] = dict(is_win=bool(w.is_win), yakuman=bool(w.yakuman), has_win_shape=bool(w.has_win_shape),
                              yaku=list(w.yaku[: w.n_yaku]), han=w.han, fu=w.fu, ron_agari=w.ron_agari,
                              tsumo_agari_oya=w.tsumo_agari_oya, tsumo_agari_ko=w.tsumo_agari_ko,
                              pao_payer=None if w.pao_payer < 0 else int(w.pao_payer))
        return out

    def apply_event(self, ev, replay=False):
        """apply_mjai_event (state/event_handler.rs, state_3p/event_handler.rs): MJAI dict or binary records.  replay=True: plus
        the missed-Ron bookkeeping of KyokuStepIterator (replay/mod.rs:129-177)."""
        recs = abi.event_records_from_mjai(ev, 3 if self.sanma else 4) if isinstance(ev, dict) else ev
        if replay:
            recs[0].pad |= 1
        self.L.orc_game_apply_event(self.h, C.byref(recs), abi.EVENT_SLOTS)

    def random_actions(self, policy_seed, global_game):
        arr = (C.c_uint64 * 4)()
        self.L.orc_game_random_actions(self.h, policy_seed, global_game, arr)
        return [arr[i] for i in range(4)]

    def greedy_actions(self, policy_seed, global_game, call_rate_256=64):
        """the greedy policy of rmj_step_greedy restated on the oracle's own lists and shanten"""
        arr = (C.c_uint64 * 4)()
        self.L.orc_game_greedy_actions(self.h, policy_seed, global_game, call_rate_256, arr)
        return [arr[i] for i in range(4)]


def bench_rollout(game_mode, rule_bits, skip_log, n_games, base_seed, policy_seed, steps_per_game, threads):
    secs = C.c_double()
    steps = lib().orc_bench_rollout(game_mode, rule_bits, int(skip_log), n_games, base_seed, policy_seed,
                                    steps_per_game, threads, C.byref(secs))
    return steps, secs.value
