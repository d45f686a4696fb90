"""Batched Riichi Mahjong environment on MI355X — Python binding of the C-ABI
(include/riichi_mi355x.h, built to riichienv_amd/libriichi_mi355x.so).

`VecRiichiEnv` is the batched counterpart of the reference's `RiichiEnv`
(riichienv-python/src/env.rs:74-872): N independent games stepped in lock-step by HIP
kernels.  There is NO CPU fallback: importing works anywhere (so the symbol table can be
checked), but every compute call raises if the HIP library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RMJ_LIB_PATH") or os.path.join(_HERE, "libriichi_mi355x.so")   # (the variable: experiment builds)
_LIB = None

# every symbol include/riichi_mi355x.h declares
EXPORTS = [
    "rmj_version", "rmj_last_error", "rmj_device_count", "rmj_create", "rmj_destroy", "rmj_reset", "rmj_step",
    "rmj_step_device", "rmj_step_random", "rmj_random_actions", "rmj_get_status", "rmj_get_legal", "rmj_get_mask",
    "rmj_get_waits", "rmj_get_scores", "rmj_get_ranks", "rmj_get_step_counts", "rmj_total_steps", "rmj_peek_state",
    "rmj_poke_state", "rmj_get_event_counts", "rmj_get_events", "rmj_format_event", "rmj_eval_hands",
    "rmj_agari_counts", "rmj_calculate_score", "rmj_shanten", "rmj_effective_tiles", "rmj_best_ukeire", "rmj_apply_events", "rmj_device_views", "rmj_step_ids_device", "rmj_clone", "rmj_copy_games", "rmj_copy_games_device",
    "rmj_scores_device", "rmj_sync", "rmj_set_stream", "rmj_encode", "rmj_encode_device", "rmj_encode_extended", "rmj_encode_extended_device", "rmj_encode_aux", "rmj_encode_aux_device", "rmj_encode_seq", "rmj_encode_seq_device", "rmj_bench_rollout", "rmj_time_rollout", "rmj_step_greedy", "rmj_time_rollout_greedy", "rmj_time_rollout_encode", "rmj_step_ids_encode_device", "rmj_step_sample_encode_device", "rmj_set_encode_row_stride", "rmj_bench_hand_kernel", "rmj_points_device", "rmj_get_points", "rmj_get_legal_compact", "rmj_get_wall_digest", "rmj_get_wall_digests",
    "rmj_bench_rollout_validated", "rmj_bench_encode", "rmj_set_rollout_streams", "rmj_total_full_path", "rmj_bench_device_alloc", "rmj_bench_device_free", "rmj_bench_device_sync",
    "rmj_random_actions_device", "rmj_peek_outputs", "rmj_sample_ids_device",
    "rmj_encode_seq_delta", "rmj_encode_seq_delta_device", "rmj_step_random_encode",
    "rmj_encode_compact_device", "rmj_step_random_encode_compact", "rmj_bench_encode_compact",
    "rmj_get_win_results",
    "rmj_drain_events", "rmj_format_events", "rmj_drain_format", "rmj_event_views", "rmj_round_track_device", "rmj_round_track_reset", "rmj_get_events_lost", "rmj_get_log_positions",
]


class RmjError(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so.7 (+ HSA runtime); the system one under
    /opt/rocm has the same soname, so whichever is loaded first serves both, and torch fails with "No HIP GPUs are
    available" when it finds the system copy already in the process.  When torch is installed but not imported yet,
    load its copy first, so that a later `import torch` (TorchVecEnv, a trainer) works whatever the import order.
    RMJ_HIP_RUNTIME=system keeps the system runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("RMJ_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_lib():
    """Load the HIP library; raises (never falls back) if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RmjError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.rmj_version.restype = C.c_char_p
    L.rmj_last_error.restype = C.c_char_p
    vp = C.c_void_p
    L.rmj_create.argtypes = [C.POINTER(abi.Config), C.POINTER(vp)]
    L.rmj_destroy.argtypes = [vp]
    L.rmj_get_wall_digest.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_char_p]
    L.rmj_get_wall_digests.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp]
    L.rmj_reset.argtypes = [vp] * 8
    L.rmj_step.argtypes = [vp, vp]
    L.rmj_clone.argtypes = [vp, C.POINTER(vp)]
    L.rmj_copy_games.argtypes = [vp, vp, vp, vp, C.c_uint32]
    L.rmj_copy_games_device.argtypes = [vp, vp, vp, vp, C.c_uint32]
    L.rmj_step_device.argtypes = [vp, vp]
    L.rmj_step_random.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_int]
    L.rmj_random_actions.argtypes = [vp, C.c_uint64, vp]
    L.rmj_get_status.argtypes = [vp, vp, vp, vp]
    L.rmj_get_legal.argtypes = [vp, vp, vp]
    L.rmj_get_mask.argtypes = [vp, vp]
    L.rmj_get_waits.argtypes = [vp, vp]
    L.rmj_get_scores.argtypes = [vp, vp]
    L.rmj_get_ranks.argtypes = [vp, vp]
    L.rmj_get_step_counts.argtypes = [vp, vp]
    L.rmj_total_steps.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rmj_peek_state.argtypes = [vp, C.c_uint32, C.POINTER(abi.StateView)]
    L.rmj_poke_state.argtypes = [vp, C.c_uint32, C.POINTER(abi.StateView)]
    L.rmj_get_event_counts.argtypes = [vp, vp]
    L.rmj_get_events.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(abi.Event), C.POINTER(C.c_uint32)]
    L.rmj_format_event.argtypes = [C.POINTER(abi.Event), C.c_uint32, C.c_int, C.c_char_p, C.c_uint32]
    L.rmj_eval_hands.argtypes = [C.c_int, C.POINTER(abi.HandCase), C.c_uint32, C.POINTER(abi.HandResult)]
    L.rmj_agari_counts.argtypes = [C.c_int, vp, C.c_uint32, vp, vp, vp]
    L.rmj_calculate_score.argtypes = [C.c_int] + [vp] * 6 + [C.c_uint32, vp]
    L.rmj_encode.argtypes = [vp, C.c_int, vp]
    L.rmj_encode_device.argtypes = [vp, C.c_int, vp]
    L.rmj_encode_extended.argtypes = [vp, C.c_int, vp]
    L.rmj_encode_extended_device.argtypes = [vp, C.c_int, vp]
    L.rmj_set_stream.argtypes = [vp, vp, C.c_int]
    L.rmj_encode_aux.argtypes = [vp, C.c_int, vp]
    L.rmj_encode_aux_device.argtypes = [vp, C.c_int, vp]
    L.rmj_encode_seq.argtypes = [vp, C.c_int, C.POINTER(abi.SeqBuffers)]
    L.rmj_encode_seq_device.argtypes = [vp, C.c_int, C.POINTER(abi.SeqBuffers)]
    L.rmj_effective_tiles.argtypes = [C.c_int, vp, C.c_uint32, C.c_int, vp]
    L.rmj_best_ukeire.argtypes = [C.c_int, vp, vp, C.c_uint32, C.c_int, vp]
    L.rmj_apply_events.argtypes = [vp, vp]
    L.rmj_shanten.argtypes = [C.c_int, vp, C.c_uint32, C.c_int, vp]
    L.rmj_bench_rollout.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(abi.BenchResult)]
    L.rmj_bench_rollout_validated.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(abi.BenchResult)]
    L.rmj_time_rollout.argtypes = [vp, C.c_uint64, C.c_uint32, C.POINTER(abi.BenchResult)]
    L.rmj_step_greedy.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32]
    L.rmj_get_legal_compact.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.rmj_points_device.argtypes = [vp, C.c_int, vp]
    L.rmj_get_points.argtypes = [vp, C.c_int, vp]
    L.rmj_bench_hand_kernel.argtypes = [C.c_int, C.c_int, vp, vp, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
    L.rmj_set_encode_row_stride.argtypes = [vp, C.c_uint32]
    L.rmj_step_ids_encode_device.argtypes = [vp, vp, C.c_int, vp]
    L.rmj_step_sample_encode_device.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_int, vp, vp]
    L.rmj_time_rollout_encode.argtypes = [vp, C.c_uint64, C.c_uint32, vp, C.POINTER(abi.BenchResult)]
    L.rmj_time_rollout_greedy.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(abi.BenchResult)]
    L.rmj_bench_encode.argtypes = [vp, C.c_int, C.c_int, vp, C.c_uint32, C.POINTER(C.c_double)]
    L.rmj_set_rollout_streams.argtypes = [vp, C.c_int]
    L.rmj_total_full_path.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rmj_bench_device_alloc.argtypes = [C.c_int, C.c_uint64, C.POINTER(vp)]
    L.rmj_bench_device_free.argtypes = [C.c_int, vp]
    L.rmj_bench_device_sync.argtypes = [C.c_int]
    L.rmj_random_actions_device.argtypes = [vp, C.c_uint64, vp]
    L.rmj_peek_outputs.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.POINTER(C.c_uint32)]
    L.rmj_sample_ids_device.argtypes = [vp, vp, C.c_uint32, C.c_uint64, vp]
    L.rmj_step_random_encode.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_int, C.c_int, vp]
    L.rmj_encode_compact_device.argtypes = [vp, vp, vp, C.c_uint32, vp]
    L.rmj_step_random_encode_compact.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_int, vp, vp, C.c_uint32, vp]
    L.rmj_bench_encode_compact.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(C.c_double)]
    L.rmj_get_win_results.argtypes = [vp, C.c_uint32, C.POINTER(abi.WinResult), C.POINTER(C.c_uint8)]
    L.rmj_encode_seq_delta.argtypes = [vp, C.c_int, C.POINTER(abi.SeqBuffers)]          # same field layout as RmjSeqBuffers
    L.rmj_encode_seq_delta_device.argtypes = [vp, C.c_int, C.POINTER(abi.SeqBuffers)]
    L.rmj_drain_events.argtypes = [vp, vp, vp, C.c_uint32, vp, C.POINTER(C.c_uint32), C.c_uint32]
    L.rmj_get_log_positions.argtypes = [vp, vp, vp]
    L.rmj_format_events.argtypes = [vp, vp, C.c_uint32, C.c_int, vp, C.c_uint64, vp, C.POINTER(C.c_uint64)]
    L.rmj_drain_format.argtypes = [vp, vp, C.c_int, vp, C.c_uint64, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), vp, C.c_uint32]
    L.rmj_event_views.argtypes = [vp, C.POINTER(abi.EventViews)]
    L.rmj_get_events_lost.argtypes = [vp, vp]
    L.rmj_round_track_device.argtypes = [vp, vp, vp, vp, vp]
    L.rmj_round_track_reset.argtypes = [vp]
    _LIB = L
    return L


def _chk(rc):
    if rc != 0:
        raise RmjError(f"rmj error {rc}: {load_lib().rmj_last_error().decode()}")


GAME_MODES = {"4p-red-single": 0, "4p-red-east": 1, "4p-red-half": 2, "3p-red-single": 3, "3p-red-east": 4,
              "3p-red-half": 5}


def _mode_id(game_mode) -> int:
    # env.rs:93-100
    if game_mode is None:
        return 0
    if isinstance(game_mode, str):
        return GAME_MODES.get(game_mode, 0)
    return int(game_mode)


def _opt(arr, dtype, shape):
    if arr is None:
        return None, None
    a = np.ascontiguousarray(arr, dtype=dtype).reshape(shape)
    return a, a.ctypes.data


class VecRiichiEnv:
    """N independent games on one GPU (one shard of a batch sharded by game index)."""

    def __init__(self, n_games, game_mode=0, seed=0, seeds=None, rule_bits=abi.RULE_TENHOU, skip_mjai_logging=False,
                 round_wind=0, device=0, game_offset=0, event_ring=64, reference_rng=None):
        """seed -> wall.  `seeds=` (one episode seed per game, used as given like RiichiEnv(seed=...)) deals every wall through the
        REFERENCE's chain - StdRng::seed_from_u64, rand's shuffle, salt, SHA-256 digest (state/wall.rs:36-67; RMJ_RULE_REFERENCE_RNG) -
        so that a seed means the wall the reference deals for it; salt / wall_digest: wall_digest(g).  `seed=` (a base seed, the games
        decorrelated by their global index: throughput runs, sharded batches) keeps the build's own counter-based shuffle, which costs a
        round start nothing.  reference_rng=True / False overrides either default (False with `seeds=`: the opt-out for throughput)."""
        self.L = load_lib()
        self.n = int(n_games)
        if reference_rng is None:
            reference_rng = seeds is not None or bool(rule_bits & abi.RULE_REFERENCE_RNG)
        rule_bits = (rule_bits | abi.RULE_REFERENCE_RNG) if reference_rng else (rule_bits & ~abi.RULE_REFERENCE_RNG)
        self.reference_rng = bool(reference_rng)
        cfg = abi.Config()
        cfg.n_games = self.n
        cfg.game_mode = _mode_id(game_mode)
        cfg.skip_mjai_logging = int(bool(skip_mjai_logging))
        cfg.round_wind = round_wind
        cfg.rule_bits = rule_bits
        cfg.device = device
        cfg.base_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.game_offset = int(game_offset)
        self._seeds = None
        if seeds is not None:
            self._seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
            assert self._seeds.shape == (self.n,)
            cfg.seeds = self._seeds.ctypes.data_as(C.POINTER(C.c_uint64))
        cfg.event_ring = event_ring
        self.event_ring = max(64, 1 << max(0, int(event_ring) - 1).bit_length())   # the library rounds the ring up to a power of two, at least 64 records (rmj_create; event_ring = 0 gives 64)
        self.game_mode = cfg.game_mode
        self.game_offset = int(game_offset)
        self.h = C.c_void_p()
        _chk(self.L.rmj_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.rmj_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- RiichiEnv.reset (env.rs:799-851), batched
    def reset(self, select=None, walls=None, oya=None, round_wind=None, scores=None, honba=None, kyotaku=None):
        n = self.n
        seats = 3 if self.game_mode >= 3 else 4
        if scores is not None:
            sc = np.asarray(scores, dtype=np.int32).reshape(n, -1)
            if sc.shape[1] != seats:
                raise ValueError(f"scores length {sc.shape[1]} does not match number of players {seats}")  # env.rs:815-823
            if seats == 3:
                sc = np.concatenate([sc, np.zeros((n, 1), np.int32)], axis=1)
            scores = sc
        if walls is not None:
            wl = np.asarray(walls, dtype=np.uint8).reshape(n, -1)
            if wl.shape[1] == 108:  # 3P walls travel in the first 108 entries of the [n][136] rows
                wl = np.concatenate([wl, np.zeros((n, 28), np.uint8)], axis=1)
            walls = wl
        keep = []
        ptrs = []
        for arr, dt, shp in ((select, np.uint8, (n,)), (walls, np.uint8, (n, 136)), (oya, np.uint8, (n,)),
                             (round_wind, np.uint8, (n,)), (scores, np.int32, (n, 4)), (honba, np.uint8, (n,)),
                             (kyotaku, np.uint32, (n,))):
            a, p = _opt(arr, dt, shp)
            keep.append(a)
            ptrs.append(p)
        _chk(self.L.rmj_reset(self.h, *ptrs))
        if select is None and getattr(self, "_cursor", None) is not None:
            self._cursor = self.log_positions()[0].copy()   # a reset of every game: the running drain cursor starts at the new logs

    # ---- RiichiEnv.step (env.rs:857-872), batched
    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.uint64).reshape(self.n, 4)
        _chk(self.L.rmj_step(self.h, a.ctypes.data))

    def clone(self):
        """rmj_clone: a new VecRiichiEnv whose games are in exactly this one's state (RiichiEnv.clone, env.rs:358-372)"""
        out = object.__new__(VecRiichiEnv)
        out.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("h", "_lc_buf", "_cursor")})   # (buffers and cursors are per environment)
        h = C.c_void_p()
        _chk(self.L.rmj_clone(self.h, C.byref(h)))
        out.h = h
        return out

    def copy_games(self, dst_idx, src, src_idx):
        """rmj_copy_games: the complete state of src's games src_idx into this environment's games dst_idx"""
        a = np.ascontiguousarray(dst_idx, dtype=np.uint32)
        b = np.ascontiguousarray(src_idx, dtype=np.uint32)
        if a.shape != b.shape:
            raise ValueError("dst_idx and src_idx must have the same length")
        _chk(self.L.rmj_copy_games(self.h, a.ctypes.data, src.h, b.ctypes.data, a.size))

    def apply_events(self, events, masked_ok=False, replay=False):
        """RiichiEnv.apply_event (env.rs:880-887) for every game: `events[g]` is an MJAI dict, pre-built records
        (abi.event_records_from_mjai) or None (no event for game g).  replay=True adds the bookkeeping of the reference's log
        walker (KyokuStepIterator, replay/mod.rs:129-177): a seat that was offered Ron on the last discard and does not win with
        this event has passed - same-turn furiten, permanent in riichi (record flag RMJ_EVF_REPLAY_PASS in `pad`)."""
        buf = (abi.Event * (abi.EVENT_SLOTS * self.n))()
        np_ = 3 if self.game_mode >= 3 else 4
        for g, ev in enumerate(events):
            if ev is None:
                continue
            recs = abi.event_records_from_mjai(ev, np_, masked_ok) if isinstance(ev, dict) else ev
            C.memmove(C.addressof(buf) + g * abi.EVENT_SLOTS * C.sizeof(abi.Event), C.addressof(recs),
                      abi.EVENT_SLOTS * C.sizeof(abi.Event))
            if replay:
                buf[g * abi.EVENT_SLOTS].pad |= 1   # on the staged copy: the caller's records stay as they were
        _chk(self.L.rmj_apply_events(self.h, C.addressof(buf)))

    def sync(self):
        """wait for everything issued on the handle's stream (rmj_sync)"""
        _chk(self.L.rmj_sync(self.h))

    def step_random(self, policy_seed, n_steps=1, auto_reset=False):
        _chk(self.L.rmj_step_random(self.h, policy_seed, n_steps, int(auto_reset)))

    def step_greedy(self, policy_seed, n_steps=1, auto_reset=False, call_rate_256=64):
        """n_steps steps of every game under the device policy that plays to win (header: rmj_step_greedy)"""
        _chk(self.L.rmj_step_greedy(self.h, policy_seed, n_steps, int(auto_reset), call_rate_256))

    def step_random_encode(self, policy_seed, n_steps, d_out_ptr, auto_reset=True, only_active=2):
        """n_steps x (step of every game + encode() of the acting seats into the device tensor at d_out_ptr): header
        rmj_step_random_encode (BASELINE configs[4])."""
        _chk(self.L.rmj_step_random_encode(self.h, policy_seed, n_steps, int(auto_reset), int(only_active), C.c_void_p(d_out_ptr)))

    def encode_compact_device(self, d_out_ptr, d_index_ptr, capacity, d_count_ptr):
        """Observation.encode() of the acting seats as one dense batch in (game, seat) order (header: rmj_encode_compact_device);
        all three pointers are device pointers, asynchronous on the handle's stream."""
        _chk(self.L.rmj_encode_compact_device(self.h, C.c_void_p(d_out_ptr), C.c_void_p(d_index_ptr), capacity, C.c_void_p(d_count_ptr)))

    def step_random_encode_compact(self, policy_seed, n_steps, d_out_ptr, d_index_ptr, capacity, d_count_ptr, auto_reset=True):
        _chk(self.L.rmj_step_random_encode_compact(self.h, policy_seed, n_steps, int(auto_reset), C.c_void_p(d_out_ptr),
                                                   C.c_void_p(d_index_ptr), capacity, C.c_void_p(d_count_ptr)))

    def bench_encode_compact(self, d_out_ptr, d_index_ptr, capacity, d_count_ptr, reps=20):
        ms = C.c_double()
        _chk(self.L.rmj_bench_encode_compact(self.h, C.c_void_p(d_out_ptr), C.c_void_p(d_index_ptr), capacity, C.c_void_p(d_count_ptr),
                                             reps, C.byref(ms)))
        return ms.value

    def random_actions(self, policy_seed):
        a = np.zeros((self.n, 4), np.uint64)
        _chk(self.L.rmj_random_actions(self.h, policy_seed, a.ctypes.data))
        return a

    # ---- observations
    def status(self):
        a, p, d = (np.zeros(self.n, np.uint8) for _ in range(3))
        _chk(self.L.rmj_get_status(self.h, a.ctypes.data, p.ctypes.data, d.ctypes.data))
        return a, p, d

    def done(self):
        return self.status()[2].astype(bool)

    def legal(self):
        l = np.zeros((self.n, 4, abi.MAX_LEGAL), np.uint64)
        c = np.zeros((self.n, 4), np.uint8)
        _chk(self.L.rmj_get_legal(self.h, l.ctypes.data, c.ctypes.data))
        return l, c

    def legal_compact(self):
        """(index [k] = game * 4 + seat, offsets [k + 1], entries [m]): the ordered legal lists of the seats that are to act, in
        (game, seat) order - rmj_get_legal_compact: ~110 B per game over PCIe instead of the 2 KB of legal().  The arrays are VIEWS of
        buffers this environment reuses: valid until its next legal_compact() call (copy them to keep them)."""
        cap_r = getattr(self, "_lc_rows", 0) or (self.n + self.n // 2 + 16)
        cap_e = getattr(self, "_lc_ents", 0) or (self.n * 24 + 1024)
        while True:
            if getattr(self, "_lc_buf", None) is None or self._lc_buf[0].size < cap_r or self._lc_buf[2].size < cap_e:
                self._lc_buf = (np.zeros(cap_r, np.uint32), np.zeros(cap_r + 1, np.uint32), np.zeros(cap_e, np.uint64))
            idx, off, ent = self._lc_buf
            nr, ne = C.c_uint32(), C.c_uint32()
            _chk(self.L.rmj_get_legal_compact(self.h, idx.ctypes.data, off.ctypes.data, ent.ctypes.data, idx.size, ent.size, C.byref(nr), C.byref(ne)))
            if nr.value <= idx.size and ne.value <= ent.size:
                self._lc_rows, self._lc_ents = idx.size, ent.size
                return idx[: nr.value], off[: nr.value + 1], ent[: ne.value]
            cap_r, cap_e = max(cap_r, nr.value + nr.value // 8), max(cap_e, ne.value + ne.value // 8)
            self._lc_buf = None

    def mask(self):
        m = np.zeros((self.n, 4, 82), np.uint8)
        _chk(self.L.rmj_get_mask(self.h, m.ctypes.data))
        return m

    def waits(self):
        w = np.zeros((self.n, 4), np.uint64)
        _chk(self.L.rmj_get_waits(self.h, w.ctypes.data))
        return w

    def scores(self):
        s = np.zeros((self.n, 4), np.int32)
        _chk(self.L.rmj_get_scores(self.h, s.ctypes.data))
        return s

    def ranks(self):
        r = np.zeros((self.n, 4), np.uint8)
        _chk(self.L.rmj_get_ranks(self.h, r.ctypes.data))
        return r

    POINT_RULES = {"basic": 0, "ouza-tyoujyo": 1, "ouza-normal": 2}

    def points(self, rule_name="basic"):
        """RiichiEnv.points(rule_name) (env.rs:691-727) of every game: float64 [n, 4] computed on the device"""
        if rule_name not in self.POINT_RULES or (self.game_mode >= 3 and rule_name != "basic"):
            raise ValueError(f"Unknown preset rule{' for 3P' if self.game_mode >= 3 else ''}: {rule_name}")
        out = np.zeros((self.n, 4), np.float64)
        _chk(self.L.rmj_get_points(self.h, self.POINT_RULES[rule_name], out.ctypes.data))
        return out

    def step_counts(self):
        s = np.zeros(self.n, np.uint64)
        _chk(self.L.rmj_get_step_counts(self.h, s.ctypes.data))
        return s

    def total_steps(self):
        t = C.c_uint64()
        _chk(self.L.rmj_total_steps(self.h, C.byref(t)))
        return t.value

    def peek(self, g) -> abi.StateView:
        v = abi.StateView()
        _chk(self.L.rmj_peek_state(self.h, g, C.byref(v)))
        return v

    def wall_digest(self, g):
        """(salt, wall_digest) of game g's wall (state/wall.rs:15-16); ("", "") without RULE_REFERENCE_RNG."""
        salt, dg = C.create_string_buffer(17), C.create_string_buffer(65)
        _chk(self.L.rmj_get_wall_digest(self.h, int(g), salt, dg))
        return salt.value.decode(), dg.value.decode()

    def wall_digests(self, first=0, n=None):
        """[(salt, wall_digest)] of games [first, first + n): one launch, one lane per game."""
        n = self.n - first if n is None else n
        salts, dgs = np.zeros((n, 17), np.uint8), np.zeros((n, 65), np.uint8)
        _chk(self.L.rmj_get_wall_digests(self.h, first, n, salts.ctypes.data, dgs.ctypes.data))
        return [(bytes(salts[i]).split(b"\0")[0].decode(), bytes(dgs[i]).split(b"\0")[0].decode()) for i in range(n)]

    def poke(self, g, v: abi.StateView):
        _chk(self.L.rmj_poke_state(self.h, g, C.byref(v)))

    def win_results(self, g):
        """RiichiEnv.win_results of game g (env.rs:606-607): {seat: dict} of the round that ended the game."""
        arr = (abi.WinResult * 4)()
        m = C.c_uint8()
        _chk(self.L.rmj_get_win_results(self.h, int(g), arr, C.byref(m)))
        out = {}
        for p in range(4):
            if (m.value >> p) & 1:
                w = arr[p]
                out[p] = dict(is_win=bool(w.is_win), yakuman=bool(w.yakuman), has_win_shape=bool(w.has_win_shape),
                              yaku=list(w.yaku[: w.n_yaku]), han=w.han, fu=w.fu, ron_agari=w.ron_agari,
                              tsumo_agari_oya=w.tsumo_agari_oya, tsumo_agari_ko=w.tsumo_agari_ko,
                              pao_payer=None if w.pao_payer < 0 else int(w.pao_payer))
        return out

    def event_counts(self):
        c = np.zeros(self.n, np.uint32)
        _chk(self.L.rmj_get_event_counts(self.h, c.ctypes.data))
        return c

    def events(self, g, first=0, max_events=1 << 16):
        buf = (abi.Event * max_events)()
        n = C.c_uint32()
        _chk(self.L.rmj_get_events(self.h, g, first, max_events, buf, C.byref(n)))
        return buf, n.value

    def mjai_log(self, g, seat=-1, first=0):
        """MJAI JSON strings of game g (seat=-1: full log; 0..3: that seat's masked view,
        state/mod.rs:2094-2148)."""
        buf, n = self.events(g, first)
        out = []
        s = C.create_string_buffer(2048)
        i = 0
        while i < n:
            used = self.L.rmj_format_event(C.cast(C.byref(buf, i * C.sizeof(abi.Event)), C.POINTER(abi.Event)), n - i, seat,
                                           s, 2048)
            if used <= 0:
                raise RmjError(f"cannot format event {i} of game {g} (type {buf[i].type})")
            out.append(s.value.decode())
            i += used
        return out

    # ---- the logs of all games at once (rmj_drain_events / rmj_format_events / rmj_drain_format)
    def log_positions(self):
        """(base, pos) [n] u32 each: every slot's record stream position and where its current game's log begins (a restart moves
        the base, never the position back: header, rmj_get_log_positions)."""
        base, pos = np.zeros(self.n, np.uint32), np.zeros(self.n, np.uint32)
        _chk(self.L.rmj_get_log_positions(self.h, base.ctypes.data, pos.ctypes.data))
        return base, pos

    def drain_events(self, cursor=None, cap_events=None, peek=False):
        """The records every slot wrote since `cursor` ([n] u32 stream positions, updated in place unless peek; None: the env's own
        running cursor, which starts at 0): (events [total] array of 32-byte records, offsets [n + 1]).  The cursor stays valid across
        auto-reset / reset restarts: a window then holds the end of one game and the start of the next."""
        cur = self._log_cursor() if cursor is None else cursor
        if cap_events is None:
            behind = (self.log_positions()[1] - cur).astype(np.int64)     # u32 wrap-safe distance
            cap_events = int(np.minimum(np.where(behind > 2 ** 31, 0, behind), self.event_ring).sum())
        ev = np.zeros((max(cap_events, 1), C.sizeof(abi.Event)), np.uint8)
        offs = np.zeros(self.n + 1, np.uint32)
        n_ev = C.c_uint32()
        _chk(self.L.rmj_drain_events(self.h, cur.ctypes.data, ev.ctypes.data, int(cap_events), offs.ctypes.data, C.byref(n_ev), 1 if peek else 0))
        return ev[: n_ev.value], offs

    def format_events(self, events, offsets, seat=-1):
        """MJAI strings of a drained buffer: a list (one entry per game) of lists of event strings."""
        n = len(offsets) - 1
        toffs = np.zeros(n + 1, np.uint64)
        need = C.c_uint64()
        ev = np.ascontiguousarray(events)
        self.L.rmj_format_events(ev.ctypes.data, offsets.ctypes.data, n, int(seat), None, 0, toffs.ctypes.data, C.byref(need))
        buf = np.zeros(max(int(need.value), 1), np.uint8)
        _chk(self.L.rmj_format_events(ev.ctypes.data, offsets.ctypes.data, n, int(seat), buf.ctypes.data, int(need.value), toffs.ctypes.data, C.byref(need)))
        return self._split_logs(buf, toffs)

    @staticmethod
    def _split_logs(buf, toffs):
        raw = buf[: int(toffs[-1])].tobytes()
        return [raw[int(toffs[g]): int(toffs[g + 1])].decode().split("\n")[:-1] for g in range(len(toffs) - 1)]

    def _log_cursor(self):
        """the env's running drain cursor (stream positions): it starts where every slot's CURRENT game began when it is first used
        (call it before stepping to have the first drain start at the games' first records), and again after a reset of all games"""
        if getattr(self, "_cursor", None) is None:
            self._cursor = self.log_positions()[0].copy()
        return self._cursor

    def drain_logs(self, seat=-1, cursor=None, timings=None, raw=False, peek=False, out=None):
        """The MJAI strings every slot logged since the last drain (RiichiEnv.mjai_log of every env, env.rs:729-739; seat >= 0: the
        seat's masked log) as a list of lists of strings - drained on the device, one copy down, formatted by host threads in C.
        Restarts in between are part of the stream (... end_game, start_game ...); what a late drain lost is counted (events_lost).
        `cursor`: explicit [n] u32 stream positions instead of the env's running cursor.  peek: cursors and loss counters are left alone.
        raw=True: (bytes buffer, text offsets [n + 1]) without splitting (one log = its events, each followed by a newline).
        timings: a list that receives [gather, copy, format] ms.  out: a uint8 array to format into when it is large enough (a text
        buffer reused from drain to drain: fresh pages cost more than the formatting)."""
        cur = self._log_cursor() if cursor is None else cursor
        toffs = np.zeros(self.n + 1, np.uint64)
        need, n_ev = C.c_uint64(), C.c_uint32()
        ms = (C.c_double * 3)()
        fl = 1 if peek else 0
        # size call (stages the records, moves nothing), then the call that formats the staging into the buffer
        self.L.rmj_drain_format(self.h, cur.ctypes.data, int(seat), None, 0, toffs.ctypes.data, C.byref(need), C.byref(n_ev), None, fl)
        buf = out if out is not None and out.dtype == np.uint8 and out.size >= int(need.value) else np.zeros(max(int(need.value), 1), np.uint8)
        _chk(self.L.rmj_drain_format(self.h, cur.ctypes.data, int(seat), buf.ctypes.data, int(buf.size), toffs.ctypes.data, C.byref(need), C.byref(n_ev), ms, fl))
        if timings is not None:
            timings[:] = [ms[0], ms[1], ms[2]]
        self.last_drain_events = n_ev.value
        return (buf, toffs) if raw else self._split_logs(buf, toffs)

    def mjai_logs(self, seat=-1):
        """The current game's log of every slot as far as its ring still holds it (RiichiEnv.mjai_log of every env): a peek from the
        slots' log bases - no cursor moves, no loss is booked."""
        return self.drain_logs(seat=seat, cursor=self.log_positions()[0].copy(), peek=True)

    def events_lost(self):
        """[n] records each game's ring lost to late drains so far (RmjEventViews.lost)"""
        out = np.zeros(self.n, np.uint32)
        _chk(self.L.rmj_get_events_lost(self.h, out.ctypes.data))
        return out

    def set_encode_row_stride(self, floats=0):
        """Row stride (floats) of the base encoder's outputs; 0 = dense 74 x W.  Header: rmj_set_encode_row_stride (rows padded to a
        multiple of 256 B - padded_row_stride() - are written 1.3-1.4 x faster)."""
        _chk(self.L.rmj_set_encode_row_stride(self.h, int(floats)))
        self.enc_stride = int(floats)

    def padded_row_stride(self):
        """74 x W rounded up to a multiple of 64 floats (256 B): 2 048 in 3P, 2 560 in 4P"""
        dense = 74 * (27 if self.game_mode >= 3 else 34)
        return (dense + 63) // 64 * 64

    def encode(self, only_active=False):
        """Observation.encode() of every seat: float32 [n, 4, 74, 34] (observation/python.rs:457-806)."""
        w = 27 if self.game_mode >= 3 else 34
        stride = getattr(self, "enc_stride", 0) or 74 * w
        out = np.zeros((self.n, 4, stride), np.float32)
        _chk(self.L.rmj_encode(self.h, int(only_active), out.ctypes.data))
        return np.ascontiguousarray(out[:, :, : 74 * w]).reshape(self.n, 4, 74, w)

    def encode_extended(self, only_active=False):
        """Observation.encode_extended() of every (game, seat): [n][4][215][34] f32 (3P: [n][4][215][27])."""
        w = 27 if self.game_mode >= 3 else 34
        out = np.zeros((self.n, 4, 215, w), np.float32)
        _chk(self.L.rmj_encode_extended(self.h, int(only_active), out.ctypes.data))
        return out

    def _encode_aux(self, which, shape):
        out = np.zeros((self.n,) + shape, np.float32)
        _chk(self.L.rmj_encode_aux(self.h, which, out.ctypes.data))
        return out

    def encode_kawa_overview(self):
        """Observation.encode_kawa_overview() of every game: [n][NP][7][W] f32 (absolute seats; header RMJ_AUX_*)."""
        return self._encode_aux(0, (3, 7, 27) if self.game_mode >= 3 else (4, 7, 34))

    def encode_yaku_possibility(self):
        """Observation.encode_yaku_possibility() of every game: [n][NP][21][2] f32."""
        return self._encode_aux(1, (3, 21, 2) if self.game_mode >= 3 else (4, 21, 2))

    def encode_furiten_ron_possibility(self):
        """Observation.encode_furiten_ron_possibility() of every game: [n][NP][21] f32."""
        return self._encode_aux(2, (3, 21) if self.game_mode >= 3 else (4, 21))

    def encode_seq(self, game_style=1):
        """Sequence (transformer) features of every (game, seat), 4P only (header: rmj_encode_seq): dict of padded arrays
        sparse [n,4,25] u16, numeric [n,4,12] f32, progression [n,256,5] u16, candidates [n,4,64,4] u16 and their lengths."""
        n = self.n
        out = dict(sparse=np.zeros((n, 4, abi.SEQ_SPARSE), np.uint16), n_sparse=np.zeros((n, 4), np.uint8),
                   numeric=np.zeros((n, 4, 12), np.float32), progression=np.zeros((n, abi.SEQ_PROG, 5), np.uint16),
                   n_progression=np.zeros(n, np.uint16), candidates=np.zeros((n, 4, abi.SEQ_CAND, 4), np.uint16),
                   n_candidates=np.zeros((n, 4), np.uint8))
        b = abi.SeqBuffers(*[out[k].ctypes.data for k in ("sparse", "n_sparse", "numeric", "progression", "n_progression",
                                                            "candidates", "n_candidates")])
        _chk(self.L.rmj_encode_seq(self.h, int(game_style), C.byref(b)))
        return out

    def encode_seq_delta(self, game_style=1):
        """The sequence features over the events of the seats' latest observation, like the reference's live environment
        (header: rmj_encode_seq_delta): progression is per seat [n,4,64,5]; seats that are not to act get empty outputs."""
        n = self.n
        out = dict(sparse=np.zeros((n, 4, abi.SEQ_SPARSE), np.uint16), n_sparse=np.zeros((n, 4), np.uint8),
                   numeric=np.zeros((n, 4, 12), np.float32), progression=np.zeros((n, 4, abi.SEQ_DELTA_PROG, 5), np.uint16),
                   n_progression=np.zeros((n, 4), np.uint16), candidates=np.zeros((n, 4, abi.SEQ_CAND, 4), np.uint16),
                   n_candidates=np.zeros((n, 4), np.uint8))
        b = abi.SeqBuffers(*[out[k].ctypes.data for k in ("sparse", "n_sparse", "numeric", "progression", "n_progression",
                                                            "candidates", "n_candidates")])
        _chk(self.L.rmj_encode_seq_delta(self.h, int(game_style), C.byref(b)))
        return out

    def bench_rollout(self, policy_seed, warmup, steps) -> abi.BenchResult:
        r = abi.BenchResult()
        _chk(self.L.rmj_bench_rollout(self.h, policy_seed, warmup, steps, C.byref(r)))
        return r

    def time_rollout(self, policy_seed, steps) -> abi.BenchResult:
        """HIP-event time of rmj_step_random(seed, steps, auto_reset) alone (header: rmj_time_rollout); counters stay 0."""
        r = abi.BenchResult()
        _chk(self.L.rmj_time_rollout(self.h, policy_seed, steps, C.byref(r)))
        return r

    def time_rollout_encode(self, policy_seed, steps, d_out_ptr) -> abi.BenchResult:
        """HIP-event time of rmj_step_random_encode(seed, steps, auto_reset, only_active = 2, d_out) (header: rmj_time_rollout_encode)"""
        r = abi.BenchResult()
        _chk(self.L.rmj_time_rollout_encode(self.h, policy_seed, steps, C.c_void_p(d_out_ptr), C.byref(r)))
        return r

    def time_rollout_greedy(self, policy_seed, steps, call_rate_256=64) -> abi.BenchResult:
        r = abi.BenchResult()
        _chk(self.L.rmj_time_rollout_greedy(self.h, policy_seed, steps, call_rate_256, C.byref(r)))
        return r

    def bench_rollout_validated(self, policy_seed, warmup, steps) -> abi.BenchResult:
        """One policy launch + one validating step launch per step (header: rmj_bench_rollout_validated)."""
        r = abi.BenchResult()
        _chk(self.L.rmj_bench_rollout_validated(self.h, policy_seed, warmup, steps, C.byref(r)))
        return r

    def bench_encode(self, d_out_ptr, reps, extended=False, only_active=2) -> float:
        """Average ms of one encoder launch into the device buffer at `d_out_ptr` (header: rmj_bench_encode)."""
        ms = C.c_double()
        _chk(self.L.rmj_bench_encode(self.h, int(extended), int(only_active), C.c_void_p(d_out_ptr), reps, C.byref(ms)))
        return ms.value

    def set_rollout_streams(self, k):
        _chk(self.L.rmj_set_rollout_streams(self.h, int(k)))

    def total_full_path(self):
        t = C.c_uint64()
        _chk(self.L.rmj_total_full_path(self.h, C.byref(t)))
        return t.value

    def peek_outputs(self, g):
        """(legal [4][64] u64, counts [4], mask [4][82], waits [4], active_mask, phase, done) of one game."""
        legal = np.zeros((4, abi.MAX_LEGAL), np.uint64)
        cnt = np.zeros(4, np.uint8)
        mask = np.zeros((4, 82), np.uint8)
        waits = np.zeros(4, np.uint64)
        st = C.c_uint32()
        _chk(self.L.rmj_peek_outputs(self.h, int(g), legal.ctypes.data, cnt.ctypes.data, mask.ctypes.data, waits.ctypes.data,
                                     C.byref(st)))
        return legal, cnt, mask, waits, st.value & 0xFF, (st.value >> 8) & 0xFF, (st.value >> 16) & 0xFF


# ---- batched hand math (kernel gate) -----------------------------------------------------
def eval_hands(cases, device=0):
    L = load_lib()
    arr = (abi.HandCase * len(cases))(*cases)
    out = (abi.HandResult * len(cases))()
    _chk(L.rmj_eval_hands(device, arr, len(cases), out))
    return list(out)


def agari_counts(counts, device=0):
    L = load_lib()
    counts = np.ascontiguousarray(counts, dtype=np.uint8)
    n = counts.shape[0]
    ag = np.zeros(n, np.uint8)
    tp = np.zeros(n, np.uint8)
    w = np.zeros(n, np.uint64)
    _chk(L.rmj_agari_counts(device, counts.ctypes.data, n, ag.ctypes.data, tp.ctypes.data, w.ctypes.data))
    return ag, tp, w


def calculate_score(han, fu, is_oya, is_tsumo, honba, num_players, device=0):
    L = load_lib()
    a = [np.ascontiguousarray(x, dtype=np.uint8) for x in (han, fu, is_oya, is_tsumo)]
    hb = np.ascontiguousarray(honba, dtype=np.uint32)
    npl = np.ascontiguousarray(num_players, dtype=np.uint8)
    n = len(a[0])
    out = np.zeros((n, 4), np.uint32)
    _chk(L.rmj_calculate_score(device, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                               hb.ctypes.data, npl.ctypes.data, n, out.ctypes.data))
    return out


def shanten(counts, sanma=False, device=0):
    """calculate_shanten / calculate_shanten_3p (shanten.rs:244-261, 470-484) over [n,34] histograms."""
    L = load_lib()
    counts = np.ascontiguousarray(counts, dtype=np.uint8)
    n = counts.shape[0]
    out = np.zeros(n, np.int8)
    _chk(L.rmj_shanten(device, counts.ctypes.data, n, int(sanma), out.ctypes.data))
    return out


def effective_tiles(counts, sanma=False, device=0):
    """calculate_effective_tiles(_3p)_with_discard on [n][34] type histograms; ValueError for a 3n hand like the
    reference's assertion (shanten.rs:304-327, 525-548)."""
    L = load_lib()
    counts = np.ascontiguousarray(counts, dtype=np.uint8).reshape(-1, 34)
    out = np.zeros(counts.shape[0], np.uint32)
    _chk(L.rmj_effective_tiles(device, counts.ctypes.data, counts.shape[0], int(sanma), out.ctypes.data))
    if (out == 0xFFFFFFFF).any():
        raise ValueError("calculate_effective_tiles_with_discard requires a 3n+1 or 3n+2 hand")
    return out


def best_ukeire(counts, visible, sanma=False, device=0):
    """calculate_best_ukeire(_3p) on [n][34] hand / visible type histograms (shanten.rs:331-405, 552-626)."""
    L = load_lib()
    counts = np.ascontiguousarray(counts, dtype=np.uint8).reshape(-1, 34)
    visible = np.ascontiguousarray(visible, dtype=np.uint8).reshape(-1, 34)
    assert counts.shape == visible.shape
    out = np.zeros(counts.shape[0], np.uint32)
    _chk(L.rmj_best_ukeire(device, counts.ctypes.data, visible.ctypes.data, counts.shape[0], int(sanma), out.ctypes.data))
    return out


def mjai_log_dicts(strings):
    return [json.loads(s) for s in strings]
