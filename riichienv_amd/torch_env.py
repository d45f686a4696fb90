"""Trainer-side batched environment: a Gym-style loop over all games of a shard with the policy
on the same GPU.  Observations are encoded straight into torch tensors (rmj_encode_device / rmj_encode_extended_device),
masks / legal lists / status are zero-copy views of the library's device slabs, actions are the policy's categorical
ids (rmj_step_ids_device = Observation.find_action + RiichiEnv.step), rewards come from scores()/ranks()
(riichienv-python/src/env.rs:401-404, 673-727).  Replaces the per-game Python loop of
riichienv-ml trainers/_ppo_worker.py:113-466.

torch is used for device memory and streams only; all game logic runs in the HIP library."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi, vecenv


class DeviceViews(C.Structure):
    _fields_ = [("n_games", C.c_uint32), ("reserved", C.c_uint32), ("status", C.c_void_p), ("nlegal", C.c_void_p),
                ("legal", C.c_void_p), ("mask", C.c_void_p), ("waits", C.c_void_p), ("stream", C.c_void_p)]


class _CudaArray:
    """Minimal __cuda_array_interface__ carrier so that torch.as_tensor wraps library-owned device memory in place."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3,
                                         "strides": None}
        self._owner = owner   # keeps the environment (and with it the allocation) alive


class TorchVecEnv:
    """`n_games` games on one GPU driven by a torch policy.

    obs(), mask, status … describe the state AFTER the last step; step(action_ids) takes an int32 tensor [n, 4] on the
    same device (-1 for seats that do not act)."""

    def __init__(self, n_games, game_mode=2, seed=0, device=0, extended=False, skip_mjai_logging=True, share_stream=True, pad_rows=True, **kw):
        """share_stream: issue the library's kernels on torch's current stream of `device` (rmj_set_stream): policy and
        environment are then ordered by the stream, without host synchronisation between them.  With False the library keeps
        its own stream and every call synchronises."""
        import torch

        self.torch = torch
        self.device = torch.device("cuda", device)
        self.env = vecenv.VecRiichiEnv(n_games, game_mode=game_mode, seed=seed, device=device,
                                       skip_mjai_logging=skip_mjai_logging, **kw)
        L = self.env.L
        L.rmj_device_views.argtypes = [C.c_void_p, C.POINTER(DeviceViews)]
        L.rmj_step_ids_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rmj_scores_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rmj_sync.argtypes = [C.c_void_p]
        self.n = int(n_games)
        self.sanma = self.env.game_mode >= 3
        self.extended = bool(extended)
        self.channels = 215 if extended else 74
        self.width = 27 if self.sanma else 34
        v = DeviceViews()
        vecenv._chk(L.rmj_device_views(self.env.h, C.byref(v)))
        wrap = lambda ptr, shape, ts: torch.as_tensor(_CudaArray(ptr, shape, ts, self), device=self.device)  # noqa: E731
        self.status_raw = wrap(v.status, (self.n,), "<u4") if hasattr(torch, "uint32") else wrap(v.status, (self.n,), "<i4")
        self.nlegal = wrap(v.nlegal, (self.n, 4), "|u1")
        self.mask = wrap(v.mask, (self.n, 4, abi.ACTION_SPACE_4P), "|u1")      # zero-copy, rewritten by every step
        self.legal = wrap(v.legal, (self.n, 4, abi.MAX_LEGAL), "<i8")          # packed actions (bit pattern of the u64)
        self.waits = wrap(v.waits, (self.n, 4), "<i8")
        # pad_rows (the default since round 5): every (game, seat) row of the 74-channel tensor padded to a multiple of 256 B
        # (rmj_set_encode_row_stride): the acting seats' rows are written 1.3-1.4 x faster (trainer loop 260 -> 274 M env.step/s); obs() then
        # returns a strided view [n, 4, 74, W] of the padded buffer (same values, same indexing; .contiguous() copies).  pad_rows=False: dense
        self.pad_rows = bool(pad_rows) and not extended
        if self.pad_rows:
            self.row_stride = self.env.padded_row_stride()
            self.env.set_encode_row_stride(self.row_stride)
            self._obs_buf = torch.zeros((self.n, 4, self.row_stride), dtype=torch.float32, device=self.device)
            self._obs = self._obs_buf[:, :, : 74 * self.width].unflatten(-1, (74, self.width))
        else:
            self.row_stride = self.channels * self.width
            self._obs_buf = torch.zeros((self.n, 4, self.channels, self.width), dtype=torch.float32, device=self.device)
            self._obs = self._obs_buf
        self._scores = torch.zeros((self.n, 4), dtype=torch.int32, device=self.device)
        self.shared = bool(share_stream)
        if self.shared:
            vecenv._chk(L.rmj_set_stream(self.env.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), 0))
        self.env.reset()

    # ---- status helpers (views, no copies beyond the arithmetic)
    def _status(self):
        s = self.status_raw.to(self.torch.int64) & 0xFFFFFFFF
        return s & 0xFF, (s >> 8) & 0xFF, ((s >> 16) & 0xFF).bool()

    def active(self):
        """bool [n, 4]: seats that must act now"""
        a, _, done = self._status()
        seats = self.torch.arange(4, device=self.device)
        return (((a[:, None] >> seats[None, :]) & 1) == 1) & ~done[:, None]

    def done(self):
        return self._status()[2]

    def obs(self, only_active=True):
        """encode() / encode_extended() into a resident tensor [n, 4, C, W].  only_active=True refreshes only the rows
        of the seats that must act (rows of the other seats keep their previous contents: mask them with active())."""
        L = self.env.L
        fn = L.rmj_encode_extended_device if self.extended else L.rmj_encode_device
        vecenv._chk(fn(self.env.h, 2 if only_active else 0, C.c_void_p(self._obs_buf.data_ptr())))
        self.sync()
        return self._obs

    def obs_compact(self, capacity=None, sync_count=True):
        """The batch a policy consumes: encode() of the acting seats only, dense, in (game, seat) order
        (rmj_encode_compact_device).  Returns (obs [k, 74, W], index [k] int32 = game * 4 + seat); with sync_count=False the
        full-capacity buffers and the device count tensor are returned instead (no host round trip: rows behind the count
        hold old data).  capacity defaults to n + n // 2 rows and grows on demand."""
        t = self.torch
        if self.extended:
            raise vecenv.RmjError("obs_compact() covers Observation.encode(); use obs() for encode_extended()")
        cap = int(capacity or getattr(self, "_cap", 0) or (self.n + self.n // 2 + 1))
        if getattr(self, "_cobs", None) is None or self._cobs.shape[0] < cap:
            self._cobs_buf = t.zeros((cap, self.row_stride), dtype=t.float32, device=self.device)
            self._cobs = self._cobs_buf[:, : 74 * self.width].unflatten(-1, (74, self.width))
            self._cidx = t.zeros((cap,), dtype=t.int32, device=self.device)
            self._ccnt = t.zeros((1,), dtype=t.int32, device=self.device)
            self._cap = cap
            if not self.shared:
                t.cuda.current_stream(self.device).synchronize()   # the buffers were zeroed on torch's stream
        self.env.encode_compact_device(self._cobs_buf.data_ptr(), self._cidx.data_ptr(), self._cap, self._ccnt.data_ptr())
        self.sync()
        if not sync_count:
            return self._cobs, self._cidx, self._ccnt
        k = int(self._ccnt.item())
        if k > self._cap:                                          # more claimants than rows: grow once and encode again
            self._cobs = None
            return self.obs_compact(capacity=k + k // 8, sync_count=True)
        return self._cobs[:k], self._cidx[:k]

    def scores(self):
        vecenv._chk(self.env.L.rmj_scores_device(self.env.h, C.c_void_p(self._scores.data_ptr()), None))
        self.sync()
        return self._scores

    def points(self, rule_name="basic"):
        """RiichiEnv.points(rule_name) (env.rs:691-727) of every game as a float64 tensor [n, 4] on the device (rmj_points_device):
        the terminal reward of a trainer loop (riichienv-ml trainers/_ppo_worker.py reads env.points / env.ranks on the host)"""
        rules = vecenv.VecRiichiEnv.POINT_RULES
        if rule_name not in rules or (self.sanma and rule_name != "basic"):
            raise ValueError(f"Unknown preset rule{' for 3P' if self.sanma else ''}: {rule_name}")
        if not hasattr(self, "_points"):
            self._points = self.torch.zeros((self.n, 4), dtype=self.torch.float64, device=self.device)
            if not self.shared:
                self.torch.cuda.current_stream(self.device).synchronize()
        vecenv._chk(self.env.L.rmj_points_device(self.env.h, rules[rule_name], C.c_void_p(self._points.data_ptr())))
        self.sync()
        return self._points

    def ranks(self):
        """ranks() of env.rs:673-689 (1 = first; ties broken by seat) computed on the device from scores()"""
        sc = self.scores().to(self.torch.int64)
        np_ = 3 if self.sanma else 4
        key = sc[:, :np_] * 8 - self.torch.arange(np_, device=self.device)[None, :]
        r = (key[:, None, :] > key[:, :, None]).sum(-1) + 1
        out = self.torch.zeros((self.n, 4), dtype=self.torch.int64, device=self.device)
        out[:, :np_] = r
        return out

    def step(self, action_ids, auto_reset=True):
        t = self.torch
        ids = action_ids.to(device=self.device, dtype=t.int32).contiguous()
        assert ids.shape == (self.n, 4)
        if not self.shared:
            t.cuda.current_stream(self.device).synchronize()  # the ids were produced on torch's stream
        vecenv._chk(self.env.L.rmj_step_ids_device(self.env.h, C.c_void_p(ids.data_ptr()), int(auto_reset)))
        self.sync()

    # ---- rewards (rmj_round_track_device): what riichienv-ml's PPO worker derives between steps on the host
    def round_track(self):
        """Call once after every step: (ended [n] u8, delta [n, 4] i32, meta [n, 4] i32, kyoku_idx [n] u8) - ended 1: a round ended in
        the last step and the next one was dealt, 2: the round and the game ended; delta = the seats' score change over that round,
        meta = (round_wind, oya, honba, riichi_sticks) when it was dealt (the GRP features chang / ju / ben / liqibang of
        trainers/_ppo_worker.py:100-116).  The first call only takes the baseline.  Resident tensors, rewritten by every call."""
        t = self.torch
        if not hasattr(self, "_rt"):
            self._rt = (t.zeros((self.n,), dtype=t.uint8, device=self.device), t.zeros((self.n, 4), dtype=t.int32, device=self.device),
                        t.zeros((self.n, 4), dtype=t.int32, device=self.device), t.zeros((self.n,), dtype=t.uint8, device=self.device))
            if not self.shared:
                t.cuda.current_stream(self.device).synchronize()
        e, d, m, k = self._rt
        vecenv._chk(self.env.L.rmj_round_track_device(self.env.h, C.c_void_p(e.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(m.data_ptr()),
                                                     C.c_void_p(k.data_ptr())))
        self.sync()
        return self._rt

    RANK_REWARDS_4P = (10.0, 4.0, -4.0, -10.0)   # trainers/_ppo_worker.py:283-291
    RANK_REWARDS_3P = (10.0, 0.0, -10.0)

    def step_rl(self, action_ids, auto_reset=True, kyoku_scale=1.0 / 1000.0, rank_rewards=None, with_obs=True):
        """One Gym-style transition of every game: (obs, reward, terminated, info).
        reward [n, 4] f32 per seat = kyoku_scale x the seat's score change over a round, paid in the step that ends the round
        (the worker's reward at a kyoku boundary, _ppo_worker.py:240-266, with the identity in place of its learned GRP model: the
        model's inputs are in info), plus the rank reward of _ppo_worker.py:283-291 in the step that ends the game (rank_rewards:
        a per-rank tuple, default 10 / 4 / -4 / -10; () for none).  terminated [n] bool: the game ended in this step (with auto_reset
        it restarts at the next one).  info: ended, delta, meta, kyoku_idx (round_track), scores, ranks of the games that ended.
        obs: the resident feature tensor (step_obs) - rows of the seats that act next; None with with_obs=False (extended encoders:
        call obs() yourself)."""
        t = self.torch
        if not hasattr(self, "_rt"):
            self.round_track()                       # baseline before the first transition
        if with_obs and not self.extended:
            obs = self.step_obs(action_ids, auto_reset=auto_reset)
        else:
            self.step(action_ids, auto_reset=auto_reset)
            obs = None
        ended, delta, meta, kidx = self.round_track()
        reward = delta.to(t.float32) * float(kyoku_scale)
        terminated = ended == 2
        rr = rank_rewards if rank_rewards is not None else (self.RANK_REWARDS_3P if self.sanma else self.RANK_REWARDS_4P)
        ranks = self.ranks()
        if len(rr):
            table = t.tensor([0.0] + list(rr) + [0.0] * (4 - len(rr)), dtype=t.float32, device=self.device)
            reward = reward + t.where(terminated[:, None], table[ranks], t.zeros_like(reward))
        info = {"ended": ended, "delta": delta, "meta": meta, "kyoku_idx": kidx, "scores": self.scores(), "ranks": ranks}
        return obs, reward, terminated, info

    def step_obs(self, action_ids, auto_reset=True):
        """step(action_ids) and obs(only_active=True) as ONE launch (rmj_step_ids_encode_device): returns the resident tensor
        [n, 4, 74, W] whose rows of the seats that are to act next have just been written.  Base encoding only."""
        if self.extended:
            raise vecenv.RmjError("step_obs() covers Observation.encode(); use step() + obs() for encode_extended()")
        t = self.torch
        ids = action_ids.to(device=self.device, dtype=t.int32).contiguous()
        assert ids.shape == (self.n, 4)
        if not self.shared:
            t.cuda.current_stream(self.device).synchronize()
        vecenv._chk(self.env.L.rmj_step_ids_encode_device(self.env.h, C.c_void_p(ids.data_ptr()), int(auto_reset), C.c_void_p(self._obs_buf.data_ptr())))
        self.sync()
        return self._obs

    def step_sample_obs(self, logits=None, seed=0, auto_reset=True):
        """sample_ids(logits, seed) + step_obs(ids) as ONE launch (rmj_step_sample_encode_device): returns (ids [n, 4] int32 - the ids the
        two calls would have produced -, the resident tensor [n, 4, 74, W] with the rows of the seats that act next).  logits: float32
        [n, 4, A'] (A' >= 82; 60 in 3P) on this device, or None for the uniform policy.  Base encoding only."""
        if self.extended:
            raise vecenv.RmjError("step_sample_obs() covers Observation.encode(); use sample_ids() + step() + obs() for encode_extended()")
        t = self.torch
        if not hasattr(self, "_ids"):
            self._ids = t.full((self.n, 4), -1, dtype=t.int32, device=self.device)
        ptr, stride = None, 0
        if logits is not None:
            assert logits.dtype == t.float32 and logits.is_contiguous() and tuple(logits.shape[:2]) == (self.n, 4)
            ptr, stride = C.c_void_p(logits.data_ptr()), int(logits.shape[2])
            if not self.shared:
                t.cuda.current_stream(self.device).synchronize()
        vecenv._chk(self.env.L.rmj_step_sample_encode_device(self.env.h, ptr, stride, int(seed) & 0xFFFFFFFFFFFFFFFF, int(auto_reset),
                                                             C.c_void_p(self._ids.data_ptr()), C.c_void_p(self._obs_buf.data_ptr())))
        self.sync()
        return self._ids, self._obs

    def bind_stream(self, stream=None):
        """Issue the library's further work on `stream` (a torch.cuda.Stream; default: torch's current stream of the device) -
        rmj_set_stream.  Work already issued is waited for first.  This is what puts the environment INSIDE a HIP graph: bind the
        side stream the graph will be captured on, run one warm-up iteration there, then capture `sample_ids` / `step_obs` /
        `step` / `obs` / `scores` / `points` together with the policy's kernels (`with torch.cuda.graph(g, stream=s): ...`) - these
        calls only launch kernels on the bound stream (no allocation, no host synchronisation), and the sampler's noise is keyed by
        every game's own step count, so each replay of the graph draws fresh actions.  `obs_compact(sync_count=True)` reads a count
        on the host and cannot be captured."""
        t = self.torch
        s = stream if stream is not None else t.cuda.current_stream(self.device)
        vecenv._chk(self.env.L.rmj_set_stream(self.env.h, C.c_void_p(s.cuda_stream), 0))
        self.shared = True
        return s

    def sync(self):
        """own stream: wait for the library's work; shared stream: nothing to do, torch's stream orders it"""
        if not self.shared:
            vecenv._chk(self.env.L.rmj_sync(self.env.h))

    def copy_games(self, dst_idx, src_env, src_idx):
        """rmj_copy_games_device: the complete state of src_env's games `src_idx` into this environment's games `dst_idx` (int32 /
        int64 tensors on this device; src_env may be self when the two index sets are disjoint) - forks for a tree search that
        lives on the GPU, saved positions, refilling slots.  Asynchronous on a shared stream; with the library's own stream the call
        returns when the copy is done (the index tensors are temporaries of torch's allocator)."""
        t = self.torch
        a = dst_idx.to(device=self.device, dtype=t.int32).contiguous()
        b = src_idx.to(device=self.device, dtype=t.int32).contiguous()
        assert a.numel() == b.numel()
        if not self.shared:
            t.cuda.current_stream(self.device).synchronize()
        L = self.env.L
        vecenv._chk(L.rmj_copy_games_device(self.env.h, C.c_void_p(a.data_ptr()), src_env.env.h, C.c_void_p(b.data_ptr()), int(a.numel())))
        # own stream: torch's caching allocator orders the reuse of `a` / `b` by torch's stream, not by the library's, so the
        # kernel must have read them before they are released (shared stream: torch's stream orders it)
        self.sync()

    def sample_ids(self, logits=None, seed=0, index=None, count=None):
        """One id per acting seat drawn from softmax(logits) over the seat's legal ids, -1 elsewhere, by ONE kernel of the
        library (rmj_sample_ids_device: Gumbel-max on the resident mask slab) - no torch indexing / multinomial in the loop.
        logits: float32 [n, 4, A'] with A' >= 82 (60 in 3P) on this device, or None for the uniform policy; with `index`
        (the second result of obs_compact(): game * 4 + seat per row) logits are the compact rows [k, A'] a policy computed
        from the compact observation batch, scattered here into the [n, 4, A'] layout (rows of seats that do not act are never
        read by the sampler).  `count`: the device count tensor of obs_compact(sync_count=False) - rows of `index` behind it hold stale
        (game, seat) values that may repeat live ones, so they are sent to a sink row instead of racing with the live rows; without
        `count` every row of `index` must be live (the sync_count=True form)."""
        t = self.torch
        if not hasattr(self, "_ids"):
            self._ids = t.full((self.n, 4), -1, dtype=t.int32, device=self.device)
        if logits is not None and index is not None:
            a = int(logits.shape[-1])
            if getattr(self, "_full_logits", None) is None or self._full_logits.shape[-1] != a:
                self._full_logits = t.zeros((self.n * 4 + 1, a), dtype=t.float32, device=self.device)   # (+ the sink row)
            k = int(index.shape[0])
            dst = index.to(t.int64)
            if count is not None:
                live = t.arange(k, device=self.device) < count.to(t.int64).reshape(-1)[0]
                dst = t.where(live, dst, t.full_like(dst, self.n * 4))
            elif k > self.n * 4:
                raise ValueError("sample_ids: more index rows than (game, seat) pairs - pass the device count of obs_compact(sync_count=False)")
            self._full_logits[dst] = logits.reshape(-1, a)[:k].to(t.float32)
            logits = self._full_logits[: self.n * 4].view(self.n, 4, a)
        ptr, stride = None, 0
        if logits is not None:
            assert logits.dtype == t.float32 and logits.is_contiguous() and tuple(logits.shape[:2]) == (self.n, 4)
            ptr, stride = C.c_void_p(logits.data_ptr()), int(logits.shape[2])
            if not self.shared:
                t.cuda.current_stream(self.device).synchronize()   # the logits were produced on torch's stream
        vecenv._chk(self.env.L.rmj_sample_ids_device(self.env.h, ptr, stride, int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                     C.c_void_p(self._ids.data_ptr())))
        self.sync()
        return self._ids

    def sample_random_ids(self, generator=None):
        """uniform choice among the legal ids of every acting seat (a masked categorical policy's baseline)"""
        t = self.torch
        m = self.mask.to(t.float32) * self.active()[:, :, None]
        has = m.sum(-1) > 0
        probs = t.where(has[:, :, None], m, t.ones_like(m))
        ids = t.multinomial(probs.view(-1, m.shape[-1]), 1, generator=generator).view(self.n, 4)
        return t.where(has, ids, t.full_like(ids, -1)).to(t.int32)


class ShardedTorchVecEnv:
    """`n_games` games as `parts` shards (by game index, like the ranks of a multi-GPU run) on `parts` torch streams of ONE
    GPU.  A step driven by an external policy ends with its slowest wave and the encoder is bound by its stores (DESIGN.md
    sections 4.6 / 4.7); with the shards on separate streams the tail of one shard's step and the encoder of another overlap
    (measured: 216 M -> 285 M env.step/s at 65 536 games with the fused sampler).  Results do not depend on the split: shard
    i holds the global games [i * n / parts, (i + 1) * n / parts) (game_offset), seeds and sampler noise are keyed by the
    global game index."""

    def __init__(self, n_games, parts=4, device=0, **kw):
        import torch

        if n_games % parts:
            raise ValueError("n_games must be a multiple of parts")
        self.torch = torch
        self.n, self.parts, self.per = int(n_games), int(parts), int(n_games) // int(parts)
        self.device = torch.device("cuda", device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.parts)]
        base = int(kw.pop("game_offset", 0))
        self.shards = []
        for i, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                self.shards.append(TorchVecEnv(self.per, device=device, share_stream=True, game_offset=base + i * self.per, **kw))

    def for_each(self, fn):
        """fn(shard, i) for every shard under the shard's stream (everything fn enqueues is ordered with the shard's kernels);
        returns the list of results.  Nothing synchronises: call synchronize() before reading results on the host."""
        out = []
        for i, (e, st) in enumerate(zip(self.shards, self.streams)):
            with self.torch.cuda.stream(st):
                out.append(fn(e, i))
        return out

    def step_policy(self, policy, auto_reset=True, compact=True):
        """One step of every game: per shard, the observations of the acting seats (obs_compact: dense rows + index + device
        count, or the [n, 4] tensor with compact=False), `policy(shard, obs...) -> int32 ids [per, 4]` (-1 = no action), step."""
        def one(e, _i):
            ids = policy(e, *e.obs_compact(sync_count=False)) if compact else policy(e, e.obs(only_active=True))
            e.step(ids, auto_reset=auto_reset)
        self.for_each(one)

    def step_policy_one_launch(self, policy, auto_reset=True):
        """One step of every game with step + encode as ONE launch per shard (TorchVecEnv.step_obs): per shard,
        `policy(shard, obs [per, 4, 74, W]) -> int32 ids [per, 4]` on the observations the previous call left, then step_obs."""
        def one(e, _i):
            if getattr(e, "_cur_obs", None) is None:
                e._cur_obs = e.obs(only_active=True)
            e._cur_obs = e.step_obs(policy(e, e._cur_obs), auto_reset=auto_reset)
        self.for_each(one)

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def scores(self):
        self.synchronize()
        out = self.torch.cat(self.for_each(lambda e, _i: e.scores().clone()))
        self.synchronize()
        return out

    def step_counts(self):
        import numpy as np

        self.synchronize()
        return np.concatenate([e.env.step_counts() for e in self.shards])


class GymVectorAdapter:
    """A Gymnasium-VectorEnv-shaped face of TorchVecEnv for single-agent learners (the hero-versus-opponents set-up of
    riichienv-ml's PPO worker, trainers/_ppo_worker.py:113-466): env i exposes ONE seat, hero[i]; the other seats are played by
    `opponent(env, active_mask) -> ids [n, 4]` (default: the library's uniform sampler over the legal ids).  reset() / step(actions)
    return what gymnasium.vector.VectorEnv returns - (obs, info) and (obs, reward, terminated, truncated, info) - with torch tensors
    on the device; gymnasium itself is not needed (and not installed in this image).

    step(actions [n] int): the hero's action id for every env, then the environment advances - opponents' turns, claims, round ends -
    until the hero is to act again in every env or its game is over.  Envs that reach the hero's turn early wait: they send no action
    (-1) and stay where they are while the others catch up.  reward [n] = the hero's share of step_rl's rewards summed over the inner
    steps; terminated [n]: the game ended (it restarts with the next step: autoreset like gymnasium's NEXT_STEP mode).
    observation: {"features": [n, 74, W] f32 (the hero's Observation.encode()), "mask": [n, A] bool}."""

    def __init__(self, env: TorchVecEnv, hero=0, opponent=None, max_inner=256):
        self.env, self.t = env, env.torch
        t = self.t
        self.num_envs = env.n
        self.hero = (t.full((env.n,), int(hero), dtype=t.int64, device=env.device) if isinstance(hero, int)
                     else hero.to(device=env.device, dtype=t.int64))
        self.opponent = opponent
        self.max_inner = int(max_inner)
        self.action_space_n = abi.ACTION_SPACE_3P if env.sanma else abi.ACTION_SPACE_4P
        self.single_observation_shape = {"features": (74, env.width), "mask": (self.action_space_n,)}
        self._seat = t.arange(4, device=env.device)[None, :]
        self._seed = 0

    def _hero_turn(self):
        return self.env.active().gather(1, self.hero[:, None])[:, 0]

    def _obs(self):
        e, t = self.env, self.t
        feats = e.obs(only_active=True)
        idx = self.hero[:, None, None, None].expand(-1, 1, feats.shape[2], feats.shape[3])
        mask = e.mask.gather(1, self.hero[:, None, None].expand(-1, 1, e.mask.shape[2]))[:, 0, : self.action_space_n] != 0
        return {"features": feats.gather(1, idx)[:, 0], "mask": mask & self._hero_turn()[:, None]}

    def _advance(self, hero_ids, reward, terminated):
        """inner steps until every env is at its hero's decision (or over); hero_ids [n]: the hero's id for the FIRST inner step, -1 = none"""
        e, t = self.env, self.t
        first = True
        for _ in range(self.max_inner):
            act = e.active()
            hero_now = act.gather(1, self.hero[:, None])[:, 0]
            over = e.done() | terminated
            wait = hero_now & ((hero_ids < 0) if first else t.ones_like(hero_now))   # at the hero's decision with nothing to send: stay
            if bool((wait | over).all()):
                break
            self._seed += 1
            ids = (self.opponent(e, act) if self.opponent is not None else e.sample_ids(None, seed=self._seed)).clone().to(t.int32)
            if first:
                ids = t.where(self._seat == self.hero[:, None], hero_ids.to(t.int32)[:, None].expand(-1, 4), ids)
            ids = t.where((wait | over)[:, None], t.full_like(ids, -1), ids)
            _o, r, term, _info = e.step_rl(ids, auto_reset=False, with_obs=False)
            reward += r.gather(1, self.hero[:, None])[:, 0]
            terminated |= term
            first = False
        return reward, terminated

    def reset(self, seed=None):
        e, t = self.env, self.t
        self._seed = int(seed or 0) * 1000003
        e.env.reset()
        if hasattr(e, "_rt"):
            vecenv._chk(e.env.L.rmj_round_track_reset(e.env.h))
        else:
            e.round_track()
        self._terminated = t.zeros((e.n,), dtype=t.bool, device=e.device)
        self._advance(t.full((e.n,), -1, dtype=t.int64, device=e.device), t.zeros((e.n,), dtype=t.float32, device=e.device),
                      t.zeros((e.n,), dtype=t.bool, device=e.device))
        return self._obs(), {}

    def step(self, actions):
        e, t = self.env, self.t
        if bool(self._terminated.any()):   # games that ended in the previous step restart now (their action is ignored)
            e.env.reset(select=self._terminated.to(t.uint8).cpu().numpy())
            vecenv._chk(e.env.L.rmj_round_track_device(e.env.h, None, None, None, None))   # re-opens the restarted games silently
            lead = self._terminated
            self._advance(t.full((e.n,), -1, dtype=t.int64, device=e.device), t.zeros((e.n,), dtype=t.float32, device=e.device), ~lead)
        acts = actions.to(device=e.device, dtype=t.int64)
        ids = t.where(self._terminated | ~self._hero_turn(), t.full_like(acts, -1), acts)
        reward = t.zeros((e.n,), dtype=t.float32, device=e.device)
        reward, terminated = self._advance(ids, reward, t.zeros((e.n,), dtype=t.bool, device=e.device))
        self._terminated = terminated
        info = {"scores": e.scores(), "ranks": e.ranks()}
        return self._obs(), reward, terminated, t.zeros_like(terminated), info
