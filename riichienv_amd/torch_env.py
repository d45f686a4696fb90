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

    def __init__(self, n_games, game_mode=2, seed=0, device=0, extended=False, skip_mjai_logging=True, share_stream=True, pad_rows=False, **kw):
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
        # pad_rows: every (game, seat) row of the 74-channel tensor padded to a multiple of 256 B (rmj_set_encode_row_stride): the
        # acting seats' rows are written 1.3-1.4 x faster; obs() then returns a strided view [n, 4, 74, W] of the padded buffer
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

    def sample_ids(self, logits=None, seed=0, index=None):
        """One id per acting seat drawn from softmax(logits) over the seat's legal ids, -1 elsewhere, by ONE kernel of the
        library (rmj_sample_ids_device: Gumbel-max on the resident mask slab) - no torch indexing / multinomial in the loop.
        logits: float32 [n, 4, A'] with A' >= 82 (60 in 3P) on this device, or None for the uniform policy; with `index`
        (the second result of obs_compact(): game * 4 + seat per row) logits are the compact rows [k, A'] a policy computed
        from the compact observation batch, scattered here into the [n, 4, A'] layout (rows of seats that do not act are never
        read by the sampler)."""
        t = self.torch
        if not hasattr(self, "_ids"):
            self._ids = t.full((self.n, 4), -1, dtype=t.int32, device=self.device)
        if logits is not None and index is not None:
            a = int(logits.shape[-1])
            if getattr(self, "_full_logits", None) is None or self._full_logits.shape[-1] != a:
                self._full_logits = t.zeros((self.n * 4, a), dtype=t.float32, device=self.device)
            k = int(index.shape[0])
            self._full_logits[index.to(t.int64)] = logits.reshape(-1, a)[:k].to(t.float32)
            logits = self._full_logits.view(self.n, 4, a)
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
