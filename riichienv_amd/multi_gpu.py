"""In-process multi-GPU host (north_star: "a host ... shards by index across the 8 GPUs of one node"): ONE process, one handle and one
host thread per device, games sharded by global index like the ranks of bench.py (riichienv_amd.shard) - no collective, no data moves
between devices.  The C-ABI is thread-compatible (a handle per thread, no device globals besides read-only tables) and ctypes releases
the GIL inside every call, so the shards' kernels run concurrently.  `devices` may name a device more than once (two shards on one
GPU: how the class is tested on a 1-GPU box); results never depend on the split - seeds and policy keys are functions of the global
game index (RmjConfig.game_offset).

No scaling curve has been measured with it (no multi-GPU node was available to the build): it is the functional counterpart of
`bench.py --gpus N`, which the driver times."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import vecenv


class MultiGpuVecEnv:
    def __init__(self, n_games, devices=None, **kw):
        if devices is None:
            devices = list(range(max(1, vecenv.load_lib().rmj_device_count())))
        self.devices = [int(d) for d in devices]
        k = len(self.devices)
        if k < 1 or n_games % k:
            raise ValueError("n_games must be a positive multiple of the number of shards")
        self.n, self.k, self.per = int(n_games), k, int(n_games) // k
        base = int(kw.pop("game_offset", 0))
        seeds = kw.pop("seeds", None)
        self.pool = ThreadPoolExecutor(max_workers=k, thread_name_prefix="rmj-shard")
        # every shard is created, driven and destroyed by ITS thread (hipSetDevice is per thread; one handle per thread)
        self._workers = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"rmj-dev{d}-{i}") for i, d in enumerate(self.devices)]

        def make(i):
            sub = None if seeds is None else np.asarray(seeds, dtype=np.uint64)[i * self.per: (i + 1) * self.per]
            return vecenv.VecRiichiEnv(self.per, device=self.devices[i], game_offset=base + i * self.per, seeds=sub, **kw)
        self.shards = self._map(lambda i, _e: make(i), with_env=False)
        self.game_mode = self.shards[0].game_mode

    # ---- plumbing
    def _map(self, fn, with_env=True):
        """fn(i, shard) on every shard's own thread, all at once; the list of results in shard order"""
        futs = [w.submit(fn, i, self.shards[i] if with_env else None) for i, w in enumerate(self._workers)]
        return [f.result() for f in futs]

    def _cat(self, parts):
        return np.concatenate(parts, axis=0)

    def owner(self, g):
        return int(g) // self.per, int(g) % self.per

    # ---- the VecRiichiEnv surface, fanned out
    def reset(self, **kw):
        def one(i, e):
            sl = slice(i * self.per, (i + 1) * self.per)
            e.reset(**{k: (None if v is None else np.asarray(v)[sl]) for k, v in kw.items()})
        self._map(one)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.uint64).reshape(self.n, 4)
        self._map(lambda i, e: e.step(a[i * self.per: (i + 1) * self.per]))

    def step_random(self, policy_seed, n_steps=1, auto_reset=False):
        self._map(lambda i, e: (e.step_random(policy_seed, n_steps, auto_reset=auto_reset), e.total_steps()))   # (total_steps: waits for the shard)

    def step_greedy(self, policy_seed, n_steps=1, auto_reset=False, call_rate_256=64):
        self._map(lambda i, e: (e.step_greedy(policy_seed, n_steps, auto_reset=auto_reset, call_rate_256=call_rate_256), e.total_steps()))

    def random_actions(self, policy_seed):
        return self._cat(self._map(lambda i, e: e.random_actions(policy_seed)))

    def status(self):
        parts = self._map(lambda i, e: e.status())
        return tuple(self._cat([p[j] for p in parts]) for j in range(3))

    def done(self):
        return self.status()[2].astype(bool)

    def legal(self):
        parts = self._map(lambda i, e: e.legal())
        return self._cat([p[0] for p in parts]), self._cat([p[1] for p in parts])

    def mask(self):
        return self._cat(self._map(lambda i, e: e.mask()))

    def waits(self):
        return self._cat(self._map(lambda i, e: e.waits()))

    def scores(self):
        return self._cat(self._map(lambda i, e: e.scores()))

    def ranks(self):
        return self._cat(self._map(lambda i, e: e.ranks()))

    def step_counts(self):
        return self._cat(self._map(lambda i, e: e.step_counts()))

    def total_steps(self):
        return int(sum(self._map(lambda i, e: e.total_steps())))

    def event_counts(self):
        return self._cat(self._map(lambda i, e: e.event_counts()))

    def peek(self, g):
        i, l = self.owner(g)
        return self._workers[i].submit(lambda: self.shards[i].peek(l)).result()

    def mjai_log(self, g, seat=-1):
        i, l = self.owner(g)
        return self._workers[i].submit(lambda: self.shards[i].mjai_log(l, seat)).result()

    def drain_logs(self, seat=-1):
        """the MJAI strings every game slot logged since the last drain, all shards at once (rmj_drain_format per shard); the
        shards' running cursors are stream positions and survive auto-reset restarts (VecRiichiEnv.drain_logs)"""
        return [log for part in self._map(lambda i, e: e.drain_logs(seat=seat)) for log in part]

    def mjai_logs(self, seat=-1):
        return [log for part in self._map(lambda i, e: e.mjai_logs(seat=seat)) for log in part]

    def close(self):
        if getattr(self, "shards", None):
            self._map(lambda i, e: e.close())
            self.shards = None
        for w in self._workers:
            w.shutdown()
        self.pool.shutdown()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
