#!/usr/bin/env python3
"""The host-buffer path of a drop-in Python agent loop: per step, fetch what an agent reads, choose on the host
(vectorised uniform choice), and hand packed actions back (rmj_step: H2D + one validating step launch).  PCIe-inclusive
env.step/s at several batch sizes, beside the device-resident rates of bench.py.  Two ways down: the full [n][4][64] list slab
(rmj_get_legal, round 2) and the compact rows of the seats that are to act (rmj_get_legal_compact, pinned staging, round 3)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riichienv_amd import abi, vecenv  # noqa: E402

out = {}
for n in (4096, 65536):
    # ---- round 3: compact rows
    env = vecenv.VecRiichiEnv(n, game_mode=2, seed=0, event_ring=64)
    env.reset()
    env.step_random(1, 100, auto_reset=True)
    rng = np.random.default_rng(0)
    K = 30
    env.legal_compact()
    t_get = t_pick = t_step = 0.0
    s0 = env.total_steps()
    t0 = time.perf_counter()
    for _ in range(K):
        a = time.perf_counter()
        idx, off, ent = env.legal_compact()
        b = time.perf_counter()
        ln = (off[1:] - off[:-1]).astype(np.int64)
        pick = off[:-1].astype(np.int64) + (rng.random(len(idx)) * ln).astype(np.int64)
        acts = np.full(n * 4, abi.NO_ACTION, np.uint64)
        acts[idx] = ent[pick]
        c = time.perf_counter()
        env.step(acts.reshape(n, 4))
        env.L.rmj_sync(env.h)
        d = time.perf_counter()
        t_get += b - a
        t_pick += c - b
        t_step += d - c
    t1 = time.perf_counter()
    steps = env.total_steps() - s0
    out[f"{n}_compact"] = {"env_steps_per_s": steps / (t1 - t0), "ms_per_iteration": (t1 - t0) / K * 1e3,
                           "fetch_compact_lists_ms": t_get / K * 1e3, "host_choice_ms": t_pick / K * 1e3, "upload_and_step_ms": t_step / K * 1e3,
                           "bytes_down_per_game": (len(idx) * 8 + len(ent) * 8 + 8) / n, "bytes_up_per_game": 32}
    env.close()
    env = vecenv.VecRiichiEnv(n, game_mode=2, seed=0, event_ring=64)
    env.reset()
    env.step_random(1, 100, auto_reset=True)
    rng = np.random.default_rng(0)
    K = 30
    t_get = t_pick = t_step = 0.0
    s0 = env.total_steps()
    t0 = time.perf_counter()
    for _ in range(K):
        a = time.perf_counter()
        act, ph, dn = env.status()
        legal, cnt = env.legal()
        b = time.perf_counter()
        acts = np.full((n, 4), abi.NO_ACTION, np.uint64)
        live = ((act[:, None] >> np.arange(4)[None, :]) & 1).astype(bool) & (cnt > 0) & (dn[:, None] == 0)
        pick = (rng.random((n, 4)) * np.maximum(cnt, 1)).astype(np.int64)
        chosen = np.take_along_axis(legal, pick[:, :, None], axis=2)[:, :, 0]
        acts[live] = chosen[live]
        c = time.perf_counter()
        env.step(acts)
        env.L.rmj_sync(env.h)
        d = time.perf_counter()
        t_get += b - a
        t_pick += c - b
        t_step += d - c
    t1 = time.perf_counter()
    steps = env.total_steps() - s0
    out[str(n)] = {"env_steps_per_s": steps / (t1 - t0), "ms_per_iteration": (t1 - t0) / K * 1e3,
                   "fetch_status_and_lists_ms": t_get / K * 1e3, "host_choice_ms": t_pick / K * 1e3,
                   "upload_and_step_ms": t_step / K * 1e3,
                   "bytes_down_per_game": 4 * 64 * 8 + 4 + 3, "bytes_up_per_game": 32}
    env.close()
print(json.dumps(out))
