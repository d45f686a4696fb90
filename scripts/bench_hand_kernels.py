#!/usr/bin/env python3
"""Kernel-gate micro-benchmarks of rows A7-A11 (the groups of the reference's
riichienv-core/benches/agari_bench.rs:142-376: is_agari / is_tenpai, find_divisions + calc via HandEvaluator, calculate_score,
shanten / ukeire): hands/s of every hand-math kernel on device-resident inputs (rmj_bench_hand_kernel: HIP events around the
launches only), the fraction of the HBM roofline at the survey's 50 B per hand evaluation (34 B counts in + 16 B result out)
and at the bytes the entry point really moves per hand (io_bytes).  roofline_frac is scored on min(io_bytes, 50) - rmj_calculate_score reads 9 B
and writes 16 B, and every working set here (26-160 MB) stays in the 256 MB Infinity Cache between the timed launches: `cache_resident`,
and the oracle beside it - one thread, and one worker process per hardware thread.

Inputs: the 1 218 fixture hands of tests/golden/agari_{4p,3p}.json tiled to 2^20 + 2^20 uniformly random
13 / 14-tile hands (seed 1).  The CPU legs run FIRST (they start worker processes; a process that has touched the GPU must not).
usage: python scripts/bench_hand_kernels.py [log2 of the hands per set, default 20] [cpu seconds per leg, default 3] [nocpu]
(nocpu: GPU legs only - for `rocprofv3 --kernel-trace --stats -- python3 scripts/bench_hand_kernels.py 20 0 nocpu`)"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import abi  # noqa: E402

HBM_PEAK = 8.0e12
B_HAND = 50
KERNELS = [("rmj_eval_hands (HandEvaluator.calc + waits)", 0), ("rmj_agari_counts (is_agari, is_tenpai, waits)", 1), ("rmj_shanten", 2),
           ("rmj_effective_tiles", 3), ("rmj_best_ukeire", 4), ("rmj_calculate_score", 5)]


def fixture_cases():
    out = []
    for name in ("agari_4p.json", "agari_3p.json"):
        with open(os.path.join(ROOT, "tests", "golden", name)) as f:
            out += [abi.hand_case_from_fixture(c) for c in json.load(f)["cases"]]
    return out


def counts_of_cases(cases):
    cnt = np.zeros((len(cases), 34), np.uint8)
    for i, c in enumerate(cases):
        for t in list(c.tiles[: c.n_tiles]):
            cnt[i, t // 4] += 1
    return cnt


def random_counts(n, seed=1):
    rng = np.random.default_rng(seed)
    size = np.where(np.arange(n) % 2 == 0, 13, 14)
    keys = rng.random((n, 136), dtype=np.float32)
    order = np.argpartition(keys, 14, axis=1)[:, :14].astype(np.int64) // 4   # 14 distinct physical tiles -> types
    cnt = np.zeros((n, 34), np.uint8)
    rows = np.repeat(np.arange(n), 14)
    keep = (np.tile(np.arange(14), n) < np.repeat(size, 14))
    np.add.at(cnt, (rows[keep], order.reshape(-1)[keep]), 1)
    return cnt


def tile_to(arr, n):
    reps = (n + len(arr) - 1) // len(arr)
    return np.ascontiguousarray(np.concatenate([arr] * reps)[:n])


_WORKER = """
import sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle
which, secs, seed = int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
sys.path.insert(0, sys.argv[1] + '/scripts')
import bench_hand_kernels as B
n, t = B.cpu_leg(which, secs, seed)
print(n, t)
"""


def cpu_leg(which, secs, seed=0):
    """hands evaluated by the oracle in ~secs seconds on this thread: (hands, seconds)"""
    from oracle import oracle

    fx = fixture_cases()
    m = 4096 if which in (0, 1, 5) else (256 if which == 2 else 32)   # (the oracle's shanten / ukeire are plain enumeration: up to seconds per hand)
    cnt = np.concatenate([tile_to(counts_of_cases(fx), m // 2), random_counts(m // 2, seed + 1)])
    cases = (fx * (m // len(fx) + 1))[:m]
    vis = np.minimum(cnt, 1)
    han = (np.arange(m) % 13 + 1).astype(np.uint8)
    fu = ((np.arange(m) % 9 + 2) * 10).astype(np.uint8)
    one = np.ones(m, np.uint8)
    calls = {0: lambda: oracle.eval_hands(cases), 1: lambda: oracle.agari_counts(cnt), 2: lambda: oracle.shanten(cnt),
             3: lambda: oracle.effective_tiles(cnt), 4: lambda: oracle.best_ukeire(cnt, vis),
             5: lambda: oracle.calculate_score(han, fu, one, one, np.zeros(m, np.uint32), one * 4)}
    f = calls[which]
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        f()
        done += m
    return done, time.perf_counter() - t0


def cpu_all(secs):
    from bench import usable_cores
    cores = usable_cores()                   # the container's CPU allowance, not the host's hardware threads
    rows = {}
    for _, which in KERNELS:
        n1, t1 = cpu_leg(which, secs)
        procs = [subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, str(which), str(secs), str(i)], stdout=subprocess.PIPE, text=True)
                 for i in range(cores)]
        outs = [p.communicate()[0].split() for p in procs]
        tot = sum(int(o[0]) for o in outs if len(o) == 2)
        mx = max(float(o[1]) for o in outs if len(o) == 2)
        rows[which] = {"one_thread_hands_per_s": n1 / t1, "all_cores_hands_per_s": tot / mx, "cores": cores}
    return rows


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    if len(sys.argv) > 3 and sys.argv[3] == "nocpu":   # under rocprofv3 (its library has touched the GPU already: no worker processes)
        cpu = {w: {"one_thread_hands_per_s": 0.0, "all_cores_hands_per_s": 0.0, "cores": 0} for _, w in KERNELS}
    else:
        cpu = cpu_all(secs)                  # before anything touches the GPU
    from riichienv_amd import vecenv
    L = vecenv.load_lib()
    n = 1 << lg
    fx = fixture_cases()
    sets = {"fixtures tiled": (tile_to(counts_of_cases(fx), n), (fx * (n // len(fx) + 1))[:n]), "random hands": (random_counts(n, 1), None)}
    out = {"hands_per_set": n, "bytes_per_hand": B_HAND, "hbm_peak_GBps": HBM_PEAK / 1e9, "rows": []}
    for sname, (cnt, cases) in sets.items():
        vis = np.ascontiguousarray(np.minimum(cnt, 1))
        for kname, which in KERNELS:
            if which == 0:
                if cases is None:
                    continue
                arr = (abi.HandCase * n)(*cases)
                a, b = C.addressof(arr), None
            elif which == 5:
                han = (np.arange(n) % 13 + 1).astype(np.uint8)
                fu = ((np.arange(n) % 9 + 2) * 10).astype(np.uint8)
                blob = np.ascontiguousarray(np.concatenate([han, fu, np.ones(n, np.uint8), np.arange(n, dtype=np.uint8) % 2, np.full(n, 4, np.uint8)]))
                hb = np.zeros(n, np.uint32)
                a, b = blob.ctypes.data, hb.ctypes.data
            else:
                a, b = cnt.ctypes.data, (vis.ctypes.data if which == 4 else None)
            ms = C.c_double()
            vecenv._chk(L.rmj_bench_hand_kernel(0, which, a, b, n, 0, 20 if which != 0 else 5, C.byref(ms)))
            rate = n / (ms.value * 1e-3)
            io = {0: C.sizeof(abi.HandCase) + C.sizeof(abi.HandResult), 1: 34 + 10, 2: 34 + 1, 3: 34 + 4, 4: 68 + 4, 5: 9 + 16}[which]
            # the roofline fraction is scored on the bytes the entry point really moves (never more than the survey's nominal 50 B), and a
            # working set that fits the 256 MB Infinity Cache is labelled cache-resident: repeated launches over it are served by the MALL, so the
            # figure is a fraction of the HBM PEAK RATE, not evidence of HBM traffic (a fraction above 1 would only say "not HBM")
            bytes_scored = min(io, B_HAND)
            ws_mb = n * io / 1e6
            row = {"kernel": kname, "inputs": sname, "ms_per_launch": ms.value, "hands_per_s": rate, "roofline_frac": rate * bytes_scored / HBM_PEAK,
                   "roofline_bytes_per_hand": bytes_scored, "nominal_50B_frac": rate * B_HAND / HBM_PEAK, "cache_resident": bool(ws_mb < 256.0),
                   "io_bytes_per_hand": io, "io_frac_of_hbm_peak": rate * io / HBM_PEAK, "working_set_MB": ws_mb,
                   "cpu_one_thread_hands_per_s": cpu[which]["one_thread_hands_per_s"], "cpu_all_cores_hands_per_s": cpu[which]["all_cores_hands_per_s"],
                   "cpu_cores": cpu[which]["cores"]}
            out["rows"].append(row)
            print(f"{kname:48s} {sname:15s} {ms.value:8.3f} ms  {rate / 1e9:7.3f} G hands/s  {100 * row['roofline_frac']:5.2f} % of 8 TB/s at the {bytes_scored} B/hand scored ({100 * row['io_frac_of_hbm_peak']:5.2f} % at the {io} B it moves; {ws_mb:.0f} MB working set, {'cache-resident' if row['cache_resident'] else 'HBM'}) | "
                  f"oracle {row['cpu_one_thread_hands_per_s'] / 1e6:7.3f} M/s one thread, {row['cpu_all_cores_hands_per_s'] / 1e6:8.2f} M/s on {row['cpu_cores']} threads", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
