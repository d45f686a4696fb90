#!/usr/bin/env python3
"""Timeline of ONE launch of the ticket rollout (k_step4_queue) on the 100 MHz clock: when waves enter, how long a ticket's pick-up
(atomic, wait for the previous chunk, acquire) and its steps take round by round, and how the launch ends (how much slot time idles
behind waves that found no ticket).  Needs the timeline build: scripts/build_qtl.sh (-DRMJ_QTL); never the shipped library.
usage: python scripts/timeline_queue.py [steps=20] [games=65536] [mode=2]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", "libriichi_mi355x_qtl.so")
ROW = 256


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    games = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    L = vecenv.load_lib()
    waves = 8192
    buf = np.zeros((waves, ROW), dtype=np.uint64)
    L.rmj_qtl_fetch(buf.ctypes.data_as(C.c_void_p), waves)
    env = vecenv.VecRiichiEnv(games, game_mode=mode, seed=0)
    env.reset()
    env.step_random(0xC0FFEE, 3000, auto_reset=True)
    L.rmj_qtl_fetch(buf.ctypes.data_as(C.c_void_p), waves)
    for rep in range(3):
        r = env.time_rollout(0xC0FFEE, steps)
        L.rmj_qtl_fetch(buf.ctypes.data_as(C.c_void_p), waves)
        ran = buf[:, 1] > 0
        b = buf[ran].astype(np.int64)
        t_in, t_out, nt = b[:, 0], b[:, 1], b[:, 2]
        z = t_in.min()
        us = lambda x: x / 100.0  # noqa: E731
        span = us(t_out.max() - z)
        print(f"== rollout of {steps} steps, {games} games: HIP events {r.total_ms * 1e3:.1f} us, queued {int(r.queued)}; {ran.sum()} waves ran, "
              f"span first entry -> last exit {span:.1f} us")
        print(f"   wave entry: p50 {us(np.median(t_in) - z):.1f}  p99 {us(np.percentile(t_in, 99) - z):.1f}  max {us(t_in.max() - z):.1f} us;  "
              f"tickets per wave: min {nt.min()} p50 {int(np.median(nt))} max {nt.max()}")
        print(f"   wave exit (before the last exit): p10 {us(t_out.max() - np.percentile(t_out, 10)):.1f}  p50 {us(t_out.max() - np.median(t_out)):.1f}  "
              f"p90 {us(t_out.max() - np.percentile(t_out, 90)):.1f} us;  idle slot time behind exits = "
              f"{(t_out.max() - t_out).sum() / max(1, (t_out.max() - z) * ran.sum()):.3f} of the launch")
        xcd = (b[:, 4] >> 32) & 7
        print("   waves per XCD:", [int((xcd == x).sum()) for x in range(8)])
        hw = b[:, 3]
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        work = np.array([sum(int(b[w, 7 + 4 * k] - b[w, 6 + 4 * k]) for k in range(int(nt[w]))) for w in range(len(b))]) / 100.0
        rate = nt / np.maximum(work, 1e-9)      # tickets per us of stepping
        print(f"   tickets per busy us, by wave: p1 {np.percentile(rate, 1):.5f} p10 {np.percentile(rate, 10):.5f} p50 {np.median(rate):.5f} p90 {np.percentile(rate, 90):.5f} p99 {np.percentile(rate, 99):.5f}")
        for name, key in (("xcd", xcd), ("se", se), ("sh", sh), ("cu", cu), ("simd", simd)):
            vals = sorted(set(key.tolist()))
            print(f"   mean tickets per wave by {name}: " + "  ".join(f"{v}:{nt[key == v].mean():.1f}({int((key == v).sum())})" for v in vals))
        place = xcd * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd          # one SIMD
        ids, inv = np.unique(place, return_inverse=True)
        per_simd_waves = np.bincount(inv)
        per_simd_tickets = np.bincount(inv, weights=nt)
        print(f"   SIMDs seen: {len(ids)}; waves per SIMD: min {per_simd_waves.min()} max {per_simd_waves.max()} hist {np.bincount(per_simd_waves).tolist()}")
        print(f"   tickets per SIMD: p1 {np.percentile(per_simd_tickets, 1):.0f} p10 {np.percentile(per_simd_tickets, 10):.0f} p50 {np.median(per_simd_tickets):.0f} "
              f"p90 {np.percentile(per_simd_tickets, 90):.0f} p99 {np.percentile(per_simd_tickets, 99):.0f}")
        for wv in sorted(set(per_simd_waves.tolist())):
            m = per_simd_waves == wv
            print(f"     SIMDs with {wv} waves: {int(m.sum())}, tickets per SIMD mean {per_simd_tickets[m].mean():.1f}, per wave {per_simd_tickets[m].mean() / wv:.1f}")
        gaps = np.array([sum(int(b[w, 6 + 4 * k] - (b[w, 7 + 4 * (k - 1)] if k else b[w, 0])) for k in range(int(nt[w]))) for w in range(len(b))]) / 100.0
        print(f"   per wave: stepping {work.mean():.1f} us, between tickets (take + wait + acquire) {gaps.mean():.1f} us, behind its last ticket {us((t_out.max() - t_out).mean()):.1f} us "
              f"of a {span:.1f} us span; wave-steps {steps * (games // 4)}: {work.sum() / (steps * (games // 4)):.2f} slot-us per wave-step")
        calls = sum(int(((b[nt > k, 4 + 4 * k] >> 40) & 0xFF).sum()) for k in range(int(nt.max())))
        rows = sum(int(((b[nt > k, 4 + 4 * k] >> 48) & 0xFFFF).sum()) for k in range(int(nt.max())))
        print(f"   calls of the step function: {calls} = {calls / nt.sum():.2f} per ticket, {calls / (steps * (games // 4)):.3f} per (quad, step); live rows per call {rows / max(calls, 1):.3f} of 4; "
              f"game-steps per live row and call {steps * games / max(rows, 1):.3f}; slot-us per call {work.sum() / max(calls, 1):.2f}")
        for k in range(int(nt.max())):
            has = nt > k
            tk, tb, te = b[has, 5 + 4 * k], b[has, 6 + 4 * k], b[has, 7 + 4 * k]
            prev_end = b[has, 7 + 4 * (k - 1)] if k else b[has, 0]
            print(f"   round {k:2d}: {has.sum():5d} waves | take after previous end p50 {us(np.median(tk - prev_end)):6.2f} p99 {us(np.percentile(tk - prev_end, 99)):6.2f} | "
                  f"wait+acquire p50 {us(np.median(tb - tk)):6.2f} p99 {us(np.percentile(tb - tk, 99)):7.2f} | steps mean {us((te - tb).mean()):7.1f} p10 {us(np.percentile(te - tb, 10)):7.1f} p50 "
                  f"{us(np.median(te - tb)):7.1f} p90 {us(np.percentile(te - tb, 90)):7.1f} max {us((te - tb).max()):7.1f} | ends at p50 {us(np.median(te) - z):7.1f} max {us(te.max() - z):7.1f}")
    env.close()


if __name__ == "__main__":
    main()
