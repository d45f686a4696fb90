#!/usr/bin/env python3
"""env.step/s and full-path share of fused rollouts under the greedy device policy (rmj_step_greedy) beside the RandomAgent's.
usage: python scripts/r03_greedy_bench.py [games] [mode] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from riichienv_amd import vecenv
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    for name, rate in (("random", None), ("greedy, no calls", 0), ("greedy, calls 25 %", 64)):
        env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=0, event_ring=64)
        env.reset()
        run = (lambda k: env.step_random(0xC0FFEE, k, auto_reset=True)) if rate is None else \
              (lambda k: env.step_greedy(0xC0FFEE, k, auto_reset=True, call_rate_256=rate))
        run(1500)
        env.sync()
        s0, f0 = env.total_steps(), env.total_full_path()
        t0 = time.perf_counter()
        run(steps)
        env.sync()
        dt = time.perf_counter() - t0
        s1, f1 = env.total_steps(), env.total_full_path()
        print(f"{name:20s}: {(s1 - s0) / dt / 1e6:8.1f} M env.step/s, {dt / steps * 1e6:7.1f} us per step of all games, "
              f"full-path share {(f1 - f0) / max(1, s1 - s0):.4f}", flush=True)
        env.close()


if __name__ == "__main__":
    main()
