#!/usr/bin/env python3
"""The driver's 20-step window on an alternative build: bench_variant_short.py <lib.so> [mode] [steps] - 6 000 pre-roll steps, 5 warm-up, then `steps`
timed by rmj_time_rollout (HIP events), best and mean of 5 windows"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.abspath(sys.argv[1])
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
env = vecenv.VecRiichiEnv(65536, game_mode=mode, seed=0)
env.reset()
env.step_random(0xC0FFEE, 6000, auto_reset=True)
env.step_random(0xC0FFEE, 5, auto_reset=True)
ms = []
for _ in range(5):
    ms.append(env.time_rollout(0xC0FFEE, steps).total_ms)
    env.step_random(0xC0FFEE, 137, auto_reset=True)
print(f"{os.path.basename(sys.argv[1])} mode {mode} {steps} steps: first window {ms[0]:.4f} ms = {65536 * steps / ms[0] / 1e3:.1f} M env.step/s, "
      f"mean of 5 windows {sum(ms) / 5:.4f} ms = {65536 * steps * 5 / sum(ms) / 1e3:.1f} M", flush=True)
