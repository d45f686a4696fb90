#!/usr/bin/env python3
"""Time the fused rollout on an alternative build of the library (tuning experiments): bench_variant.py <lib.so> [mode] [policy]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.abspath(sys.argv[1])
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
policy = sys.argv[3] if len(sys.argv) > 3 else "random"
env = vecenv.VecRiichiEnv(65536, game_mode=mode, seed=0)
env.reset()
if policy == "random":
    env.step_random(0xC0FFEE, 600, auto_reset=True)
    r = env.bench_rollout(0xC0FFEE, 0, 1000)
    ms, steps = r.total_ms, r.env_steps
else:
    env.step_greedy(0xC0FFEE, 1500, auto_reset=True, call_rate_256=64)
    s0 = env.total_steps()
    r = env.time_rollout_greedy(0xC0FFEE, 1000, 64)
    ms, steps = r.total_ms, env.total_steps() - s0
print(f"{os.path.basename(sys.argv[1])} mode {mode} {policy}: {ms / 1000 * 1e3:.1f} us per step, {steps / ms / 1e3:.1f} M env.step/s", flush=True)
