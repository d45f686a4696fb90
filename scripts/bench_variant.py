#!/usr/bin/env python3
"""Time k_step on an alternative build of the library (tuning experiments): bench_variant.py <lib.so> [mode]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.abspath(sys.argv[1])
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
env = vecenv.VecRiichiEnv(65536, game_mode=mode, seed=0)
env.reset()
env.step_random(0xC0FFEE, 200, auto_reset=True)
r = env.bench_rollout(0xC0FFEE, 0, 1000)
print(f"{os.path.basename(sys.argv[1])} mode {mode}: kernel {r.step_kernel_ms:.4f} ms, {r.env_steps / r.total_ms / 1e3:.1f} M env.step/s")
