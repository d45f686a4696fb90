#!/usr/bin/env python3
"""Debug aid for the accounting build (-DRMJ_CUTS): greedy rollouts of growing length, then single cut launches (mark 43)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", sys.argv[1])
L = vecenv.load_lib()
if hasattr(L, "rmj_prof_set_cut"):
    L.rmj_prof_set_cut.argtypes = [C.c_int, C.c_int, C.c_int]
    L.rmj_prof_set_cut(-1, -1, -1)
env = vecenv.VecRiichiEnv(int(sys.argv[2]), game_mode=2, seed=0)
env.reset()
for k in (1, 2, 8, 40, 120):
    t0 = time.time()
    env.step_greedy(0xC0FFEE, k, auto_reset=True, call_rate_256=64)
    n = env.total_steps()
    print("rollout", k, "steps ok", round(time.time() - t0, 3), "s, total", n, flush=True)
env.set_rollout_streams(1)
for cut in [int(x) for x in sys.argv[3:]]:
    e = env.clone()
    e.set_rollout_streams(1)
    L.rmj_prof_set_cut(cut, -1, -1)
    e.step_greedy(0xC0FFEE, 1, auto_reset=True, call_rate_256=64)
    e.total_steps()
    L.rmj_prof_set_cut(-1, -1, -1)
    print("cut", cut, "ok", flush=True)
    e.close()
