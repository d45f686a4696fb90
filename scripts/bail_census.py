#!/usr/bin/env python3
"""Why rows of k_step4 leave tier 0: census of the R4BAIL sites over a steady-state rollout (accounting build)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RMJ_STEP4"] = "1"
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", "libriichi_mi355x_cuts.so")
L = vecenv.load_lib()
L.rmj_prof_set_cut.argtypes = [C.c_int, C.c_int, C.c_int]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
env = vecenv.VecRiichiEnv(65536, game_mode=mode, seed=0)
env.reset()
L.rmj_prof_set_cut(-1, -1, -1)
env.step_random(0xC0FFEE, 400, auto_reset=True)
buf = (C.c_uint32 * 32)()
L.rmj_prof_bail_census(buf, 1)
s0 = env.total_steps()
env.step_random(0xC0FFEE, 200, auto_reset=True)
steps = env.total_steps() - s0
L.rmj_prof_bail_census(buf, 0)
NAMES = {1: "event stage full", 2: "refill: possible wait (tables <= 0)", 3: "refill(hist): possible wait", 4: "sufuurenta", 5: "suukansansen", 6: "suucha riichi",
         7: "exhaustive draw", 8: "pending kan dora at discard", 9: "ron possible on the discard", 10: "claim list too long", 11: "actor holds 13 / riichi stage",
         12: "tsumo check: odd hand", 13: "complete hand (tsumo)", 14: "riichi possible (14-tile shanten <= 0)", 15: "ankan available", 16: "ankan in riichi",
         17: "act list too long", 18: "finished game (restart)", 19: "no action / no tile", 20: "discard: tile not found / unsorted", 21: "kita: precondition",
         22: "kita: ron possible", 23: "riichi / kan / tsumo / kyushu action", 24: "ron settlement", 25: "daiminkan response", 26: "pending kan resolves"}
tot = sum(buf)
print(f"mode {mode}: {steps} game-steps, {tot} bails ({100.0 * tot / steps:.2f} %)")
for i in sorted(range(32), key=lambda k: -buf[k]):
    if buf[i]:
        print(f"  {buf[i]:8d}  {100.0 * buf[i] / steps:5.2f} %  {NAMES.get(i, i)}")
