#!/usr/bin/env python3
"""Why rows of k_step4 leave tier 0: census of the R4BAIL sites over a steady-state rollout (accounting build)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("RMJ_CENSUS_FUSED") != "1":   # default: one launch per step (k_step4<false>, the rich tier 0); RMJ_CENSUS_FUSED=1: the fused rollout kernels (lean tier 0 under the RandomAgent)
    os.environ["RMJ_STEP4"] = "1"
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", os.environ.get("RMJ_CENSUS_LIB", "libriichi_mi355x_cuts.so"))   # or a -DRMJ_CENSUS build: the shipped code + counters
L = vecenv.load_lib()
if hasattr(L, "rmj_prof_set_cut"):
    L.rmj_prof_set_cut.argtypes = [C.c_int, C.c_int, C.c_int]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
policy = sys.argv[2] if len(sys.argv) > 2 else "random"      # random | greedy
rate = int(sys.argv[3]) if len(sys.argv) > 3 else 64          # greedy: call rate / 256
env = vecenv.VecRiichiEnv(65536, game_mode=mode, seed=0)
env.reset()
if hasattr(L, "rmj_prof_set_cut"):
    L.rmj_prof_set_cut(-1, -1, -1)
run = (lambda k: env.step_random(0xC0FFEE, k, auto_reset=True)) if policy == "random" else \
      (lambda k: env.step_greedy(0xC0FFEE, k, auto_reset=True, call_rate_256=rate))
run(int(os.environ.get("RMJ_CENSUS_WARM", "800")))
f0 = env.total_full_path()
buf = (C.c_uint32 * 32)()
L.rmj_prof_bail_census(buf, 1)
s0 = env.total_steps()
run(200)
steps = env.total_steps() - s0
L.rmj_prof_bail_census(buf, 0)
print(f"full-path steps by the games' own counters: {env.total_full_path() - f0} of {steps} game-steps")
NAMES = {1: "event stage full", 2: "refill: possible wait (tables <= 0)", 3: "refill(hist): possible wait", 4: "sufuurenta", 5: "suukansansen", 6: "suucha riichi",
         7: "exhaustive draw", 8: "pending kan dora at discard", 9: "ron possible on the discard", 10: "claim list too long", 11: "actor holds 13 / riichi stage",
         12: "tsumo check: odd hand", 13: "complete hand (tsumo)", 14: "riichi possible (14-tile shanten <= 0)", 15: "ankan available", 16: "ankan in riichi",
         17: "act list too long", 18: "finished game (restart)", 19: "no action / no tile", 20: "discard: tile not found / unsorted", 21: "kita: precondition",
         22: "kita: ron possible", 23: "riichi / kan / tsumo / kyushu action", 24: "ron settlement", 25: "daiminkan response", 26: "pending kan resolves", 27: "action from a seat that is not to act", 28: "illegal action", 29: "kan without a replacement tile", 30: "kan: a seat could rob the tile", 31: "dealt by r4_round_end, first list by the full path", 0: "(no site recorded)"}
tot = sum(buf)
print(f"mode {mode}, policy {policy}" + (f" (call rate {rate}/256)" if policy != "random" else "") + f": {steps} game-steps, {tot} bails ({100.0 * tot / steps:.2f} %)")
for i in sorted(range(32), key=lambda k: -buf[k]):
    if buf[i]:
        print(f"  {buf[i]:8d}  {100.0 * buf[i] / steps:5.2f} %  {NAMES.get(i, i)}")
