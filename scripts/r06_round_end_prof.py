#!/usr/bin/env python3
"""round 6: where a round end's time goes in the fused rollout (build: -DRMJ_RE_PROF, riichienv_amd/libriichi_mi355x_reprof.so; never the shipped library).
Ticks of the 100 MHz clock around r4_round_end and step4_pass2, summed over all waves of ONE 1 000-step rollout of 65 536 games."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import abi, vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", "libriichi_mi355x_reprof.so")
L = vecenv.load_lib()
env = vecenv.VecRiichiEnv(65536, game_mode=int(sys.argv[1]) if len(sys.argv) > 1 else 2, seed=0, rule_bits=abi.RULE_TENHOU, event_ring=64)
env.reset()
env.step_random(0xC0FFEE, 3000, auto_reset=True)
buf = (C.c_ulonglong * 24)()
env.sync()                       # (the counters are zeroed by a copy outside the handle's stream)
L.rmj_debug_re_prof(buf, 1)
steps = 1000
r = env.time_rollout(0xC0FFEE, steps)
L.rmj_debug_re_prof(buf, 0)
v = list(buf)
slots = 6144
print(f"rollout {steps} steps: {r.total_ms:.2f} ms on {slots} wave slots = {r.total_ms * 1e3 * slots / 1e6:.2f} slot-seconds... per wave {r.total_ms * 1e3:.0f} us")
us = lambda t: t / 100.0  # noqa: E731
print(f"r4_round_end: {v[1]} calls, {us(v[0]) / max(v[1], 1):.1f} us each, {us(v[0]) / slots / (r.total_ms * 1e3) * 100:.1f} % of the slot time "
      f"(before the wall loop {us(v[0] - v[5]) / max(v[1], 1):.1f} us, wall + deal + hand sorts + records {us(v[5]) / max(v[1], 1):.1f} us for {v[6] / max(v[1], 1):.2f} games per call)")
print("rows by mode at step4_finish_rounds (0 none, 1 draw, 2 restart):", v[8:11])
print("r4_round_end phases, us per call: entry + stale wait-cache refills %.1f | payments + ryukyoku record %.1f | decision + end records %.1f | restart + round reset %.1f" % tuple(us(v[k]) / max(v[1], 1) for k in (12, 13, 14, 15)))
print("call -> first statement of r4_round_end: %.2f us per call (the callee's prologue: callee-saved registers to scratch)" % (us(v[16] - v[17]) / max(v[1], 1)))
print("inside r4_round_end: draw rows, restart rows, new-round rows, rows of finished games:", v[20:24])
print(f"step4_pass2: {v[3]} calls, {us(v[2]) / max(v[3], 1):.1f} us each, {us(v[2]) / slots / (r.total_ms * 1e3) * 100:.1f} % of the slot time")
