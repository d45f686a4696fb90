#!/usr/bin/env python3
"""round 6: what do round ends cost the fused rollout?  20-step windows right after a reset (every wall has 70 draws left: no round can end for ~45 steps) against
windows in steady state (a round ends in ~4.7 % of the quad-calls), same binary, same process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riichienv_amd import abi, vecenv  # noqa: E402

env = vecenv.VecRiichiEnv(65536, game_mode=2, seed=0, rule_bits=abi.RULE_TENHOU, event_ring=64)
env.reset()
env.step_random(0xC0FFEE, 6000, auto_reset=True)
env.time_rollout(0xC0FFEE, 5)
steady = [env.time_rollout(0xC0FFEE, 20).total_ms for _ in range(8)]
fresh = []
for rep in range(8):
    env.reset()
    env.time_rollout(0xC0FFEE, 10)             # first go-arounds (kyushu / first-turn rules)
    fresh.append(env.time_rollout(0xC0FFEE, 20).total_ms)
    fresh.append(env.time_rollout(0xC0FFEE, 20).total_ms)
med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
print(f"steady state: kernel median {med(steady):.4f} ms per 20-step window ({65536 * 20 / med(steady) / 1e3:.0f} M env.step/s)")
print(f"steps 10-50 after a reset (no round ends): kernel median {med(fresh):.4f} ms ({65536 * 20 / med(fresh) / 1e3:.0f} M env.step/s)")
