#!/usr/bin/env python3
"""round 6: does the host work between the warm-up and the timed 20-step window (two counter reads = two small launches + copies + synchronisations) cost the window
anything?  Alternates (a) warm-up -> reads -> barrier -> timed window (bench.py's order) and (b) reads -> warm-up -> barrier -> timed window."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riichienv_amd import abi, vecenv  # noqa: E402

env = vecenv.VecRiichiEnv(65536, game_mode=2, seed=0, rule_bits=abi.RULE_TENHOU, event_ring=64)
env.reset()
env.step_random(0xC0FFEE, 6000, auto_reset=True)
env.sync()
L = env.L
dsync = (lambda: L.rmj_bench_device_sync(0)) if len(sys.argv) > 1 and sys.argv[1] == "device" else env.sync
res = {"a": [], "b": []}
for rep in range(24):
    v = "ab"[rep % 2]
    if v == "b":
        env.total_full_path(); env.total_steps()
    if len(sys.argv) > 2 and sys.argv[2] == "same_entry":
        env.time_rollout(0xC0FFEE, 5)        # the warm-up through the entry point of the timed region
    else:
        env.step_random(0xC0FFEE, 5, auto_reset=True)
    if v == "a":
        env.total_full_path(); env.total_steps()
    dsync()
    t0 = time.perf_counter()
    r = env.time_rollout(0xC0FFEE, 20)
    dsync()
    t1 = time.perf_counter()
    res[v].append((65536 * 20 / (t1 - t0) / 1e6, r.total_ms))
print("windows in order (M env.step/s):", " ".join("%s%.0f" % (v, x[0]) for i in range(12) for v, x in (("a", res["a"][i]), ("b", res["b"][i]))))
for v in "ab":
    w = sorted(x[0] for x in res[v]); k = sorted(x[1] for x in res[v])
    print(f"{sys.argv[1] if len(sys.argv) > 1 else chr(115) + chr(116) + chr(114) + chr(101) + chr(97) + chr(109)} sync, order {v}: window median {w[len(w)//2]:.1f} M (min {w[0]:.1f} max {w[-1]:.1f}); kernel median {k[len(k)//2]:.4f} ms (min {k[0]:.4f})")
