"""Experiment: does overlapping the tail of one launch with the body of another pay?  The 65 536 games are held by
K handles (one HIP stream each) that are stepped independently; compared with one handle / one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riichienv_amd import vecenv

N, STEPS = 65536, 1000
for k in (1, 2, 4):
    envs = [vecenv.VecRiichiEnv(N // k, game_mode=2, seed=1000 * i) for i in range(k)]
    for e in envs:
        e.reset()
        e.step_random(0xC0FFEE, 300, auto_reset=True)
    before = sum(e.total_steps() for e in envs)
    t0 = time.perf_counter()
    CH = 50
    for _ in range(STEPS // CH):
        for e in envs:
            e.step_random(0xC0FFEE, CH, auto_reset=True)
    after = sum(e.total_steps() for e in envs)   # synchronises every stream
    dt = time.perf_counter() - t0
    print(f"{k} stream(s) x {N // k} games: {dt / STEPS * 1e6:.1f} us per step of all games, {(after - before) / dt / 1e6:.1f} M env.step/s")
    for e in envs:
        e.close()
