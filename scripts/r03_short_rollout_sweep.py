#!/usr/bin/env python3
"""Short fused rollouts: env.step/s of a K-step rollout of 65 536 steady-state games against the shortest ticket the library may cut
(RMJ_QUEUE_MIN_CHUNK; 0 here = tickets off, every wave keeps its quad).  usage: python scripts/r03_short_rollout_sweep.py [floors, comma-separated] [rollout lengths, comma-separated]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from riichienv_amd import vecenv
    floors = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 2, 4, 8, 16]
    ks = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [8, 20, 40, 100, 300]
    for floor in floors:
        if floor == 0:
            os.environ["RMJ_QUEUE_CHUNK"] = "0"
        else:
            os.environ.pop("RMJ_QUEUE_CHUNK", None)
            os.environ["RMJ_QUEUE_MIN_CHUNK"] = str(floor)
        env = vecenv.VecRiichiEnv(65536, game_mode=2, seed=0, event_ring=64)
        env.reset()
        env.step_random(0xC0FFEE, 600, auto_reset=True)
        row = []
        for k in ks:
            best = 0.0
            for _ in range(5):
                r = env.bench_rollout(0xC0FFEE, 0, k)
                best = max(best, r.env_steps / (r.total_ms * 1e-3))
            row.append(f"{k}: {best / 1e6:7.1f} M (queued {int(r.queued)})")
        print(f"min chunk {floor:2d} | " + " | ".join(row), flush=True)
        env.close()


if __name__ == "__main__":
    main()
