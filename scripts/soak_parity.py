#!/usr/bin/env python3
"""Parity soak (not part of the test suite): long device-policy rollouts of every game mode under both rule sets and
several seeds, compared with the oracle game by game - final state, legal lists, masks, waits, step counts and the whole
MJAI log of every game.  usage: python scripts/soak_parity.py [games] [steps] [seeds] [first seed index]
(RMJ_QUEUE_FORCE=1 with >= 256 games runs the rollouts as (quad, chunk) tickets, kernel k_step4_queue, instead of one quad per wave;
RMJ_SOAK_POLICY=greedy: the device policy that plays to win, rmj_step_greedy, against the oracle's twin orc_game_greedy_actions)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from riichienv_amd import abi, vecenv  # noqa: E402
from riichienv_amd.shard import game_seed  # noqa: E402
from tests.test_gpu_step import _compare  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
first = (int(sys.argv[4]) if len(sys.argv) > 4 else 0) + int(os.environ.get("RMJ_SOAK_OFFSET", "0"))   # (RMJ_SOAK_OFFSET: the same run plan on other seeds)
GREEDY = os.environ.get("RMJ_SOAK_POLICY", "random") == "greedy"
RATE = int(os.environ.get("RMJ_SOAK_CALL_RATE", "64"))
EXTRA = int(os.environ.get("RMJ_SOAK_RULE_EXTRA", "0"))   # e.g. 256 = abi.RULE_REFERENCE_RNG on top of both rule sets
total = 0
t0 = time.time()
for mode in range(6):
    for rule, rname in ((abi.RULE_TENHOU | EXTRA, "tenhou"), (abi.RULE_MJSOUL | EXTRA, "mjsoul")):
        for k in range(first, first + seeds):
            seed, pseed = 7000 + 131 * k + mode, 0xA5A5 + 977 * k
            env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=8192)
            games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g), rule_bits=rule) for g in range(n)]
            env.reset()
            for o in games:
                o.reset()
            if GREEDY:
                env.step_greedy(pseed, steps, auto_reset=True, call_rate_256=RATE)
            else:
                env.step_random(pseed, steps, auto_reset=True)
            for g, o in enumerate(games):
                for _ in range(steps):
                    if o.status()[2]:
                        o.reset()
                        continue
                    o.step(o.greedy_actions(pseed, g, RATE) if GREEDY else o.random_actions(pseed, g))
            _compare(env, games, range(n), steps)
            assert list(env.step_counts()) == [o.step_count for o in games]
            cnt = env.event_counts()
            for g in range(n):
                if cnt[g] <= 8192:   # the whole log is still in the ring
                    assert env.mjai_log(g) == games[g].log(), (mode, rname, k, g)
            total += n * steps
            print(f"mode {mode} {rname} seed {seed}: {n} games x {steps} steps ok ({time.time() - t0:.0f} s)", flush=True)
            env.close()
print(f"soak ok: {total} game-steps compared")
