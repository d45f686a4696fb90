#!/usr/bin/env python3
"""Parity soak under a policy that WINS (not part of the test suite): the oracle plays hanchan with the tenpai-seeking policy
of tests/mjsoul_util.greedy_actions (every win / riichi / kan / kita taken, shanten-greedy discards, some calls), the same packed
actions go to rmj_step, and device and oracle are compared after every step (status, ordered legal lists, masks, waits; the full
state of a rotating sample) and at the end (every state, every whole MJAI log).  The uniform RandomAgent of soak_parity.py wins
once in ~250 rounds; this policy ends most rounds with a win, so riichi / ippatsu / ura, Ron with several claimants, chankan,
rinshan, pao and the yaku checks of the step kernel's claim code run thousands of times.
usage: python scripts/soak_greedy.py [games] [max steps] [seed] [call rate, default 0.25]"""
import collections
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from riichienv_amd import abi, vecenv  # noqa: E402
from riichienv_amd.shard import game_seed  # noqa: E402
from tests.mjsoul_util import greedy_actions  # noqa: E402
from tests.test_gpu_step import _compare  # noqa: E402


def run(mode, rule, n, max_steps, seed, call_rate=0.25):
    sanma = mode >= 3
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=16384)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g), rule_bits=rule) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    rng = np.random.default_rng(seed)
    steps = 0
    for step in range(1, max_steps + 1):
        acts = np.full((n, 4), abi.NO_ACTION, dtype=np.uint64)
        live = 0
        for g, o in enumerate(games):
            if o.status()[2]:
                continue
            live += 1
            a = greedy_actions(o, rng, sanma, call_rate)
            acts[g] = a
            o.step(a)
        if not live:
            break
        steps += live
        env.step(acts)
        _compare(env, games, range(n), step, check_state=False)
        _compare(env, games, [step % n, (step * 7) % n], step, check_state=True)
    _compare(env, games, range(n), -1, check_state=True)
    tally = collections.Counter()
    logs = []
    for g, o in enumerate(games):
        log = o.log()
        dev = env.mjai_log(g)
        assert dev == log, (mode, g)
        logs.append([json.loads(s) for s in dev])
        for e in logs[-1]:
            tally[e["type"] if e["type"] != "ryukyoku" else "ryukyoku:" + e.get("reason", "")] += 1
    env.close()
    # every win of the device's logs, reconstructed by the replay stack and evaluated by rmj_eval_hands, pays what the log booked
    from scripts.parity_coverage import census_logs

    c = census_logs(logs, f"mode {mode}")
    assert c["bad"] == 0, c["wins"]
    return steps, tally, c


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 555
    call_rate = float(sys.argv[4]) if len(sys.argv) > 4 else 0.25
    t0 = time.time()
    total = 0
    for k, (mode, rule, name) in enumerate(((2, abi.RULE_TENHOU, "4p-red-half tenhou"), (2, abi.RULE_MJSOUL, "4p-red-half mjsoul"),
                                            (5, abi.RULE_MJSOUL, "3p-red-half mjsoul"), (5, abi.RULE_TENHOU, "3p-red-half tenhou"),
                                            (0, abi.RULE_TENHOU, "4p-red-single tenhou"), (4, abi.RULE_MJSOUL, "3p-red-east mjsoul"))):
        steps, tally, cen = run(mode, rule, n, max_steps, seed + 17 * k, call_rate)
        total += steps
        keep = {k: v for k, v in sorted(tally.items()) if k in ("hora", "reach", "reach_accepted", "ankan", "kakan", "daiminkan", "kita", "pon", "chi", "start_kyoku")
                or k.startswith("ryukyoku")}
        print(f"{name}: {n} games, {steps} game-steps ok ({time.time() - t0:.0f} s) {json.dumps(keep)}", flush=True)
        print(f"  wins reconstructed and re-scored on the GPU: {cen['checked']}, payments = deltas for all; kinds {json.dumps(cen['wins'])}; yaku ids {json.dumps(cen['yaku'])}",
              flush=True)
    print(f"greedy soak ok: {total} game-steps compared step by step")


if __name__ == "__main__":
    main()
