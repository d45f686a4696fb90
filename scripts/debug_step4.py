#!/usr/bin/env python3
"""Step-by-step comparison of the four-games-per-wave kernel (RMJ_STEP4=1) with the oracle: prints the first divergence."""
import os
import sys

os.environ.setdefault("RMJ_STEP4", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle  # noqa: E402
from riichienv_amd import abi, vecenv  # noqa: E402
from riichienv_amd.shard import game_seed  # noqa: E402
from tests.parity_util import diff_dict, fmt_action, normalize_view  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
seed, pseed = 5, 4242
env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=4096)
games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
env.reset()
for o in games:
    o.reset()
bad = 0
for step in range(steps):
    acts = [o.random_actions(pseed, g) for g, o in enumerate(games)]
    env.step_random(pseed, 1, auto_reset=True)
    for g, o in enumerate(games):
        if o.status()[2]:
            o.reset()
        else:
            o.step(acts[g])
    legal, cnt = env.legal()
    mask = env.mask()
    waits = env.waits()
    act, ph, dn = env.status()
    for g, o in enumerate(games):
        msgs = []
        oa, op, od = o.status()
        if (act[g], ph[g], dn[g]) != (oa, op, od):
            msgs.append(f"status dev {(act[g], ph[g], dn[g])} oracle {(oa, op, od)}")
        d = diff_dict(normalize_view(env.peek(g)), normalize_view(o.peek()))
        if d:
            msgs.append("state " + "; ".join(d[:8]))
        for s in range(4):
            if (oa >> s) & 1 and not od:
                ol = o.legal(s)
                gl = [int(x) for x in legal[g, s, : cnt[g, s]]]
                if gl != ol:
                    msgs.append(f"legal seat {s}: dev {[fmt_action(a) for a in gl]} oracle {[fmt_action(a) for a in ol]}")
                if not (mask[g, s] == o.mask(s)).all():
                    msgs.append(f"mask seat {s}")
                if int(waits[g, s]) != o.waits(s):
                    msgs.append(f"waits seat {s}: {int(waits[g, s]):x} vs {o.waits(s):x}")
            elif cnt[g, s] != 0 or mask[g, s].sum() != 0:
                msgs.append(f"inactive seat {s} has outputs")
        if env.mjai_log(g)[-3:] != o.log()[-3:]:
            msgs.append(f"log tail dev {env.mjai_log(g)[-3:]} oracle {o.log()[-3:]}")
        if msgs:
            print(f"step {step} game {g}: actions {[fmt_action(a) for a in acts[g] if a != abi.NO_ACTION]}")
            for m in msgs:
                print("   ", m)
            bad += 1
    if bad:
        break
print("divergences:", bad, "after", step + 1, "steps; full-path steps", env.total_full_path(), "of", env.total_steps())
