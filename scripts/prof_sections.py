#!/usr/bin/env python3
"""Section timing of k_step: runs the bench workload on the profiling build (libriichi_mi355x_prof.so, -DRMJ_PROFILE)
and prints wave cycles per section.  Build: scripts/build_prof.sh.  Not part of the product path."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", "libriichi_mi355x_prof.so")

NAMES = {0: "load_state", 1: "policy pick", 2: "validate", 3: "WaitAct pre (find/remove/sort, other acts)",
         4: "resolve_discard head", 5: "claims x3", 6: "post-claims (deal_next)", 7: "WaitResponse branch",
         8: "(gap) step->finalize", 9: "act_legal: waits+tsumo", 10: "act_legal: discard/riichi probe",
         11: "act_legal: kan", 12: "act_legal: kyushu/kita", 13: "finalize: pre-publication", 14: "finalize: mask/list/status",
         15: "store_state",          20: "pre: hand_find", 21: "pre: remove_at", 22: "pre: sort", 23: "deal: accept/abortive", 24: "deal: deal_next", 25: "FULL PATH (ol_step_full)", 26: "fast-path tail before bail",
         16: "claims A: cache refill", 17: "claims B: ron eligibility", 18: "claims C: pon/kan", 19: "claims D: chi", 27: "claims E: tail", 28: "FULL PATH total (overlaps inner sections)"}


def main():
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    L = vecenv.load_lib()
    cyc = (C.c_uint64 * 32)()
    cnt = (C.c_uint64 * 32)()
    L.rmj_prof_fetch(games, cyc, cnt, 1)   # allocates the per-game accumulation buffer before any kernel runs
    env = vecenv.VecRiichiEnv(games, game_mode=mode, seed=0)
    env.reset()
    env.step_random(0xC0FFEE, 200, auto_reset=True)
    L.rmj_prof_fetch(games, cyc, cnt, 1)
    K = 300
    r = env.bench_rollout(0xC0FFEE, 0, K)
    L.rmj_prof_fetch(games, cyc, cnt, 0)
    waves = games * K
    tot = sum(cyc)
    out = {"kernel_ms": r.step_kernel_ms, "cycles_per_wave": tot / waves, "sections": {}}
    print(f"kernel {r.step_kernel_ms:.4f} ms (instrumented), {tot / waves:.0f} cycles per wave-step")
    for i in range(32):
        if cnt[i]:
            print(f"{i:2d} {NAMES.get(i, '?'):45s} visits/step {cnt[i] / waves:6.3f}  cyc/visit {cyc[i] / cnt[i]:8.0f}  "
                  f"cyc/step {cyc[i] / waves:8.0f}  {100.0 * cyc[i] / tot:5.1f}%")
            out["sections"][NAMES.get(i, str(i))] = {"visits_per_step": cnt[i] / waves, "cycles_per_visit": cyc[i] / cnt[i],
                                                      "cycles_per_step": cyc[i] / waves}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"prof_sections_mode{mode}.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
