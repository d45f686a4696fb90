#!/usr/bin/env python3
"""Instruction accounting of k_step by section (accounting build, -DRMJ_CUTS: scripts/build_cuts.sh).

run (GPU box):   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/cuts -- \
                     python3 scripts/valu_sections.py run
then:            python3 scripts/valu_sections.py report gpurun_out/cuts

`run` warms the bench workload up, then launches k_step once per cut point: waves end at PROF mark `cut`, at mark 26 (the
hand-over to the full path, so that the numbers describe the fast path) and at mark 13 (before anything is published or
stored), so every launch sees the same game states.  `report` prints, per mark, the average over all waves of the
instructions executed with that cut in place (a wave that does not pass the mark runs on to mark 13 / 26): the
difference between two consecutive marks of one path is the cost of the section between them per average game-step."""
import csv
import ctypes as C
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CUTS = [0, 1, 2, 20, 22, 3, 4, 16, 17, 18, 19, 27, 5, 23, 24, 6, 7, 8, 9, 10, 11, 12, 13]
NAMES = {-1: "(whole fast path; full path cut)", 0: "load_state", 1: "policy pick", 2: "validate", 20: "pre: hand_find", 22: "pre: sort",
         3: "WaitAct pre", 4: "resolve_discard head", 16: "claims A: cache refill", 17: "claims B: ron eligibility",
         18: "claims C: pon/kan", 19: "claims D: chi", 27: "claims E: tail", 5: "claims (all)", 23: "deal: accept/abortive",
         24: "deal: deal_next", 6: "post-claims", 7: "WaitResponse branch", 8: "step -> finalize", 9: "act_legal: waits+tsumo",
         10: "act_legal: discard/riichi probe", 11: "act_legal: kan", 12: "act_legal: kyushu/kita", 13: "finalize: pre-publication",
         14: "finalize: mask/list/status", 15: "store_state"}
GAMES = 65536


def run():
    os.environ["RMJ_STEP_STREAMS"] = "1"
    from riichienv_amd import vecenv
    vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", "libriichi_mi355x_cuts.so")
    L = vecenv.load_lib()
    L.rmj_prof_set_cut.argtypes = [C.c_int, C.c_int, C.c_int]
    env = vecenv.VecRiichiEnv(GAMES, game_mode=int(os.environ.get("RMJ_MODE", "2")), seed=0)
    env.reset()
    L.rmj_prof_set_cut(-1, -1, -1)
    env.step_random(0xC0FFEE, 500, auto_reset=True)
    env.total_steps()
    for cut in CUTS:
        L.rmj_prof_set_cut(cut, 26, 13)
        env.step_random(0xC0FFEE, 1, auto_reset=True)
        env.total_steps()
    L.rmj_prof_set_cut(-1, -1, -1)
    env.step_random(0xC0FFEE, 1, auto_reset=True)      # reference launch, same states: everything, stores included
    env.total_steps()


def report(root):
    rows = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "k_step" not in r.get("Kernel_Name", ""):
                    continue
                d = int(r["Dispatch_Id"])
                rows.setdefault(d, {})
                rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(rows)[-(len(CUTS) + 1):]
    full = rows[ids[-1]]
    out = {"games": GAMES, "whole_step_per_wave": {k: v / GAMES for k, v in full.items()}, "reach_mark": {}}
    print("whole step (fast + full path, publication and stores), per wave:", {k: round(v / GAMES, 1) for k, v in full.items()})
    for cut, d in zip(CUTS, ids[:-1]):
        per = {k: v / GAMES for k, v in rows[d].items()}
        out["reach_mark"][str(cut)] = {"name": NAMES[cut], **per}
        print(f"{cut:3d} {NAMES[cut]:36s} VALU {per.get('SQ_INSTS_VALU', 0):7.1f}  SALU {per.get('SQ_INSTS_SALU', 0):7.1f}  LDS {per.get('SQ_INSTS_LDS', 0):6.1f}")
    json.dump(out, open(os.path.join(root, "valu_sections.json"), "w"), indent=1)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
