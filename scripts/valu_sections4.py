#!/usr/bin/env python3
"""Instruction accounting of k_step4 (four games per wave) by section, accounting build -DRMJ_CUTS (scripts/build_cuts.sh).

run (GPU box):   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/cuts4 -- \
                     python3 scripts/valu_sections4.py run
then:            python3 scripts/valu_sections4.py report gpurun_out/cuts4

The marks end the WAVE (all four rows) where the first row reaches them: marks inside the WaitAct branch (42, 50..56) give
the cost of the discard path up to there for a wave whose rows are in that branch, marks at the convergence points (40,
41, 43, 45, 46, 47) the cost of everything before them.  Nothing is stored by a cut launch: every launch sees the same
game states (the rollout is warmed up with the uncut kernel)."""
import csv
import ctypes as C
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CUTS = [40, 41, 42, 50, 51, 52, 53, 54, 55, 56, 43, 60, 61, 45, 46, 47]
NAMES = {40: "records loaded", 41: "policy pick", 42: "discard: find + sort", 50: "discard: bookkeeping + dahai event", 51: "claims A: wait-cache refill",
         52: "claims B: ron eligibility", 53: "claims C: pon / kan lists", 54: "claims D: chi lists", 55: "claims E: pass / lengths",
         56: "no claims: riichi accept, abortive check, next draw", 43: "both phase branches done (incl. WaitResponse)",
         60: "act list: tsumo check", 61: "act list: discards + riichi bound", 45: "act list: kan / kyushu / kita", 46: "publication: lists, masks, status, events",
         47: "records stored"}
GAMES = 65536


def run():
    from riichienv_amd import vecenv
    vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", "libriichi_mi355x_cuts.so")
    L = vecenv.load_lib()
    L.rmj_prof_set_cut.argtypes = [C.c_int, C.c_int, C.c_int]
    env = vecenv.VecRiichiEnv(GAMES, game_mode=int(os.environ.get("RMJ_MODE", "2")), seed=0)
    env.reset()
    L.rmj_prof_set_cut(-1, -1, -1)
    greedy = os.environ.get("RMJ_POLICY", "random") == "greedy"      # RMJ_POLICY=greedy: the sections under rmj_step_greedy
    step = (lambda k: env.step_greedy(0xC0FFEE, k, auto_reset=True, call_rate_256=64)) if greedy else \
           (lambda k: env.step_random(0xC0FFEE, k, auto_reset=True))
    step(1000 if greedy else 500)       # warm-up as ONE fused launch (a launch per step under the counters takes ~0.5 s each)
    env.total_steps()
    env.set_rollout_streams(1)          # from here on every step is its own launch of k_step4<false>
    base = env
    cuts = list(CUTS)   # (round 4: the marks are an asm s_endpgm - the noreturn builtin inside divergent control flow was what broke mark 43 under the greedy instantiation)
    for cut in cuts + [-1]:          # (-1: reference launch, same states: everything, stores and bails included)
        # a cut behind the publication leaves new lists next to old records: every cut runs on a fresh copy of the same states
        env = base.clone()
        env.set_rollout_streams(1)
        step = (lambda k: env.step_greedy(0xC0FFEE, k, auto_reset=True, call_rate_256=64)) if greedy else \
               (lambda k: env.step_random(0xC0FFEE, k, auto_reset=True))
        L.rmj_prof_set_cut(cut, -1, -1)
        step(1)
        env.total_steps()
        L.rmj_prof_set_cut(-1, -1, -1)
        env.close()


def report(root):
    rows = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "k_step4" not in r.get("Kernel_Name", ""):
                    continue
                d = int(r["Dispatch_Id"])
                rows.setdefault(d, {})
                rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    cuts = list(CUTS)
    ids = sorted(rows)[-(len(cuts) + 1):]
    W = GAMES // 4
    full = rows[ids[-1]]
    out = {"games": GAMES, "waves": W, "whole_step_per_wave": {k: v / W for k, v in full.items()}, "reach_mark": {}}
    print("whole step (tier 0 + bailed games, publication and stores), per wave of four games:", {k: round(v / W, 1) for k, v in full.items()})
    for cut, d in zip(cuts, ids[:-1]):
        per = {k: v / W for k, v in rows[d].items()}
        out["reach_mark"][str(cut)] = {"name": NAMES[cut], **per}
        line = f"{cut:3d} {NAMES[cut]:52s} VALU {per.get('SQ_INSTS_VALU', 0):7.1f}  SALU {per.get('SQ_INSTS_SALU', 0):7.1f}  LDS {per.get('SQ_INSTS_LDS', 0):6.1f}"
        if per.get("SQ_ACTIVE_INST_VALU"):   # lane use: a second run with --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU (cumulative up to the mark)
            line += f"  lanes/VALU-cycle {per['SQ_THREAD_CYCLES_VALU'] / per['SQ_ACTIVE_INST_VALU']:5.1f} of 64"
        print(line)
    # lane use of each section = difference of the cumulative counters between consecutive marks (marks in program order)
    prev = None
    for cut, d in zip(cuts, ids[:-1]):
        cur = rows[d]
        if prev is not None and cur.get("SQ_ACTIVE_INST_VALU") and cur["SQ_ACTIVE_INST_VALU"] > prev[1].get("SQ_ACTIVE_INST_VALU", 0):
            da = cur["SQ_ACTIVE_INST_VALU"] - prev[1]["SQ_ACTIVE_INST_VALU"]
            dt = cur["SQ_THREAD_CYCLES_VALU"] - prev[1]["SQ_THREAD_CYCLES_VALU"]
            print(f"    section {prev[0]:3d} -> {cut:3d}: {dt / da:5.1f} lanes per VALU cycle ({dt / da / 64:.2f}), {da / W:8.1f} VALU cycles per wave")
        prev = (cut, cur)
    json.dump(out, open(os.path.join(root, "valu_sections4.json"), "w"), indent=1)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
