#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs of several --pmc passes: mean counter value per launch of one kernel."""
import csv
import glob
import json
import os
import sys

root, tag, kern = sys.argv[1], sys.argv[2], sys.argv[3]
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
steps_per_launch = int(sys.argv[5]) if len(sys.argv) > 5 else 1     # fused rollouts: one launch = that many steps of every game
games_per_wave = int(sys.argv[6]) if len(sys.argv) > 6 else 1
games_known = int(sys.argv[7]) if len(sys.argv) > 7 else 0            # games of a launch when known (SQ_WAVES over-counts fused rollouts)
out = {}
for f in glob.glob(os.path.join(root, tag + "_p*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kern not in row.get("Kernel_Name", ""):
                continue
            key = (row["Counter_Name"], row.get("Dispatch_Id"))
            acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
    per = {}
    for (name, _), v in acc.items():
        per.setdefault(name, []).append(v)
    for name, vals in per.items():
        out[name] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
doc = {"command": "rocprofv3 --pmc <group> -- python3 <bench command> (one pass per counter group, scripts/r02_profiles.sh / "
                  "scripts/pmc_collect.sh)", "kernel": kern, "mode": mode, "counters": out}
waves = out.get("SQ_WAVES", {}).get("mean_per_launch")
if games_known:
    doc["sq_waves_counter"] = waves
    waves = games_known / games_per_wave
if waves:
    doc["games_per_launch"] = int(round(waves)) * games_per_wave
    doc["games_per_wave"] = games_per_wave
    doc["steps_per_launch"] = steps_per_launch
    # per wave AND per step: a wave of a fused rollout lives for steps_per_launch steps
    doc["per_wave"] = {k: v["mean_per_launch"] / waves / steps_per_launch for k, v in out.items()}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    f, w = out["FETCH_SIZE"]["mean_per_launch"], out["WRITE_SIZE"]["mean_per_launch"]
    doc["hbm_traffic"] = {"fetch_kb": f, "write_kb": w,
                          "note": "FETCH_SIZE is reported in 64 B units on gfx950 but scaled as 32 B by the tool: doubled per "
                                  "MI355X_MICROARCH.md; WRITE_SIZE as is",
                          "bytes_per_launch": (2.0 * f + w) * 1024.0,
                          "bytes_per_step": (2.0 * f + w) * 1024.0 / steps_per_launch}
print(json.dumps(doc, indent=1))
