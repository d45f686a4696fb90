#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs of several --pmc passes: mean counter value per launch of one kernel."""
import csv
import glob
import json
import os
import sys

root, tag, kern = sys.argv[1], sys.argv[2], sys.argv[3]
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
print(json.dumps({"kernel": kern, "counters": out}, indent=1))
