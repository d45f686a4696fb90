#!/usr/bin/env python3
"""Sums a rocprofv3 --pmc counter_collection.csv by kernel: counters per launch and per wave.  usage: pmc_by_kernel.py <csv> [name filter]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
    calls[name].add(row["Dispatch_Id"])
for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    n = len(calls[name])
    waves = c.get("SQ_WAVES", 0) / n if n else 0
    line = f"{name[:70]:70s} launches {n:4d} waves/launch {waves:9.0f}"
    for k, v in sorted(c.items()):
        if k != "SQ_WAVES":
            line += f"  {k} {v / n / max(waves, 1):9.1f}/wave"
    print(line)
