#!/usr/bin/env python3
"""Which path do the behavioural scenarios take on the GPU?  Runs every scenario of tests/scenarios.py through the HIP path alone and prints
the game's full-path step count against its step count (rows settled / dealt in row form leave the full-path counter alone)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RMJ_ROWS", "4")
from tests.env_adapters import GpuEnv  # noqa: E402
from tests.scenarios import SCENARIOS, SCENARIOS_3P  # noqa: E402

envs = []


def make(**kw):
    e = GpuEnv(**kw)
    envs.append(e)
    return e


tot = [0, 0]
for sc in SCENARIOS + SCENARIOS_3P:
    envs.clear()
    try:
        sc(make)
    except Exception as ex:   # (scenarios assert against the oracle's twin elsewhere; here only the path census matters)
        print(f"{sc.__name__:62s} raised {type(ex).__name__}")
        continue
    steps = sum(int(e.e.total_steps()) for e in envs)
    full = sum(int(e.e.total_full_path()) for e in envs)
    tot[0] += steps
    tot[1] += full
    print(f"{sc.__name__:62s} steps {steps:4d}  full-path steps {full:4d}")
print(f"total: {tot[0]} steps, {tot[1]} through the full path")
