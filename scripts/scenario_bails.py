#!/usr/bin/env python3
"""Why do scenario steps leave tier 0?  (-DRMJ_CENSUS build)  usage: scenario_bails.py name [name ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RMJ_ROWS", "4")
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", "libcensus.so")
from tests import scenarios  # noqa: E402
from tests.env_adapters import GpuEnv  # noqa: E402

L = vecenv.load_lib()
buf = (C.c_uint32 * 32)()
for name in sys.argv[1:]:
    L.rmj_prof_bail_census(buf, 1)
    getattr(scenarios, name)(lambda **kw: GpuEnv(**kw))
    L.rmj_prof_bail_census(buf, 0)
    print(name, {i: buf[i] for i in range(32) if buf[i]})
