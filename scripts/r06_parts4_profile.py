#!/usr/bin/env python3
"""round 6: only the 4-shard leg of scripts/bench_torch_env.py (for rocprofv3 --kernel-trace --stats: which kernel of that loop changed between builds)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_torch_env as B  # noqa: E402

B.run_parts(65536, 4)
