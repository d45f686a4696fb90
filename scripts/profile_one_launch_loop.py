#!/usr/bin/env python3
"""The trainer loop in its fastest one-stream form (fused masked sampler + step and encode as one launch, padded rows) alone, for
`rocprofv3 --kernel-trace --stats -- python3 scripts/profile_one_launch_loop.py`: which kernels make up an iteration."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_torch_env as B  # noqa: E402

B.run_one_launch(int(sys.argv[1]) if len(sys.argv) > 1 else 65536, pad=True)
