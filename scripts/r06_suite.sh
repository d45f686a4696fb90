#!/bin/bash
# round 6: the GPU suite, the default bench line and the hand-kernel gate on the library as built by __graft_entry__.build()
cd "$(dirname "$0")/.." && export PYTHONPATH=.
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/suite.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06/suite.log
tail -5 gpurun_out/r06/suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err; tail -c 1500 gpurun_out/r06/bench_default.json
python scripts/bench_hand_kernels.py > gpurun_out/r06/hand_kernels.txt 2>&1; grep -i "hands/s" gpurun_out/r06/hand_kernels.txt | cut -c1-120
