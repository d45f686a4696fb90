#!/bin/bash
# round 2, first GPU pass: GPU test suite (incl. the full-size sampled parity tests), bench lines, lane-utilisation PMC
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
T=${1:-r02a}
python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/${T}_fullsize.log 2>&1; echo "fullsize rc=$?"; tail -5 gpurun_out/${T}_fullsize.log
python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py > gpurun_out/${T}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/${T}_gputests.log
python3 bench.py > gpurun_out/${T}_bench_n1.json 2> gpurun_out/${T}_bench_n1.err; echo "bench rc=$?"; cat gpurun_out/${T}_bench_n1.json
python3 bench.py --mode 5 --encode --steps 300 --warmup 300 --no-cpu-baseline > gpurun_out/${T}_bench_3p_encode.json 2> gpurun_out/${T}_bench_3p_encode.err; echo "bench enc rc=$?"; cat gpurun_out/${T}_bench_3p_encode.json
python3 bench.py --gpus 2 --steps 50 --warmup 50 --no-cpu-baseline > gpurun_out/${T}_bench_gpus2.out 2> gpurun_out/${T}_bench_gpus2.err; echo "bench --gpus 2 on a 1-GPU box rc=$? (expected 3)"; tail -2 gpurun_out/${T}_bench_gpus2.err
rocprofv3 -L > gpurun_out/${T}_counters_avail.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/${T}_lane_p1 -- python3 bench.py --steps 50 --warmup 300 --no-cpu-baseline --no-extras > gpurun_out/${T}_lane_p1.log 2>&1; echo "lane pmc rc=$?"
python3 scripts/pmc_summary.py $R/gpurun_out ${T}_lane k_step > gpurun_out/${T}_lane_summary.json; cat gpurun_out/${T}_lane_summary.json | head -40
