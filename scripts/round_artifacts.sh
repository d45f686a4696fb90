#!/bin/bash
# Round artifacts on the GPU box: default bench line, 3P bench line, rocprofv3 kernel stats, PMC summary.
# usage: bash scripts/round_artifacts.sh <tag>     (outputs under gpurun_out/<tag>_*)
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
python3 bench.py --mode 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_3p_mode5.json 2>> gpurun_out/${TAG}_bench_n1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 bench.py --steps 500 --warmup 100 --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_stats/*/*kernel_trace.csv
bash scripts/pmc_collect.sh ${TAG}_pmc 2 > /dev/null 2>&1
cat gpurun_out/${TAG}_bench_n1.json gpurun_out/${TAG}_bench_3p_mode5.json
head -5 gpurun_out/${TAG}_kernel_stats.csv
