#!/bin/bash
# PMC passes over the bench command (one rocprofv3 --pmc run per counter group, no tracing flags), then a summary JSON.
# usage (on the GPU box): bash scripts/pmc_collect.sh <tag> [mode]
TAG=${1:-pmc}; MODE=${2:-2}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_p$i -- python3 bench.py --steps 50 --warmup 20 --no-cpu-baseline --mode $MODE > $R/gpurun_out/${TAG}_p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
python3 scripts/pmc_summary.py $R/gpurun_out $TAG k_step > $R/gpurun_out/${TAG}_summary.json
cat $R/gpurun_out/${TAG}_summary.json
