#!/bin/bash
# PMC passes for the stall picture of k_step (issue stalls, LDS pipe): bash scripts/pmc_stalls.sh <tag>
TAG=${1:-stalls}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
i=0
for grp in "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_EXP_GDS" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_p$i -- python3 bench.py --steps 50 --warmup 20 --no-cpu-baseline > $R/gpurun_out/${TAG}_p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
python3 scripts/pmc_summary.py $R/gpurun_out $TAG k_step > $R/gpurun_out/${TAG}_summary.json
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/${TAG}_summary.json"))
print({k:round(v["mean_per_launch"]/16384,1) for k,v in d["counters"].items()})
PY
