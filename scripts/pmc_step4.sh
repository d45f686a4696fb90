#!/bin/bash
# PMC passes for the four-games-per-wave step kernel (RMJ_STEP4=1): instruction mix, wave cycles, HBM traffic
TAG=${1:-s4}
export TMPDIR=/tmp RMJ_STEP4=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc_p$i -- python3 bench.py --steps 50 --warmup 300 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_pmc_p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmc k_step4 2 > $R/gpurun_out/${TAG}_pmc_k_step4.json
rm -rf gpurun_out/${TAG}_pmc_p*/
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 bench.py --steps 500 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_stats.log 2>&1
find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_stats
head -4 gpurun_out/${TAG}_kernel_stats.csv
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_pmc_k_step4.json')); print(d.get('hbm_traffic')); print({k:round(v,1) for k,v in d['per_wave'].items()})"
