#!/bin/bash
# rocprofv3 counter passes (one group per run, --pmc alone) + kernel stats of one bench.py command.
# usage: scripts/r03_pmc.sh TAG KERNEL_SUBSTRING STEPS_PER_LAUNCH MODE -- <extra bench.py args>
# Every launch of the kernel must be a rollout of exactly STEPS_PER_LAUNCH steps: --steps N --warmup N --preroll (k * N).
TAG=$1; KERN=$2; N=$3; MODE=$4; shift 5
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc_p$i -- python3 bench.py --mode $MODE --steps $N --warmup $N --preroll $((3 * N)) --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_pmc_p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmc "$KERN" $MODE $N 4 65536 > $R/gpurun_out/${TAG}_pmc.json
rm -rf gpurun_out/${TAG}_pmc_p*/
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 bench.py --mode $MODE --steps 1000 --warmup 1000 --no-cpu-baseline --no-extras "$@" > gpurun_out/${TAG}_stats.log 2>&1
find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_stats
tail -1 gpurun_out/${TAG}_stats.log | cut -c1-300
head -4 gpurun_out/${TAG}_kernel_stats.csv
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_pmc.json')); print(d.get('hbm_traffic')); print({k:round(v,1) for k,v in d['per_wave'].items()})"
