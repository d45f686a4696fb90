#!/bin/bash
# round 6: the counter passes of the fused RandomAgent rollout only (a subset of scripts/r05_profiles.sh), summary -> gpurun_out/<tag>_pmc_k_step4.json
TAG=${1:-r06q}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_k_step4_p$i -- python3 bench.py --mode 2 --steps 300 --warmup 300 --preroll 300 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_k_step4_p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_k_step4 "k_step4_queue<0>" 2 300 4 65536 > $R/gpurun_out/${TAG}_pmc_k_step4.json
rm -rf gpurun_out/${TAG}_k_step4_p*/ gpurun_out/${TAG}_k_step4_p*.log
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_pmc_k_step4.json')); print(d.get('hbm_traffic')); print({k:round(v,1) for k,v in d['per_wave'].items()})"
