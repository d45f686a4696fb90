#!/bin/bash
# the greedy-policy part of the round-3 profile artifacts (after the last change to the rich tier 0)
TAG=${1:-r03}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_flags.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --policy greedy --no-cpu-baseline > gpurun_out/${TAG}_bench_greedy.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --policy greedy --mode 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_greedy_3p.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --mode 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_3p_mode5.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_st_g -- python3 bench.py --policy greedy --steps 1000 --warmup 1000 --preroll 1000 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_st_g.log 2>&1
find gpurun_out/${TAG}_st_g -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_greedy.csv \;
rm -rf gpurun_out/${TAG}_st_g
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_kg_p$i -- python3 bench.py --policy greedy --steps 300 --warmup 300 --preroll 300 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_kg_p$i.log 2>&1
done
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_kg "k_step4_queue<1>" 2 300 4 65536 > $R/gpurun_out/${TAG}_pmc_k_step4_greedy.json
rm -rf gpurun_out/${TAG}_kg_p*/ gpurun_out/${TAG}_kg_p*.log
if [ -f riichienv_amd/libriichi_mi355x_cuts.so ]; then
  RMJ_POLICY=greedy timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/${TAG}_cuts4g -- python3 scripts/valu_sections4.py run > gpurun_out/${TAG}_cuts4g.log 2>&1
  RMJ_POLICY=greedy python3 scripts/valu_sections4.py report gpurun_out/${TAG}_cuts4g > gpurun_out/${TAG}_k_step4_sections_greedy.txt
  cp gpurun_out/${TAG}_cuts4g/valu_sections4.json gpurun_out/${TAG}_valu_sections_greedy.json
  rm -rf gpurun_out/${TAG}_cuts4g
  for m in 2 5; do python3 scripts/bail_census.py $m greedy 64 2>/dev/null; python3 scripts/bail_census.py $m random 2>/dev/null; done > gpurun_out/${TAG}_bail_census_all.txt
fi
cat gpurun_out/${TAG}_k_step4_sections_greedy.txt; python3 -c "
import json
for f in ('bench_n1','bench_driver_flags','bench_greedy','bench_greedy_3p','bench_3p_mode5'):
    d=json.load(open('gpurun_out/${TAG}_'+f+'.json')); print(f, round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2), d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['full_path_frac'],4), {k:round(v['value']/1e6,1) for k,v in d.items() if isinstance(v,dict) and 'value' in v})
d=json.load(open('gpurun_out/${TAG}_pmc_k_step4_greedy.json')); print({k:round(v,1) for k,v in d['per_wave'].items()}, d['hbm_traffic']['bytes_per_step'])"
