#!/bin/bash
# Round-2 profile artifacts on the GPU box (outputs under gpurun_out/<tag>_*; the summaries are copied to profiles/ by hand):
#   bench lines (headline, 3P, 3P + feature tensor), rocprofv3 kernel stats of those commands, FETCH/WRITE + instruction PMC
#   passes for k_step4 (fused rollout) and the encoders, section accounting and bail census (accounting build), trainer
#   loop and host-path rates.
TAG=${1:-r02}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
python3 bench.py --mode 5 --encode --steps 400 --warmup 300 --no-cpu-baseline > gpurun_out/${TAG}_bench_3p_encode.json 2>> gpurun_out/${TAG}_bench_n1.err
python3 bench.py --mode 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_3p_mode5.json 2>> gpurun_out/${TAG}_bench_n1.err
python3 bench.py --games 524288 --steps 500 --warmup 300 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_524288.json 2>> gpurun_out/${TAG}_bench_n1.err
python3 bench.py --games 4096 --mode 0 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_4096_mode0.json 2>> gpurun_out/${TAG}_bench_n1.err
# kernel stats: the headline command (warmup == steps: every k_step4_queue launch is a rollout of exactly 1000 steps) ...
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 bench.py --steps 1000 --warmup 1000 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_stats.log 2>&1
find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_stats
# ... and the feature-output command
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_enc -- python3 bench.py --mode 5 --encode --steps 300 --warmup 300 --no-cpu-baseline > gpurun_out/${TAG}_stats_enc.log 2>&1
find gpurun_out/${TAG}_stats_enc -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_3p_encode.csv \;
rm -rf gpurun_out/${TAG}_stats_enc
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_ext -- python3 scripts/bench_encoders.py > gpurun_out/${TAG}_bench_encoders.json 2> gpurun_out/${TAG}_stats_ext.log
find gpurun_out/${TAG}_stats_ext -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_encoders.csv \;
rm -rf gpurun_out/${TAG}_stats_ext
# PMC: the fused rollout kernel k_step4_queue (one group per run, no tracing flags; warmup == steps = 300)
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc_p$i -- python3 bench.py --steps 300 --warmup 300 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_pmc_p$i.log 2>&1
  echo "k_step4 pass $i ($grp) rc=$?"
done
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmc "k_step4_queue" 2 300 4 65536 > $R/gpurun_out/${TAG}_pmc_k_step4.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmcenc_p$i -- python3 scripts/bench_encoders.py > $R/gpurun_out/${TAG}_pmcenc_p$i.log 2>&1
  echo "encoders pass $i ($grp) rc=$?"
done
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmcenc "k_encode_base<false, false>" 2 > $R/gpurun_out/${TAG}_pmc_k_encode_4p.json
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmcenc "k_encode_base<true, false>" 5 > $R/gpurun_out/${TAG}_pmc_k_encode_3p.json
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmcenc "k_encode_base<false, true>" 2 > $R/gpurun_out/${TAG}_pmc_k_encode_compact_4p.json
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmcenc "k_encode_base<true, true>" 5 > $R/gpurun_out/${TAG}_pmc_k_encode_compact_3p.json
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmcenc "k_encode_ext<false>" 2 > $R/gpurun_out/${TAG}_pmc_k_encode_ext_4p.json
python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_pmcenc "k_encode_ext<true>" 5 > $R/gpurun_out/${TAG}_pmc_k_encode_ext_3p.json
rm -rf gpurun_out/${TAG}_pmc_p* gpurun_out/${TAG}_pmcenc_p*
# accounting build: sections of one step (per-step launches of k_step4<false>) and the bail census
if [ -f riichienv_amd/libriichi_mi355x_cuts.so ]; then
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/${TAG}_cuts4 -- python3 scripts/valu_sections4.py run > gpurun_out/${TAG}_cuts4.log 2>&1
  python3 scripts/valu_sections4.py report gpurun_out/${TAG}_cuts4 > gpurun_out/${TAG}_k_step4_sections.txt
  cp gpurun_out/${TAG}_cuts4/valu_sections4.json gpurun_out/${TAG}_k_step4_valu_sections.json
  rm -rf gpurun_out/${TAG}_cuts4
  python3 scripts/bail_census.py 2 2>/dev/null > gpurun_out/${TAG}_bail_census.txt
  python3 scripts/bail_census.py 5 2>/dev/null >> gpurun_out/${TAG}_bail_census.txt
fi
# timeline build: what a launch per step spends its time on
if [ -f riichienv_amd/libriichi_mi355x_tl4.so ]; then
  python3 scripts/timeline4.py 65536 2 2>/dev/null > gpurun_out/${TAG}_k_step4_launch_timeline.json
fi
python3 scripts/bench_torch_env.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_torch_loop.txt
python3 scripts/bench_torch_env.py 65536 ext 2>&1 | grep -v amdgpu >> gpurun_out/${TAG}_torch_loop.txt
python3 scripts/bench_host_path.py 2>/dev/null > gpurun_out/${TAG}_host_path.json
cat gpurun_out/${TAG}_bench_n1.json; head -4 gpurun_out/${TAG}_kernel_stats.csv; head -4 gpurun_out/${TAG}_kernel_stats_3p_encode.csv
