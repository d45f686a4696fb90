#!/bin/bash
# Round-6 profile artifacts on the GPU box (outputs under gpurun_out/r06_*; the summaries are copied to profiles/ by hand).
TAG=${1:-r06}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_flags.json 2> gpurun_out/${TAG}_bench.err        # the driver's command: every leg in one line
python3 bench.py > gpurun_out/${TAG}_bench_default_flags.json 2>> gpurun_out/${TAG}_bench.err                      # default flags (2 000 timed steps)
python3 bench.py --policy greedy --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_greedy.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --mode 5 --encode --no-cpu-baseline > gpurun_out/${TAG}_bench_3p_encode.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --mode 5 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_3p_mode5.json 2>> gpurun_out/${TAG}_bench.err
stats() { # tag, program args...
  local t=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_st_$t -- python3 "$@" > gpurun_out/${TAG}_st_$t.log 2>&1
  find gpurun_out/${TAG}_st_$t -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats$t.csv \;
  rm -rf gpurun_out/${TAG}_st_$t
}
# every k_step4_queue launch of these is a rollout of exactly --steps steps (warmup == steps == preroll)
stats "" bench.py --steps 1000 --warmup 1000 --preroll 1000 --no-cpu-baseline --no-extras
stats _driver_flags bench.py --steps 20 --warmup 20 --preroll 6000 --no-cpu-baseline --no-extras
stats _3p_encode bench.py --mode 5 --encode --steps 300 --warmup 300 --preroll 300 --no-cpu-baseline
stats _greedy bench.py --policy greedy --steps 1000 --warmup 1000 --preroll 1000 --no-cpu-baseline --no-extras
stats _single_stream bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-configs
pmc() { # tag, kernel substring, mode, bench args...
  local t=$1 k=$2 m=$3; shift 3
  local i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_${t}_p$i -- python3 bench.py --mode $m --steps 300 --warmup 300 --preroll 300 --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/${TAG}_${t}_p$i.log 2>&1
    echo "$t pass $i ($grp) rc=$?"
  done
  python3 scripts/pmc_summary.py $R/gpurun_out ${TAG}_${t} "$k" $m 300 4 65536 > $R/gpurun_out/${TAG}_pmc_${t}.json
  rm -rf gpurun_out/${TAG}_${t}_p*/ gpurun_out/${TAG}_${t}_p*.log
}
pmc k_step4 "k_step4_queue<0>" 2
pmc k_step4_greedy "k_step4_queue<1>" 2 --policy greedy
pmc k_step4_enc "k_step4_queue_enc" 5 --encode
python3 scripts/bench_hand_kernels.py 20 3 > gpurun_out/${TAG}_hand_kernels.txt 2>&1
python3 scripts/bench_torch_env.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_torch_loop.txt
cut -c1-600 gpurun_out/${TAG}_bench_driver_flags.json; head -4 gpurun_out/${TAG}_kernel_stats.csv; head -3 gpurun_out/${TAG}_kernel_stats_driver_flags.csv
