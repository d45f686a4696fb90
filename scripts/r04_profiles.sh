#!/bin/bash
# Round-4 profile artifacts on the GPU box (outputs under gpurun_out/r04_*; the summaries are copied to profiles/ by hand).
# Needs the accounting builds beside the library: libcensus.so (-DRMJ_CENSUS) and libriichi_mi355x_tl4.so (scripts/build_tl4.sh).
TAG=${1:-r04}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_driver_flags.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --policy greedy --no-cpu-baseline > gpurun_out/${TAG}_bench_greedy.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --policy greedy --mode 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_greedy_3p.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --mode 5 --encode --no-cpu-baseline > gpurun_out/${TAG}_bench_3p_encode.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --mode 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_3p_mode5.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --games 524288 --steps 500 --warmup 300 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_524288.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --games 4096 --mode 0 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_4096_mode0.json 2>> gpurun_out/${TAG}_bench.err
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
stats _single_stream bench.py --steps 300 --warmup 20 --no-cpu-baseline
stats _hand_kernels scripts/bench_hand_kernels.py 20 0 nocpu
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
python3 scripts/bench_torch_env.py 65536 ext 2>&1 | grep -v amdgpu >> gpurun_out/${TAG}_torch_loop.txt
if [ -f riichienv_amd/libcensus.so ]; then
  export RMJ_CENSUS_LIB=libcensus.so
  { for m in 2 5; do
      echo "== mode $m, per-step kernel (k_step4<false>: rich tier 0), RandomAgent"; python3 scripts/bail_census.py $m random 2>/dev/null
      echo "== mode $m, per-step kernel, greedy policy"; python3 scripts/bail_census.py $m greedy 64 2>/dev/null
      echo "== mode $m, fused rollout (lean tier 0), RandomAgent"; RMJ_CENSUS_FUSED=1 python3 scripts/bail_census.py $m random 2>/dev/null
      echo "== mode $m, fused rollout (rich tier 0), greedy policy"; RMJ_CENSUS_FUSED=1 python3 scripts/bail_census.py $m greedy 64 2>/dev/null
    done; } > gpurun_out/${TAG}_bail_census.txt
fi
if [ -f riichienv_amd/libriichi_mi355x_tl4.so ]; then
  python3 scripts/timeline4.py > gpurun_out/${TAG}_timeline4.txt 2>&1
fi
cut -c1-700 gpurun_out/${TAG}_bench_n1.json; head -4 gpurun_out/${TAG}_kernel_stats.csv; head -3 gpurun_out/${TAG}_kernel_stats_driver_flags.csv
