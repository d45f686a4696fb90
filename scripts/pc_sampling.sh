set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --output-format csv -d $R/gpurun_out/pcs_ht -- python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline > $R/gpurun_out/pcs_ht.log 2>&1
echo "host_trap rc=$?"
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --output-format csv -d $R/gpurun_out/pcs_st -- python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline > $R/gpurun_out/pcs_st.log 2>&1
echo "stochastic rc=$?"
ls -la $R/gpurun_out/pcs_ht/* $R/gpurun_out/pcs_st/* | head -30
tail -5 $R/gpurun_out/pcs_ht.log $R/gpurun_out/pcs_st.log
