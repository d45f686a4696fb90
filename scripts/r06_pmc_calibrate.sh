#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=.
rm -rf gpurun_out/r06_cal; mkdir -p gpurun_out/r06_cal
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r06_cal/w -- python3 scripts/pmc_calibrate_write.py run > gpurun_out/r06_cal/w.log 2>&1; echo "write pass rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r06_cal/f -- python3 scripts/pmc_calibrate_write.py run > gpurun_out/r06_cal/f.log 2>&1; echo "fetch pass rc=$?"
python3 scripts/pmc_calibrate_write.py report gpurun_out/r06_cal | tee gpurun_out/r06_pmc_write_calibration.txt
rm -rf gpurun_out/r06_cal/w gpurun_out/r06_cal/f
