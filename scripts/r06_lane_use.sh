#!/bin/bash
# Lane use by section of the step (round 6: the same account on the round-6 binary): the accounting build (-DRMJ_CUTS) ends the wave at a section
# mark; instruction counts and SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU accumulated up to each mark, differences = the section's own.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for pol in random greedy; do
  rm -rf gpurun_out/lu_$pol; mkdir -p gpurun_out/lu_$pol
  RMJ_POLICY=$pol timeout 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/lu_$pol/a -- python3 scripts/valu_sections4.py run > gpurun_out/lu_$pol/a.log 2>&1
  RMJ_POLICY=$pol timeout 500 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/lu_$pol/b -- python3 scripts/valu_sections4.py run > gpurun_out/lu_$pol/b.log 2>&1
  { echo "== k_step4<false> (one launch per step, rich tier 0), policy $pol, 65 536 games: counters accumulated up to each section mark, per wave of four games"
    RMJ_POLICY=$pol python3 scripts/valu_sections4.py report gpurun_out/lu_$pol; } > gpurun_out/r06_lane_use_$pol.txt 2>&1
  rm -rf gpurun_out/lu_$pol
done
cat gpurun_out/r06_lane_use_random.txt
