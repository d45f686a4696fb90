#!/bin/bash
# round 6: kernel statistics of the 4-shard trainer loop under two libraries (RMJ_LIB_PATH is read by riichienv_amd.vecenv at import)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; export PYTHONPATH=$R
for lib in libriichi_exp_r05.so libriichi_mi355x.so; do
  export RMJ_LIB_PATH=$R/riichienv_amd/$lib
  rm -rf gpurun_out/p4_$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p4_$lib -- python3 scripts/r06_parts4_profile.py > gpurun_out/p4_$lib.log 2>&1
  echo "== $lib: $(grep 'shards' gpurun_out/p4_$lib.log | cut -c1-120)"
  find gpurun_out/p4_$lib -name "*kernel_stats.csv" -exec python3 -c "
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]: print('   %-60s calls %6s avg %9.1f us  total %8.2f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))" {} \;
  rm -rf gpurun_out/p4_$lib
done
