#!/bin/bash
# HBM traffic counters of scripts/bench_variant.py on an alternative build: scripts/pmc_variant.sh <lib.so> <tag>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for grp in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  g=$(echo $grp | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pv_$2_$g -- python3 scripts/bench_variant.py $1 2 random > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob("$R/gpurun_out/pv_$2_$g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step4_queue" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in tot: print("$2", k, "launches", n[k], "total", tot[k])
PY
  rm -rf $R/gpurun_out/pv_$2_$g
done
