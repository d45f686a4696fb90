#!/bin/bash
# round 6: vector / scalar / LDS instructions per wave-step of the fused RandomAgent rollout for experiment builds (one --pmc pass each); usage: scripts/r06_pmc_insts.sh <lib> ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; export PYTHONPATH=$R
for lib in "$@"; do
  export RMJ_LIB_PATH=$R/riichienv_amd/$lib
  rm -rf gpurun_out/pi_$lib
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pi_$lib -- python3 bench.py --mode 2 --steps 300 --warmup 300 --preroll 300 --no-cpu-baseline --no-extras > gpurun_out/pi_$lib.log 2>&1
  python3 - "$lib" <<'PY'
import csv, glob, sys, collections
lib = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/pi_{lib}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step4_queue<0>" in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
last = acc[max(acc)]                       # the timed launch: 300 steps of 65 536 games = 16 384 quads x 300 wave-steps
per = {k: v / (16384 * 300) for k, v in last.items()}
print("%-28s per wave-step: VALU %.1f  SALU %.1f  LDS %.1f  branch %.1f" % (lib, per.get("SQ_INSTS_VALU", 0), per.get("SQ_INSTS_SALU", 0), per.get("SQ_INSTS_LDS", 0), per.get("SQ_INSTS_BRANCH", 0)))
PY
  rm -rf gpurun_out/pi_$lib gpurun_out/pi_$lib.log
done
