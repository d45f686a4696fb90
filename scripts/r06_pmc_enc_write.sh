#!/bin/bash
# WRITE_SIZE of the fused 3P step + encode rollout (configs[4]), dense rows and rows padded to 256 B, per step of all games, next to the bytes it must write
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=.
# the rows a step really writes: acting seats per game-step, averaged over 200 steps of the same pre-rolled rollout (a WaitResponse step writes one row per seat that was offered a claim)
ROWS=$(python3 - <<PY
import numpy as np
from riichienv_amd import vecenv
e = vecenv.VecRiichiEnv(65536, game_mode=5, seed=0, event_ring=64); e.reset(); e.step_random(0xC0FFEE, 600, auto_reset=True)
tot = 0
for _ in range(200):
    e.step_random(0xC0FFEE, 1, auto_reset=True)
    a, _, d = e.status()
    tot += int(np.unpackbits(a[d == 0][:, None], axis=1).sum())
print(tot / 200.0)
PY
)
echo "acting seats (= rows written) per step of all 65 536 games, mean of 200 steps: $ROWS"
for v in dense padded; do
  extra=""; [ $v = padded ] && extra="--padded-rows"
  rm -rf gpurun_out/r06_encw_$v
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r06_encw_$v -- python3 bench.py --mode 5 --encode --steps 300 --warmup 300 --preroll 300 --no-cpu-baseline --no-extras $extra > gpurun_out/r06_encw_$v.log 2>&1
  python3 - <<PY
import csv, glob, json
acc={}
for f in glob.glob("gpurun_out/r06_encw_$v/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_step4_queue_enc" in row["Kernel_Name"] and row["Counter_Name"]=="WRITE_SIZE":
            acc[row["Dispatch_Id"]]=acc.get(row["Dispatch_Id"],0.0)+float(row["Counter_Value"])
line=json.loads([l for l in open("gpurun_out/r06_encw_$v.log") if l.startswith("{\"metric\"")][-1])
acting=float("$ROWS")
vals=sorted(acc.values())
per_step=vals[len(vals)//2]*1024/300
need=acting*7992+65536*(640+64)+acting*14*8
print("$v rows: WRITE_SIZE %.1f MB per step of all games (median of %d launches of 300 steps); must write ~%.1f MB (%.0f acting seats x 7 992 B + 65 536 x (640 B record + 64 B events) + lists): ratio %.3f; rollout %.1f M env.step/s" % (per_step/1e6, len(vals), need/1e6, acting, per_step/need, line["value"]/1e6))
PY
  rm -rf gpurun_out/r06_encw_$v
done
