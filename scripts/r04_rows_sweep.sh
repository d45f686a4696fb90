#!/bin/bash
# games per wave (RMJ_ROWS = 4 / 2 / 1) against the batch size: fused RandomAgent rollouts of 4p-red-single (configs[1]) and 4p-red-half games
cd ${GRAFT_REPO_ROOT:-.}
run() { python3 bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M env.step/s', d['roofline']['kernel'])"; }
for n in 2048 4096 8192 16384 32768; do for r in 4 2 1; do echo -n "games $n mode 0 rows $r: "; ( export RMJ_ROWS=$r; run --games $n --mode 0 ); done; done
for n in 4096 16384; do for r in 4 2; do echo -n "games $n mode 2 rows $r single_stream: "; ( export RMJ_ROWS=$r RMJ_STEP_STREAMS=1; run --games $n --mode 2 --steps 300 ); done; done
