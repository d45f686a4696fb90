#!/bin/bash
# Parity soak of the round's final build: scripts/soak_parity.py runs (every game of every mode / rule set against the oracle: final
# states, lists, masks, waits, step counts, whole MJAI logs), 16 processes at a time on one GPU box.
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r04_soak; mkdir -p $OUT
run() { # tag, env assignments..., -- args
  local tag=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( env "${envs[@]}" python3 scripts/soak_parity.py "$@" > $OUT/$tag.log 2>&1; echo "$tag rc=$?" >> $OUT/_done.txt ) &
}
: > $OUT/_done.txt
# wave 1: RandomAgent rollouts (fused, tickets forced, one launch per step, fewer games per wave)
for k in 500 501 502 503; do run random_$k RMJ_QUEUE_FORCE=0 -- 512 6000 1 $k; done
for k in 510 511; do run random_tickets_$k RMJ_QUEUE_FORCE=1 -- 512 6000 1 $k; done
run random_perstep_520 RMJ_STEP4=1 -- 256 3000 1 520
run random_rows1_521 RMJ_ROWS=1 -- 256 6000 1 521
run random_rows2_522 RMJ_ROWS=2 -- 256 6000 1 522
# greedy policy (wins, riichi, kans: settlements and yaku checks between the passes), call rates 0 .. 256 of 256
for kr in "600 0" "664 64" "728 128" "855 255" "632 32" "792 192"; do set -- $kr; run greedy_$1 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=$2 RMJ_QUEUE_FORCE=0 -- 512 4000 1 $1; done
run greedy_tickets_665 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=64 RMJ_QUEUE_FORCE=1 -- 512 4000 1 665
wait
# wave 2
for kr in "700 96" "701 160" "702 224" "703 16"; do set -- $kr; run greedy_$1 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=$2 RMJ_QUEUE_FORCE=0 -- 512 4000 1 $1; done
run greedy_tickets_729 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=128 RMJ_QUEUE_FORCE=1 -- 512 4000 1 729
run greedy_perstep_666 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=64 RMJ_STEP4=1 -- 256 2000 1 666
run greedy_perstep_730 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=200 RMJ_STEP4=1 -- 256 2000 1 730
run greedy_rows1_667 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=64 RMJ_ROWS=1 -- 256 4000 1 667
run greedy_rows2_668 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=64 RMJ_ROWS=2 -- 256 4000 1 668
for k in 504 505 506 507; do run random_$k RMJ_QUEUE_FORCE=0 -- 512 6000 1 $k; done
for k in 512 513; do run random_tickets_$k RMJ_QUEUE_FORCE=1 -- 512 6000 1 $k; done
wait
{ echo "# parity soak of the final build of round 4 (scripts/r04_soak.sh): every line = one scripts/soak_parity.py run, all twelve (mode, rule set) configurations"
  for f in $OUT/*.log; do echo "== $(basename $f .log): $(grep -c ' ok (' $f) configurations ok; $(tail -1 $f)"; done
  cat $OUT/_done.txt | sort
  python3 - <<PY
import glob,re
t=0
for f in glob.glob("$OUT/*.log"):
    m=re.search(r"soak ok: (\d+) game-steps", open(f).read())
    if m: t+=int(m.group(1))
print("total game-steps compared equal:", t)
PY
} > gpurun_out/r04_parity_soak_final.log
tail -45 gpurun_out/r04_parity_soak_final.log
