#!/bin/bash
# Parity soak of round 6's final build (-disable-machine-licm, row_ballot16, 32-bit policy keys, the group-residue tests of the fused rollouts): scripts/soak_parity.py runs (every game of every mode / rule set against the oracle: final
# states, lists, masks, waits, step counts, whole MJAI logs), 16 processes at a time on one GPU box.
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r06_soak2; mkdir -p $OUT
run() { # tag, env assignments..., -- args
  local tag=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( env "${envs[@]}" python3 scripts/soak_parity.py "$@" > $OUT/$tag.log 2>&1; echo "$tag rc=$?" >> $OUT/_done.txt ) &
}
: > $OUT/_done.txt
# wave 1: the ticket rollout of round 5 (tickets counted in calls, no L1 invalidate at pick-up, quads hop between CUs): short and long tickets
for kc in "2510 2" "2511 3" "2512 5" "2513 8"; do set -- $kc; run random_tickets_$1 RMJ_QUEUE_FORCE=1 RMJ_QUEUE_MIN_CHUNK=$2 RMJ_QUEUE_CHUNK=$2 -- 512 6000 1 $1; done
for kc in "2514 5" "2515 32"; do set -- $kc; run random_tickets_tail_$1 RMJ_QUEUE_FORCE=1 RMJ_QUEUE_TAIL=1 RMJ_QUEUE_CHUNK=$2 -- 512 6000 1 $1; done
for kr in "2665 64 3" "2729 128 6" "2793 192 32"; do set -- $kr; run greedy_tickets_$1 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=$2 RMJ_QUEUE_FORCE=1 RMJ_QUEUE_CHUNK=$3 RMJ_QUEUE_MIN_CHUNK=$3 -- 512 4000 1 $1; done
for k in 2500 2501; do run random_$k RMJ_QUEUE_FORCE=0 -- 512 6000 1 $k; done
run random_refrng_530 RMJ_SOAK_RULE_EXTRA=256 RMJ_QUEUE_FORCE=1 -- 512 6000 1 2530
run random_refrng_531 RMJ_SOAK_RULE_EXTRA=256 RMJ_QUEUE_FORCE=0 -- 512 6000 1 2531
run greedy_refrng_532 RMJ_SOAK_RULE_EXTRA=256 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=64 -- 512 4000 1 2532
run random_perstep_520 RMJ_STEP4=1 -- 256 3000 1 2520
run random_perstep_refrng_533 RMJ_SOAK_RULE_EXTRA=256 RMJ_STEP4=1 -- 256 3000 1 2533
wait
# wave 2
for kc in "2516 4" "2517 7" "2518 16"; do set -- $kc; run random_tickets_$1 RMJ_QUEUE_FORCE=1 RMJ_QUEUE_MIN_CHUNK=$2 RMJ_QUEUE_CHUNK=$2 -- 512 6000 1 $1; done
for kr in "2600 0" "2664 64" "2855 255"; do set -- $kr; run greedy_$1 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=$2 RMJ_QUEUE_FORCE=0 -- 512 4000 1 $1; done
run greedy_tickets_666 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=64 RMJ_QUEUE_FORCE=1 -- 512 4000 1 2666
run greedy_perstep_667 RMJ_SOAK_POLICY=greedy RMJ_SOAK_CALL_RATE=64 RMJ_STEP4=1 -- 256 2000 1 2667
run random_rows1_521 RMJ_ROWS=1 -- 256 6000 1 2521
run random_rows2_522 RMJ_ROWS=2 -- 256 6000 1 2522
for k in 2502 2503; do run random_$k RMJ_QUEUE_FORCE=0 -- 512 6000 1 $k; done
run random_tickets_refrng_534 RMJ_SOAK_RULE_EXTRA=256 RMJ_QUEUE_FORCE=1 RMJ_QUEUE_MIN_CHUNK=3 RMJ_QUEUE_CHUNK=3 -- 512 6000 1 2534
wait
{ echo "# parity soak of the round-6 build (-disable-machine-licm, v_perm row ballots, 32-bit policy keys, group-residue tests in the fused rollouts; the ticket schedules and RMJ_RULE_REFERENCE_RNG as in round 5, a second set of seeds) (scripts/r06_soak2.sh): every line = one scripts/soak_parity.py run, all twelve (mode, rule set) configurations"
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
} > gpurun_out/r06_parity_soak_second_seeds.log
tail -45 gpurun_out/r06_parity_soak_second_seeds.log
