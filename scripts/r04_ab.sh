#!/bin/bash
# A/B on one box: the build vs variant libraries riichienv_amd/libvar_*.so (fused 1 000-step rollouts, 65 536 games, RandomAgent and greedy)
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  for lib in riichienv_amd/libriichi_mi355x.so riichienv_amd/libvar_*.so; do
    [ -f $lib ] || continue
    python3 scripts/bench_variant.py $lib 2 random 2>&1 | tail -1
    python3 scripts/bench_variant.py $lib 2 greedy 2>&1 | tail -1
  done
done
