#!/bin/bash
# A/B on one box: row-form settlements (the build) vs the settlements in the full path (-DRMJ_ROW_SETTLE=0) vs the build before them
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  for lib in riichienv_amd/libriichi_mi355x.so riichienv_amd/libvar_nosettle.so riichienv_amd/libvar_head.so; do
    [ -f $lib ] || continue
    python3 scripts/bench_variant.py $lib 2 random 2>&1 | tail -1
    python3 scripts/bench_variant.py $lib 2 greedy 2>&1 | tail -1
  done
done
