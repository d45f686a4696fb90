#!/bin/bash
# A/B: the step function of the fused rollouts out of line (shipped) against inlined into the rollout loop with every loop-invariant input -
# the lane id included - laundered per iteration (-DRMJ_INLINE_STEP -> riichienv_amd/libvar_outofline.so)
cd "$(dirname "$0")/.." && export PYTHONPATH=.
one() { python bench.py --steps $1 --warmup 5 --games $2 --mode $3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  %6d games mode %d %4d steps: %.1f M  kernel %.4f ms  %s' % (d['config']['games_per_gpu'], $3, d['steps'], d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['kernel']))"; }
for rep in 1 2; do for lib in libriichi_mi355x.so libvar_outofline.so; do
  echo "== $lib"
  export RMJ_LIB_PATH=riichienv_amd/$lib
  one 20 65536 2; one 1000 65536 2; one 300 524288 2; one 2000 4096 0; one 1000 65536 5
  python bench.py --steps 300 --warmup 5 --policy greedy --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  greedy 300 steps: %.1f M  kernel %.4f ms  %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['kernel']))"
done; done
