#!/bin/bash
# round 5 A/Bs: (1) the step + encode rollout inlined into its loop (-DRMJ_INLINE_ENC=1 -> libvar_encinl.so) against the shipped out-of-line form;
# (2) what RMJ_RULE_REFERENCE_RNG costs a rollout (bench.py --reference-rng: every round start deals through ChaCha12 + the serial swap pass)
cd "$(dirname "$0")/.." && export PYTHONPATH=.
enc() { python bench.py --steps 300 --warmup 5 --mode $1 --encode --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  mode $1 + encode 300 steps: %.1f M  kernel %.4f ms  %s  frac %.3f' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['kernel'], d['roofline']['frac']))"; }
for rep in 1 2; do for lib in libriichi_mi355x.so libvar_encinl.so; do
  echo "== $lib"; export RMJ_LIB_PATH=riichienv_amd/$lib
  enc 5; enc 2
done; done
unset RMJ_LIB_PATH
one() { python bench.py --steps $1 --warmup 5 --mode $2 $3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  mode $2 $3 %4d steps: %.1f M  kernel %.4f ms' % (d['steps'], d['value']/1e6, d['roofline']['kernel_ms']))"; }
for rep in 1 2; do
  echo "== seed -> wall: the build's shuffle / the reference's chain"
  one 1000 2 ""; one 1000 2 --reference-rng; one 1000 5 ""; one 1000 5 --reference-rng; one 1000 0 ""; one 1000 0 --reference-rng
  python bench.py --steps 300 --warmup 5 --policy greedy --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  greedy 300 steps: %.1f M' % (d['value']/1e6))"
  python bench.py --steps 300 --warmup 5 --policy greedy --reference-rng --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  greedy --reference-rng 300 steps: %.1f M' % (d['value']/1e6))"
done
