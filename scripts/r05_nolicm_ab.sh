#!/bin/bash
# A/B: the shipped library against the same sources compiled with -mllvm -disable-machine-licm (the rollout loops around the inlined step:
# MachineLICM hoists constant materialisations out of them, and the register allocator then spills what it hoisted)
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for rep in 1 2; do for lib in libriichi_mi355x.so libvar_nolicm.so; do
  echo "== $lib"
  RMJ_LIB_PATH=riichienv_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  window %.1f M | long %.1f M | greedy %.1f M | single_stream %.1f M | validated %.1f M' % (d['value']/1e6, d['long_rollout']['value']/1e6, d['greedy_policy']['value']/1e6, d['single_stream']['value']/1e6, d['validated_actions']['value']/1e6))
print('  ' + ' | '.join('%s %.1f M' % (c['config'][:10], c['value']/1e6) for c in d['configs']) + ' | encode frac %.3f | format %.0f M ev/s' % (d['configs'][2]['roofline_encode']['frac'], d['log_drain']['format_events_per_s']/1e6))"
  RMJ_LIB_PATH=riichienv_amd/$lib python scripts/bench_hand_kernels.py 2>/dev/null | grep -i "hands/s\|G \|M " | head -8
done; done
