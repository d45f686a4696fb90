#!/bin/bash
# round 5 A/B: s_setprio 3 around the complete state machine (ol_step_full) of a per-step launch (-DRMJ_FULL_PRIO=1 -> libvar_fprio.so): do the few
# long waves that bound k_step4<false> (journal r04 section 4) end earlier when they issue ahead of their SIMD's other waves?
cd "$(dirname "$0")/.." && export PYTHONPATH=.
one() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  window %.1f M  single_stream %.1f M (%.1f us)  validated %.1f M (%.1f us)  long %.1f M' % (d['value']/1e6, d['single_stream']['value']/1e6, d['single_stream']['ms_per_step']*1e3, d['validated_actions']['value']/1e6, d['validated_actions']['ms_per_step']*1e3, d['long_rollout']['value']/1e6))"; }
for rep in 1 2 3; do for lib in libriichi_mi355x.so libvar_fprio.so; do
  echo "== $lib"; export RMJ_LIB_PATH=riichienv_amd/$lib; one
done; done
