#!/bin/bash
# round 5 A/B: heavy-first order of the per-step kernel (journal r05 section 10): without the "a seat waits without a riichi" hint
# (-DRMJ_HEAVY_TENPAI=0 -> libvar_noheavytenpai.so), with it for valid wait caches only (=1: what ships -> libriichi_mi355x.so / libvar_heavytenpai1.so), and with stale caches
# under an old shanten bound <= 1 too (=2: measured and rejected, 163 us per launch: journal r05 section 10)
cd "$(dirname "$0")/.." && export PYTHONPATH=.
one() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  window %.1f M  single_stream %.1f M (%.1f us)  validated %.1f M (%.1f us)  long %.1f M' % (d['value']/1e6, d['single_stream']['value']/1e6, d['single_stream']['ms_per_step']*1e3, d['validated_actions']['value']/1e6, d['validated_actions']['ms_per_step']*1e3, d['long_rollout']['value']/1e6))"; }
for rep in 1 2 3; do for lib in libriichi_mi355x.so libvar_heavytenpai1.so libvar_noheavytenpai.so; do
  echo "== $lib"; export RMJ_LIB_PATH=riichienv_amd/$lib; one
done; done
