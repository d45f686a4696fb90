#!/bin/bash
# round 6: the one-launch-per-step legs over 300 steps for experiment builds (RMJ_LIB_PATH); usage: scripts/r06_exp_single.sh <lib> [<lib> ...]
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for rep in 1 2; do
for lib in "$@"; do
RMJ_LIB_PATH=$PWD/riichienv_amd/$lib timeout 400 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib  300-step %.1f M | long %.1f M | single_stream %.1f M (%.1f us) | validated %.1f M (%.1f us)' % (d['value']/1e6, d['long_rollout']['value']/1e6, d['single_stream']['value']/1e6, d['single_stream']['ms_per_step']*1e3, d['validated_actions']['value']/1e6, d['validated_actions']['ms_per_step']*1e3))"
done
done
