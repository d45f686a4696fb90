#!/bin/bash
# round 6: one bench line (no config legs, 1 000-step long rollout) per experiment build; usage: scripts/r06_exp_flags.sh <lib> [<lib> ...]
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for lib in "$@"; do
RMJ_LIB_PATH=$PWD/riichienv_amd/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-32s window %.1f M | long %.1f M | greedy %.1f M | single %.1f M | validated %.1f M' % ('$lib', d['value']/1e6, d['long_rollout']['value']/1e6, d['greedy_policy']['value']/1e6, d['single_stream']['value']/1e6, d['validated_actions']['value']/1e6))"
done
