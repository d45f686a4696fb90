#!/bin/bash
# round 6: bench line incl. the config legs for experiment builds (RMJ_LIB_PATH); usage: scripts/r06_exp_cfg.sh <lib> [<lib> ...]
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for rep in 1 2; do
for lib in "$@"; do
RMJ_LIB_PATH=$PWD/riichienv_amd/$lib timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib  window %.1f M | long %.1f M | greedy %.1f M | single %.1f M | validated %.1f M | ' % (d['value']/1e6, d['long_rollout']['value']/1e6, d['greedy_policy']['value']/1e6, d['single_stream']['value']/1e6, d['validated_actions']['value']/1e6) + ' | '.join('%s %.1f M' % (c['config'][:10], c['value']/1e6) for c in d['configs']))"
done
done
