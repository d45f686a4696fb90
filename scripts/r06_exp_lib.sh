#!/bin/bash
# round 6: the short bench line of experiment builds of the library (RMJ_LIB_PATH); usage: scripts/r06_exp_lib.sh <lib> [<lib> ...]
cd "$(dirname "$0")/.." && export PYTHONPATH=.
mkdir -p gpurun_out/r06
for lib in "$@"; do
for rep in 1 2; do
RMJ_LIB_PATH=$PWD/riichienv_amd/$lib timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib  window %.1f M | long %.1f M | greedy %.1f M | single_stream %.1f M | validated %.1f M | refrng %.1f M' % (d['value']/1e6, d['long_rollout']['value']/1e6, d['greedy_policy']['value']/1e6, d['single_stream']['value']/1e6, d['validated_actions']['value']/1e6, d.get('reference_rng',{}).get('value',0)/1e6))"
done
done
