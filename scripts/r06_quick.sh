#!/bin/bash
# round 6: the GPU suite (stop at the first failure) and the short bench line of the library as built; usage: scripts/r06_quick.sh <tag> [notests]
cd "$(dirname "$0")/.." && export PYTHONPATH=.
T=${1:-x}; mkdir -p gpurun_out/r06
if [ "$2" != "notests" ]; then python -m pytest tests -m gpu -x -q > gpurun_out/r06/suite_$T.log 2>&1; tail -4 gpurun_out/r06/suite_$T.log; fi
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r06/bench_$T.$rep.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  window %.1f M | long %.1f M | greedy %.1f M | single_stream %.1f M | validated %.1f M | refrng %.1f M' % (d['value']/1e6, d['long_rollout']['value']/1e6, d['greedy_policy']['value']/1e6, d['single_stream']['value']/1e6, d['validated_actions']['value']/1e6, d.get('reference_rng',{}).get('value',0)/1e6))
print('  ' + ' | '.join('%s %.1f M' % (c['config'][:10], c['value']/1e6) for c in d['configs']))"
done
