#!/bin/bash
# round 6: the driver's 20-step window against the shortest ticket (RMJ_QUEUE_MIN_CHUNK, calls per ticket) and the tail schedule
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for rep in 1 2 3; do
for mc in 3 4 5 6 7 10; do
RMJ_QUEUE_MIN_CHUNK=$mc timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('min_chunk $mc: window %.1f M  kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done
done
