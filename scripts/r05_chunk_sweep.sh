#!/bin/bash
# call-based tickets (round 5): the driver's 20-step window and longer rollouts against the ticket schedule
# (RMJ_QUEUE_TAIL: lengths descend towards the expected end, RMJ_QUEUE_MIN_CHUNK: the first ticket's calls of a short rollout)
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for tail in 1 0; do for c in 4 6 8 10; do
  echo "== RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_MIN_CHUNK=$c"
  for i in 1 2; do RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_MIN_CHUNK=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  20 steps: %.1f M  kernel %.4f ms  %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['kernel']))"; done
  RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_MIN_CHUNK=$c python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(' 100 steps: %.1f M  kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done; done
for tail in 1 0; do for q in 64 128; do
  echo "== RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_CHUNK=$q"
  RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_CHUNK=$q python bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1000 steps: %.1f M  kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done; done
