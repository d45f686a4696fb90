#!/bin/bash
# call-based tickets (round 5): the driver's 20-step window and longer rollouts against the ticket schedule
# (RMJ_QUEUE_TAIL: lengths descend towards the expected end, RMJ_QUEUE_MIN_CHUNK: the first ticket's calls of a short rollout)
cd "$(dirname "$0")/.." && export PYTHONPATH=.
one() { python bench.py --steps $1 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  %4d steps: %.1f M  kernel %.4f ms  %s' % (d['steps'], d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['kernel']))"; }
for tail in 0 1; do for c in 2 3 4 5 6; do
  echo "== RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_MIN_CHUNK=$c"
  export RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_MIN_CHUNK=$c
  one 20; one 20; one 20; one 50; one 100; one 300
done; done
unset RMJ_QUEUE_MIN_CHUNK
for tail in 0 1; do for q in 16 32 64 128; do
  echo "== RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_CHUNK=$q"
  export RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_CHUNK=$q
  one 300; one 1000; one 1000
done; done
