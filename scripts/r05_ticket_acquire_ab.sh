#!/bin/bash
# timing experiment: the ticket rollout A/B: the shipped library (no L1 invalidate at ticket pick-up: tier 0 reads past the L1) against -DRMJ_Q_ACQ (an agent-scope acquire per pick-up, as in rounds 2-4)
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for lib in libriichi_mi355x.so libvar_acq.so; do for tail in 0 1; do for c in 4 8; do
  echo "== $lib RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_MIN_CHUNK=$c"
  for i in 1 2; do RMJ_LIB_PATH=riichienv_amd/$lib RMJ_QUEUE_TAIL=$tail RMJ_QUEUE_MIN_CHUNK=$c python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  20 steps: %.1f M  kernel %.4f ms  %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['kernel']))"; done
done; done
  RMJ_LIB_PATH=riichienv_amd/$lib python bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1000 steps: %.1f M  kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done
