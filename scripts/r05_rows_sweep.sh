#!/bin/bash
# round 5: games per wave (RMJ_ROWS = 1 / 2 / 4) against batch size once more (the thresholds of rmj_api.hip are from round 4: <= 2 560 games one per wave,
# <= 6 144 two): fused 4p-red-single rollouts of 2 000 steps, M env.step/s
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for games in 1024 2048 3072 4096 6144 8192 12288 16384; do
  line="games $games:"
  for rows in 1 2 4; do
    v=$(RMJ_ROWS=$rows timeout 200 python bench.py --games $games --mode 0 --steps 2000 --warmup 50 --preroll 300 --no-cpu-baseline --no-extras --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f' % (d['value']/1e6))")
    line="$line  rows=$rows $v M"
  done
  echo "$line"
done
