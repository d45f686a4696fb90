#!/bin/bash
# heavy-first launch order of the per-step kernel: single_stream / validated_actions / trainer loop with RMJ_HEAVY_FIRST = 0 | 1
cd ${GRAFT_REPO_ROOT:-.}
for hf in 0 1; do
  export RMJ_HEAVY_FIRST=$hf
  python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('RMJ_HEAVY_FIRST=$hf', 'single_stream', round(d['single_stream']['value']/1e6,1), 'validated', round(d['validated_actions']['value']/1e6,1), 'headline', round(d['value']/1e6,1))"
  python3 scripts/bench_torch_env.py 2>&1 | grep -v amdgpu | sed -n 1,3p
done
