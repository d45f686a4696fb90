#!/bin/bash
cd "$(dirname "$0")/.." && export PYTHONPATH=.
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_seq_features.py tests/test_gpu_compat_sanma.py tests/test_gpu_multi.py tests/test_gpu_replay.py -q -x > gpurun_out/r06/suite_c.log 2>&1; tail -5 gpurun_out/r06/suite_c.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r06/bench_c.err | tee gpurun_out/r06/bench_c.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  window %.1f M | long %.1f M | greedy %.1f M | single_stream %.1f M | validated %.1f M | refrng %.1f M' % (d['value']/1e6, d['long_rollout']['value']/1e6, d['greedy_policy']['value']/1e6, d['single_stream']['value']/1e6, d['validated_actions']['value']/1e6, d.get('reference_rng',{}).get('value',0)/1e6))
e=d['external_policy']; print('  external policy: one stream %.1f M (%.1f us/step) | alternating halves %.1f M (%.1f us/step)' % (e['one_stream']['value']/1e6, e['one_stream']['ms_per_step']*1e3, e['alternating_halves']['value']/1e6, e['alternating_halves']['ms_per_step']*1e3))"
tail -3 gpurun_out/r06/bench_c.err
timeout 600 python scripts/bench_torch_env.py net 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/torch_net.txt
