#!/bin/bash
# round 5 A/B: the step + encode kernels (k_step4_act_enc, k_step4_enc, k_step4_queue_enc) at five waves per SIMD with the 256-entry value table
# (libvar_lut256.so: 7 280 B of LDS per wave), at five with the 64-entry table (shipped build before the switch), and at six (-DRMJ_STEP4_ENC_WAVES=6 ->
# libvar_encw6.so: 80 VGPR, 6 512 B): trainer loop (one stream, padded rows) and the 3P step + encode rollout of configs[4]
cd "$(dirname "$0")/.." && export PYTHONPATH=.
enc() { timeout 200 python bench.py --steps 300 --warmup 5 --mode 5 --encode --no-cpu-baseline --no-extras --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  3P + encode 300 steps: %.1f M' % (d['value']/1e6))"; }
for rep in 1 2; do for lib in ${LIBS:-libvar_lut256.so libriichi_mi355x.so libvar_encw6.so}; do echo "== $lib"; export RMJ_LIB_PATH=riichienv_amd/$lib; timeout 120 python scripts/profile_one_launch_loop.py 2>/dev/null | sed "s/.*shared stream: /  /"; enc; done; done
