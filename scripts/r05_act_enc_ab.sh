#!/bin/bash
# round 5: what bounds the trainer loop's step + encode launch (k_step4_act_enc)?  The loop of scripts/profile_one_launch_loop.py (fused masked sampler +
# step and encode as one launch, padded rows, one stream) on the shipped library, with the encoder's stores removed (-DRMJ_ENC_NOSTORE -> libvar_nostore.so)
# and with every other wave started 7 / 14 us late (-DRMJ_ACT_ENC_STAGGER=2 / 4: do the waves of a generation alternate between step and store phases in lock-step?)
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for rep in 1 2; do for lib in libriichi_mi355x.so libvar_nostore.so libvar_stagger2.so libvar_stagger4.so; do echo "== $lib"; RMJ_LIB_PATH=riichienv_amd/$lib timeout 120 python scripts/profile_one_launch_loop.py 2>/dev/null | sed 's/.*shared stream: /  /'; done; done
