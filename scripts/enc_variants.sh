#!/bin/bash
# encoder occupancy variants (built with -DRMJ_ENC_WAVES=<n> into riichienv_amd/libriichi_mi355x_encw<n>.so) + parity of the default
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize.py tests/test_gpu_torch_env.py tests/test_gpu_replay.py tests/test_gpu_replay_3p.py -x -q -m gpu -k "encode or feature or torch or replay" > gpurun_out/enc_parity.log 2>&1
tail -3 gpurun_out/enc_parity.log
python3 scripts/bench_encoders.py > gpurun_out/enc_w8.json 2>gpurun_out/enc_w8.err; cat gpurun_out/enc_w8.json
for w in 0 6 7; do
  if [ -f riichienv_amd/libriichi_mi355x_encw$w.so ]; then python3 scripts/bench_encoders.py encw$w > gpurun_out/enc_w$w.json 2>gpurun_out/enc_w$w.err; cat gpurun_out/enc_w$w.json; fi
done
python3 bench.py --mode 5 --encode --steps 400 --warmup 400 --no-cpu-baseline > gpurun_out/enc_bench3p.json 2>gpurun_out/enc_bench3p.err; cat gpurun_out/enc_bench3p.json
