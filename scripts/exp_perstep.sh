#!/bin/bash
# where does the per-step-launch time go?  early game (no round ends) vs steady state, kernel durations vs wall time; fill-rate calibration
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export RMJ_STEP4=1
echo "early (20 steps after reset)"; python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('launches_in_flight'))"
echo "steady"; python3 $R/bench.py --steps 1000 --warmup 1000 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('launches_in_flight'))"
export RMJ_STEP_STREAMS=1
echo "steady, 1 stream"; python3 $R/bench.py --steps 1000 --warmup 1000 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('launches_in_flight'))"
echo "early, 1 stream"; python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('launches_in_flight'))"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/exp_ps -o ps -- python3 $R/bench.py --steps 300 --warmup 300 --no-cpu-baseline --no-extras > $R/gpurun_out/exp_ps.log 2>&1
find $R/gpurun_out/exp_ps -name "*kernel_stats.csv" | head -1 | xargs head -5
unset RMJ_STEP4 RMJ_STEP_STREAMS
python3 - <<'PY'
import torch, time
for mb in (660, 2600):
    x = torch.empty(mb * 1000 * 1000 // 4, dtype=torch.float32, device="cuda:0")
    for fn, name in ((lambda: x.zero_(), "zero_"), (lambda: x.fill_(1.0), "fill_")):
        fn(); torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print(name, mb, "MB", ms, "ms", x.numel() * 4 / ms / 1e6, "GB/s")
    y = torch.empty_like(x)
    y.copy_(x); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): y.copy_(x)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print("copy", mb, "MB", ms, "ms", 2 * x.numel() * 4 / ms / 1e6, "GB/s (read+write)")
    del x, y
PY
