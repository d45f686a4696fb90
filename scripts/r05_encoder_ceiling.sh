#!/bin/bash
# round 5: what bounds k_encode_base?  The shipped kernel (5 waves per SIMD), the same compiled for 8 waves per SIMD (-DRMJ_ENC_WAVES=8 ->
# libriichi_mi355x_encw8.so), its memory side alone (-DRMJ_ENC_NOCOMPUTE -> _encnc.so: record in, zero-coded bytes through the same emit loop out),
# and a plain device fill of the same buffers (torch.Tensor.fill_) as the write-only rate of the part.
cd "$(dirname "$0")/.." && export PYTHONPATH=.
for v in "" encw8 encnc "" encw8 encnc; do echo "== variant [${v:-shipped}]"; timeout 150 python scripts/bench_encoders.py $v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  ' + '  '.join('%s %.1f us %.2f TB/s' % (k.replace('k_encode','enc'), v['kernel_ms']*1e3, v['achieved_GBps']/1e3) for k,v in d.items() if 'ext' not in k))"; done
python3 - <<'PY'
import torch
for rows, stride in ((65792, 2516), (65792, 2560), (65593, 1998), (65593, 2048)):
    b = torch.empty((rows, stride), dtype=torch.float32, device="cuda:0")
    for _ in range(5): b.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): b.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    print("  torch fill_ of %d x %d floats (%.0f MB): %.1f us  %.2f TB/s" % (rows, stride, b.numel() * 4 / 1e6, ms * 1e3, b.numel() * 4 / (ms * 1e-3) / 1e12))
PY
