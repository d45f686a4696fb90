#!/bin/bash
python3 -m pytest tests/test_gpu_encode_compact.py -x -q 2>&1 | grep -E "assert|Error|passed|failed" | head -8
for v in "" e0 e3 e5 e6 "" e3; do
  echo "variant [$v]"; python3 scripts/bench_encoders.py $v 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v['kernel_ms'],4), round(v['frac_of_8TBps'],3)) for k,v in d.items() if 'ext' not in k})"
done
