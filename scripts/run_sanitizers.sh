#!/bin/bash
# CPU sanitizers (GPU-side AddressSanitizer / XNACK runs do not exist on this pool):
#   1. the oracle - the arbiter of every parity claim - rebuilt with -fsanitize=address,undefined (make -C oracle asan) and the whole
#      CPU test suite (-m "not gpu") run against it (RMJ_ORACLE_LIB + libasan preloaded into python);
#   2. the host side of the C-ABI (riichienv_amd/csrc/rmj_host.h: record <-> view, the MJAI formatter) as its own translation unit
#      under the same flags, fed random and adversarial inputs (tests/host_san/host_san.cpp).
# usage: scripts/run_sanitizers.sh [log file, default profiles/r04_sanitizers.log]
set -u
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r04_sanitizers.log}
ASAN_LIB=$(gcc -print-file-name=libasan.so)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
{
  echo "== $(date -u +%Y-%m-%dT%H:%MZ) $(gcc --version | head -1); flags: $SAN"
  echo "== host translation unit (tests/host_san/host_san.cpp)"
  g++ -O1 -g -std=c++17 -Wall -Wextra -pthread $SAN tests/host_san/host_san.cpp -o /tmp/rmj_host_san && \
    ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 /tmp/rmj_host_san
  echo "host_san exit code: $?"
  echo "== oracle under ASan + UBSan: make -C oracle asan, then pytest -m 'not gpu'"
  make -s -C oracle asan && \
    LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    RMJ_ORACLE_LIB=liboracle_asan.so python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
  echo "pytest exit code: ${PIPESTATUS[0]}"
} 2>&1 | tee "$LOG"
grep -q "ERROR: AddressSanitizer\|runtime error:" "$LOG" && { echo "sanitizer findings in $LOG"; exit 1; }
exit 0
