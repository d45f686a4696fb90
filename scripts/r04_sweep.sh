cd ${GRAFT_REPO_ROOT:-.}
run() { python3 bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M', d['roofline']['kernel'], round(d['roofline']['kernel_ms'],4), 'ms')"; }
for e in "" "RMJ_QUEUE_MIN_CHUNK=10" "RMJ_QUEUE_MIN_CHUNK=5" "RMJ_QUEUE_MIN_CHUNK=20" "RMJ_QUEUE_CHUNK=0"; do echo "driver window [$e]"; for i in 1 2; do env $e python3 -c "print(end='')"; ( export $e >/dev/null 2>&1; run --steps 20 --warmup 5 ); done; done
echo "4096 games mode 0 [default]"; run --games 4096 --mode 0
echo "4096 games mode 0 [RMJ_STEP4=0]"; ( export RMJ_STEP4=0; run --games 4096 --mode 0 )
echo "4096 games mode 0 [RMJ_QUEUE_CHUNK=0]"; ( export RMJ_QUEUE_CHUNK=0; run --games 4096 --mode 0 )
python3 scripts/bench_torch_env.py 2>&1 | grep -v amdgpu | tail -8
