cd $GRAFT_REPO_ROOT && export PYTHONPATH=.
r() { timeout 200 python bench.py --steps 1000 --warmup 5 --no-cpu-baseline --no-extras --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  random 1000 steps: %.1f M' % (d['value']/1e6))"; }
g() { timeout 200 python bench.py --steps 300 --warmup 5 --policy greedy --no-cpu-baseline --no-extras --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  greedy 300 steps: %.1f M' % (d['value']/1e6))"; }
for rep in 1 2 3; do for lib in libriichi_mi355x.so libvar_noabskip.so; do echo "== $lib"; export RMJ_LIB_PATH=riichienv_amd/$lib; r; g; done; done
