import sys
sys.path.insert(0, "/root/repo")
from riichienv_amd import vecenv
for n in (8192, 65536):
    env = vecenv.VecRiichiEnv(n, game_mode=2, seed=0)
    env.reset()
    env.step_random(0xC0FFEE, 8, auto_reset=True)
    r = env.bench_rollout(0xC0FFEE, 0, 24)      # steps 8..32 of the first kyoku: no round can end yet
    print(f"n={n} early (no round ends): kernel {r.step_kernel_ms*1e3:.1f} us")
    env.step_random(0xC0FFEE, 1000, auto_reset=True)
    r = env.bench_rollout(0xC0FFEE, 0, 200)
    print(f"n={n} steady state:           kernel {r.step_kernel_ms*1e3:.1f} us")
