import os, sys
sys.path.insert(0, "/root/repo")
from riichienv_amd import vecenv
for n in (8192, 16384, 32768, 65536, 131072, 262144, 524288):
    env = vecenv.VecRiichiEnv(n, game_mode=2, seed=0)
    env.reset()
    env.step_random(0xC0FFEE, 200, auto_reset=True)
    r = env.bench_rollout(0xC0FFEE, 0, 500)
    print(f"n={n}: kernel {r.step_kernel_ms*1e3:.1f} us, {r.env_steps / r.total_ms / 1e3:.1f} M env.step/s, {r.step_kernel_ms*1e6/n:.3f} ns/game")
    env.close()
