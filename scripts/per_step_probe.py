"""300 per-step launches of the whole batch on one stream (what an external policy drives): for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riichienv_amd import vecenv
env = vecenv.VecRiichiEnv(65536, game_mode=2, seed=0)
env.reset()
env.step_random(0xC0FFEE, 2000, auto_reset=True)
env.set_rollout_streams(1)
r = env.bench_rollout(0xC0FFEE, 0, 300)
print("per step", r.total_ms / 300 * 1e3, "us", r.env_steps / r.total_ms / 1e3, "M env.step/s")
