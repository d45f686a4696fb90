"""Share of the game-steps that took the full path (the games' own counters), for a library given on the command line."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riichienv_amd import vecenv
if len(sys.argv) > 1:
    vecenv.LIB_PATH = os.path.abspath(sys.argv[1])
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
env = vecenv.VecRiichiEnv(65536, game_mode=mode, seed=0)
env.reset()
env.step_random(0xC0FFEE, 1000, auto_reset=True)
for k in (200, 64, 20):
    f0, s0 = env.total_full_path(), env.total_steps()
    env.step_random(0xC0FFEE, k, auto_reset=True)
    f1, s1 = env.total_full_path(), env.total_steps()
    print(os.path.basename(vecenv.LIB_PATH), "QUEUE_CHUNK", os.environ.get("RMJ_QUEUE_CHUNK"), k, "steps: full", f1 - f0, "of", s1 - s0, f"{(f1 - f0) / (s1 - s0):.5f}", flush=True)
